pub mod encoder;
