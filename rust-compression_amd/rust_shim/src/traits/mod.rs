pub mod decoder;
pub mod encoder;
