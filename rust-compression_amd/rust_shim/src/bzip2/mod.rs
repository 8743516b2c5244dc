pub mod decoder;
pub mod encoder;
pub mod error;
