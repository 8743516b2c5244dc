//! `ZlibEncoder` (the reference's `src/zlib/encoder.rs:55-157`): 78 DA (or 78 F9 + the dictionary's
//! Adler-32 with `with_dict`), the Deflate stream, Adler-32 big endian.
use crate::deflate::encoder::{deflate_family, DeflateFamilyEncoder};
use crate::ffi;

pub struct ZlibEncoder(DeflateFamilyEncoder);

impl ZlibEncoder {
    /// `ZlibEncoder::with_dict` (src/zlib/encoder.rs:74-93)
    pub fn with_dict(dict: &[u8]) -> Self {
        ZlibEncoder(DeflateFamilyEncoder::with_kind_and_dict(ffi::DF_KIND_ZLIB, dict))
    }
}
deflate_family!(ZlibEncoder, ffi::DF_KIND_ZLIB);
