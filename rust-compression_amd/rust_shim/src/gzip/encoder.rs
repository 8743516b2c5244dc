//! `GZipEncoder` (the reference's `src/gzip/encoder.rs:50-135`): the 10-byte header, the Deflate
//! stream, CRC-32 and ISIZE little endian.
use crate::deflate::encoder::{deflate_family, DeflateFamilyEncoder};
use crate::ffi;

pub struct GZipEncoder(DeflateFamilyEncoder);

deflate_family!(GZipEncoder, ffi::DF_KIND_GZIP);
