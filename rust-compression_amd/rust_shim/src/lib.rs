//! MI355X drop-in for the `compression` crate (chalharu/rust-compression 0.1.5): the crate's own
//! surface -- `Action`, `CompressionError`, the `Encoder` / `Decoder` traits with their `encode()` /
//! `decode()` iterator adaptors, and the codec types re-exported by `prelude` -- with the codecs bound
//! to hand-written HIP kernels through the C ABI of `include/bz2_mi355x.h` (`src/ffi.rs`).
//!
//!     use compression::prelude::*;
//!     let out = data.iter().cloned()
//!         .encode(&mut BZip2Encoder::new(9), Action::Finish)
//!         .collect::<Result<Vec<_>, _>>();
//!
//! Output is bit-identical to the reference's encoders / decoders (checked against the C oracle in
//! this repository).  NOT compiled in this repository's image (no Rust toolchain there); module
//! layout, feature flags and `prelude` follow the reference (its `src/lib.rs:48-117`, `Cargo.toml:28-38`).
//! What the reference has and this crate does not: `lzhuf`, the standalone `LzssEncoder` /
//! `LzssDecoder`, and the Deflate-family DECODERS (`Deflater`, `GZipDecoder`, `ZlibDecoder`).
#![cfg_attr(not(feature = "std"), no_std)]
#[cfg(not(feature = "std"))]
extern crate alloc;

mod action;
mod error;
mod traits;

#[cfg(feature = "mi355x")]
mod ffi;
#[cfg(feature = "mi355x")]
pub mod mi355x;

#[cfg(all(feature = "bzip2", feature = "mi355x"))]
mod bzip2;
#[cfg(all(feature = "deflate", feature = "mi355x"))]
mod deflate;
#[cfg(all(feature = "gzip", feature = "mi355x"))]
mod gzip;
#[cfg(all(feature = "zlib", feature = "mi355x"))]
mod zlib;

/// The reference's `prelude` (its `src/lib.rs:70-117`), name for name.
pub mod prelude {
    pub use crate::action::Action;

    #[cfg(all(feature = "bzip2", feature = "mi355x"))]
    pub use crate::bzip2::decoder::BZip2Decoder;
    #[cfg(all(feature = "bzip2", feature = "mi355x"))]
    pub use crate::bzip2::encoder::BZip2Encoder;
    #[cfg(all(feature = "bzip2", feature = "mi355x"))]
    pub use crate::bzip2::error::BZip2Error;

    #[cfg(all(feature = "deflate", feature = "mi355x"))]
    pub use crate::deflate::encoder::Inflater;
    #[cfg(all(feature = "gzip", feature = "mi355x"))]
    pub use crate::gzip::encoder::GZipEncoder;
    #[cfg(all(feature = "zlib", feature = "mi355x"))]
    pub use crate::zlib::encoder::ZlibEncoder;

    pub use crate::error::CompressionError;
    pub use crate::traits::decoder::{DecodeExt, DecodeIterator, Decoder};
    pub use crate::traits::encoder::{EncodeExt, EncodeIterator, Encoder};
}
