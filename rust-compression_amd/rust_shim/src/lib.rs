//! Rust shim: `BZip2Encoder` of the `compression` crate re-implemented over the MI355X C ABI
//! (include/bz2_mi355x.h).  NOT compiled in this repository's image (no Rust toolchain); it is
//! the binding a maintainer drops into the crate as `src/bzip2/encoder.rs` behind a
//! `feature = "mi355x"` gate.  The trait, `Action`, `CompressionError` and `EncodeExt` stay the
//! crate's own (src/traits/encoder.rs, src/action.rs, src/error.rs), so callers do not change:
//!
//!     let out = data.iter().cloned()
//!         .encode(&mut BZip2Encoder::new(9), Action::Finish)
//!         .collect::<Result<Vec<_>, _>>();
use crate::action::Action;
use crate::error::CompressionError;
use crate::traits::encoder::Encoder;
use core::ffi::c_void;

#[allow(non_camel_case_types)]
type bz_enc = c_void;

#[link(name = "bz2_mi355x")]
extern "C" {
    fn bz_enc_create(out: *mut *mut bz_enc, level: i32, device: i32) -> i32;
    fn bz_enc_write(e: *mut bz_enc, data: *const u8, n: usize) -> i32;
    fn bz_enc_end(e: *mut bz_enc, action: i32) -> i32;
    fn bz_enc_read(e: *mut bz_enc, out: *mut u8, cap: usize) -> isize;
    fn bz_enc_pending(e: *const bz_enc) -> usize;
    fn bz_enc_destroy(e: *mut bz_enc);
}

const CHUNK: usize = 1 << 20;

pub struct BZip2Encoder {
    h: *mut bz_enc,
    ready: Vec<u8>,
    pos: usize,
    chunk: Vec<u8>,
}

impl Default for BZip2Encoder {
    fn default() -> Self {
        Self::new(9)
    }
}

impl BZip2Encoder {
    pub fn new(level: usize) -> Self {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { bz_enc_create(&mut h, level as i32, 0) };
        if rc != 0 {
            panic!("invalid level"); // src/bzip2/encoder.rs:59-61
        }
        Self { h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    fn refill(&mut self) -> Result<usize, CompressionError> {
        self.ready.resize(1 << 16, 0);
        let k = unsafe { bz_enc_read(self.h, self.ready.as_mut_ptr(), self.ready.len()) };
        if k < 0 {
            return Err(map_err(k as i32));
        }
        self.ready.truncate(k as usize);
        self.pos = 0;
        Ok(k as usize)
    }
}

fn map_err(rc: i32) -> CompressionError {
    match rc {
        -1 => CompressionError::DataError,
        -2 => CompressionError::UnexpectedEof,
        _ => CompressionError::Unexpected,
    }
}

fn action_code(a: Action) -> i32 {
    match a {
        Action::Run => 0,
        Action::Flush => 1,
        Action::Finish => 2,
    }
}

impl Encoder for BZip2Encoder {
    type Error = CompressionError;
    type In = u8;
    type Out = u8;

    fn next<I: Iterator<Item = u8>>(
        &mut self,
        iter: &mut I,
        action: Action,
    ) -> Option<Result<u8, CompressionError>> {
        if self.pos == self.ready.len() {
            match self.refill() {
                Err(e) => return Some(Err(e)),
                Ok(0) => {
                    loop {
                        self.chunk.clear();
                        while self.chunk.len() < CHUNK {
                            match iter.next() {
                                Some(b) => self.chunk.push(b),
                                None => break,
                            }
                        }
                        let exhausted = self.chunk.len() < CHUNK;
                        if !self.chunk.is_empty() {
                            let rc = unsafe { bz_enc_write(self.h, self.chunk.as_ptr(), self.chunk.len()) };
                            if rc != 0 {
                                return Some(Err(map_err(rc)));
                            }
                            if unsafe { bz_enc_pending(self.h) } > 0 {
                                break;
                            }
                        }
                        if exhausted {
                            let rc = unsafe { bz_enc_end(self.h, action_code(action)) };
                            if rc != 0 {
                                return Some(Err(map_err(rc)));
                            }
                            break;
                        }
                    }
                    match self.refill() {
                        Err(e) => return Some(Err(e)),
                        Ok(0) => return None,
                        Ok(_) => {}
                    }
                }
                Ok(_) => {}
            }
        }
        let b = self.ready[self.pos];
        self.pos += 1;
        Some(Ok(b))
    }
}

impl Drop for BZip2Encoder {
    fn drop(&mut self) {
        unsafe { bz_enc_destroy(self.h) }
    }
}

// ---- decoder: `BZip2Decoder` (src/bzip2/decoder.rs:583-612) over section 3 of the C ABI -------------
// Dropped into the crate as `src/bzip2/decoder.rs` behind the same feature gate; `Decoder`,
// `DecodeExt` and `BZip2Error` stay the crate's own (src/traits/decoder.rs, src/bzip2/error.rs).
use crate::bzip2::error::BZip2Error;
use crate::traits::decoder::Decoder;

#[allow(non_camel_case_types)]
type bz_dec = c_void;

#[link(name = "bz2_mi355x")]
extern "C" {
    fn bz_dec_create(out: *mut *mut bz_dec, device: i32) -> i32;          // BZip2Decoder::new      decoder.rs:588-594
    fn bz_dec_write(d: *mut bz_dec, data: *const u8, n: usize) -> i32;    // iterator yields bytes
    fn bz_dec_end(d: *mut bz_dec) -> i32;                                 // iterator returns None: decode
    fn bz_dec_read(d: *mut bz_dec, out: *mut u8, cap: usize) -> isize;    // items handed out       decoder.rs:604-612
    fn bz_dec_destroy(d: *mut bz_dec);
}

pub struct BZip2Decoder {
    h: *mut bz_dec,
    ready: Vec<u8>,
    pos: usize,
    ended: bool,
    failed: bool,
}

impl Default for BZip2Decoder {
    fn default() -> Self {
        Self::new()
    }
}

impl BZip2Decoder {
    pub fn new() -> Self {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { bz_dec_create(&mut h, 0) };
        assert!(rc == 0, "bz2_mi355x: no usable gfx950 device");
        Self { h, ready: Vec::new(), pos: 0, ended: false, failed: false }
    }
}

impl Drop for BZip2Decoder {
    fn drop(&mut self) {
        unsafe { bz_dec_destroy(self.h) }
    }
}

fn map_bz_err(rc: i32) -> BZip2Error {
    match rc {
        -1 => BZip2Error::DataError,
        -4 => BZip2Error::DataErrorMagicFirst,
        -5 => BZip2Error::DataErrorMagic,
        -2 => BZip2Error::UnexpectedEof,
        _ => BZip2Error::Unexpected,
    }
}

impl Decoder for BZip2Decoder {
    type Input = u8;
    type Output = u8;
    type Error = BZip2Error;

    fn next<I: Iterator<Item = u8>>(&mut self, iter: &mut I) -> Option<Result<u8, BZip2Error>> {
        while self.pos == self.ready.len() {
            self.ready.resize(1 << 16, 0);
            let k = unsafe { bz_dec_read(self.h, self.ready.as_mut_ptr(), self.ready.len()) };
            if k < 0 {
                self.ready.clear();
                self.pos = 0;
                if self.failed {
                    return None;
                }
                self.failed = true;
                return Some(Err(map_bz_err(k as i32)));
            }
            self.ready.truncate(k as usize);
            self.pos = 0;
            if k > 0 {
                break;
            }
            if self.ended {
                return None; // 0 after the end: the clean end
            }
            // nothing ready yet: hand over more input (the reference pulls bytes on demand; the bytes
            // are the same).  The library decodes complete records every BZ_DEC_CHUNK bytes.
            let chunk: Vec<u8> = iter.by_ref().take(CHUNK).collect();
            if !chunk.is_empty() {
                let rc = unsafe { bz_dec_write(self.h, chunk.as_ptr(), chunk.len()) };
                if rc != 0 {
                    return Some(Err(map_bz_err(rc)));
                }
            }
            if chunk.len() < CHUNK {
                self.ended = true;
                unsafe { bz_dec_end(self.h) }; // the verdict follows the last byte out of bz_dec_read
            }
        }
        let b = self.ready[self.pos];
        self.pos += 1;
        Some(Ok(b))
    }
}


// ---------------------------------------------------------------------------------------------
// Deflate / zlib / gzip encoders (include/bz2_mi355x.h section 4): `Inflater`
// (src/deflate/encoder.rs:92-260), `ZlibEncoder` (src/zlib/encoder.rs:55-157), `GZipEncoder`
// (src/gzip/encoder.rs:50-135).  One body, three constructors.
// ---------------------------------------------------------------------------------------------
#[link(name = "bz2_mi355x")]
extern "C" {
    fn df_enc_create(out: *mut *mut c_void, kind: i32, device: i32) -> i32;
    fn df_enc_create_dict(out: *mut *mut c_void, kind: i32, device: i32, dict: *const u8, dict_len: usize) -> i32;
    fn df_enc_write(e: *mut c_void, data: *const u8, n: usize) -> i32;
    fn df_enc_end(e: *mut c_void, action: i32) -> i32;
    fn df_enc_read(e: *mut c_void, out: *mut u8, cap: usize) -> isize;
    fn df_enc_destroy(e: *mut c_void);
}

pub struct DeflateFamilyEncoder {
    h: *mut c_void,
    ready: Vec<u8>,
    pos: usize,
    chunk: Vec<u8>,
}

/// `Inflater::new()` -- the reference's name for its Deflate encoder
pub struct Inflater(DeflateFamilyEncoder);
pub struct ZlibEncoder(DeflateFamilyEncoder);
pub struct GZipEncoder(DeflateFamilyEncoder);

impl DeflateFamilyEncoder {
    fn with_kind_and_dict(kind: i32, dict: &[u8]) -> Self {
        let mut h: *mut c_void = core::ptr::null_mut();
        let rc = unsafe { df_enc_create_dict(&mut h, kind, 0, dict.as_ptr(), dict.len()) };
        assert!(rc == 0, "bz2_mi355x: no usable MI355X (the path has no CPU fallback)");
        Self { h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    fn with_kind(kind: i32) -> Self {
        let mut h: *mut c_void = core::ptr::null_mut();
        let rc = unsafe { df_enc_create(&mut h, kind, 0) };
        assert!(rc == 0, "bz2_mi355x: no usable MI355X (the path has no CPU fallback)");
        Self { h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    fn next<I: Iterator<Item = u8>>(&mut self, iter: &mut I, action: Action) -> Option<Result<u8, CompressionError>> {
        while self.pos == self.ready.len() {
            self.ready.resize(1 << 16, 0);
            let k = unsafe { df_enc_read(self.h, self.ready.as_mut_ptr(), self.ready.len()) };
            if k < 0 {
                self.ready.clear();
                self.pos = 0;
                return Some(Err(map_err(k as i32)));
            }
            self.ready.truncate(k as usize);
            self.pos = 0;
            if k > 0 {
                break;
            }
            // nothing ready: move the rest of this iterator in, then tell the library it ended
            loop {
                self.chunk.clear();
                self.chunk.extend(iter.by_ref().take(CHUNK));
                if !self.chunk.is_empty() {
                    let rc = unsafe { df_enc_write(self.h, self.chunk.as_ptr(), self.chunk.len()) };
                    if rc != 0 {
                        return Some(Err(map_err(rc)));
                    }
                }
                if self.chunk.len() < CHUNK {
                    break;
                }
            }
            // Flush is not offered by the library (BZ_E_PARAM): surfaces as CompressionError::Unexpected
            let rc = unsafe { df_enc_end(self.h, action_code(action)) };
            if rc != 0 {
                return Some(Err(map_err(rc)));
            }
            self.ready.resize(1 << 16, 0);
            let k = unsafe { df_enc_read(self.h, self.ready.as_mut_ptr(), self.ready.len()) };
            if k <= 0 {
                self.ready.clear();
                self.pos = 0;
                return if k < 0 { Some(Err(map_err(k as i32))) } else { None };
            }
            self.ready.truncate(k as usize);
            self.pos = 0;
        }
        let b = self.ready[self.pos];
        self.pos += 1;
        Some(Ok(b))
    }
}

impl Drop for DeflateFamilyEncoder {
    fn drop(&mut self) {
        unsafe { df_enc_destroy(self.h) }
    }
}

macro_rules! deflate_family {
    ($name:ident, $kind:expr) => {
        impl $name {
            pub fn new() -> Self {
                $name(DeflateFamilyEncoder::with_kind($kind))
            }
        }
        impl Default for $name {
            fn default() -> Self {
                Self::new()
            }
        }
        impl Encoder for $name {
            type Error = CompressionError;
            type In = u8;
            type Out = u8;
            fn next<I: Iterator<Item = u8>>(&mut self, iter: &mut I, action: Action) -> Option<Result<u8, CompressionError>> {
                self.0.next(iter, action)
            }
        }
    };
}
impl Inflater {
    /// `Inflater::with_dict` (src/deflate/encoder.rs:134-153)
    pub fn with_dict(dict: &[u8]) -> Self {
        Inflater(DeflateFamilyEncoder::with_kind_and_dict(0, dict))
    }
}
impl ZlibEncoder {
    /// `ZlibEncoder::with_dict` (src/zlib/encoder.rs:74-93)
    pub fn with_dict(dict: &[u8]) -> Self {
        ZlibEncoder(DeflateFamilyEncoder::with_kind_and_dict(1, dict))
    }
}
deflate_family!(Inflater, 0);
deflate_family!(ZlibEncoder, 1);
deflate_family!(GZipEncoder, 2);
