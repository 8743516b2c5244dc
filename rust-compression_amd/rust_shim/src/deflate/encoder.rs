//! `Inflater` -- the reference's name for its Deflate ENCODER (`src/deflate/encoder.rs:92-260`) -- over
//! section 4 of the C ABI (`df_enc_*`), and the body the zlib / gzip wrappers share
//! (`src/zlib/encoder.rs:55-157`, `src/gzip/encoder.rs:50-135`).  `Action::Run` accumulates,
//! `Action::Finish` produces the stream (one window and one bit string run through all of the input).
#[cfg(not(feature = "std"))]
use alloc::vec::Vec;
use core::ffi::c_void;

use crate::action::Action;
use crate::error::CompressionError;
use crate::ffi::{
    self, df_enc_create, df_enc_create_dict, df_enc_destroy, df_enc_end, df_enc_finished, df_enc_read, df_enc_write,
};

const CHUNK: usize = 1 << 20;

fn map_err(rc: i32) -> CompressionError {
    CompressionError::from_status(rc)
}

fn action_code(a: Action) -> i32 {
    a.code()
}

pub struct DeflateFamilyEncoder {
    kind: i32,
    h: *mut c_void,
    ready: Vec<u8>,
    pos: usize,
    chunk: Vec<u8>,
}

impl DeflateFamilyEncoder {
    pub(crate) fn with_kind_and_dict(kind: i32, dict: &[u8]) -> Self {
        let mut h: *mut c_void = core::ptr::null_mut();
        let rc = unsafe { df_enc_create_dict(&mut h, kind, 0, dict.as_ptr(), dict.len()) };
        if rc != 0 {
            crate::mi355x::note_status(rc);
            panic!("bz2_mi355x: cannot create a Deflate encoder context: {:?}", crate::mi355x::Status::from_code(rc));
        }
        Self { kind, h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    pub(crate) fn with_kind(kind: i32) -> Self {
        let mut h: *mut c_void = core::ptr::null_mut();
        let rc = unsafe { df_enc_create(&mut h, kind, 0) };
        if rc != 0 {
            crate::mi355x::note_status(rc);
            panic!("bz2_mi355x: cannot create a Deflate encoder context: {:?}", crate::mi355x::Status::from_code(rc));
        }
        Self { kind, h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    pub(crate) fn next<I: Iterator<Item = u8>>(&mut self, iter: &mut I, action: Action) -> Option<Result<u8, CompressionError>> {
        // Action::Flush: a byte-aligned segment for Inflater.  The zlib / gzip wrappers end their container at the
        // first None of the inner Inflater whatever the Action (src/zlib/encoder.rs:138-150): the library writes
        // header + what the Inflater yields + trailer, and from then on the iterator is not pulled (:130-136).
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
            if self.kind != ffi::DF_KIND_DEFLATE && unsafe { df_enc_finished(self.h) } != 0 {
                self.ready.clear();
                return None;
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

unsafe impl Send for DeflateFamilyEncoder {}

/// implements `new`, `Default` and `Encoder` for a wrapper of `DeflateFamilyEncoder`
macro_rules! deflate_family {
    ($name:ident, $kind:expr) => {
        impl $name {
            pub fn new() -> Self {
                $name($crate::deflate::encoder::DeflateFamilyEncoder::with_kind($kind))
            }
        }
        impl Default for $name {
            fn default() -> Self {
                Self::new()
            }
        }
        impl $crate::traits::encoder::Encoder for $name {
            type Error = $crate::error::CompressionError;
            type In = u8;
            type Out = u8;
            fn next<I: Iterator<Item = u8>>(
                &mut self,
                iter: &mut I,
                action: $crate::action::Action,
            ) -> Option<Result<u8, $crate::error::CompressionError>> {
                self.0.next(iter, action)
            }
        }
    };
}
pub(crate) use deflate_family;

/// `Inflater::new()` (src/deflate/encoder.rs:113-132)
pub struct Inflater(DeflateFamilyEncoder);

impl Inflater {
    /// `Inflater::with_dict` (src/deflate/encoder.rs:134-153)
    pub fn with_dict(dict: &[u8]) -> Self {
        Inflater(DeflateFamilyEncoder::with_kind_and_dict(ffi::DF_KIND_DEFLATE, dict))
    }
}
deflate_family!(Inflater, ffi::DF_KIND_DEFLATE);
