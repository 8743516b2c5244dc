//! `BZip2Decoder` (the reference's `src/bzip2/decoder.rs:583-612`) over section 3 of the C ABI: the
//! streaming context `bz_dec_*`.  Compressed bytes go in as the iterator yields them, decoded bytes
//! come out in order, an error (if any) after the bytes in front of it -- the reference's item sequence.
#[cfg(not(feature = "std"))]
use alloc::vec::Vec;

use crate::bzip2::error::BZip2Error;
use crate::ffi::{self, bz_dec, bz_dec_create, bz_dec_destroy, bz_dec_end, bz_dec_read, bz_dec_write};
use crate::mi355x::Status;
use crate::traits::decoder::Decoder;

const CHUNK: usize = 1 << 20;

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
    /// `BZip2Decoder::new()` (src/bzip2/decoder.rs:588-594).  Panics when no MI355X is usable: the
    /// reference's constructor cannot fail and this crate has no CPU path to fall back to
    /// (`try_new` returns the status instead).
    pub fn new() -> Self {
        match Self::try_new(0) {
            Ok(d) => d,
            Err(s) => panic!("bz2_mi355x: cannot create a decoder context: {:?}", s),
        }
    }

    pub fn try_new(device: usize) -> Result<Self, Status> {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { bz_dec_create(&mut h, device as i32) };
        if rc != ffi::BZ_OK {
            crate::mi355x::note_status(rc);
            return Err(Status::from_code(rc));
        }
        Ok(Self { h, ready: Vec::new(), pos: 0, ended: false, failed: false })
    }
}

impl Drop for BZip2Decoder {
    fn drop(&mut self) {
        unsafe { bz_dec_destroy(self.h) }
    }
}

fn map_bz_err(rc: i32) -> BZip2Error {
    BZip2Error::from_status(rc)
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

unsafe impl Send for BZip2Decoder {}
