//! `BZip2Encoder` (the reference's `src/bzip2/encoder.rs:40-159`) over section 1 of the C ABI: the
//! streaming context `bz_enc_*`.  `Encoder::next` drains `bz_enc_read`; when nothing is ready it moves
//! up to CHUNK bytes of the input iterator into `bz_enc_write` (the library uploads and encodes chunks
//! while more input is being written) and, when the iterator is exhausted, tells the library the
//! caller's `Action` with `bz_enc_end`, which replays the reference's Run / Flush / Finish behaviour.
#[cfg(not(feature = "std"))]
use alloc::vec::Vec;

use crate::action::Action;
use crate::error::CompressionError;
use crate::ffi::{
    self, bz_enc, bz_enc_create, bz_enc_create_multi, bz_enc_destroy, bz_enc_end, bz_enc_pending, bz_enc_read, bz_enc_write,
};
use crate::mi355x::Status;
use crate::traits::encoder::Encoder;

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
    /// `BZip2Encoder::new(level)` (src/bzip2/encoder.rs:58-72): panics unless 1 <= level <= 9, and for
    /// nothing else -- creating the context does not touch the GPU; a missing device shows up as the
    /// first item, `Err(CompressionError::Unexpected)`, with `mi355x::last_status() == Some(Status::NoGpu)`.
    pub fn new(level: usize) -> Self {
        match Self::try_new(level, 0) {
            Ok(e) => e,
            Err(Status::InvalidParameter) => panic!("invalid level"), // src/bzip2/encoder.rs:59-61
            Err(s) => panic!("bz2_mi355x: cannot create an encoder context: {:?}", s), // (allocation of the host object)
        }
    }

    /// The same without the panic, on a chosen device.
    pub fn try_new(level: usize, device: usize) -> Result<Self, Status> {
        if !(1..=9).contains(&level) {
            return Err(Status::InvalidParameter);
        }
        let mut h = core::ptr::null_mut();
        let rc = unsafe { bz_enc_create(&mut h, level as i32, device as i32) };
        if rc != ffi::BZ_OK {
            crate::mi355x::note_status(rc);
            return Err(Status::from_code(rc));
        }
        Ok(Self { h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) })
    }

    /// The same encoder over several GPUs of this process (`bz_enc_create_multi`): chunks of the input go round two
    /// lanes per listed device, the tail of a chunk's input crosses devices over xGMI, and the stream is the one
    /// `BZip2Encoder::new(level)` writes.  Panics like `new` for an invalid level or an empty list.
    pub fn with_devices(level: usize, devices: &[i32]) -> Self {
        if !(1..=9).contains(&level) {
            panic!("invalid level"); // src/bzip2/encoder.rs:59-61
        }
        let mut h = core::ptr::null_mut();
        let rc = unsafe { bz_enc_create_multi(&mut h, level as i32, devices.as_ptr(), devices.len() as i32) };
        if rc != ffi::BZ_OK {
            crate::mi355x::note_status(rc);
            panic!("bz2_mi355x: cannot create an encoder context: {:?}", Status::from_code(rc));
        }
        Self { h, ready: Vec::new(), pos: 0, chunk: Vec::with_capacity(CHUNK) }
    }

    /// Self-check (not in the reference, whose sequential code cannot write a stream that does not decode): every
    /// job's blocks are decoded again on the device and compared with the input they cover before their bytes are
    /// handed out (`bz_enc_set_verify`).  Returns `self` for chaining: `BZip2Encoder::new(9).verified(true)`.
    pub fn verified(self, on: bool) -> Self {
        unsafe { ffi::bz_enc_set_verify(self.h, on as i32) };
        self
    }

    /// (blocks checked, jobs redone, redone jobs that failed again, nanoseconds spent checking)
    pub fn verify_stats(&self) -> [u64; 4] {
        let mut s = [0u64; 4];
        unsafe { ffi::bz_enc_verify_stats(self.h, s.as_mut_ptr()) };
        s
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
    CompressionError::from_status(rc)
}

fn action_code(a: Action) -> i32 {
    a.code()
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

// the context is used by one caller at a time (`&mut self`), from any thread
unsafe impl Send for BZip2Encoder {}
