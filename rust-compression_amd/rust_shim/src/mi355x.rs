//! What the reference has no place for: which device, and why a call failed when the reason is not
//! one of `CompressionError`'s three variants.
use core::sync::atomic::{AtomicI32, Ordering};

use crate::ffi;

static LAST_STATUS: AtomicI32 = AtomicI32::new(0);

/// A status of the C ABI that the crate's error types cannot express.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Status {
    /// no usable gfx950 (MI355X) device or HIP runtime: this crate has no CPU path
    NoGpu,
    /// device or host memory exhausted
    NoMemory,
    /// `level` outside 1..=9 (the reference panics: `BZip2Encoder::new`)
    InvalidParameter,
    /// any other status (the raw BZ_E_* code)
    Other(i32),
}

impl Status {
    pub fn from_code(rc: i32) -> Self {
        match rc {
            ffi::BZ_E_NOGPU => Status::NoGpu,
            ffi::BZ_E_NOMEM => Status::NoMemory,
            ffi::BZ_E_PARAM => Status::InvalidParameter,
            other => Status::Other(other),
        }
    }
}

/// gfx950 devices this process can use (0: none -- every codec call will fail loudly).
pub fn device_count() -> usize {
    let n = unsafe { ffi::bz_device_count() };
    if n > 0 { n as usize } else { 0 }
}

/// The last non-zero status any codec of this process received from the library (0: none yet).
/// `CompressionError::Unexpected` from an encoder usually means "look here": e.g. `Status::NoGpu`.
pub fn last_status() -> Option<Status> {
    match LAST_STATUS.load(Ordering::Relaxed) {
        0 => None,
        rc => Some(Status::from_code(rc)),
    }
}

pub(crate) fn note_status(rc: i32) {
    if rc != 0 {
        LAST_STATUS.store(rc, Ordering::Relaxed);
    }
}
