//! `CompressionError` (the reference's `src/error.rs:10-15`): errors are values, the codecs never
//! panic on data.
use core::fmt;

#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub enum CompressionError {
    DataError,
    UnexpectedEof,
    Unexpected,
}

impl CompressionError {
    fn text(&self) -> &'static str {
        match *self {
            CompressionError::DataError => "data integrity error in data",
            CompressionError::UnexpectedEof => "file ends unexpectedly",
            CompressionError::Unexpected => "unexpected error",
        }
    }

    /// BZ_E_* status of the C ABI -> the crate's error.  Statuses the reference has no variant for
    /// (no GPU, out of memory, a HIP failure) are `Unexpected`; `mi355x::last_status()` keeps the code.
    pub(crate) fn from_status(rc: i32) -> Self {
        crate::mi355x::note_status(rc);
        match rc {
            -1 | -4 | -5 => CompressionError::DataError,
            -2 => CompressionError::UnexpectedEof,
            _ => CompressionError::Unexpected,
        }
    }
}

impl fmt::Display for CompressionError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.write_str(self.text())
    }
}

#[cfg(feature = "std")]
impl std::error::Error for CompressionError {}
