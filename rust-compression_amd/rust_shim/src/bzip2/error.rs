//! `BZip2Error` (the reference's `src/bzip2/error.rs:5-11`) and its conversion to `CompressionError`
//! (`:44-52`: both magic variants are data errors).
use core::fmt;

use crate::error::CompressionError;

#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub enum BZip2Error {
    DataError,
    DataErrorMagicFirst,
    DataErrorMagic,
    UnexpectedEof,
    Unexpected,
}

impl BZip2Error {
    /// BZ_E_* verdict of the decode entry points -> the reference's variant
    pub(crate) fn from_status(rc: i32) -> Self {
        crate::mi355x::note_status(rc);
        match rc {
            -1 => BZip2Error::DataError,
            -4 => BZip2Error::DataErrorMagicFirst,
            -5 => BZip2Error::DataErrorMagic,
            -2 => BZip2Error::UnexpectedEof,
            _ => BZip2Error::Unexpected,
        }
    }
}

impl fmt::Display for BZip2Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.write_str(match *self {
            BZip2Error::DataError => "data integrity error in data",
            BZip2Error::DataErrorMagicFirst => "bad magic number (file not created by bzip2)",
            BZip2Error::DataErrorMagic => "trailing garbage after EOF ignored",
            BZip2Error::UnexpectedEof => "file ends unexpectedly",
            BZip2Error::Unexpected => "unexpected error",
        })
    }
}

#[cfg(feature = "std")]
impl std::error::Error for BZip2Error {}

impl From<BZip2Error> for CompressionError {
    fn from(e: BZip2Error) -> Self {
        match e {
            BZip2Error::DataError | BZip2Error::DataErrorMagicFirst | BZip2Error::DataErrorMagic => CompressionError::DataError,
            BZip2Error::UnexpectedEof => CompressionError::UnexpectedEof,
            BZip2Error::Unexpected => CompressionError::Unexpected,
        }
    }
}
