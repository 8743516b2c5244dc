//! The C ABI of libbz2_mi355x.so (include/bz2_mi355x.h), as far as the codec types need it.
//! Linking is set up by build.rs.
#![allow(non_camel_case_types)]
use core::ffi::c_void;

pub type bz_enc = c_void;
pub type bz_dec = c_void;
pub type df_enc = c_void;

pub const BZ_OK: i32 = 0;
pub const BZ_E_PARAM: i32 = -6; // invalid level: the reference panics (src/bzip2/encoder.rs:59-61)
pub const BZ_E_NOGPU: i32 = -7; // no gfx950 device / HIP runtime unusable: the path has no CPU fallback
pub const BZ_E_NOMEM: i32 = -8;

pub const DF_KIND_DEFLATE: i32 = 0;
pub const DF_KIND_ZLIB: i32 = 1;
pub const DF_KIND_GZIP: i32 = 2;

extern "C" {
    pub fn bz_device_count() -> i32;
    pub fn bz_strerror(code: i32) -> *const core::ffi::c_char;

    // section 1: BZip2Encoder (src/bzip2/encoder.rs:40-159)
    pub fn bz_enc_create(out: *mut *mut bz_enc, level: i32, device: i32) -> i32;
    pub fn bz_enc_create_multi(out: *mut *mut bz_enc, level: i32, devices: *const i32, n_devices: i32) -> i32;
    pub fn bz_enc_write(e: *mut bz_enc, data: *const u8, n: usize) -> i32;
    pub fn bz_enc_end(e: *mut bz_enc, action: i32) -> i32;
    pub fn bz_enc_read(e: *mut bz_enc, out: *mut u8, cap: usize) -> isize;
    pub fn bz_enc_pending(e: *const bz_enc) -> usize;
    pub fn bz_enc_destroy(e: *mut bz_enc);
    pub fn bz_enc_set_verify(e: *mut bz_enc, on: i32) -> i32;
    pub fn bz_enc_verify_stats(e: *mut bz_enc, out: *mut u64) -> i32;

    // section 3: BZip2Decoder (src/bzip2/decoder.rs:583-612)
    pub fn bz_dec_create(out: *mut *mut bz_dec, device: i32) -> i32;
    pub fn bz_dec_write(d: *mut bz_dec, data: *const u8, n: usize) -> i32;
    pub fn bz_dec_end(d: *mut bz_dec) -> i32;
    pub fn bz_dec_read(d: *mut bz_dec, out: *mut u8, cap: usize) -> isize;
    pub fn bz_dec_destroy(d: *mut bz_dec);

    // section 4: Inflater / ZlibEncoder / GZipEncoder
    pub fn df_enc_create(out: *mut *mut df_enc, kind: i32, device: i32) -> i32;
    pub fn df_enc_create_dict(out: *mut *mut df_enc, kind: i32, device: i32, dict: *const u8, dict_len: usize) -> i32;
    pub fn df_enc_write(e: *mut df_enc, data: *const u8, n: usize) -> i32;
    pub fn df_enc_end(e: *mut df_enc, action: i32) -> i32;
    pub fn df_enc_finished(e: *const df_enc) -> i32;
    pub fn df_enc_read(e: *mut df_enc, out: *mut u8, cap: usize) -> isize;
    pub fn df_enc_destroy(e: *mut df_enc);
}
