"""CPU-side checks of the boundary: the HIP library builds for gfx950, loads, exports every
symbol include/bz2_mi355x.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, product


def _declared():
    src = open(os.path.join(ROOT, "include", "bz2_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:bz|df)_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    L = pkg.lib()
    names = _declared()
    assert len(names) >= 20
    rccl = [n for n in names if n.startswith("bz_rccl_")]  # the transport library's (it alone links librccl)
    names = [n for n in names if not n.startswith("bz_rccl_")]
    for n in names:
        assert hasattr(L, n), n
    assert sorted(pkg.EXPORTS) == names
    R = pkg.rccl_lib()
    for n in rccl:
        assert hasattr(R, n), n
    assert sorted(pkg.RCCL_EXPORTS) == rccl


def test_no_oracle_in_product():
    """The product path must not reach into oracle/ (it would void every parity claim)."""
    base = os.path.join(ROOT, "rust-compression_amd")
    for dp, _, files in os.walk(base):
        if "build" in dp.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "bz2oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_parameter_errors_without_gpu(pkg):
    L = pkg.lib()
    h = ctypes.c_void_p()
    assert L.bz_enc_create(ctypes.byref(h), 0, 0) == pkg.BZ_E_PARAM
    assert L.bz_enc_create(ctypes.byref(h), 10, 0) == pkg.BZ_E_PARAM
    assert L.bz_strerror(pkg.BZ_E_NOGPU).decode().startswith("no usable gfx950")
    assert L.bz_encode_bound(0) > 14
    with pytest.raises(ValueError):
        pkg.BZip2Encoder(0)


def test_fails_loudly_without_gpu(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.CompressionError) as ei:
        pkg.compress(b"hello", 9)
    assert ei.value.kind == "NoGpu"
    with pytest.raises(pkg.CompressionError):
        pkg.GpuEngine(0, 4)
    enc = pkg.BZip2Encoder(9)           # creating the context does not touch the GPU
    with pytest.raises(pkg.CompressionError) as ei:
        enc.write(b"hello")             # the first bytes start the upload pipeline: that needs the device
    assert ei.value.kind == "NoGpu"
    with pytest.raises(pkg.CompressionError):
        enc.end(pkg.Action.FINISH)


def test_rust_crate_ffi_matches_the_library(pkg):
    """rust_shim/ (the crate skeleton; no Rust toolchain here to build it): every extern "C" function
    its src/ffi.rs declares is exported by the library and declared in the C header with the same
    number of parameters; the crate has the reference's feature flags and prelude names."""
    base = os.path.join(ROOT, "rust-compression_amd", "rust_shim")
    ffi = open(os.path.join(base, "src", "ffi.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bz2_mi355x.h")).read(), flags=re.S)
    L = pkg.lib()
    decls = re.findall(r"pub fn ((?:bz|df)_[a-z0-9_]+)\(([^)]*)\)", ffi)
    assert len(decls) >= 18
    for name, params in decls:
        assert hasattr(L, name), name
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, hdr)
        assert m, name
        n_c = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        n_rs = 0 if not params.strip() else params.count(",") + 1
        assert n_c == n_rs, (name, n_c, n_rs)
    cargo = open(os.path.join(base, "Cargo.toml")).read()
    for feat in ("default", "all", "bzip2", "gzip", "deflate", "zlib", "std", "docs", "mi355x"):
        assert re.search(r"^%s\s*=" % feat, cargo, flags=re.M), feat
    lib_rs = open(os.path.join(base, "src", "lib.rs")).read()
    for name in ("Action", "BZip2Decoder", "BZip2Encoder", "BZip2Error", "Inflater", "GZipEncoder", "ZlibEncoder",
                 "CompressionError", "DecodeExt", "DecodeIterator", "Decoder", "EncodeExt", "EncodeIterator", "Encoder"):
        assert re.search(r"pub use [a-z0-9_:]+(::|\{[^}]*)\b%s\b" % name, lib_rs), name
    assert os.path.exists(os.path.join(base, "build.rs"))
