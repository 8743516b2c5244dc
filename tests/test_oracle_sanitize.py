"""The CPU oracle under AddressSanitizer + UBSan (gcc, CPU only): the restatements must not read or write out
of bounds on the inputs the parity tests lean on.  GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r'''
import ctypes as C, os, random, sys
L = C.CDLL(os.path.join(sys.argv[1], "oracle", "libbz2oracle_asan.so"))
u8p = C.POINTER(C.c_uint8)
L.bzo_encode_bound.restype = C.c_size_t
L.bzo_encode_bound.argtypes = [C.c_size_t]
L.bzo_encode_buffer.restype = C.c_long
L.bzo_encode_buffer.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L.bzo_decode_buffer.restype = C.c_long
L.dfo_encode.restype = C.c_long
L.dfo_encode.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u8p, C.c_size_t]
L.dfo_lzss_tokens.restype = C.c_size_t
L.dfo_lzss_tokens.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
rnd = random.Random(4)
words = [bytes(rnd.choice(b"abcdefgh") for _ in range(rnd.randint(1, 7))) for _ in range(40)]
cases = [b"", b"a", b"a" * 300, bytes(range(256)) * 5, b" ".join(rnd.choice(words) for _ in range(20000)),
         bytes(rnd.getrandbits(8) for _ in range(70000)), b"ab" * 40000,
         bytes((rnd.getrandbits(8) & rnd.getrandbits(8) & rnd.getrandbits(8)) for _ in range(20000))]
for d in cases:
    cap = L.bzo_encode_bound(len(d)) + 64
    out = (C.c_uint8 * cap)()
    n = L.bzo_encode_buffer(1, 0, d, len(d), out, cap, None, 0, None)
    assert n > 0
    for kind in (0, 1, 2):
        cap2 = len(d) + len(d) // 8 + 1024
        o2 = (C.c_uint8 * cap2)()
        m = L.dfo_encode(kind, d, len(d), b"xyz" if kind == 1 else None, 3 if kind == 1 else 0, o2, cap2)
        assert m > 0
    toks = (C.c_uint32 * (2 * (len(d) + 16)))()
    L.dfo_lzss_tokens(d, len(d), d[:1000], min(len(d), 1000), 1, 0x10000, 256, 3, 3, toks, len(d) + 16)
print("sanitized oracle ok")
'''


def test_oracle_under_asan_ubsan(tmp_path):
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libbz2oracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    script = tmp_path / "run.py"
    script.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
