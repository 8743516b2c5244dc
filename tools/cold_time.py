#!/usr/bin/env python3
"""First and later bz_encode_buffer calls of a fresh process on the 1 GiB corpus read from a file:
tools/cold_time.py <file> [calls]; BZ_ENC_CHUNK_MIB / BZ_ENC_TRACE steer the library."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("rust-compression_amd")
L = pkg.lib()
h = np.fromfile(sys.argv[1], dtype=np.uint8)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ts = []
for i in range(calls):
    outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    t0 = time.perf_counter()
    rc = L.bz_encode_buffer(9, 0, ctypes.cast(h.ctypes.data, ctypes.c_char_p), h.size, ctypes.byref(outp), ctypes.byref(outn))
    ts.append(time.perf_counter() - t0)
    L.bz_free(outp)
    assert rc == 0
print("BZ_ENC_CHUNK_MIB=%s: calls (ms): %s" % (os.environ.get("BZ_ENC_CHUNK_MIB"), " ".join("%.1f" % (t * 1e3) for t in ts)), flush=True)
