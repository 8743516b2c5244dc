#!/usr/bin/env python3
"""One small stream through bz_decode_buffer, several times: tools/small_dec.py <file.bz2 or raw file> [calls]
(a raw file is compressed by the library first; for kernel traces and latency figures)"""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rust-compression_amd")
data = open(sys.argv[1], "rb").read()
z = data if data[:3] == b"BZh" else pkg.compress(data, 9)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    out = pkg.decompress(z)
    ts.append(time.perf_counter() - t0)
print("%s: %d -> %d bytes, calls (ms): %s" % (os.path.basename(sys.argv[1]), len(z), len(out[0]), " ".join("%.2f" % (t * 1e3) for t in ts)))
