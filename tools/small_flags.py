#!/usr/bin/env python3
"""tools/small_flags.py <file> <hipSetDeviceFlags value> [calls]: one small input through bz_encode_buffer with the
device's scheduling flag set first (0 auto, 1 spin, 2 yield, 4 blocking sync)."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flag = int(sys.argv[2])
pkg = importlib.import_module("rust-compression_amd")
L = pkg.lib()
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = ctypes.CDLL(path)
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
data = open(sys.argv[1], "rb").read()
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ts = []
for i in range(calls):
    outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    t0 = time.perf_counter()
    r = L.bz_encode_buffer(9, 0, data, len(data), ctypes.byref(outp), ctypes.byref(outn))
    assert r == 0, r
    ts.append(time.perf_counter() - t0)
    L.bz_free(outp)
ts = sorted(ts[1:])
print("flags %d (rc %d) %s: median %.2f ms, min %.2f" % (flag, rc, os.path.basename(sys.argv[1]), ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
