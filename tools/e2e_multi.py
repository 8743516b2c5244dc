#!/usr/bin/env python3
"""Host buffer -> host buffer through bz_encode_buffer_multi for several device lists (run on the GPU box):
tools/e2e_multi.py <mib> "0" "0,0" "0,0,0,0" ...   BZ_ENC_TRACE=1 prints the phases of every job."""
import ctypes, hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lists = [[int(x) for x in a.split(",")] for a in sys.argv[2:]] or [[0]]
data = corpus.corpus_bytes(mib << 20)
L = pkg.lib()
src = ctypes.c_char_p(data)
for devices in lists:
    devs = (ctypes.c_int * len(devices))(*devices)
    best, sha = None, None
    for it in range(3):
        outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        t0 = time.perf_counter()
        rc = L.bz_encode_buffer_multi(9, devs, len(devices), src, len(data), ctypes.byref(outp), ctypes.byref(outn))
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        sha = hashlib.sha256(ctypes.string_at(outp, outn.value)).hexdigest()[:16]
        L.bz_free(outp)
        if it and (best is None or dt < best):
            best = dt
    print("devices %s: %.1f ms  %.0f MB/s  sha %s" % (devices, best * 1e3, len(data) / best / 1e6, sha), flush=True)
    pkg.release_cached_resources()
