#!/usr/bin/env python3
"""Host buffer -> host buffer Deflate timing (df_encode_buffer), run on the GPU box: tools/df_e2e.py [MiB]"""
import ctypes, hashlib, importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    h = corpus.corpus_numpy(mib << 20)
    n = h.size
    L = pkg.lib()
    for rep in range(4):
        dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        t0 = time.perf_counter()
        rc = L.df_encode_buffer(0, 0, ctypes.cast(h.ctypes.data, ctypes.c_char_p), n, ctypes.byref(dp), ctypes.byref(dn))
        dt = time.perf_counter() - t0
        sha = hashlib.sha256(memoryview((ctypes.c_uint8 * dn.value).from_address(ctypes.addressof(dp.contents)))).hexdigest()[:16] if rc == 0 else None
        L.bz_free(dp)
        print("df_encode_buffer rc %d: %.1f ms = %.0f MB/s (%d bytes, sha %s)" % (rc, dt * 1e3, n / dt / 1e6, dn.value, sha), flush=True)


if __name__ == "__main__":
    main()
