#!/usr/bin/env python3
"""bz_encode_buffer and bz_decode_buffer over a range of sizes (run on the GPU box): median of 3 warm calls each."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")


def main():
    h = corpus.corpus_numpy(1 << 30)
    L = pkg.lib()
    for mib in (1, 4, 16, 48, 64, 96, 128, 200, 256, 384, 512, 768, 1024):
        n = mib << 20
        ts, ds, z = [], [], None
        for rep in range(4):
            outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
            t0 = time.perf_counter()
            assert L.bz_encode_buffer(9, 0, ctypes.cast(h.ctypes.data, ctypes.c_char_p), n, ctypes.byref(outp), ctypes.byref(outn)) == 0
            ts.append(time.perf_counter() - t0)
            if z is None:
                z = ctypes.string_at(outp, outn.value)
            L.bz_free(outp)
        for rep in range(4):
            dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
            t0 = time.perf_counter()
            assert L.bz_decode_buffer(0, z, len(z), ctypes.byref(dp), ctypes.byref(dn)) == 0 and dn.value == n
            ds.append(time.perf_counter() - t0)
            L.bz_free(dp)
        te, td = sorted(ts[1:])[1], sorted(ds[1:])[1]
        print("%5d MiB: encode %7.1f ms = %6.0f MB/s; decode %7.1f ms = %6.0f MB/s" % (mib, te * 1e3, n / te / 1e6, td * 1e3, n / td / 1e6), flush=True)


if __name__ == "__main__":
    main()
