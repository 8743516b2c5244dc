#!/usr/bin/env python3
"""The streaming encoder fed in pieces of several sizes through raw pointers (run on the GPU box):
tools/stream_pieces.py [MiB]"""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    h = corpus.corpus_numpy(mib << 20)
    n = h.size
    L = pkg.lib()
    L.bz_enc_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.bz_enc_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    buf = (ctypes.c_uint8 * (4 << 20))()
    base = h.ctypes.data
    for piece in (4 << 10, 64 << 10, 1 << 20, 16 << 20):
        for rep in range(2):
            hd = ctypes.c_void_p()
            t0 = time.perf_counter()
            assert L.bz_enc_create(ctypes.byref(hd), 9, 0) == 0
            tot = 0
            polls = 0
            for i in range(0, n, piece):
                assert L.bz_enc_write(hd, base + i, min(piece, n - i)) == 0
                if (i // piece) % max(1, (1 << 20) // piece) == 0:  # (poll for output about once per MiB)
                    while True:
                        k = L.bz_enc_read(hd, buf, len(buf))
                        polls += 1
                        if k <= 0:
                            break
                        tot += k
            assert L.bz_enc_end(hd, 2) == 0
            while True:
                k = L.bz_enc_read(hd, buf, len(buf))
                if k <= 0:
                    break
                tot += k
            dt = time.perf_counter() - t0
            L.bz_enc_destroy(hd)
        print("pieces of %7d bytes: %.1f ms = %.0f MB/s (%d bytes out)" % (piece, dt * 1e3, n / dt / 1e6, tot), flush=True)


if __name__ == "__main__":
    main()
