#!/usr/bin/env python3
"""Host buffer -> host buffer DECODE timing (bz_decode_buffer, the streaming bz_dec_*), run on the GPU box:
tools/dec_e2e.py [MiB]"""
import ctypes, hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    h = corpus.corpus_numpy(mib << 20)
    n = h.size
    L = pkg.lib()
    outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    assert L.bz_encode_buffer(9, 0, ctypes.cast(h.ctypes.data, ctypes.c_char_p), n, ctypes.byref(outp), ctypes.byref(outn)) == 0
    z = ctypes.string_at(outp, outn.value)
    L.bz_free(outp)
    want = hashlib.sha256(memoryview(h)).hexdigest()
    for rep in range(4):
        dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        t0 = time.perf_counter()
        rc = L.bz_decode_buffer(0, z, len(z), ctypes.byref(dp), ctypes.byref(dn))
        dt = time.perf_counter() - t0
        ok = rc == 0 and dn.value == n
        if rep == 0 and ok:
            ok = hashlib.sha256(memoryview((ctypes.c_uint8 * dn.value).from_address(ctypes.addressof(dp.contents)))).hexdigest() == want
        L.bz_free(dp)
        print("bz_decode_buffer rc %d ok %s: %.1f ms = %.0f MB/s (decoded bytes)" % (rc, ok, dt * 1e3, n / dt / 1e6), flush=True)
    # the streaming context: the stream written in 1 MiB pieces, the decoded bytes read in 4 MiB pieces as they come
    L.bz_dec_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.bz_dec_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.bz_dec_read.restype = ctypes.c_long
    zb = ctypes.create_string_buffer(z, len(z))
    base = ctypes.addressof(zb)
    sink = (ctypes.c_uint8 * (4 << 20))()
    for rep in range(3):
        hd = ctypes.c_void_p()
        t0 = time.perf_counter()
        assert L.bz_dec_create(ctypes.byref(hd), 0) == 0
        got = 0
        for i in range(0, len(z), 1 << 20):
            assert L.bz_dec_write(hd, base + i, min(1 << 20, len(z) - i)) == 0
            while True:
                k = L.bz_dec_read(hd, sink, len(sink))
                if k <= 0:
                    break
                got += k
        rc = L.bz_dec_end(hd)
        while True:
            k = L.bz_dec_read(hd, sink, len(sink))
            if k <= 0:
                break
            got += k
        dt = time.perf_counter() - t0
        L.bz_dec_destroy(hd)
        print("bz_dec_* streaming rc %d: %.1f ms = %.0f MB/s (%d decoded bytes)" % (rc, dt * 1e3, n / dt / 1e6, got), flush=True)


if __name__ == "__main__":
    main()
