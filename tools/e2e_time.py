#!/usr/bin/env python3
"""Host buffer -> host buffer timing of bz_encode_buffer / the streaming context (PCIe inside the clock).
usage: tools/e2e_time.py [MiB] ; BZ_ENC_CHUNK_MIB / BZ_ENC_TRACE steer the library."""
import ctypes, hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
host = corpus.corpus_bytes(mib << 20)
n = len(host)
L = pkg.lib()
pkg.compress(host[:8 << 20], 9)
for rep in range(3):
    outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    t0 = time.perf_counter()
    rc = L.bz_encode_buffer(9, 0, host, n, ctypes.byref(outp), ctypes.byref(outn))
    dt = time.perf_counter() - t0
    sha = hashlib.sha256(ctypes.string_at(outp, outn.value)).hexdigest()[:16]
    L.bz_free(outp)
    print("bz_encode_buffer rc %d: %.1f ms = %.0f MB/s  (%d bytes, sha %s)" % (rc, dt * 1e3, n / dt / 1e6, outn.value, sha), flush=True)
enc = pkg.BZip2Encoder(9)
t0 = time.perf_counter()
mv = memoryview(host)
tot = 0
for i in range(0, n, 1 << 20):
    enc.write(mv[i:i + (1 << 20)])
    tot += len(enc.read_available())
enc.end(pkg.Action.FINISH)
tot += len(enc.read_all())
dt = time.perf_counter() - t0
print("streaming 1 MiB pieces: %.1f ms = %.0f MB/s (%d bytes)" % (dt * 1e3, n / dt / 1e6, tot))

# the same through raw ctypes pointers (no Python-side copies of the pieces)
del enc  # (its engines and staging buffers go back to the library's cache for the next context)
h = ctypes.c_void_p()
assert L.bz_enc_create(ctypes.byref(h), 9, 0) == 0
base = ctypes.cast(ctypes.c_char_p(host), ctypes.c_void_p).value
L.bz_enc_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
buf = (ctypes.c_uint8 * (4 << 20))()
t0 = time.perf_counter()
tot = 0
for i in range(0, n, 1 << 20):
    assert L.bz_enc_write(h, base + i, min(1 << 20, n - i)) == 0
    while True:
        k = L.bz_enc_read(h, buf, len(buf))
        if k <= 0:
            break
        tot += k
assert L.bz_enc_end(h, 2) == 0
while True:
    k = L.bz_enc_read(h, buf, len(buf))
    if k <= 0:
        break
    tot += k
dt = time.perf_counter() - t0
L.bz_enc_destroy(h)
print("streaming 1 MiB pieces, raw pointers: %.1f ms = %.0f MB/s (%d bytes)" % (dt * 1e3, n / dt / 1e6, tot))
