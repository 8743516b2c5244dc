#!/usr/bin/env python3
"""Inputs far from text at full size (run on the GPU box): 1 GiB of zeros, of long runs, of 255/256-byte runs; HBM-resident
encode, the stream checked by CPython's bz2 (libbzip2) against the input, and by the library's own decoder."""
import bz2, hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module("rust-compression_amd")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 1 << 30
rng = np.random.default_rng(5)


def runs(lengths):
    out = np.empty(n, dtype=np.uint8)
    pos, v = 0, 0
    ls = rng.choice(lengths, size=n // min(lengths) + 1)
    vals = rng.integers(0, 7, size=ls.size, dtype=np.uint8)
    for L, b in zip(ls, vals):
        if pos >= n:
            break
        out[pos:pos + L] = b
        pos += L
    return out


cases = {"zeros": np.zeros(n, dtype=np.uint8),
         "runs of 255/256/257/1000": None, "runs of 4/5/6": None}
eng = pkg.GpuEngine(0, 1400)
for name in cases:
    if name == "zeros":
        h = cases[name]
    elif name.startswith("runs of 255"):
        h = runs([255, 256, 257, 1000])
    else:
        h = runs([4, 5, 6]) if n <= (256 << 20) else np.tile(runs([4, 5, 6])[:64 << 20], n // (64 << 20))
    d = torch.from_numpy(h).to(dev)
    cap = (pkg.encode_bound(n) + 15) & ~15
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    k = eng.encode_device(9, d.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = eng.encode_device(9, d.data_ptr(), n, d_out.data_ptr(), cap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    z = bytes(d_out[:k].cpu().numpy())
    t1 = time.perf_counter()
    back = bz2.decompress(z)
    ok_cpu = len(back) == n and hashlib.sha256(back).digest() == hashlib.sha256(memoryview(h)).digest()
    cpu_s = time.perf_counter() - t1
    d_dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    kk, v = eng.decode_device(d_out.data_ptr(), k, d_dec.data_ptr(), n + 64)
    ok_gpu = v == 0 and kk == n and bool(torch.equal(d_dec[:n], d))
    print("%s: %d -> %d bytes, encode %.1f ms = %.0f MB/s, stages %s, blocks %d, cuts %s; libbzip2 decodes it to the input: %s (%.1f s); own decoder: %s" % (
        name, n, k, dt * 1e3, n / dt / 1e6, {a: round(b * 1e3, 1) for a, b in eng.timings().items()}, len(eng.block_stats()), eng.cut_stats(), ok_cpu, cpu_s, ok_gpu), flush=True)
    del d, d_out, d_dec
