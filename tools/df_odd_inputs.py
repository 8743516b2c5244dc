#!/usr/bin/env python3
"""Deflate on inputs far from text (run on the GPU box): random bytes, zeros, long runs, 256 MiB each, HBM-resident;
CPython's zlib inflates the stream back to the input."""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module("rust-compression_amd")
dev = torch.device("cuda", 0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
rng = np.random.default_rng(3)
cases = {"random": rng.integers(0, 256, n, dtype=np.uint8), "zeros": np.zeros(n, dtype=np.uint8),
         "runs": np.repeat(rng.integers(0, 4, n // 700 + 1, dtype=np.uint8), 700)[:n],
         "two symbols": rng.integers(0, 2, n, dtype=np.uint8)}
eng = pkg.GpuEngine(0, 8)
for name, h in cases.items():
    d = torch.from_numpy(np.ascontiguousarray(h)).to(dev)
    cap = (pkg.lib().df_encode_bound(n) + 15) & ~15
    d_o = torch.empty(cap, dtype=torch.uint8, device=dev)
    k = eng.deflate_encode_device(pkg.DEFLATE, d.data_ptr(), n, d_o.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = eng.deflate_encode_device(pkg.DEFLATE, d.data_ptr(), n, d_o.data_ptr(), cap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    z = bytes(d_o[:k].cpu().numpy())
    try:
        ok = zlib.decompress(z, -15) == h.tobytes()
    except Exception as e:  # noqa: BLE001
        ok = "zlib: %r" % (e,)
    print("%s: %d -> %d bytes, %.1f ms = %.0f MB/s, stages %s; zlib inflates it to the input: %s" % (
        name, n, k, dt * 1e3, n / dt / 1e6, {a: round(b * 1e3, 1) for a, b in eng.deflate_timings().items()}, ok), flush=True)
