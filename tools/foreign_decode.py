#!/usr/bin/env python3
"""Streams made by libbzip2 (CPython's bz2) decoded by the library at size (run on the GPU box): text, random bytes and
long runs, levels 1 and 9, also two streams concatenated; bytes and verdict against the input."""
import bz2, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import corpus
pkg = importlib.import_module("rust-compression_amd")
dev = torch.device("cuda", 0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mib << 20
rng = np.random.default_rng(9)
text = bytes(corpus.corpus_numpy(n))
inputs = {"text": text, "random": bytes(rng.integers(0, 256, n // 4, dtype=np.uint8)),
          "runs": b"".join(bytes([int(b)]) * int(L) for b, L in zip(rng.integers(0, 5, 20000), rng.choice([1, 3, 4, 5, 255, 256, 3000], 20000)))}
eng = pkg.GpuEngine(0, 64)
for name, data in inputs.items():
    for level in (1, 9):
        t0 = time.perf_counter()
        z = bz2.compress(data, level)
        cs = time.perf_counter() - t0
        if name == "text" and level == 9:
            z = z + bz2.compress(data[:1 << 20], 5)  # a second stream behind it (decoder.rs:503-516)
            want = data + data[:1 << 20]
        else:
            want = data
        d_z = torch.zeros(len(z) + 64, dtype=torch.uint8, device=dev)
        d_z[:len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8).to(dev)
        d_o = torch.empty(len(want) + 64, dtype=torch.uint8, device=dev)
        eng.decode_device(d_z.data_ptr(), len(z), d_o.data_ptr(), len(want) + 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k, v = eng.decode_device(d_z.data_ptr(), len(z), d_o.data_ptr(), len(want) + 64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = v == 0 and k == len(want) and bytes(d_o[:k].cpu().numpy()) == want
        print("%s level %d: libbzip2 %d -> %d bytes in %.1f s; GPU decode %.1f ms = %.0f MB/s, equals the input: %s" % (
            name, level, len(want), len(z), cs, dt * 1e3, len(want) / dt / 1e6, ok), flush=True)
