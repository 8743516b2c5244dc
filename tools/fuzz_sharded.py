#!/usr/bin/env python3
"""Randomised parity of the SHARDED encode on one GPU: a `world`-rank bz_gpu_encode_sharded job is played rank by rank
(sharded.replay_job: every rank sees what the ranks in front of it would have sent) and rank 0's stream is compared
with the oracle's, byte for byte.  Inputs with long runs (a block covers more input than a slab holds), chunk and
block ends on and around slab edges, fewer blocks than ranks.  usage: fuzz_sharded.py [seconds] [seed]"""
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("rust-compression_amd")
sharded = importlib.import_module("rust-compression_amd.sharded")
from oracle import oracle
import corpus

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
dev = torch.device("cuda", 0)
TEXT = corpus.chapter(2, 4 << 20)


def gen():
    kind = rng.randrange(6)
    n = rng.choice([0, 1, 4095, 4096, 4097, 20000, 99981, 100500, 300000, 1_000_000, 2_500_000])
    n = max(0, n + rng.randrange(-40, 41)) if n > 100 else n
    if kind == 0:
        off = rng.randrange(0, len(TEXT) - n - 1)
        return TEXT[off:off + n]
    k = rng.choice([1, 2, 3, 5, 200])
    if kind == 1:
        return bytes(rng.randrange(k) for _ in range(min(n, 400000)))
    out = bytearray()
    lens = {2: [1, 2, 3, 4, 5, 6, 254, 255, 256, 257], 3: [255, 256, 510, 1000, 4096, 5000], 4: [1, 1, 1, 2, 4, 5, 3000, 70000],
            5: [4, 5, 255, 259, 260]}[kind]
    while len(out) < n:
        out += bytes([rng.randrange(k)]) * rng.choice(lens)
        if kind == 4 and rng.randrange(40) == 0:
            off = rng.randrange(0, len(TEXT) - 50000)
            out += TEXT[off:off + rng.randrange(1, 50000)]
    return bytes(out[:n])


eng = pkg.GpuEngine(0, 64)
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    data = gen()
    n = len(data)
    level = rng.choice([1, 1, 1, 2, 3, 9])
    world = rng.choice([2, 2, 3, 4, 5, 8])
    windows = rng.randrange(3) > 0
    want = oracle.encode(data, level)
    d_in = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    if n:
        d_in[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    cap = (pkg.encode_bound(n) + 15) & ~15
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    k, _ = sharded.replay_job(eng, level, d_in, n, world, d_out, cap, windows=windows)
    got = bytes(d_out[:k].cpu().numpy())
    if got != want:
        path = "/tmp/fuzz_sharded_fail_%d_%d.bin" % (seed, cases)
        open(path, "wb").write(data)
        print("MISMATCH: case %d seed %d: n %d level %d world %d windows %s (input kept in %s)" % (cases, seed, n, level, world, windows, path))
        sys.exit(1)
    cases += 1
st = eng.cut_stats()
print("fuzz_sharded ok: %d jobs in %.0f s (seed %d); cuts %s" % (cases, time.time() - t0, seed, st))
sys.exit(1 if st["fell_back"] else 0)
