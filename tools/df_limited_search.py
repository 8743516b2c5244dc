"""Searches seeded inputs for Deflate blocks whose Huffman tree is deeper than its limit (the package-merge
path of make_table), checking parity with the oracle on the way; prints the seeds that take it.  Run on a GPU box."""
import sys, importlib, random
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
import torch
eng = pkg.GpuEngine(0, 1)
def enc(data):
    n = len(data)
    tin = torch.frombuffer(bytearray(data) + bytearray(16), dtype=torch.uint8).cuda()
    cap = pkg.deflate_bound(n)
    tout = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    k = eng.deflate_encode_device(0, tin.data_ptr(), n, tout.data_ptr(), cap)
    return bytes(tout[:k].cpu().numpy())
found = []
for seed in range(400):
    rnd = random.Random(seed)
    kind = seed % 4
    n = rnd.choice([300, 2000, 20000, 70000])
    if kind == 0:
        lam = rnd.choice([0.5, 0.7, 1.0, 1.5])
        d = bytes(min(255, int(rnd.expovariate(lam))) for _ in range(n))
    elif kind == 1:  # fibonacci-weighted symbols, shuffled
        w = [1, 1]
        while len(w) < rnd.choice([12, 18, 24]): w.append(w[-1] + w[-2])
        pool = [s for s, c in enumerate(w) for _ in range(c)]
        rnd.shuffle(pool); d = bytes(pool[:n])
    elif kind == 2:
        d = bytes(int(abs(rnd.gauss(0, rnd.choice([1, 2, 4])))) & 255 for _ in range(n))
    else:
        d = bytes((rnd.getrandbits(8) & rnd.getrandbits(8) & rnd.getrandbits(8)) for _ in range(n))
    g = enc(d)
    st = eng.deflate_stats()
    ok = g == oracle.deflate_encode(d)
    if not ok: print("MISMATCH", seed); sys.exit(1)
    if st["limited_tables"]: found.append((seed, kind, n, st["limited_tables"], st["blocks"]))
print(len(found), found[:12])
