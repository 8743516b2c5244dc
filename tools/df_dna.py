"""Deflate on random text over four symbols (every chain is full, matches are short: the heaviest input for
the match kernel): parity with the oracle on 4 MiB, stage timings on 256 MiB."""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
rng = np.random.default_rng(7)
n = 256 << 20
big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]
d = big[:4 << 20].tobytes()
t0 = time.time(); want = oracle.deflate_encode(d); t1 = time.time()
got = pkg.deflate_compress(d)
print("parity on 4 MiB:", got == want, "oracle %.2f MB/s" % (len(d) / (t1 - t0) / 1e6), "ratio %.4f" % (len(got) / len(d)))
dev = torch.device("cuda", 0)
tin = torch.from_numpy(big.copy()).to(dev)
eng = pkg.GpuEngine(0, 1)
cap = pkg.deflate_bound(n)
tout = torch.zeros(cap, dtype=torch.uint8, device=dev)
for _ in range(2):
    k = eng.deflate_encode_device(0, tin.data_ptr(), n, tout.data_ptr(), cap)
print({a: round(b * 1e3, 2) for a, b in eng.deflate_timings().items()}, "MB/s %.0f" % (n / eng.deflate_timings()["total"] / 1e6))
z = bytes(tout[:k].cpu().numpy())
print("inflates:", zlib.decompress(z, -15) == big.tobytes())
