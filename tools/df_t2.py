"""Deflate on the stress corpus T2 (a 4 KiB paragraph repeated: every chain is full, every match is 258 long):
parity with the oracle on 8 MiB, stage timings on 256 MiB."""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
import corpus
d = corpus.stress_t2(8 << 20)
t0 = time.time(); want = oracle.deflate_encode(d); t1 = time.time()
got = pkg.deflate_compress(d)
print("parity on 8 MiB:", got == want, "oracle %.1f MB/s" % (len(d) / (t1 - t0) / 1e6), "ratio %.4f" % (len(got) / len(d)))
n = 256 << 20
big = corpus.stress_t2(n)
dev = torch.device("cuda", 0)
tin = torch.frombuffer(bytearray(big), dtype=torch.uint8).to(dev)
eng = pkg.GpuEngine(0, 1)
cap = pkg.deflate_bound(n)
tout = torch.zeros(cap, dtype=torch.uint8, device=dev)
for _ in range(2):
    k = eng.deflate_encode_device(0, tin.data_ptr(), n, tout.data_ptr(), cap)
print({a: round(b * 1e3, 2) for a, b in eng.deflate_timings().items()}, "MB/s %.0f" % (n / eng.deflate_timings()["total"] / 1e6))
z = bytes(tout[:k].cpu().numpy())
print("inflates:", zlib.decompress(z, -15) == big)
