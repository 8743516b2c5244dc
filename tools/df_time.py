"""Stage timings of the Deflate path on the bench corpus, no checks (for timing experiments with build variants)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rust-compression_amd")
import corpus
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
dev = torch.device("cuda", 0)
d_in = corpus.corpus_on_device(n, dev)
eng = pkg.GpuEngine(0, 1)
cap = pkg.deflate_bound(n)
d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
for _ in range(3):
    k = eng.deflate_encode_device(0, d_in.data_ptr(), n, d_out.data_ptr(), cap)
print({a: round(b * 1e3, 2) for a, b in eng.deflate_timings().items()}, k)
