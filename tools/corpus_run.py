#!/usr/bin/env python3
"""One corpus of corpus.MATRIX (or "text") through bz_gpu_encode_device, for profiling runs:
tools/corpus_run.py <name> [MiB] [steps]   (rocprofv3 --kernel-trace --stats -- python3 tools/corpus_run.py binary 256 2)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import corpus
pkg = importlib.import_module("rust-compression_amd")
name = sys.argv[1]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
import numpy as np
h = corpus.matrix_corpus(name, 256 << 20)[:mib << 20] if name in corpus.MATRIX else (np.frombuffer(corpus.stress_t2(mib << 20), dtype=np.uint8).copy() if name == "t2" else None)
t = torch.from_numpy(h).cuda() if h is not None else corpus.corpus_on_device(mib << 20, torch.device("cuda", 0))
n = t.numel()
eng = pkg.GpuEngine(0, int(os.environ.get("BZ_ENGINE_BLOCKS", "400")))
cap = (pkg.encode_bound(n) + 15) & ~15
o = torch.empty(cap, dtype=torch.uint8, device="cuda")
eng.encode_device(9, t.data_ptr(), n, o.data_ptr(), cap)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    k = eng.encode_device(9, t.data_ptr(), n, o.data_ptr(), cap)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("%s %d MiB: %.2f ms = %.0f MB/s, %d rounds, stages %s" % (name, mib, dt * 1e3, n / dt / 1e6, eng.bwt_stats()["rounds"],
                                                               {a: round(b * 1e3, 2) for a, b in eng.timings().items()}))
if os.environ.get("KPROF"):
    eng.profile(True)
    eng.encode_device(9, t.data_ptr(), n, o.data_ptr(), cap)
    torch.cuda.synchronize()
    kp = eng.kernel_profile()
    for k, v in sorted(kp.items(), key=lambda kv: -kv[1]["seconds"])[:25]:
        print("  %-28s %5d launches %8.3f ms" % (k, v["launches"], v["seconds"] * 1e3))
    print("  unordered after round:", eng.bwt_stats()["unordered_after_round"])
