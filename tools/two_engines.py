#!/usr/bin/env python3
"""Two engines side by side on one GPU, each over half of the corpus (HBM-resident), against one engine over all of it:
does the chip do more when the latency-bound stages of one half run beside the sort of the other?  tools/two_engines.py [MiB]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import corpus
pkg = importlib.import_module("rust-compression_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
t = corpus.corpus_on_device(mib << 20, dev)
n = t.numel()
cap = (pkg.encode_bound(n) + 15) & ~15
o = torch.empty(cap, dtype=torch.uint8, device=dev)
o2 = torch.empty(cap, dtype=torch.uint8, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


one = pkg.GpuEngine(0, 1400)
dt = timed(lambda: one.encode_device(9, t.data_ptr(), n, o.data_ptr(), cap))
print("one engine, %d MiB: %.2f ms = %.0f MB/s" % (mib, dt * 1e3, n / dt / 1e6), flush=True)
for parts in (2, 3, 4):
    engs = [pkg.GpuEngine(0, 1400 // parts + 64) for _ in range(parts)]
    outs = [torch.empty(cap // parts + 4096, dtype=torch.uint8, device=dev) for _ in range(parts)]
    per = (n // parts) & ~4095

    def side(i):
        lo = i * per
        k = per if i + 1 < parts else n - lo
        engs[i].encode_device(9, t.data_ptr() + lo, k, outs[i].data_ptr(), outs[i].numel())

    def both():
        th = [threading.Thread(target=side, args=(i,)) for i in range(parts)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    dt = timed(both)
    print("%d engines side by side, %d MiB each: %.2f ms = %.0f MB/s" % (parts, per >> 20, dt * 1e3, n / dt / 1e6), flush=True)

    # staggered: engine i starts i * (step / parts) later, so that its sort meets the others' tails
    def staggered():
        th = []
        for i in range(parts):
            x = threading.Thread(target=side, args=(i,))
            x.start()
            th.append(x)
            time.sleep(0.012)
        for x in th:
            x.join()

    dt = timed(staggered)
    print("  ... started 12 ms apart: %.2f ms = %.0f MB/s" % (dt * 1e3, n / dt / 1e6), flush=True)
    for e in engs:
        e.close()
