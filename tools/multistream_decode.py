#!/usr/bin/env python3
"""A pbzip2-style file -- every 900 000 input bytes a stream of their own, concatenated (decoder.rs:503-516 reads on
behind a stream's end) -- decoded by the library (run on the GPU box): tools/multistream_decode.py [MiB]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import corpus
pkg = importlib.import_module("rust-compression_amd")
dev = torch.device("cuda", 0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
d_in = corpus.corpus_on_device(n, dev)
eng = pkg.GpuEngine(0, 64)
piece = 900_000
cap = (pkg.encode_bound(piece) + 15) & ~15
d_o = torch.empty(cap, dtype=torch.uint8, device=dev)
parts = []
t0 = time.perf_counter()
for off in range(0, n, piece):
    k = min(piece, n - off)
    src = d_in[off:off + k].clone()  # (16-byte aligned copy)
    z = eng.encode_device(9, src.data_ptr(), k, d_o.data_ptr(), cap)
    parts.append(bytes(d_o[:z].cpu().numpy()))
zs = b"".join(parts)
print("%d streams, %d bytes, made in %.1f s" % (len(parts), len(zs), time.perf_counter() - t0), flush=True)
d_z = torch.zeros(len(zs) + 64, dtype=torch.uint8, device=dev)
d_z[:len(zs)] = torch.frombuffer(bytearray(zs), dtype=torch.uint8).to(dev)
d_dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
eng2 = pkg.GpuEngine(0, 64)
eng2.decode_device(d_z.data_ptr(), len(zs), d_dec.data_ptr(), n + 64)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    k, v = eng2.decode_device(d_z.data_ptr(), len(zs), d_dec.data_ptr(), n + 64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("decode: %.1f ms = %.0f MB/s, verdict %d, equals the input: %s, stages %s" % (
        dt * 1e3, n / dt / 1e6, v, bool(k == n and torch.equal(d_dec[:n], d_in)), {a: round(b * 1e3, 1) for a, b in eng2.decode_timings().items()}), flush=True)
