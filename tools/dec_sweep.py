#!/usr/bin/env python3
"""Decode-stage timing sweep over environment knobs (run on the GPU box)."""
import importlib, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("rust-compression_amd")
import corpus
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
d_in = corpus.corpus_on_device(mib << 20, dev)
n = d_in.numel()
eng = pkg.GpuEngine(0, min(n // 800000 + 8, 1400))
cap = (pkg.encode_bound(n) + 15) & ~15
d_z = torch.empty(cap, dtype=torch.uint8, device=dev)
zn = eng.encode_device(9, d_in.data_ptr(), n, d_z.data_ptr(), cap)
d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
for knob, vals in (("BZ_DEC_WALK_WGS", sys.argv[2].split(",") if len(sys.argv) > 2 else ["1024"]),):
    for v in vals:
        os.environ[knob] = v
        for rep in range(2):
            r = eng.decode_device(d_z.data_ptr(), zn, d_out.data_ptr(), n)
        ok = torch.equal(d_out[:n], d_in)
        print(knob, v, r, ok, {k: round(x * 1e3, 2) for k, x in eng.decode_timings().items()}, flush=True)
