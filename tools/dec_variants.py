#!/usr/bin/env python3
"""Decode with the library as built (a build variant is swapped in by tools/variant_run.sh): stage and kernel times
over BZ_DEC_WALK_WGS.  usage: dec_variants.py <tag> [mib] [wgs,wgs,...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("rust-compression_amd")
import corpus
tag = sys.argv[1]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
wgs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["256"]
dev = torch.device("cuda", 0)
d_in = corpus.corpus_on_device(mib << 20, dev)
n = d_in.numel()
eng = pkg.GpuEngine(0, min(n // (90000 * int(os.environ.get('BZ_LEVEL', '9'))) + 64, 12000))
cap = (pkg.encode_bound(n) + 15) & ~15
d_z = torch.empty(cap, dtype=torch.uint8, device=dev)
zn = eng.encode_device(int(os.environ.get('BZ_LEVEL', '9')), d_in.data_ptr(), n, d_z.data_ptr(), cap)
d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
for v in wgs:
    os.environ["BZ_DEC_WALK_WGS"] = v
    eng.decode_device(d_z.data_ptr(), zn, d_out.data_ptr(), n)
    eng.profile(True)
    for rep in range(3):
        eng.decode_device(d_z.data_ptr(), zn, d_out.data_ptr(), n)
    kp = eng.kernel_profile()
    eng.profile(False)
    ok = torch.equal(d_out[:n], d_in)
    st = {k: round(x * 1e3, 2) for k, x in eng.decode_timings().items()}
    ks = {k: round(x["seconds"] * 1e3 / max(x["launches"], 1), 3) for k, x in kp.items() if k.startswith("k_dec") and x["launches"]}
    print(tag, "wgs", v, "ok" if ok else "MISMATCH", st, ks, flush=True)
