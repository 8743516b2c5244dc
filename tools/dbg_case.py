import sys, importlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
d = open(sys.argv[1], "rb").read()
lvl = 1
want = oracle.encode(d, lvl)
got = pkg.compress(d, lvl)
print("stream equal:", got == want, len(got), len(want))
rle, be, ie, crcs = oracle.rle1_blocks(d, lvl)
eng = pkg.GpuEngine(0, 16)
b0 = 0
for i, e in enumerate(be):
    blk = rle[b0:e]
    b0 = e
    sa = eng.debug_bwt(blk)
    ref = oracle.bwt(blk)
    bad = [k for k in range(len(blk)) if sa[k] != ref[k]]
    print("block", i, "n", len(blk), "rounds", eng.bwt_stats()["rounds"], "mismatching positions", len(bad), bad[:10])
    if bad:
        k = bad[0]
        print("  at", k, "got", sa[k:k+6], "want", ref[k:k+6])
        x, y = sa[k], ref[k]
        n = len(blk)
        t = blk + blk
        l = 0
        while l < n and t[x + l] == t[y + l]:
            l += 1
        print("  lcp of the two:", l, "bytes", t[x+l] if l < n else None, t[y+l] if l < n else None)
