#!/usr/bin/env python3
"""CPU census of what a doubling round leaves unordered in one block, and what a round that ranks the members of SMALL
groups from the first differences of their start-order neighbours would order (the sizing behind k_link_scan /
k_group_rank).  Uses the oracle's rotation order (test infrastructure).

  tools/group_census.py <corpus name> [block index] [depths ...]

For depth h the unordered rotations are the members of maximal SA ranges whose neighbours agree on >= h symbols.  A group
of k <= KMAX members, taken in START order x_1 < ... < x_k with links (x_t, x_t+1), is *decided* when for every pair i < j
the true common prefix of rot(x_i), rot(x_j) equals the smallest link prefix between them (then T[x_i + L] != T[x_j + L]
says which is smaller)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "binary"
bi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
depths = [int(x) for x in sys.argv[3:]] or [8, 16, 32, 64]
KMAX = int(os.environ.get("KMAX", "15"))
CAP = int(os.environ.get("CAP", "65536"))

raw = corpus.matrix_corpus(name, 8 << 20).tobytes() if name in corpus.MATRIX else corpus.stress_t2(8 << 20)
rle, be, ie, crcs = oracle.rle1_blocks(raw, 9)
b0 = be[bi - 1] if bi else 0
blk = rle[b0:be[bi]]
n = len(blk)
print("%s block %d: n = %d" % (name, bi, n))
sa = np.array(oracle.bwt(blk), dtype=np.int64)
T = np.frombuffer(blk, dtype=np.uint8)
T2 = np.concatenate([T, T])
# cyclic LCP of SA neighbours (Kasai over rotations, capped at n)
rank = np.empty(n, dtype=np.int64)
rank[sa] = np.arange(n)
lcp = np.zeros(n, dtype=np.int64)
h = 0
sal = sa.tolist()
rkl = rank.tolist()
Tl = T2.tolist()
lcpl = [0] * n
for i in range(n):
    r = rkl[i]
    if r > 0:
        j = sal[r - 1]
        while h < n and Tl[i + h] == Tl[j + h]:
            h += 1
        lcpl[r] = h
        if h > 0:
            h -= 1
    else:
        h = 0
lcp = np.array(lcpl, dtype=np.int64)
for depth in depths:
    same = lcp >= depth  # same[r]: SA[r-1], SA[r] in one group
    # group boundaries
    starts = np.flatnonzero(~same)
    sizes = np.diff(np.append(starts, n))
    unordered = int(sizes[sizes > 1].sum())
    hist = {}
    for lo, hi in [(2, 2), (3, 4), (5, 8), (9, 15), (16, 64), (65, 1024), (1025, n)]:
        m = (sizes >= lo) & (sizes <= hi)
        hist["%d-%d" % (lo, hi)] = int(sizes[m].sum())
    decided = undecided = capped = 0
    left_hist = {}
    for s, k in zip(starts[(sizes > 1) & (sizes <= KMAX)].tolist(), sizes[(sizes > 1) & (sizes <= KMAX)].tolist()):
        mem = sa[s:s + k]
        order = np.argsort(mem)
        pos = s + order  # SA positions in start order
        # true lcp of pair = min lcp over SA range between
        def tl(p, q):
            a, b = (p, q) if p < q else (q, p)
            return int(lcp[a + 1:b + 1].min())
        links = [tl(pos[t], pos[t + 1]) for t in range(k - 1)]
        if max(links) - depth > CAP:
            capped += k
            continue
        ok = True
        for i in range(k):
            for j in range(i + 1, k):
                if tl(pos[i], pos[j]) != min(links[i:j]):
                    ok = False
                    break
            if not ok:
                break
        if ok:
            decided += k
        else:
            undecided += k
    print("depth %4d: unordered %7d (%.1f %%) by group size %s; groups <= %d: decided %d, undecided %d, beyond the cap %d -> left %d (%.1f %%)"
          % (depth, unordered, 100.0 * unordered / n, hist, KMAX, decided, undecided, capped, unordered - decided, 100.0 * (unordered - decided) / n))
