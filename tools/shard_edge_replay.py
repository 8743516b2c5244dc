#!/usr/bin/env python3
"""Sharded encode, the case of a RUN ACROSS THE EDGE OF A SLAB that also closes a block (engine.hip:557).

RLE1 gives a run's count byte to the run's LAST input byte, so a slab's image can end in the middle of a chunk; if that
chunk is the one that brings a block to its 100000 * level - 19 bytes, the cut lies behind the slab.  Rule (csrc/k_rle1.hip,
"kernels H" and k_rle_cuts): a rank closes a block only with a chunk that ENDS inside its slab; the block whose closing
chunk ends behind the slab begins here and is the NEXT rank's first one (that rank codes the bytes in front of its slab
afresh, from the block's start).  Before round 5 the rank in front cut behind its slab and every rank returned BZ_E_PARAM.

Inputs: `p` alternating bytes (chunks of one byte) and then zeros (chunks of 255 bytes = 5 image bytes), level 1; (world, p,
n) chosen so that an edge 4096 * t0 falls m = 1, 2, 3, 4 or more bytes into the chunk that closes a block -- the model below
checks that for the library's own split (bz_shard_slab_tiles) and fails if a case no longer hits.  Every job is played rank
by rank on one GPU (sharded.replay_job), with windows and on the whole input; streams == oracle.encode.
BZ_CUT_TABLES=0 in the environment runs the chain kernel instead of the tables.  usage: shard_edge_replay.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("rust-compression_amd")
sharded = importlib.import_module("rust-compression_amd.sharded")
from oracle import oracle

L = 100000 - 19  # level 1 (encoder.rs:186)
CASES = [(2, 56, 30527488), (2, 471, 20307968), (2, 472, 20307968), (2, 473, 20307968),
         (3, 51, 30527488), (3, 233, 15196160), (3, 234, 15196160), (3, 235, 15196160)]


def closing_chunks(p, n):
    """input ranges [a, b) of the chunks that close a block: p one-byte chunks, then zero chunks of 255 bytes"""
    out, s, k, nz = [], 0, 0, (n - p) // 255
    while True:
        kk = max(k, -(-(s + L - p) // 5) - 1)  # first zero chunk whose end p + 5 (kk + 1) reaches s + L
        if kk >= nz:
            return out
        out.append((p + 255 * kk, p + 255 * kk + 255))
        s, k = p + 5 * (kk + 1), kk + 1


def depth_into_closing_chunk(world, p, n):
    for r in range(1, world):
        edge = sharded.slab_tiles(n, r, world)[0] * 4096
        for a, b in closing_chunks(p, n):
            if a < edge < b:
                return r, edge - a
    return None


def main():
    dev = torch.device("cuda", 0)
    eng = pkg.GpuEngine(0, 64)
    depths = []
    for world, p, n in CASES:
        hit = depth_into_closing_chunk(world, p, n)
        assert hit is not None, "case (%d, %d, %d) no longer puts a slab edge inside a closing chunk (BZ_SHARD_SKEW changed?)" % (world, p, n)
        depths.append(min(hit[1], 5))
        data = (b"ab" * (p // 2 + 1))[:p] + bytes(n - p)
        want = oracle.encode(data, 1)
        d_in = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        d_in[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        cap = (pkg.encode_bound(n) + 15) & ~15
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        for windows in (True, False):
            k, _ = sharded.replay_job(eng, 1, d_in, n, world, d_out, cap, windows=windows)
            got = bytes(d_out[:k].cpu().numpy())
            if got != want:
                print("MISMATCH: world %d p %d n %d windows %s: edge of rank %d lies %d bytes into a closing chunk" % (world, p, n, windows, hit[0], hit[1]))
                return 1
    assert set(depths) == {1, 2, 3, 4, 5}, depths
    st = eng.cut_stats()
    print("shard_edge_replay ok: %d jobs, edge depths %s; cuts %s" % (2 * len(CASES), depths, st))
    tables = os.environ.get("BZ_CUT_TABLES", "1") != "0"
    return 1 if (st["fell_back"] or (tables and not st["from_tables"])) else 0


if __name__ == "__main__":
    sys.exit(main())
