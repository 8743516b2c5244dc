#!/usr/bin/env python3
"""bench_deflate.py -- Deflate ENCODE throughput on MI355X (SURVEY.md row f-2 / BASELINE.json
configs[4]; the contract benchmark of the north-star path is bench.py).

A "step" is one pass of the path (hash chains -> matches -> parse -> blocks + Huffman tables ->
emission) over the synthetic corpus, input and stream resident in HBM.  Prints ONE JSON line.
N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench_deflate.py --gpus N): replicas only --
one 32 KiB window and one bit string run through a whole input, so the path does not shard inside a
stream (DESIGN_deflate.md); every rank encodes its own GiB as its own stream, the only collective is the
barrier / max of the timing.
"""
import argparse
import importlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--cpu-sample-mib", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="(accepted for tools/profile.sh; this bench has no extras)")
    ap.add_argument("--verify-full", action="store_true", help="inflate the whole stream with zlib and compare (slow)")
    ap.add_argument("--force-replicas", action="store_true", help="run the N > 1 code path with one rank")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("bench_deflate.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(local_rank)
    replicas = world > 1 or args.force_replicas
    if replicas:  # one independent stream per GPU: no data-path collective, only the barrier of the timing
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pkg = importlib.import_module("rust-compression_amd")
    import corpus

    n = args.mib << 20
    d_in = corpus.corpus_on_device(n * world, dev)[rank * n:(rank + 1) * n].clone() if world > 1 else corpus.corpus_on_device(n, dev)
    eng = pkg.GpuEngine(local_rank, 1)
    cap = pkg.deflate_bound(n)
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    state = {}

    def step():
        state["len"] = eng.deflate_encode_device(args.kind, d_in.data_ptr(), n, d_out.data_ptr(), cap)

    def sync():
        torch.cuda.synchronize()
        if replicas:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if replicas:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    zn = state["len"]
    stages = eng.deflate_timings()
    stats = eng.deflate_stats()
    # check outside the timed region: the head of the stream inflates to the head of the corpus
    z = bytes(d_out[:min(zn, 64 << 20)].cpu().numpy())
    do = zlib.decompressobj(-15 if args.kind == 0 else (15 if args.kind == 1 else 31))
    head = do.decompress(z, 64 << 20)
    ok = len(head) > 0 and head == bytes(d_in[:len(head)].cpu().numpy())
    # dominant kernel: k_df_match2, one launch per step (HIP events around it on the engine's stream)
    match_s = stages["matches"]
    alg = 9 * n  # 4 B sorted position + 1 B text in, 4 B match word out per position
    achieved = alg / match_s / 1e9 if match_s > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic_deflate.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("k_df_match2")
        except Exception:
            traffic = None
    result = {
        "metric": "Deflate encode MB/s (input bytes, HBM-resident in and out)",
        "value": round(n * world * args.steps / dt / 1e6, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
        "config": {"workload": "%d MiB synthetic repeating-text corpus, Inflater (window 32 KiB, chains of 255, lazy 3), "
                               "kind %d%s" % (args.mib, args.kind, ", one independent stream per GPU (replicas)" if replicas else ""),
                   "out_bytes": zn, "ratio": round(zn / n, 4)},
        "roofline": {"bound": "hbm", "kernel": "k_df_match2", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "launches": 1,
                     "avg_launch_ms": round(match_s * 1e3, 3), "algorithmic_bytes_per_launch": alg,
                     "note": "60 G candidate pairs per GiB at 16-17 vector instructions per 64 of them: bound by vector "
                             "instruction issue, not HBM (DESIGN_deflate.md)"},
        "kernel_seconds_last_step": {k: round(v, 5) for k, v in stages.items()},
        "deflate_stats": stats,
        "checks": {"head_inflates_to_input": bool(ok)},
    }
    if args.verify_full:
        do = zlib.decompressobj(-15 if args.kind == 0 else (15 if args.kind == 1 else 31))
        zall = bytes(d_out[:zn].cpu().numpy())
        host = bytes(d_in.cpu().numpy())
        pos, okf = 0, True
        for off in range(0, len(zall), 16 << 20):
            piece = do.decompress(zall[off:off + (16 << 20)])
            okf = okf and piece == host[pos:pos + len(piece)]
            pos += len(piece)
        result["checks"]["whole_stream_inflates_to_input"] = bool(okf and pos == n and do.eof)
    if not args.no_cpu_baseline:
        from oracle import oracle
        smp = min(args.cpu_sample_mib << 20, n)
        host = bytes(d_in[:smp].cpu().numpy())
        oracle.lib()
        c0 = time.perf_counter()
        ref = oracle.deflate_encode(host, args.kind)
        cdt = time.perf_counter() - c0
        d_s = torch.zeros(pkg.deflate_bound(smp), dtype=torch.uint8, device=dev)
        k = eng.deflate_encode_device(args.kind, d_in.data_ptr(), smp, d_s.data_ptr(), d_s.numel())
        result["cpu_baseline"] = {"value": round(smp / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                                  "sample": "first %d MiB of the same corpus, oracle/deflate_oracle.c (C restatement of "
                                            "the reference algorithm, single thread like the reference)" % (smp >> 20)}
        result["checks"]["gpu_equals_oracle_on_cpu_sample"] = bool(bytes(d_s[:k].cpu().numpy()) == ref)
    if replicas:
        okt = torch.tensor([1 if all(result["checks"].values()) else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        result["checks"]["every_replica_ok"] = bool(okt.item() == 1)
    if rank == 0:
        print(json.dumps(result))
    if replicas:
        dist.barrier()
        dist.destroy_process_group()
    if not all(result["checks"].values()):
        sys.exit(3)


if __name__ == "__main__":
    main()
