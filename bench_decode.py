#!/usr/bin/env python3
"""bench_decode.py -- BZip2 DECODE throughput on MI355X (SURVEY.md row a18 / BASELINE.json
configs[3]; the contract benchmark of the north-star path is bench.py).

A "step" is one pass of the decode path (magic scan -> Huffman -> zero runs + inverse MTF ->
inverse BWT -> RLE1 undo -> CRC check) over the .bz2 stream of the synthetic corpus, compressed
stream and decoded bytes resident in HBM.  The stream is produced by this library's encoder
(bit-identical to the reference's) outside the timed region.  Prints ONE JSON line.

N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench_decode.py --gpus N): weak
scaling at 1 GiB of decoded bytes per GPU.  Rank r encodes GiB r of the corpus as its own stream;
the streams are concatenated (a multi-stream .bz2 file, src/bzip2/decoder.rs:503-516) and every
rank holds the whole file.  Timed: bz_gpu_decode_device_sharded -- rank r rebuilds the r-th share
of the blocks into its own slice; two small all-gathers over RCCL are the only traffic.  Every
rank checks its slice against the corpus at the slice's offset.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="(accepted for tools/profile.sh; this bench has no extras)")
    ap.add_argument("--corpus", default="text", choices=["text", "t2"])
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("bench_decode.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    if world > 1 or args.force_sharded:
        return main_sharded(args, rank, world, local_rank)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pkg = importlib.import_module("rust-compression_amd")
    import corpus

    total = args.mib << 20
    if args.corpus == "t2":
        d_in = torch.frombuffer(bytearray(corpus.stress_t2(total)), dtype=torch.uint8).to(dev)
    else:
        d_in = corpus.corpus_on_device(total, dev)
    n = d_in.numel()
    eng = pkg.GpuEngine(0, min(n // 800000 + 8, 1400))
    cap = (pkg.encode_bound(n) + 15) & ~15
    d_z = torch.empty(cap, dtype=torch.uint8, device=dev)
    zn = eng.encode_device(args.level, d_in.data_ptr(), n, d_z.data_ptr(), cap)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)

    state = {}

    def step():
        state["res"] = eng.decode_device(d_z.data_ptr(), zn, d_out.data_ptr(), n)

    for _ in range(args.warmup):
        step()
    eng.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out_len, verdict = state["res"]
    same = bool(verdict == 0 and out_len == n and torch.equal(d_out[:n], d_in))
    stages = eng.decode_timings()
    kprof = {k: v for k, v in eng.kernel_profile().items() if k.startswith("k_dec") and v["launches"]}
    eng.profile(False)
    # dominant kernel group by measured time (HIP events around its launches on the engine's stream)
    dname, dk = max(kprof.items(), key=lambda kv: kv[1]["seconds"])
    achieved = dk["bytes"] / dk["seconds"] / 1e9 if dk["seconds"] > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic_decode.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(dname)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "launches": dk["launches"],
                "avg_launch_ms": round(dk["seconds"] / dk["launches"] * 1e3, 4),
                "algorithmic_bytes_per_launch": dk["bytes"] // dk["launches"],
                "note": "random 4-byte loads over a 3.6 MB vector per block: bounded by 64-byte L2-miss sectors, see DESIGN_decode.md"}
    result = {
        "metric": "BZip2 decode MB/s (decoded bytes, HBM-resident in and out)",
        "value": round(n * args.steps / dt / 1e6, 2), "unit": "MB/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
        "config": {"workload": "%d MiB %s corpus, level-%d stream of %d bytes" % (n >> 20, args.corpus, args.level, zn)},
        "roofline": roofline,
        "kernels": {k: {"launches": v["launches"], "ms": round(v["seconds"] * 1e3, 3),
                        "GBps": round(v["bytes"] / v["seconds"] / 1e9, 1) if v["seconds"] else 0} for k, v in kprof.items()},
        "kernel_seconds_last_step": {k: round(v, 5) for k, v in stages.items()},
        "decode_stats": eng.decode_stats(),
        "checks": {"decoded_equals_input": same},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle
        smp = min(args.cpu_sample_mib << 20, n)
        d_s = torch.empty((pkg.encode_bound(smp) + 15) & ~15, dtype=torch.uint8, device=dev)
        k = eng.encode_device(args.level, d_in.data_ptr(), smp, d_s.data_ptr(), d_s.numel())
        z = bytes(d_s[:k].cpu().numpy())
        oracle.lib()
        c0 = time.perf_counter()
        ref, st = oracle.decode(z, smp + 1024)
        cdt = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": round(smp / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                                  "sample": "stream of the first %d MiB of the same corpus, oracle/bz2_oracle.c decoder "
                                            "restatement, single thread like the reference" % (smp >> 20)}
        result["checks"]["oracle_decodes_sample"] = bool(st == 0 and ref == bytes(d_in[:smp].cpu().numpy()))
    print(json.dumps(result))
    if not all(result["checks"].values()):
        sys.exit(3)


def main_sharded(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pkg = importlib.import_module("rust-compression_amd")
    sharded = importlib.import_module("rust-compression_amd.sharded")
    import corpus

    per = args.mib << 20
    total = per * world
    d_all = corpus.corpus_on_device(total, dev)          # the whole decoded file, for the check
    eng = pkg.GpuEngine(local_rank, min(per // 800000 + 8, 1400))
    # my GiB as one stream; all streams to every rank
    cap = (pkg.encode_bound(per) + 15) & ~15
    d_z = torch.zeros(cap, dtype=torch.uint8, device=dev)
    zn = eng.encode_device(args.level, d_all[rank * per:].data_ptr(), per, d_z.data_ptr(), cap)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([zn], dtype=torch.int64, device=dev))
    lens = [int(x.item()) for x in lens]
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, d_z)
    d_file = torch.cat([b[:k] for b, k in zip(bufs, lens)] + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    nfile = sum(lens)
    del bufs
    out_cap = 2 * per + (4 << 20)                        # a rank's share of blocks is about 1/world of the file
    d_out = torch.zeros(out_cap + 64, dtype=torch.uint8, device=dev)
    gather = sharded.allgather_bytes(rank, world, dev)
    state = {}

    def step():
        state["res"] = eng.decode_device_sharded(d_file.data_ptr(), nfile, d_out.data_ptr(), out_cap, rank, world, gather)

    def sync():
        torch.cuda.synchronize()
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
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    n_mine, off, tot, verdict = state["res"]
    ok = bool(verdict == 0 and tot == total and torch.equal(d_out[:n_mine], d_all[off:off + n_mine]))
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    stages = eng.decode_timings()
    if rank == 0:
        result = {
            "metric": "BZip2 decode MB/s (decoded bytes, HBM-resident in and out)",
            "value": round(total * args.steps / dt / 1e6, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": "%d x %d MiB text corpus, %d level-%d streams concatenated (%d bytes), blocks sharded "
                                   "over ranks by contiguous ranges" % (world, args.mib, world, args.level, nfile)},
            "kernel_seconds_last_step_rank0": {k: round(v, 5) for k, v in stages.items()},
            "decode_stats_rank0": eng.decode_stats(),
            "checks": {"every_slice_equals_corpus": bool(okt.item() == 1)},
        }
        print(json.dumps(result))
    dist.barrier()
    dist.destroy_process_group()
    if okt.item() != 1:
        sys.exit(3)


if __name__ == "__main__":
    main()
