#!/usr/bin/env python3
"""bench.py -- BZip2 level-9 encode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (RLE1+split+CRC -> BWT -> MTF/ZLE -> Huffman -> bit
emission -> stream assembly) over the synthetic corpus, input and output resident in HBM.
N = 1: BASELINE.json configs[1] (1 GiB repeating text, level 9, one MI355X).
N > 1: configs[2] scaled weakly (1 GiB per GPU): every rank holds the corpus but works on its own
slab of it (rust-compression_amd/sharded.py: the RLE1 split is sharded by input tiles, the cut
chain is an 8-byte hand-off from rank to rank), encodes the blocks that end in its slab, and the
block bit strings are gathered to rank 0 over RCCL (torch.distributed "nccl"), which assembles the
serial stream.

Prints ONE JSON line on rank 0.  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import bz2
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib-per-gpu", type=int, default=1024, help="corpus MiB per GPU (default: the 1 GiB config)")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus", default="text", choices=["text", "t2"])
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the partition / encode_blocks / exchange / assemble path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("rust-compression_amd")  # after torch: shares its HIP runtime
    import corpus
    sharded = importlib.import_module("rust-compression_amd.sharded")

    total = args.mib_per_gpu * world << 20
    if args.corpus == "t2":
        host = corpus.stress_t2(total)
        d_in = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    else:
        d_in = corpus.corpus_on_device(total, dev)
    n = d_in.numel()
    est_blocks = n // 800000 + 8
    local_blocks = (est_blocks + world - 1) // world + 2
    eng = pkg.GpuEngine(local_rank, min(local_blocks, 1400))
    cap = (pkg.encode_bound(n) + 15) & ~15
    d_out = torch.empty(cap if rank == 0 else 16, dtype=torch.uint8, device=dev)

    state = {}

    def step_single():
        state["out_len"] = eng.encode_device(args.level, d_in.data_ptr(), n, d_out.data_ptr(), cap)

    # multi-GPU buffers
    if world > 1 or args.force_sharded:
        cap_words = pkg.encode_bound(n // world + (2 << 20)) // 4 + 4 * local_blocks + 64
        d_packed = torch.empty(cap_words, dtype=torch.int32, device=dev)
        d_all = torch.empty((world, cap_words), dtype=torch.int32, device=dev) if rank == 0 else None

    def step_multi():
        nb = sharded.partition(eng, args.level, d_in.data_ptr(), n, rank, world, dev)
        woff, blen, crc, used = eng.encode_blocks(0, 1, nb, d_packed.data_ptr(), cap_words)
        res = sharded.exchange(woff, blen, crc, d_packed, used, rank, world, dev, d_all)
        if rank == 0:
            buf, w_off, b_len, crcs = res
            out_len, _, _, _ = eng.assemble(args.level, buf.data_ptr(), w_off, b_len, crcs, d_out.data_ptr(), cap)
            state["out_len"] = out_len

    step = step_single if (world == 1 and not args.force_sharded) else step_multi

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.profile(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    kprof = eng.kernel_profile()
    stages = eng.timings()
    bstats = eng.bwt_stats()
    eng.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    result = None
    if rank == 0:
        out_len = state["out_len"]
        out = bytes(d_out[:out_len].cpu().numpy())
        # size-independent checks outside the timed region: the stream decodes, and its head is the corpus
        try:
            head = bz2.BZ2Decompressor().decompress(out[:min(len(out), 48 << 20)], 32 << 20)
            ok_head = len(head) > 0 and head == bytes(d_in[:len(head)].cpu().numpy())
        except (OSError, ValueError, EOFError):
            ok_head = False
        value = n * args.steps / dt / 1e6
        # dominant kernel by measured time
        dom = max(kprof.items(), key=lambda kv: kv[1]["seconds"])
        dname, d = dom
        avg = d["seconds"] / max(d["launches"], 1)
        achieved = d["bytes"] / d["seconds"] / 1e9 if d["seconds"] > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dname)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "launches": d["launches"], "avg_launch_ms": round(avg * 1e3, 4),
                    "algorithmic_bytes_per_launch": d["bytes"] // max(d["launches"], 1)}
        pipeline_bytes = 24 * n + out_len  # whole-pipeline algorithmic traffic
        result = {
            "metric": "BZip2 level-%d encode MB/s (input bytes, HBM-resident in and out)" % args.level,
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": ("%d MiB synthetic repeating-text corpus (16 MiB Zipf chapters), level %d, "
                                    "%d KB blocks" % (n >> 20, args.level, args.level * 100)) if args.corpus == "text"
                       else "%d MiB stress T2 (4 KiB paragraph repeated)" % (n >> 20),
                       "blocks": len(out) and (n // (args.level * 100000 - 19)) + 1, "parallelism": "input slabs x%d, blocks in stream order, RCCL gather to rank 0" % world,
                       "out_bytes": out_len, "ratio": round(out_len / n, 4)},
            "roofline": roofline,
            "pipeline_algorithmic_GBps_per_gpu": round(pipeline_bytes * args.steps / dt / 1e9 / world, 2),
            "kernel_seconds_last_step_rank0": {k: round(v, 5) for k, v in stages.items()},
            "bwt": bstats,
            "kernels": {k: {"launches": v["launches"], "ms": round(v["seconds"] * 1e3, 3),
                            "GBps": round(v["bytes"] / v["seconds"] / 1e9, 1) if v["seconds"] else 0}
                        for k, v in kprof.items()},
            "stream_sha256": hashlib.sha256(out).hexdigest(),
            "checks": {"head_decodes_to_input": bool(ok_head)},
        }
        if not args.no_cpu_baseline:
            # cpu_baseline leg: the oracle (a C restatement of the reference algorithm, 1 thread like the
            # reference) on a bounded sample of the same corpus; its output doubles as a parity check.
            from oracle import oracle
            smp = min(args.cpu_sample_mib << 20, n)
            sample = bytes(d_in[:smp].cpu().numpy())
            oracle.lib()
            c0 = time.perf_counter()
            ref = oracle.encode(sample, args.level)
            cdt = time.perf_counter() - c0
            d_s = torch.empty((pkg.encode_bound(smp) + 15) & ~15, dtype=torch.uint8, device=dev)
            k = eng.encode_device(args.level, d_in.data_ptr(), smp, d_s.data_ptr(), d_s.numel())
            same = bytes(d_s[:k].cpu().numpy()) == ref
            result["cpu_baseline"] = {"value": round(smp / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                                      "sample": "first %d MiB of the same corpus, oracle/bz2_oracle.c (C restatement "
                                                "of the reference algorithm, single thread like the reference)" % (smp >> 20),
                                      "host_cpus": os.cpu_count()}
            result["checks"]["gpu_equals_oracle_on_cpu_sample"] = bool(same)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
        if not all(result["checks"].values()):
            sys.exit(3)


if __name__ == "__main__":
    main()
