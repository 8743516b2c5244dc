#!/usr/bin/env python3
"""bench.py -- BZip2 level-9 encode throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (RLE1+split+CRC -> BWT -> MTF/ZLE -> Huffman -> bit
emission -> stream assembly) over the synthetic corpus, input and output resident in HBM.
N = 1: BASELINE.json configs[1] (1 GiB repeating text, level 9, one MI355X).
N > 1: configs[2] scaled weakly (1 GiB per GPU): one process per GPU; every rank holds the corpus but
works on its own slab of it (bz_gpu_encode_sharded: the RLE1 split is sharded by input tiles, the cut
chain is a 16-byte hand-off from rank to rank), encodes the blocks that end in its slab, and the block
bit strings are gathered to rank 0 over RCCL (torch.distributed "nccl" behind the library's four
transport callbacks), which assembles the serial stream.

`python3 bench.py --gpus N` starts the N rank processes itself (fresh children, before anything
touches a GPU); under `python -m torch.distributed.run ... bench.py --gpus N` the ranks already exist
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  Rank 0 prints ONE JSON line.

Besides the headline (value, roofline, cpu_baseline, per-step median / min) the line carries, outside the timed
region: end_to_end (host buffer -> host buffer through the C ABI; at N > 1 ONE process drives all N GPUs through
bz_encode_buffer_multi -- the drop-in surface's own multi-GPU path, BASELINE.json's "end-to-end" figure),
extra.decode (configs[3]; at N > 1 all ranks together) and, at N = 1, extra.deflate (configs[4]), the stress
corpus T2 and the all-cores CPU baseline; at N > 1 a second short leg over the library's own RCCL transport.

No rank waits for ever: a watchdog thread ends the process with a non-zero status when the run makes no
progress for --hang-timeout seconds (a peer died inside a collective), process groups are created with that
timeout, and the extras behind the headline have their own deadline after which rank 0 prints the headline
without them.
"""
import argparse
import ctypes
import bz2
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib-per-gpu", type=int, default=1024, help="corpus MiB per GPU (default: the 1 GiB config)")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--corpus", default="text", choices=["text", "t2"])
    ap.add_argument("--force-sharded", action="store_true", help="run bz_gpu_encode_sharded even with one rank")
    ap.add_argument("--transport", default="torch", choices=["torch", "rccl"],
                    help="N > 1: who carries the four transport callbacks of bz_gpu_encode_sharded -- torch.distributed "
                         "(backend nccl = RCCL; the default) or the library's own RCCL transport "
                         "(libbz2_mi355x_rccl.so: ncclAllGather / ncclSend / ncclRecv from C, no Python in the data path)")
    ap.add_argument("--hang-timeout", type=float, default=900.0,
                    help="seconds without progress after which a rank gives up with exit status 5")
    ap.add_argument("--extras-timeout", type=float, default=600.0,
                    help="N > 1: seconds the legs behind the headline may take before rank 0 prints the headline without them")
    ap.add_argument("--preflight-timeout", type=float, default=240.0,
                    help="N > 1: seconds the checks in front of the timed steps may take (transport self-test over the real "
                         "communicator, peer copies between the devices) before rank 0 prints a line that says where they stood")
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks share the visible GPUs (rank r -> device r mod count) and talk over gloo: lets "
                         "the N > 1 path run on a box with fewer GPUs than ranks (not a scaling measurement)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N from a plain shell: start N rank processes (this process never touches a GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:  # a rank died: its peers would wait in a collective for ever
                    q.terminate()
    sys.exit(rc)


class Watchdog:
    """Ends the process (status 5) when beat() has not been called for `timeout` seconds: a rank whose peer died
    inside a collective must not wait for ever.  A thread, because the main thread may be blocked in C."""

    def __init__(self, timeout, rank=0):
        import threading
        self.timeout, self.rank = float(timeout), rank
        self.last, self.label = time.monotonic(), "start"
        self.on_expire = None
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def beat(self, label):
        self.last, self.label = time.monotonic(), label

    def _run(self):
        while True:
            time.sleep(min(1.0, self.timeout / 4))
            if time.monotonic() - self.last > self.timeout:
                if self.on_expire is not None:
                    self.on_expire()
                sys.stderr.write("bench.py: rank %d made no progress for %.0f s in '%s': giving up\n"
                                 % (self.rank, self.timeout, self.label))
                sys.stderr.flush()
                os._exit(5)


def step_stats(times):
    """per-step wall times of the timed steps -> median / min / max in ms"""
    t = sorted(times)
    if not t:
        return None
    med = t[len(t) // 2] if len(t) % 2 else 0.5 * (t[len(t) // 2 - 1] + t[len(t) // 2])
    return {"median": round(med * 1e3, 3), "min": round(t[0] * 1e3, 3), "max": round(t[-1] * 1e3, 3), "n": len(t)}


def timed(fn, reps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def roofline_of(kprof, names, pmc):
    """roofline object of the slowest of `names` (kernels with launches) from the in-library HIP-event profile"""
    cand = {k: v for k, v in kprof.items() if k in names and v["launches"] and v["seconds"] > 0}
    if not cand:
        return None
    name, d = max(cand.items(), key=lambda kv: kv[1]["seconds"])
    achieved = d["bytes"] / d["seconds"] / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": pmc.get(name), "launches": d["launches"],
            "avg_launch_ms": round(d["seconds"] / d["launches"] * 1e3, 4),
            "algorithmic_bytes_per_launch": d["bytes"] // d["launches"]}


def kernel_sources_sha16():
    """SHA-256 (16 hex digits) of the kernel sources -- what tools/summarize_prof.py stores beside the PMC traffic it condenses."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    hh = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(here, "rust-compression_amd", "csrc", "k_*.hip")) + [os.path.join(here, "rust-compression_amd", "csrc", "bzgpu.h")]):
        with open(fn, "rb") as f:
            hh.update(f.read())
    return hh.hexdigest()[:16]


def load_json(*path):
    p = os.path.join(ROOT, *path)
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return {}


def preflight(args, torch, dist, ctl, pkg, comm, rank, world, ndev, share, native, wire, dog, config_stub):
    """N > 1, before anything is timed: the first contact with a multi-GPU node must be DIAGNOSABLE (VERDICT r4 item 6 --
    RCCL with more than one rank and peer copies between two devices have only ever run in one-GPU emulation here).
      (1) the communicator counts N ranks (ncclCommCount for the library's transport; a sum of ones over the process group
          -- on the devices when the backend is nccl -- for torch.distributed's);
      (2) bz_shard_comm_selftest over the REAL communicator with device buffers: the exchanges of bz_gpu_encode_sharded
          (all-gather, the rank-to-rank chain, the variable-length gather) with known patterns, verdict shared by all ranks;
      (3) rank 0: one hipMemcpyPeerAsync round trip per neighbour pair of the devices the end-to-end leg will drive
          (bz_peer_copy_selftest; its failure only skips that leg -- the sharded path does not use peer copies).
    A failure of (1) or (2), or no answer within --preflight-timeout, ends the run: rank 0 prints ONE JSON line with
    "preflight" saying which stage and which ranks, every rank leaves with status 4 (a plain exit of a process that has
    touched the GPU -- nothing is re-executed)."""
    import threading
    rec = {"backend": dist.get_backend(), "transport": "library RCCL" if native else "torch.distributed", "world": world,
           "devices_visible": ndev, "ranks_share_gpus": bool(share), "stage": "start"}
    try:  # (what is linked: the version string a maintainer asks for first when a collective misbehaves)
        rec["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        rec["rccl_version"] = "unknown (%r)" % (e,)
    rec["hip_version"] = getattr(torch.version, "hip", None)

    def line(status):
        out = dict(config_stub)
        out.update({"value": None, "preflight": dict(rec, status=status)})
        return json.dumps(out)

    def expired():
        if rank == 0:
            print(line("no answer within %.0f s in stage '%s'" % (args.preflight_timeout, rec["stage"])), flush=True)
        sys.stderr.write("bench.py: rank %d: preflight stood in stage '%s' for %.0f s: giving up\n" % (rank, rec["stage"], args.preflight_timeout))
        sys.stderr.flush()
        os._exit(4)
    timer = threading.Timer(args.preflight_timeout, expired)
    timer.daemon = True
    timer.start()
    mine = {"rank": rank, "errors": []}
    try:
        rec["stage"] = "communicator count"
        if native:
            mine["comm_count"] = int(comm.count())
        else:
            one = torch.ones(1, dtype=torch.int64, device=wire)
            dist.all_reduce(one)
            if wire.type == "cuda":
                torch.cuda.synchronize()
            mine["comm_count"] = int(one.item())
        dog.beat("preflight: communicator count")
        rec["stage"] = "transport self-test"
        mine["selftest_status"] = int(pkg.lib().bz_shard_comm_selftest(ctypes.byref(comm.struct), 0))
        mine["errors"] += [repr(e) for e in getattr(comm, "errors", [])]
        if os.environ.get("BZ_BENCH_PREFLIGHT_FAIL") == str(rank):  # (tests: this rank's transport reports an error)
            mine["errors"].append("injected by BZ_BENCH_PREFLIGHT_FAIL")
        dog.beat("preflight: transport self-test")
    except Exception as e:  # noqa: BLE001 -- reported in the line, whatever it is
        mine["errors"].append(repr(e))
    if rank == 0:
        rec["stage"] = "peer copies"
        devices = [r % ndev for r in range(world)] if share else list(range(world))
        peer = (ctypes.c_int * world)()
        ms = (ctypes.c_double * world)()
        try:
            prc = int(pkg.lib().bz_peer_copy_selftest((ctypes.c_int * world)(*devices), world, 1 << 20, peer, ms))
        except Exception as e:  # noqa: BLE001
            prc = -1
            mine["errors"].append(repr(e))
        # hipDeviceCanAccessPeer for ALL pairs of the visible devices (a query: no context is made on the other devices)
        try:
            rec["peer_access_matrix"] = [[1 if (i == j or torch.cuda.can_device_access_peer(i, j)) else 0 for j in range(ndev)] for i in range(ndev)]
        except Exception as e:  # noqa: BLE001
            rec["peer_access_matrix"] = "unavailable (%r)" % (e,)
        rec["peer_copies"] = {"devices": devices, "status": prc, "peer_access": list(peer), "round_trip_ms": [round(x, 3) for x in ms],
                              "note": "devices[i] -> devices[i + 1 mod N] and back, 1 MiB; peer_access 1: direct (xGMI), 0: through "
                                      "the host, -1: both on one device"}
        dog.beat("preflight: peer copies")
    rec["stage"] = "verdicts"
    every = [None] * world
    dist.all_gather_object(every, mine, group=ctl)
    timer.cancel()
    rec["ranks"] = every
    bad = [m["rank"] for m in every if m.get("comm_count") != world or m.get("selftest_status") != 0 or m["errors"]]
    rec["stage"] = "done"
    if bad:
        if rank == 0:
            print(line("FAILED on rank(s) %s" % bad), flush=True)
        dist.barrier(group=ctl)
        sys.exit(4)
    rec["status"] = "ok"
    return rec


ENC_KERNELS = ("k_radix_hist", "k_radix_scan", "k_radix_scatter", "k_group_flags", "k_group_apply", "k_last_column",
               "k_radix_scatter_lb", "k_ghist_text", "k_ghist_scan", "k_bucket_sort", "k_rank_place", "k_group_refine")
DEC_KERNELS = ("k_dec_block", "k_dec_mtf", "k_dec_tsort", "k_dec_walk_lengths", "k_dec_place", "k_dec_rle", "k_dec_crc")


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dog = Watchdog(args.hang_timeout, rank)
    import datetime
    import torch
    import torch.distributed as dist
    dog.beat("torch imported")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    share = args.share_gpu or ndev < world
    dev_index = local_rank % ndev if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    native = args.transport == "rccl" and not share
    ctl = None  # control-plane group (gloo): waits that must not keep a GPU busy, the extras' bookkeeping
    if world > 1:
        tmo = datetime.timedelta(seconds=args.hang_timeout)
        if share or native:  # (native transport: torch.distributed only carries the id, the barrier and the timing)
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
            ctl = dist.group.WORLD
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            ctl = dist.new_group(backend="gloo", timeout=tmo)
    dog.beat("process group up")

    pkg = importlib.import_module("rust-compression_amd")  # after torch: shares its HIP runtime
    import corpus
    sharded = importlib.import_module("rust-compression_amd.sharded")

    total = args.mib_per_gpu * world << 20
    n = total
    # N > 1: a rank holds only its WINDOW of the corpus (its slab, one block's worth of input in front of it, a tile
    # behind: bz_shard_window) -- not N GiB of HBM per rank for bytes it never reads
    win_off, win_bytes = pkg.shard_window(args.level, total, rank, world) if world > 1 else (0, total)
    chapters = {}  # chapters of the text corpus this rank has generated so far (a chapter is ~3 s of Python)

    def corpus_slice(off, nbytes):
        """bytes [off, off + nbytes) of the run's corpus on this rank's device"""
        if args.corpus == "t2":
            return torch.frombuffer(bytearray(corpus.t2_slice(off, nbytes)), dtype=torch.uint8).to(dev)
        for c in corpus.slice_chapters(off, nbytes):
            if c not in chapters:
                chapters[c] = corpus.chapter(c)
        return corpus.slice_on_device(off, nbytes, dev, chs=chapters)

    d_in = corpus_slice(win_off, win_bytes)  # (rank 0's window starts at byte 0; with one rank it is the whole corpus)
    est_blocks = n // 800000 + 8
    local_blocks = (est_blocks + world - 1) // world + 2
    eng = pkg.GpuEngine(dev_index, min(local_blocks, 1400))
    cap = (pkg.encode_bound(n) + 15) & ~15
    d_out = torch.empty(cap if rank == 0 else 16, dtype=torch.uint8, device=dev)
    state = {}

    def step_single():
        state["out_len"] = eng.encode_device(args.level, d_in.data_ptr(), n, d_out.data_ptr(), cap)

    multi = world > 1 or args.force_sharded
    if multi:
        if native:
            ids = [pkg.rccl_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            comm = pkg.RcclComm(ids[0], rank, world, dev_index)
        else:
            comm = sharded.TorchComm(rank, world, dev)
        cap_words = pkg.encode_bound(n // world + (48 << 20)) // 4 + 4 * local_blocks + 64
        d_packed = comm.register(torch.empty(cap_words, dtype=torch.int32, device=dev))
        gather_words = pkg.encode_bound(n) // 4 + 4 * est_blocks + 64
        d_gather = comm.register(torch.empty(gather_words, dtype=torch.int32, device=dev)) if rank == 0 else None

    def step_multi():
        k = eng.encode_sharded_window(args.level, d_in.data_ptr(), win_off, win_bytes, n, comm, d_out.data_ptr(),
                                      cap if rank == 0 else 16, packed=(d_packed.data_ptr(), cap_words),
                                      gather=(d_gather.data_ptr(), gather_words) if rank == 0 else None)
        if rank == 0:
            state["out_len"] = k

    step = step_multi if multi else step_single

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    wire = torch.device("cpu") if (share or native or world == 1) else dev
    pre = None
    if world > 1:
        pre = preflight(args, torch, dist, ctl, pkg, comm, rank, world, ndev, share, native, wire, dog,
                        {"metric": "BZip2 level-%d encode MB/s" % args.level, "unit": "MB/s", "n_gpus": world, "steps": args.steps,
                         "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "data": "synthetic"})
    die_at = os.environ.get("BZ_BENCH_DIE")  # tests: "<rank>:<step>" -- that rank dies in front of that timed step
    die_rank, die_step = (int(x) for x in die_at.split(":")) if die_at else (-1, -1)
    dog.beat("corpus and buffers ready")
    for _ in range(args.warmup):
        step()
        dog.beat("warm-up step")
    eng.profile(True)
    sync()
    step_times = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if rank == die_rank and i == die_step:
            os._exit(9)
        s0 = time.perf_counter()
        step()  # (returns with the engine's stream drained: no extra synchronisation inside the timed region)
        step_times.append(time.perf_counter() - s0)
        dog.beat("timed step %d" % i)
    sync()
    dt = time.perf_counter() - t0
    kprof = eng.kernel_profile()
    shard_t = eng.shard_timings() if multi else None
    stages = eng.timings()
    bstats = eng.bwt_stats()
    nblocks_rank = len(eng.block_stats())
    eng.profile(False)
    rank_ms = None
    if world > 1:
        # every rank's own clock over the timed steps (a straggler GPU shows in the one line the driver keeps)
        mine_t = [torch.zeros(1, dtype=torch.float64, device=wire) for _ in range(world)]
        dist.all_gather(mine_t, torch.tensor([dt], dtype=torch.float64, device=wire))
        rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in mine_t]
        tmax = torch.tensor([dt], dtype=torch.float64, device=wire)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = torch.tensor([float(nblocks_rank)], dtype=torch.float64, device=wire)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax.item())
        nblocks = int(tsum.item())
        # the serial chain across the ranks: every rank's link (cut arrives -> own cut handed on), last timed step
        links = [torch.zeros(4, dtype=torch.float64, device=wire) for _ in range(world)]
        dist.all_gather(links, torch.tensor([shard_t[k] for k in ("wait_for_cut_ms", "chain_link_ms", "gather_ms", "assemble_ms")],
                                            dtype=torch.float64, device=wire))
        shard_all = [[round(float(x), 3) for x in t.tolist()] for t in links]
    else:
        nblocks = nblocks_rank

    result = None
    if rank == 0:
        golden = load_json("tests", "golden", "corpus_hashes.json")
        pmc = load_json("profiles", "pmc_traffic.json")
        out_len = state["out_len"]
        out = bytes(d_out[:out_len].cpu().numpy())
        sha = hashlib.sha256(out).hexdigest()
        # size-independent checks outside the timed region: the stream decodes, and its head is the corpus
        try:
            head = bz2.BZ2Decompressor().decompress(out[:min(len(out), 48 << 20)], 32 << 20)
            ok_head = len(head) > 0 and head == bytes(d_in[:len(head)].cpu().numpy())
        except (OSError, ValueError, EOFError):
            ok_head = False
        checks = {"head_decodes_to_input": bool(ok_head)}
        gkey = None
        if args.level == 9 and n % (1 << 20) == 0:
            size = "%dgib" % (n >> 30) if n % (1 << 30) == 0 else "%dmib" % (n >> 20)
            gkey = "bzip2_l9_%s_%s" % (args.corpus, size)
        if gkey in golden:  # the oracle's stream for this exact corpus (tests/golden/make_corpus_hashes.py)
            checks["stream_sha_equals_oracle_golden"] = bool(golden[gkey]["sha256"] == sha and golden[gkey]["bytes"] == out_len)
        value = n * args.steps / dt / 1e6
        roofline = roofline_of(kprof, ENC_KERNELS, pmc)
        # whole-pipeline algorithmic traffic (24 N_in + N_out) over the kernel time, per GPU
        pipeline_bytes = 24 * n + out_len
        pipe = pipeline_bytes * args.steps / dt / 1e9 / world
        roofline["pipeline_8d"] = {"achieved": round(pipe, 2), "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBPS, 5),
                                   "bytes_per_input_byte": round(pipeline_bytes / n, 3),
                                   "definition": "(24 N_in + N_out) / step time, per GPU (SURVEY.md 8(d))"}
        if "__total_bytes_per_step" in pmc:
            roofline["hbm_traffic_bytes_per_input_byte"] = round(pmc["__total_bytes_per_step"] / pmc["__input_bytes"], 1)
            roofline["hbm_traffic_source"] = pmc.get("__source")
            # (the counters are a committed file, not a measurement of this run: say whether it belongs to the kernels that ran)
            roofline["hbm_traffic_is_of_these_kernels"] = bool(pmc.get("__kernel_sources_sha16") == kernel_sources_sha16())
        result = {
            "metric": "BZip2 level-%d encode MB/s (input bytes, HBM-resident in and out: the kernels' rate; value_end_to_end is "
                      "host buffer to host buffer at the C ABI)" % args.level,
            "value": round(value, 2), "value_is": "hbm_resident", "value_hbm_resident": round(value, 2),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "step_ms": step_stats(step_times),
            "ms_per_step_of_every_rank": rank_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": ("%d MiB synthetic repeating-text corpus (16 MiB Zipf chapters), level %d, "
                                    "%d KB blocks" % (n >> 20, args.level, args.level * 100)) if args.corpus == "text"
                       else "%d MiB stress T2 (4 KiB paragraph repeated)" % (n >> 20),
                       "input_bytes": n, "blocks": nblocks,
                       "parallelism": ("input slabs x%d (bz_gpu_encode_sharded), blocks in stream order, %s gather to rank 0"
                                       % (world, "gloo (ranks share GPUs: not a scaling run)" if share and world > 1
                                          else ("RCCL (library transport)" if native else "RCCL (torch.distributed)")))
                       if multi else "one engine, one GPU",
                       "out_bytes": out_len, "ratio": round(out_len / n, 4),
                       "ranks": {"world": world, "backend": (dist.get_backend() if world > 1 else None),
                                 "rccl_comm_count": (comm.count() if (multi and native) else (pre["ranks"][0]["comm_count"] if pre else None)),
                                 "rccl_comm_count_source": ("ncclCommCount" if (multi and native) else
                                                            ("sum of ones over the %s process group" % dist.get_backend() if pre else None))}},
            "roofline": roofline,
            "kernel_seconds_last_step_rank0": {k: round(v, 5) for k, v in stages.items()},
            "bwt": bstats,
            "kernels": {k: {"launches": v["launches"], "ms": round(v["seconds"] * 1e3, 3),
                            "GBps": round(v["bytes"] / v["seconds"] / 1e9, 1) if v["seconds"] else 0}
                        for k, v in kprof.items() if v["launches"]},
            "stream_sha256": sha,
            "checks": checks,
        }
        if pre is not None:
            result["preflight"] = pre
        if world > 1:
            chain = [t[1] for t in shard_all[:-1]]  # (the last rank hands nothing on)
            result["shard_chain"] = {"chain_ms_per_link": round(sum(chain) / len(chain), 3), "links_ms": chain,
                                     "last_rank_waited_ms": shard_all[-1][0], "gather_ms_rank0": shard_all[0][2],
                                     "assemble_ms_rank0": shard_all[0][3],
                                     "note": "a link = from the arrival of the cut handed over by the rank before to the hand-on of "
                                             "this rank's own: the look-ups in the tables of candidate cuts the rank filled while it "
                                             "waited (BZ_CUT_TABLES=0: left halo, tile offsets and the chain kernel); the links are the one "
                                             "serial thing across the ranks.  Ranks that SHARE a GPU wait for each other's kernels "
                                             "here: extra.shard_link_replay of the one-GPU line has the link on an idle GPU"}
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
            sample_stream = bytes(d_s[:k].cpu().numpy())
            same = sample_stream == ref
            result["cpu_baseline"] = {"value": round(smp / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                                      "sample": "first %d MiB of the same corpus, oracle/bz2_oracle.c (C restatement "
                                                "of the reference algorithm, single thread like the reference)" % (smp >> 20),
                                      "host_cpus": os.cpu_count()}
            checks["gpu_equals_oracle_on_cpu_sample"] = bool(same)
            if not args.no_extras and world == 1:
                result["cpu_baseline_all_cores"] = all_cores_baseline(oracle, sample, args.level)
            dog.beat("cpu baseline")
        if not args.no_extras and world == 1 and args.corpus == "text":
            args._beat = dog.beat
            extras(result, args, pkg, eng, torch, dev, d_in, n, d_out, out_len, golden, corpus)
    if world > 1 and not args.no_extras:
        # The legs behind the headline.  They have only ever run in one-GPU emulation here, so they get a deadline of
        # their own: when it passes, rank 0 prints the headline with what is there and every rank leaves with status 0
        # (the main watchdog would end the run with status 5 and no line at all).
        import threading
        leg = {"name": "start"}

        def give_up():
            if rank == 0:
                result.setdefault("extra", {})["unfinished"] = "leg '%s' did not finish within %.0f s" % (leg["name"], args.extras_timeout)
                print(json.dumps(result), flush=True)
            os._exit(0)
        timer = threading.Timer(args.extras_timeout, give_up)
        timer.daemon = True
        timer.start()
        dog.beat("extras")
        # (1) BASELINE.json configs[3] at this size: the stream just assembled, decoded by all ranks together
        # (bz_gpu_decode_device_sharded: every rank rebuilds its contiguous share of the blocks; three small
        # all-gathers are the only traffic) and compared with the corpus on every rank.
        leg["name"] = "sharded decode"
        try:
            dec = sharded_decode_extra(torch, dist, pkg, sharded, eng, dev, wire, rank, world, corpus_slice, n, d_out,
                                       state.get("out_len", 0))
        except Exception as e:  # (reported, never fatal for the headline)
            dec = {"error": repr(e)}
        dog.beat("sharded decode done")
        if rank == 0:
            result.setdefault("extra", {})["decode"] = dec
            if "round_trip_equals_input_on_every_rank" in dec:
                result["checks"]["decode_sharded_round_trip"] = bool(dec["round_trip_equals_input_on_every_rank"])
        # (2) the same encode over the library's own RCCL transport (libbz2_mi355x_rccl.so: ncclAllGather / ncclSend /
        # ncclRecv from C), a short leg with its own SHA check -- one GPU per rank only
        if multi and not native and not share:
            leg["name"] = "library RCCL transport"
            try:
                lib_leg = rccl_library_leg(torch, dist, ctl, pkg, eng, dev_index, rank, world, args.level, d_in, (win_off, win_bytes), n, d_out, cap,
                                           d_packed, cap_words, d_gather if rank == 0 else None, gather_words,
                                           result["stream_sha256"] if rank == 0 else None)
            except Exception as e:
                lib_leg = {"error": repr(e)}
            dog.beat("library transport done")
            if rank == 0:
                result["extra"]["rccl_library_transport"] = lib_leg
                if "stream_equals_headline_stream" in lib_leg:
                    result["checks"]["rccl_library_transport_stream"] = bool(lib_leg["stream_equals_headline_stream"])
        # (3) END TO END through the drop-in surface: ONE process (rank 0) drives all the GPUs of the run through
        # bz_encode_buffer_multi == BZip2Encoder::with_devices, host buffer -> host buffer, H2D / D2H and the xGMI
        # hand-over of chunk tails inside the clock; the other ranks wait on the host (gloo), their GPUs idle
        leg["name"] = "end to end over all devices"
        if rank == 0 and pre["peer_copies"]["status"] != 0:
            result["end_to_end"] = {"skipped": "preflight: the peer copies between the devices failed (preflight.peer_copies)"}
        elif rank == 0:
            devices = [r % ndev for r in range(world)] if share else list(range(world))
            try:
                import numpy as np
                h_all = (np.frombuffer(corpus.stress_t2(n), dtype=np.uint8) if args.corpus == "t2"
                         else corpus.corpus_numpy(n))  # pageable caller memory: the whole N GiB, on rank 0 only
                result["end_to_end"] = end_to_end_buffer(pkg, args.level, devices, h_all, n, result["stream_sha256"],
                                                         result["checks"], result["value"], repeats=3)
                result["value_end_to_end"] = result["end_to_end"]["bz_encode_buffer_multi"]
                del h_all
            except Exception as e:
                result["end_to_end"] = {"error": repr(e)}
        dist.barrier(group=ctl)
        dog.beat("end to end done")
        timer.cancel()
    if world > 1:
        dist.barrier(group=ctl)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
        if not all(result["checks"].values()):
            sys.exit(3)


def rccl_library_leg(torch, dist, ctl, pkg, eng, dev_index, rank, world, level, d_in, win, n, d_out, cap, d_packed, cap_words,
                     d_gather, gather_words, want_sha):
    """bz_gpu_encode_sharded over the library's own RCCL transport: 1 warm-up + 2 timed steps, stream compared with the
    headline's.  torch.distributed only carries the communicator id and the barriers (gloo)."""
    ids = [pkg.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0, group=ctl)
    comm = pkg.RcclComm(ids[0], rank, world, dev_index)
    st = {}

    def step():
        k = eng.encode_sharded_window(level, d_in.data_ptr(), win[0], win[1], n, comm, d_out.data_ptr(), cap if rank == 0 else 16,
                                      packed=(d_packed.data_ptr(), cap_words),
                                      gather=(d_gather.data_ptr(), gather_words) if rank == 0 else None)
        st["k"] = k

    def sync():
        torch.cuda.synchronize()
        dist.barrier(group=ctl)
    step()
    sync()
    t0 = time.perf_counter()
    for _ in range(2):
        step()
    sync()
    ldt = (time.perf_counter() - t0) / 2
    t = torch.tensor([ldt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
    count = comm.count()
    out = None
    if rank == 0:
        sha = hashlib.sha256(bytes(d_out[:st["k"]].cpu().numpy())).hexdigest()
        out = {"value": round(n / float(t.item()) / 1e6, 2), "unit": "MB/s", "ms_per_step": round(float(t.item()) * 1e3, 3),
               "steps": 2, "rccl_comm_count": count, "stream_equals_headline_stream": bool(sha == want_sha)}
    comm.close()
    return out


def end_to_end_buffer(pkg, level, devices, h_in, n, want_sha, checks, hbm_value, repeats=5):
    """Host buffer -> host buffer through bz_encode_buffer_multi over `devices` (one process): one untimed call (engines,
    pinned staging, device buffers: its time is first_call_s), then `repeats` timed ones -- median / min / max, the
    per-phase times of the median call (bz_encode_buffer_last_phases); every stream is compared with the device
    stream's SHA-256."""
    L = pkg.lib()
    src = ctypes.cast(h_in.ctypes.data, ctypes.c_char_p)
    devs = (ctypes.c_int * len(devices))(*devices)
    times, phases, ok = [], [], True
    for it in range(repeats + 1):
        outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        c0 = time.perf_counter()
        rc = L.bz_encode_buffer_multi(level, devs, len(devices), src, n, ctypes.byref(outp), ctypes.byref(outn))
        times.append(time.perf_counter() - c0)
        if rc != 0:
            raise RuntimeError("bz_encode_buffer_multi: status %d" % rc)
        phases.append(pkg.last_call_phases())
        if it in (1, repeats):  # (hashing 226 MB takes longer than encoding 1 GiB: the first and the last timed call)
            got = hashlib.sha256(memoryview((ctypes.c_uint8 * outn.value).from_address(ctypes.addressof(outp.contents)))).hexdigest()
            ok = ok and got == want_sha
        L.bz_free(outp)
    checks["end_to_end_buffer_equals_device_stream"] = bool(ok)
    timed = times[1:]
    order = sorted(range(len(timed)), key=lambda i: timed[i])
    med_i = order[len(order) // 2]
    stats = step_stats(timed)
    rate = n / (stats["median"] * 1e-3) / 1e6
    return {"unit": "MB/s", "devices": list(devices), "bz_encode_buffer_multi": round(rate, 2),
            "best_call": round(n / timed[order[0]] / 1e6, 2), "calls_ms": stats,
            "first_call_s": round(times[0], 3), "first_call_over_median": round(times[0] / (stats["median"] * 1e-3), 2),
            "fraction_of_hbm_resident_rate": round(rate / hbm_value, 3),
            "phases_ms_of_the_median_call": phases[1 + med_i],
            "note": "host buffer in -> host buffer out, ONE process over %d device(s) (two lanes each), H2D / D2H inside the "
                    "clock; pageable caller memory on both sides; median of %d calls behind one untimed call; phases are "
                    "summed over the call's jobs (ENCODE runs side by side on the lanes, SPLIT and ASSEMBLE are serial "
                    "sections)" % (len(devices), repeats)}


def sharded_decode_extra(torch, dist, pkg, sharded, eng, dev, wire, rank, world, corpus_slice, n, d_out, out_len):
    ln = torch.tensor([out_len if rank == 0 else 0], dtype=torch.int64, device=wire)
    dist.broadcast(ln, src=0)
    zlen = int(ln.item())
    d_z = torch.zeros(((zlen + 3) // 4) * 4 + 64, dtype=torch.uint8, device=dev)
    if rank == 0:
        d_z[:zlen] = d_out[:zlen]
    if wire.type == "cuda":
        dist.broadcast(d_z, src=0)
    else:  # gloo: through the host
        h = d_z.cpu()
        dist.broadcast(h, src=0)
        d_z.copy_(h)
    cap = n // world + n // (4 * world) + (64 << 20)
    d_dec = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    gather = sharded.allgather_bytes(rank, world, dev)
    res = {}

    def run():
        res["r"] = eng.decode_device_sharded(d_z.data_ptr(), zlen, d_dec.data_ptr(), cap, rank, world, gather)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    run()  # warm-up: workspace
    sync()
    t0 = time.perf_counter()
    for _ in range(2):
        run()
    sync()
    ddt = (time.perf_counter() - t0) / 2
    k, off, tot, verdict = res["r"]
    ok = verdict == 0 and tot == n and bool(torch.equal(d_dec[:k], corpus_slice(off, k)))  # (this rank's slice of the corpus)
    flag = torch.tensor([1.0 if ok else 0.0, ddt], dtype=torch.float64, device=wire)
    worst = flag.clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MIN)
    slow = flag.clone()
    dist.all_reduce(slow, op=dist.ReduceOp.MAX)
    return {"metric": "BZip2 decode MB/s (decoded bytes; every rank rebuilds its share of the blocks, HBM-resident)",
            "value": round(n / float(slow[1]) / 1e6, 2), "unit": "MB/s", "n_gpus": world,
            "ms_per_step": round(float(slow[1]) * 1e3, 3), "steps": 2,
            "round_trip_equals_input_on_every_rank": bool(float(worst[0]) == 1.0)}


def all_cores_baseline(oracle, sample, level):
    """the oracle block-parallel on all host cores -- independent 16 MiB pieces of the
    sample, one per thread at a time (ctypes releases the GIL); a throughput figure, not one stream."""
    from concurrent.futures import ThreadPoolExecutor
    threads = min(os.cpu_count() or 1, 256)
    piece = 16 << 20
    pieces = [sample[i:i + piece] for i in range(0, len(sample), piece)] or [sample]
    work = [pieces[i % len(pieces)] for i in range(max(threads, len(pieces)))]
    c0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda p: len(oracle.encode(p, level)), work))
    cdt = time.perf_counter() - c0
    return {"value": round(sum(len(p) for p in work) / cdt / 1e6, 2), "unit": "MB/s", "cores": threads, "kind": "port",
            "sample": "%d independent 16 MiB pieces of the corpus, one oracle encoder per thread (block-parallel "
                      "throughput of the reference algorithm; the pieces are separate streams)" % len(work)}


def extras(result, args, pkg, eng, torch, dev, d_in, n, d_out, out_len, golden, corpus):
    """Non-headline measurements, each outside the headline's timed region and bounded in time."""
    import numpy as np
    from oracle import oracle
    pmc_dec = load_json("profiles", "pmc_traffic_decode.json")
    pmc_df = load_json("profiles", "pmc_traffic_deflate.json")
    checks = result["checks"]

    def sync():
        torch.cuda.synchronize()

    # ---- decode (BASELINE.json configs[3]: BZip2Decoder on the stream just produced, HBM -> HBM)
    d_dec = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    eng.decode_device(d_out.data_ptr(), out_len, d_dec.data_ptr(), n + 64)  # warm-up: workspace
    eng.profile(True)
    res = {}

    def dec_step():
        res["r"] = eng.decode_device(d_out.data_ptr(), out_len, d_dec.data_ptr(), n + 64)
    ddt = timed(dec_step, 3, sync)
    kp = eng.kernel_profile()
    eng.profile(False)
    k, verdict = res["r"]
    checks["decode_round_trip_equals_input"] = bool(verdict == 0 and k == n and torch.equal(d_dec[:n], d_in))
    dec = {"metric": "BZip2 decode MB/s (decoded bytes, HBM-resident in and out)", "value": round(n / ddt / 1e6, 2),
           "unit": "MB/s", "ms_per_step": round(ddt * 1e3, 3), "steps": 3,
           "stages_s": {a: round(b, 5) for a, b in eng.decode_timings().items()},
           "roofline": roofline_of(kp, DEC_KERNELS, pmc_dec)}
    if not args.no_cpu_baseline:
        smp = min(args.cpu_sample_mib << 20, n)
        d_s = torch.empty((pkg.encode_bound(smp) + 15) & ~15, dtype=torch.uint8, device=dev)
        ks = eng.encode_device(args.level, d_in.data_ptr(), smp, d_s.data_ptr(), d_s.numel())
        sstream = bytes(d_s[:ks].cpu().numpy())
        c0 = time.perf_counter()
        back, v = oracle.decode(sstream)
        cdt = time.perf_counter() - c0
        dec["cpu_baseline"] = {"value": round(len(back) / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                               "sample": "the stream of the first %d MiB, oracle decoder (C restatement of BZip2Decoder)" % (smp >> 20)}
        kk, vv = eng.decode_device(d_s.data_ptr(), ks, d_dec.data_ptr(), n + 64)
        checks["decode_equals_oracle_on_cpu_sample"] = bool(v == 0 and vv == 0 and kk == len(back) and
                                                            bytes(d_dec[:kk].cpu().numpy()) == back)
    # the same stream host buffer -> host buffer (bz_decode_buffer: pageable caller memory, H2D / D2H inside the clock)
    z_host = d_out[:out_len].cpu().numpy()
    L = pkg.lib()
    calls, e2e_ok = [], True
    for rep in range(6):
        dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        c0 = time.perf_counter()
        rc = L.bz_decode_buffer(dev.index or 0, ctypes.cast(z_host.ctypes.data, ctypes.c_char_p), int(out_len), ctypes.byref(dp), ctypes.byref(dn))
        calls.append(time.perf_counter() - c0)
        e2e_ok = e2e_ok and rc == 0 and dn.value == n
        if rep == 0 and e2e_ok:
            back = torch.frombuffer((ctypes.c_uint8 * n).from_address(ctypes.addressof(dp.contents)), dtype=torch.uint8)
            e2e_ok = bool(torch.equal(back.to(dev), d_in[:n]))
            del back
        L.bz_free(dp)
    checks["decode_host_to_host_equals_input"] = bool(e2e_ok)
    st5 = step_stats(calls[1:])
    dec["end_to_end"] = {"bz_decode_buffer": round(n / (st5["median"] * 1e-3) / 1e6, 2), "unit": "MB/s (decoded bytes)", "calls_ms": st5,
                         "first_call_s": round(calls[0], 4),
                         "fraction_of_hbm_resident_rate": round(n / (st5["median"] * 1e-3) / 1e6 / dec["value"], 3),
                         "note": "host buffer in -> host buffer out, pageable caller memory, median of 5 calls behind one untimed call"}
    # the streaming context (bz_dec_write / bz_dec_end / bz_dec_read through raw pointers: the loop a Rust or C host runs):
    # the stream written in 1 MiB pieces, the decoded bytes read in 4 MiB pieces as they come
    wr, rd = L.bz_dec_write, L.bz_dec_read
    saved_w, saved_r, saved_rr = wr.argtypes, rd.argtypes, rd.restype
    wr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rd.restype = ctypes.c_long
    zbase = z_host.ctypes.data
    sink4 = (ctypes.c_uint8 * (4 << 20))()
    s_calls, s_ok = [], True
    for rep in range(4):
        hd = ctypes.c_void_p()
        c0 = time.perf_counter()
        s_ok = s_ok and L.bz_dec_create(ctypes.byref(hd), dev.index or 0) == 0
        got = 0
        # (the untimed first stream keeps what it reads, in the order it reads it, and is compared with the corpus: bytes mixed
        # up between the context's lanes, its sub-batch segments or recycled buffers would keep the COUNT right)
        full = np.empty(n + len(sink4), dtype=np.uint8) if rep == 0 else None
        for i in range(0, int(out_len), 1 << 20):
            s_ok = s_ok and wr(hd, zbase + i, min(1 << 20, int(out_len) - i)) == 0
            while True:
                k = rd(hd, (full.ctypes.data + min(got, n)) if full is not None else sink4, len(sink4))
                if k <= 0:
                    break
                got += k
        rc_end = L.bz_dec_end(hd)
        last = 0
        while True:
            k = rd(hd, (full.ctypes.data + min(got, n)) if full is not None else sink4, len(sink4))
            if k <= 0:
                last = k
                break
            got += k
        s_calls.append(time.perf_counter() - c0)
        L.bz_dec_destroy(hd)
        s_ok = s_ok and rc_end == 0 and last == 0 and got == n
        if full is not None:
            s_ok = s_ok and got == n and bool(torch.equal(torch.from_numpy(full[:n]).to(dev), d_in[:n]))
            del full
    wr.argtypes, rd.argtypes, rd.restype = saved_w, saved_r, saved_rr
    checks["decode_streaming_context_yields_every_byte"] = bool(s_ok)  # (count and verdicts of every stream, CONTENT of the first)
    ss = step_stats(s_calls[1:])
    dec["end_to_end"]["bz_dec_write_read"] = round(n / (ss["median"] * 1e-3) / 1e6, 2)
    dec["end_to_end"]["streaming_calls_ms"] = ss
    dec["end_to_end"]["streaming_note"] = ("bz_dec_write in 1 MiB pieces, bz_dec_read in 4 MiB pieces between the writes and behind bz_dec_end; "
                                           "median of 3 streams behind one untimed one")
    del d_dec, z_host

    # ---- Deflate (BASELINE.json configs[4]: Inflater on the same corpus, HBM -> HBM); one call takes < 2 GiB
    result["extra"] = {"decode": dec}
    if n < (1 << 31):
        dcap = (pkg.lib().df_encode_bound(n) + 15) & ~15
        d_df = torch.empty(dcap, dtype=torch.uint8, device=dev)
        eng.deflate_encode_device(pkg.DEFLATE, d_in.data_ptr(), n, d_df.data_ptr(), dcap)  # warm-up: workspace
        st = {}

        def df_step():
            st["k"] = eng.deflate_encode_device(pkg.DEFLATE, d_in.data_ptr(), n, d_df.data_ptr(), dcap)
        fdt = timed(df_step, 2, sync)
        dft = eng.deflate_timings()
        dsha = hashlib.sha256(bytes(d_df[:st["k"]].cpu().numpy())).hexdigest()
        size = "%dgib" % (n >> 30) if n % (1 << 30) == 0 else "%dmib" % (n >> 20)
        gk = "deflate_text_%s" % size
        if gk in golden:
            checks["deflate_sha_equals_oracle_golden"] = bool(golden[gk]["sha256"] == dsha and golden[gk]["bytes"] == st["k"])
        # dominant kernel: the match finder; 9 algorithmic bytes per position (sorted position in, text, match word
        # out: DESIGN_deflate.md)
        ach = 9.0 * n / dft["matches"] / 1e9 if dft["matches"] > 0 else 0.0
        df = {"metric": "Deflate (Inflater) encode MB/s (input bytes, HBM-resident in and out)",
              "value": round(n / fdt / 1e6, 2), "unit": "MB/s", "ms_per_step": round(fdt * 1e3, 3), "steps": 2,
              "out_bytes": st["k"], "ratio": round(st["k"] / n, 4), "stream_sha256": dsha,
              "stages_s": {a: round(b, 5) for a, b in dft.items()},
              "roofline": {"bound": "hbm", "kernel": "k_df_match2", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": pmc_df.get("k_df_match2"),
                           "avg_launch_ms": round(dft["matches"] * 1e3, 3), "algorithmic_bytes_per_launch": 9 * n,
                           "note": "VALU-issue-bound, not HBM-bound (DESIGN_deflate.md)"}}
        if not args.no_cpu_baseline:
            smp = min(16 << 20, n)
            sample = bytes(d_in[:smp].cpu().numpy())
            c0 = time.perf_counter()
            ref = oracle.deflate_encode(sample, 0)
            cdt = time.perf_counter() - c0
            ks = eng.deflate_encode_device(pkg.DEFLATE, d_in.data_ptr(), smp, d_df.data_ptr(), dcap)
            checks["deflate_equals_oracle_on_cpu_sample"] = bool(bytes(d_df[:ks].cpu().numpy()) == ref)
            df["cpu_baseline"] = {"value": round(smp / cdt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                                  "sample": "first %d MiB of the corpus, oracle/deflate_oracle.c (C restatement of Inflater)" % (smp >> 20)}
        # the same host buffer -> host buffer (df_encode_buffer)
        h_df = d_in[:n].cpu().numpy()
        calls, df_ok = [], True
        for rep in range(4):
            dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
            c0 = time.perf_counter()
            rc = pkg.lib().df_encode_buffer(pkg.DEFLATE, dev.index or 0, ctypes.cast(h_df.ctypes.data, ctypes.c_char_p), n, ctypes.byref(dp),
                                            ctypes.byref(dn))
            calls.append(time.perf_counter() - c0)
            df_ok = df_ok and rc == 0 and dn.value == st["k"]
            if rep == 0 and df_ok:
                df_ok = hashlib.sha256(memoryview((ctypes.c_uint8 * dn.value).from_address(ctypes.addressof(dp.contents)))).hexdigest() == dsha
            pkg.lib().bz_free(dp)
        checks["deflate_host_to_host_equals_device_stream"] = bool(df_ok)
        st3 = step_stats(calls[1:])
        df["end_to_end"] = {"df_encode_buffer": round(n / (st3["median"] * 1e-3) / 1e6, 2), "unit": "MB/s", "calls_ms": st3,
                            "first_call_s": round(calls[0], 4), "fraction_of_hbm_resident_rate": round(n / (st3["median"] * 1e-3) / 1e6 / df["value"], 3),
                            "note": "host buffer in -> host buffer out, pageable caller memory, median of 3 calls behind one untimed call"}
        del h_df
        result["extra"]["deflate"] = df
        del d_df

    # ---- stress corpus T2 (4 KiB paragraph repeated: deep LCPs, 18 doubling rounds), 256 MiB, median of three steps
    t2n = min(n, 256 << 20)
    d_t2 = torch.frombuffer(bytearray(corpus.stress_t2(t2n)), dtype=torch.uint8).to(dev)
    res2 = {}

    def t2_step():
        res2["k"] = eng.encode_device(args.level, d_t2.data_ptr(), t2n, d_out.data_ptr(), d_out.numel())
    t2_step()
    tdt = sorted(timed(t2_step, 1, sync) for _ in range(3))[1]  # (median of three: one step alone swung by 10 % from run to run)
    t2 = bytes(d_out[:res2["k"]].cpu().numpy())
    try:
        ok_t2 = bz2.decompress(t2) == bytes(d_t2.cpu().numpy())
    except (OSError, ValueError, EOFError):
        ok_t2 = False
    checks["t2_stress_decodes_to_input"] = bool(ok_t2)
    result["t2_stress"] = {"value": round(t2n / tdt / 1e6, 2), "unit": "MB/s", "mib": t2n >> 20,
                           "rounds": eng.bwt_stats()["rounds"], "out_bytes": res2["k"]}
    del d_t2
    # ... and at the headline's size, in one batch like the headline (VERDICT r4 item 2 quotes T2 at 1 GiB): median of three
    # steps behind one untimed one, the stream's SHA-256 against the oracle's golden for that corpus
    if n >= (1 << 30) and args.level == 9:
        t2g = 1 << 30
        d_t2 = torch.frombuffer(bytearray(corpus.stress_t2(t2g)), dtype=torch.uint8).to(dev)
        res3 = {}

        def t2_big():
            res3["k"] = eng.encode_device(9, d_t2.data_ptr(), t2g, d_out.data_ptr(), d_out.numel())
        t2_big()
        tds = sorted(timed(t2_big, 1, sync) for _ in range(3))
        sha_t2 = hashlib.sha256(bytes(d_out[:res3["k"]].cpu().numpy())).hexdigest()
        want_t2 = (load_json("tests", "golden", "corpus_hashes.json") or {}).get("bzip2_l9_t2_1gib")
        want_t2 = want_t2.get("sha256") if isinstance(want_t2, dict) else want_t2
        checks["t2_1gib_stream_equals_oracle_golden"] = bool(want_t2) and sha_t2 == want_t2
        result["t2_stress"]["at_1gib"] = {"value": round(t2g / tds[1] / 1e6, 2), "unit": "MB/s", "ms_per_step": round(tds[1] * 1e3, 3),
                                          "rounds": eng.bwt_stats()["rounds"], "out_bytes": res3["k"], "sha256": sha_t2}
        del d_t2

    # ---- end to end: host buffer -> host buffer through the C ABI (PCIe both ways inside the clock)
    import numpy as np
    h_in = d_in.cpu().numpy()  # pageable caller memory
    e2e = end_to_end_buffer(pkg, args.level, [dev.index], h_in, n, result["stream_sha256"], checks, result["value"])
    # the streaming context in 1 MiB pieces (bz_enc_write / bz_enc_read through raw pointers: the loop a
    # Rust or C host runs; Python-level byte objects would add a copy per piece)
    L = pkg.lib()
    write = L.bz_enc_write
    saved = write.argtypes
    write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    base = h_in.ctypes.data
    sink = (ctypes.c_uint8 * (out_len + (8 << 20)))()  # the caller's output buffer (touched: no page faults in the clock)
    ctypes.memset(sink, 0, len(sink))
    sink_addr = ctypes.addressof(sink)
    read = L.bz_enc_read
    saved_r = read.argtypes
    read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    piece = 1 << 20

    def stream_once():
        h = ctypes.c_void_p()
        assert L.bz_enc_create(ctypes.byref(h), args.level, dev.index) == 0
        c0 = time.perf_counter()
        got_n = 0
        ok = True
        for i in range(0, n, piece):
            ok = ok and write(h, base + i, min(piece, n - i)) == 0
            while True:
                k = read(h, sink_addr + got_n, len(sink) - got_n)
                if k <= 0:
                    break
                got_n += k
        ok = ok and L.bz_enc_end(h, int(pkg.Action.FINISH)) == 0
        while True:
            k = read(h, sink_addr + got_n, len(sink) - got_n)
            if k <= 0:
                break
            got_n += k
        dt = time.perf_counter() - c0
        L.bz_enc_destroy(h)
        return ok, got_n, dt

    # one untimed run (a streaming context fills whole 384 MiB chunks: larger jobs than the one-shot call's balanced
    # ones, the engines' batch workspace grows to them once), then five timed ones
    _, _, s_first = stream_once()
    s_times, s_ok = [], True
    for it in range(5):
        ok, got_n, dt = stream_once()
        s_times.append(dt)
        s_ok = s_ok and ok and got_n == out_len
    s_ok = s_ok and hashlib.sha256(memoryview(sink)[:got_n]).hexdigest() == result["stream_sha256"]
    write.argtypes = saved
    read.argtypes = saved_r
    checks["end_to_end_streaming_equals_device_stream"] = bool(s_ok)
    s_stats = step_stats(s_times)
    s_rate = n / (s_stats["median"] * 1e-3) / 1e6
    result["end_to_end"] = dict(e2e, bz_encode_buffer=e2e["bz_encode_buffer_multi"],
                                bz_enc_write_read_1MiB_pieces=round(s_rate, 2), streaming_calls_ms=s_stats,
                                streaming_first_call_s=round(s_first, 3),
                                fraction_of_hbm_resident_rate=round(max(e2e["bz_encode_buffer_multi"], s_rate) / result["value"], 3))
    result["value_end_to_end"] = e2e["bz_encode_buffer_multi"]  # SURVEY.md 8(d)'s metric: host buffer to host buffer at the C ABI
    dog_beat = getattr(args, "_beat", lambda label: None)
    dog_beat("end to end")

    # ---- cold start: the FIRST call of a fresh process (HIP runtime, code objects, every hipMalloc / hipHostMalloc of
    # the pipeline inside the clock), what a Rust caller of BZip2Encoder::new(9) pays once per process
    result["end_to_end"]["cold"] = cold_start_leg(h_in, n, args.level, dev.index, result["stream_sha256"], checks)
    dog_beat("cold start")

    # ---- small inputs: latency of one warm bz_encode_buffer call, next to the oracle on the same bytes
    result["end_to_end"]["small"] = small_inputs_leg(pkg, oracle, args, h_in, checks, dev.index)
    del sink
    dog_beat("small inputs")

    # ---- the self-check on (bz_gpu_engine_set_verify): every block decoded on the device and compared before the call returns
    eng.set_verify(True)
    v0 = eng.verify_stats()
    st = {}

    def v_step():
        st["k"] = eng.encode_device(args.level, d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel())
    v_step()
    vdt = timed(v_step, 2, sync)
    v1 = eng.verify_stats()
    eng.set_verify(False)
    vsha = hashlib.sha256(bytes(d_out[:st["k"]].cpu().numpy())).hexdigest()
    checks["verified_stream_equals_device_stream"] = bool(vsha == result["stream_sha256"])
    checks["self_check_never_fired"] = bool(v1["jobs_redone"] == v0["jobs_redone"] and v1["jobs_failed_again"] == 0)
    result["extra"]["verify"] = {"metric": "BZip2 level-%d encode MB/s with the self-check on (HBM-resident)" % args.level,
                                 "value": round(n / vdt / 1e6, 2), "unit": "MB/s", "ms_per_step": round(vdt * 1e3, 3), "steps": 2,
                                 "fraction_of_unchecked_rate": round(n / vdt / 1e6 / result["value"], 3),
                                 "blocks_checked": v1["blocks_checked"] - v0["blocks_checked"],
                                 "jobs_redone": v1["jobs_redone"] - v0["jobs_redone"],
                                 "check_ms_per_step": round((v1["nanoseconds"] - v0["nanoseconds"]) / 3 * 1e-6, 3)}
    dog_beat("self-check")

    # ---- the corpus matrix: inputs that are not Zipf text
    del h_in
    result["extra"]["corpora"] = corpora_leg(pkg, eng, torch, dev, args, d_out, golden, corpus, checks, result["value"])
    dog_beat("corpus matrix")

    # ---- a link of the cut chain between ranks, measured on a GPU that does nothing else
    result["extra"]["shard_link_replay"] = shard_link_replay_leg(pkg, eng, torch, dev, args, corpus, d_out, checks)
    dog_beat("shard link replay")


_COLD = r"""
import ctypes, hashlib, importlib, json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
t_imp = time.perf_counter()
pkg = importlib.import_module("rust-compression_amd")
L = pkg.lib()
h_in = np.fromfile(%(path)r, dtype=np.uint8)
n = h_in.size
mode, level, device = %(mode)r, %(level)d, %(device)d
out = {}
if mode == "oneshot":
    for tag in ("first", "second"):
        outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        c0 = time.perf_counter()
        rc = L.bz_encode_buffer(level, device, ctypes.cast(h_in.ctypes.data, ctypes.c_char_p), n, ctypes.byref(outp), ctypes.byref(outn))
        out[tag + "_call_s"] = round(time.perf_counter() - c0, 4)
        assert rc == 0, rc
        if tag == "first":
            out["sha"] = hashlib.sha256(memoryview((ctypes.c_uint8 * outn.value).from_address(ctypes.addressof(outp.contents)))).hexdigest()
            out["phases_ms_first_call"] = pkg.last_call_phases()
        L.bz_free(outp)
else:
    L.bz_enc_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.bz_enc_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    sink = np.zeros(n // 3 + (8 << 20), dtype=np.uint8)
    for tag in ("first", "second"):
        h = ctypes.c_void_p()
        c0 = time.perf_counter()
        assert L.bz_enc_create(ctypes.byref(h), level, device) == 0
        got, first_byte = 0, None
        for i in range(0, n, 1 << 20):
            assert L.bz_enc_write(h, h_in.ctypes.data + i, min(1 << 20, n - i)) == 0
            while True:
                k = L.bz_enc_read(h, sink.ctypes.data + got, sink.size - got)
                if k <= 0:
                    break
                if first_byte is None:
                    first_byte = time.perf_counter() - c0
                got += k
        assert L.bz_enc_end(h, 2) == 0
        while True:
            k = L.bz_enc_read(h, sink.ctypes.data + got, sink.size - got)
            if k <= 0:
                break
            got += k
        out[tag + "_call_s"] = round(time.perf_counter() - c0, 4)
        out[tag + "_first_output_byte_s"] = round(first_byte or 0.0, 4)
        if tag == "first":
            out["sha"] = hashlib.sha256(memoryview(sink)[:got]).hexdigest()
        L.bz_enc_destroy(h)
print("RESULT " + json.dumps(out))
"""


def cold_start_leg(h_in, n, level, device, want_sha, checks):
    """bz_encode_buffer and the streaming context, each as the first (and then the second) call of a FRESH process: the
    corpus travels through a file in shared memory, so the children pay nothing but the library's own start-up."""
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(shm, "bz2_mi355x_bench_%d.bin" % os.getpid())
    out = {}
    try:
        h_in[:n].tofile(path)
        for mode in ("oneshot", "streaming"):
            p = subprocess.run([sys.executable, "-c", _COLD % {"root": ROOT, "path": path, "mode": mode, "level": level,
                                                                "device": device}],
                               capture_output=True, text=True, timeout=600)
            line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                out[mode] = {"error": (p.stderr or p.stdout)[-500:]}
                checks["cold_%s_equals_device_stream" % mode] = False
                continue
            r = json.loads(line[0][7:])
            checks["cold_%s_equals_device_stream" % mode] = bool(r.pop("sha") == want_sha)
            r["first_over_second"] = round(r["first_call_s"] / r["second_call_s"], 2)
            r["first_call_MBps"] = round(n / r["first_call_s"] / 1e6, 1)
            out[mode] = r
    finally:
        if os.path.exists(path):
            os.unlink(path)
    out["note"] = ("first and second call of a fresh process (no HIP context, no cached engines): bz_encode_buffer on the whole "
                   "corpus; bz_enc_create + write in 1 MiB pieces + read")
    return out


def small_inputs_leg(pkg, oracle, args, h_in, checks, device):
    """One warm bz_encode_buffer call on small inputs (median of 11), the oracle on the same bytes beside it:
    the reference's sample1 (98 KB, one block), one level-9 block of the corpus, ten blocks."""
    L = pkg.lib()
    gold = os.path.join(ROOT, "tests", "golden", "sample1.ref")
    cases = [("sample1.ref (data/sample1.ref, 98 696 B, 1 block)", open(gold, "rb").read())] if os.path.exists(gold) else []
    cases += [("corpus, 900 000 B (1 block)", bytes(h_in[:900_000])), ("corpus, 9 000 000 B (10 blocks)", bytes(h_in[:9_000_000]))]
    out, same = [], True
    for name, data in cases:
        ts = []
        for it in range(12):
            outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
            c0 = time.perf_counter()
            rc = L.bz_encode_buffer(args.level, device, data, len(data), ctypes.byref(outp), ctypes.byref(outn))
            dt = time.perf_counter() - c0
            z = ctypes.string_at(outp, outn.value) if rc == 0 else b""
            L.bz_free(outp)
            if it:
                ts.append(dt)
        c0 = time.perf_counter()
        ref = oracle.encode(data, args.level)
        odt = time.perf_counter() - c0
        same = same and z == ref
        ts.sort()
        out.append({"input": name, "bytes": len(data), "gpu_call_us": round(ts[len(ts) // 2] * 1e6, 1),
                    "gpu_call_us_min": round(ts[0] * 1e6, 1), "oracle_us": round(odt * 1e6, 1),
                    "gpu_over_oracle": round(ts[len(ts) // 2] / odt, 3)})
    checks["small_inputs_equal_oracle"] = bool(same)
    return out


def shard_link_replay_leg(pkg, eng, torch, dev, args, corpus, d_out, checks):
    """The ranks of a 2-rank and of an 8-rank job (mib_per_gpu each) played one after the other on this GPU by
    sharded.ReplayComm: every rank sees exactly what the ranks in front of it would have sent, so its cuts, its block
    records and its timings are the real job's, taken on an idle GPU.  Per rank: ms from the call's entry until it is
    ready for the cut of the rank in front (scan, counts, offsets, image and cut tables: no rank waits for another
    there), and the LINK -- from the hop's arrival to the hand-on -- which is all that is serial across the ranks."""
    sharded = importlib.import_module("rust-compression_amd.sharded")
    out = {"note": "ranks replayed one at a time on one idle GPU; link = hop in -> hop out (cuts_ms: the table look-ups and their "
                   "copy back alone); before_the_cut_ms runs on all ranks at once in a real job"}
    per = args.mib_per_gpu << 20
    for world in (2, 8):
        n = world * per
        hist, ranks, blocks = {}, [], 0
        for rank in range(world):
            off, nbytes = pkg.shard_window(args.level, n, rank, world)
            d_win = corpus.slice_on_device(off, nbytes, dev)
            cap = d_out.numel() if rank == 0 else 16
            for _ in range(2):  # (warm, then the one that counts: buffers sized, same hop replayed)
                comm = sharded.ReplayComm(rank, world, dev, hist)
                eng.encode_sharded_window(args.level, d_win.data_ptr(), off, nbytes, n, comm, d_out.data_ptr(), cap)
                assert not comm.errors, comm.errors
            ph = eng.shard_phases()
            blocks += len(eng.block_stats())
            ranks.append({"rank": rank, "link_ms": ph["chain_link_ms"], "cuts_ms": ph["cuts_ms"],
                          "before_the_cut_ms": ph["before_the_cut_ms"], "call_ms": ph["call_ms"]})
            del d_win
        cs = eng.cut_stats()
        links = [r["link_ms"] for r in ranks[:-1]]
        out["world_%d" % world] = {"input_gib": n >> 30, "blocks": blocks, "chain_ms_per_link": round(sum(links) / len(links), 3),
                                   "max_link_ms": max(links), "ranks": ranks}
        checks["replay_world_%d_cuts_from_tables" % world] = bool(cs["fell_back"] == 0 and cs["from_tables"] > 0)
    return out


def corpora_leg(pkg, eng, torch, dev, args, d_out, golden, corpus, checks, text_value):
    """256 MiB each of the corpora of corpus.MATRIX (uniform random bytes, 4-symbol DNA with repeats, the reference's
    binary fixtures tiled, text blocks interleaved 2:1 with deep-repeat blocks, fixed-width log lines) through
    bz_gpu_encode_device: MB/s (one warm step, then two timed), BWT rounds, HBM-resident; the stream of the first 32 MiB
    against the oracle's committed golden (tests/golden/corpus_hashes.json, made by tests/golden/make_corpus_hashes.py)."""
    import numpy as np
    out = {}
    nbytes = min(256 << 20, args.mib_per_gpu << 20)
    for name in corpus.MATRIX:
        h = corpus.matrix_corpus(name, 256 << 20)[:nbytes]
        d = torch.from_numpy(h).to(dev)
        torch.cuda.synchronize()
        st = {}

        def step():
            st["k"] = eng.encode_device(args.level, d.data_ptr(), nbytes, d_out.data_ptr(), d_out.numel())
        step()
        dt = timed(step, 2, torch.cuda.synchronize)
        bst = eng.bwt_stats()
        stages = eng.timings()
        rec = {"value": round(nbytes / dt / 1e6, 2), "unit": "MB/s", "mib": nbytes >> 20, "ms_per_step": round(dt * 1e3, 3),
               "out_bytes": st["k"], "ratio": round(st["k"] / nbytes, 4), "bwt_rounds": bst["rounds"],
               "fused_fallbacks": bst["fused_fallbacks"], "over_text_rate": round(nbytes / dt / 1e6 / text_value, 3),
               "stages_s": {a: round(b, 5) for a, b in stages.items()}}
        gk = "bzip2_l9_%s_32mib" % name
        pre = 32 << 20
        if args.level == 9 and gk in golden and nbytes >= pre:
            k = eng.encode_device(9, d.data_ptr(), pre, d_out.data_ptr(), d_out.numel())
            sha = hashlib.sha256(bytes(d_out[:k].cpu().numpy())).hexdigest()
            same_input = hashlib.sha256(memoryview(h[:pre])).hexdigest() == golden[gk]["input_sha256"]
            rec["first_32mib_equals_oracle_golden"] = bool(same_input and sha == golden[gk]["sha256"] and k == golden[gk]["bytes"])
            checks["corpus_%s_equals_oracle_golden" % name] = rec["first_32mib_equals_oracle_golden"]
        out[name] = rec
        del d, h
    return out


if __name__ == "__main__":
    main()
