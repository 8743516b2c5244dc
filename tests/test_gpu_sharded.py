"""The multi-GPU encode path in REAL processes: `world` ranks, each with its own engine and HIP
context, share the one GPU of the test box and run bz_gpu_encode_sharded end to end -- slab split,
all-gather of the slab run starts, the send/recv cut chain, block encode, the variable-length
gather, assembly on rank 0 -- with torch.distributed (gloo: two ranks cannot share a device under
RCCL) carrying the four C callbacks.  Rank 0's stream must equal the oracle's, byte for byte, and
the single-engine stream.  bench.py --gpus N runs the same code over RCCL, one GPU per rank."""
import importlib
import os
import random
import socket
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _text(n, seed):
    import corpus
    return corpus.chapter(seed, max(n, 1 << 16))[:n]


def _input(kind):
    if kind == "mixed":
        rng = random.Random(77)
        runs = b"".join(bytes([rng.randrange(3)]) * rng.randint(1, 900) for _ in range(1500))
        return _text(300_000, 7) + runs + _text(260_000, 8) + b"z" * 70_000 + _text(100_001, 9), 1
    if kind == "text9":
        return _text(5_000_000, 3), 9
    if kind == "runs":  # long runs: a level-1 block covers megabytes of input, more than some slabs hold
        rng = random.Random(5)
        runs = b"".join(bytes([rng.randrange(2)]) * rng.choice([1, 2, 3, 4, 5, 254, 255, 256, 700, 3000]) for _ in range(9000))
        return runs + _text(150_000, 4) + runs[:1_000_000], 1
    if kind == "tiny":
        return b"abc" * 1000, 9      # fewer blocks than ranks: some ranks own none
    raise KeyError(kind)


def _worker(rank, world, port, kind, tables, q):
    sys.path.insert(0, ROOT)
    os.environ["BZ_CUT_TABLES"] = "1" if tables else "0"  # (read by the library once, at its first partition)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("rust-compression_amd")
        sharded = importlib.import_module("rust-compression_amd.sharded")
        dev = torch.device("cuda", 0)
        data, level = _input(kind)
        n = len(data)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        eng = pkg.GpuEngine(0, 8)
        comm = sharded.TorchComm(rank, world, dev)
        cap = (pkg.encode_bound(n) + 15) & ~15
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        # once with the engine's own buffers (pointers the transport has never seen), once with
        # registered caller-owned ones
        k1 = eng.encode_sharded(level, d_in.data_ptr(), n, comm, d_out.data_ptr(), cap)
        out1 = bytes(d_out[:k1].cpu().numpy())
        words = cap // 4 + 64
        packed = comm.register(torch.empty(words, dtype=torch.int32, device=dev))
        gather = comm.register(torch.empty(words * world, dtype=torch.int32, device=dev)) if rank == 0 else None
        k2 = eng.encode_sharded(level, d_in.data_ptr(), n, comm, d_out.data_ptr(), cap,
                                packed=(packed.data_ptr(), words),
                                gather=(gather.data_ptr(), words * world) if rank == 0 else None)
        out2 = bytes(d_out[:k2].cpu().numpy())
        single = None
        if rank == 0:
            ks = eng.encode_device(level, d_in.data_ptr(), n, d_out.data_ptr(), cap)
            single = bytes(d_out[:ks].cpu().numpy())
        q.put((rank, out1, out2, single, list(comm.errors), eng.cut_stats()))
        eng.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,tables", [(2, "mixed", 1), (3, "mixed", 1), (2, "text9", 1), (4, "tiny", 1), (3, "runs", 1),
                                               (4, "runs", 1), (3, "mixed", 0), (3, "runs", 0)])
def test_sharded_encode_in_real_processes(oracle, world, kind, tables):
    """tables = 1: every rank's cuts come from its tables of candidate cuts, filled before the cut of the rank in front
    arrives (k_rle1.hip "kernels H"); 0: from the chain kernel, started when it arrives (BZ_CUT_TABLES=0)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, tables, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    data, level = _input(kind)
    want = oracle.encode(data, level)
    assert [g[4] for g in got] == [[]] * world
    assert got[0][1] == want and got[0][2] == want and got[0][3] == want
    for r in range(1, world):
        assert got[r][1] == b"" and got[r][2] == b""
    for r in range(world):
        st = got[r][5]
        assert st["fell_back"] == 0 and (st["from_tables"] >= 2 if tables else st["from_tables"] == 0), (r, st)


def test_replayed_jobs_fuzz():
    """tools/fuzz_sharded.py for 25 s: 2-8 rank jobs played rank by rank on the one GPU (sharded.replay_job), windows and
    whole inputs, long runs, slabs smaller than a block; every stream == the oracle's, every cut from the tables."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sharded.py"), "25", "5"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "fuzz_sharded ok" in p.stdout


@pytest.mark.parametrize("tables", [True, False])
def test_run_across_a_slab_edge_closes_the_next_ranks_block(tables):
    """tools/shard_edge_replay.py: slab edges 1, 2, 3, 4 and more bytes into the 255-byte chunk that closes a block (a run's
    count byte belongs to its last input byte, so the slab's image ends in mid-chunk): the block is the next rank's first
    one; streams == oracle with the cut tables and with the chain kernel (every rank returned BZ_E_PARAM)."""
    import subprocess
    env = dict(os.environ, BZ_CUT_TABLES="1" if tables else "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_edge_replay.py")], capture_output=True, text=True, timeout=900,
                       env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "shard_edge_replay ok" in p.stdout


def test_rccl_transport_library_single_rank(pkg, oracle):
    """libbz2_mi355x_rccl.so (the callbacks over RCCL, implemented in C): a one-rank communicator on the test
    box's one GPU goes through the library's transport self-test (all-gather, the variable-length gather of
    device memory -- its own part) and carries a sharded encode.  More ranks need more GPUs: RCCL refuses two
    ranks on one device (the driver's multi-GPU run is where `bench.py --gpus N --transport rccl` can run)."""
    import ctypes
    import torch
    comm = pkg.RcclComm(pkg.rccl_unique_id(), 0, 1, 0)
    assert pkg.lib().bz_shard_comm_selftest(ctypes.byref(comm.struct), 0) == 0
    data, level = _input("mixed")
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    eng = pkg.GpuEngine(0, 16)
    cap = (pkg.encode_bound(len(data)) + 15) & ~15
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    k = eng.encode_sharded(level, d_in.data_ptr(), len(data), comm, d_out.data_ptr(), cap)
    assert bytes(d_out[:k].cpu().numpy()) == oracle.encode(data, level)
    eng.close()
    comm.close()


def _bench_ranks(world, extra_args, env_extra, timeout):
    """bench.py's ranks started by hand (as torch.distributed.run would: RANK / WORLD_SIZE / MASTER_* in the
    environment), so that nobody but the ranks themselves reacts to a peer's death."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu"] + extra_args,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
            o += "\n<<killed by the test: still running after %d s>>" % timeout
        outs.append((p.returncode, o, e))
    return outs


def test_bench_two_ranks_line_is_self_sufficient():
    """bench.py --gpus 2 (two real rank processes sharing the box's one GPU, gloo): the N > 1 line carries
    cpu_baseline, roofline.pipeline_8d, per-step median / min, the rank count, the sharded decode round trip and
    the single-process end-to-end figure over the run's devices -- everything a SCALE line needs."""
    import json
    outs = _bench_ranks(2, ["--steps", "2", "--warmup", "1", "--mib-per-gpu", "32", "--cpu-sample-mib", "8",
                            "--hang-timeout", "240"], {}, 900)
    assert [o[0] for o in outs] == [0, 0], outs[0][2][-2000:] + outs[1][2][-2000:]
    lines0 = [x for x in outs[0][1].splitlines() if x.startswith("{")]
    assert len(lines0) == 1                              # ONE JSON line, from rank 0 only
    assert not [x for x in outs[1][1].splitlines() if x.startswith("{")]
    line = json.loads(lines0[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1
    assert line["roofline"]["pipeline_8d"]["frac"] > 0 and line["roofline"]["frac"] > 0
    assert line["step_ms"]["n"] == 2 and line["step_ms"]["min"] <= line["step_ms"]["median"]
    assert line["config"]["ranks"]["world"] == 2 and line["config"]["ranks"]["rccl_comm_count"] == 2
    assert line["value_is"] == "hbm_resident" and line["value_hbm_resident"] == line["value"]
    # the preflight in front of the timed steps: communicator count, the library's transport self-test over
    # the real communicator with device buffers, one peer round trip per neighbour pair of the end-to-end leg's devices
    pre = line["preflight"]
    assert pre["status"] == "ok" and pre["world"] == 2 and pre["backend"] == "gloo"
    assert [r["rank"] for r in pre["ranks"]] == [0, 1]
    assert all(r["comm_count"] == 2 and r["selftest_status"] == 0 and r["errors"] == [] for r in pre["ranks"])
    # round 6: what is linked, who can reach whom, and every rank's own clock in the one line
    assert pre["rccl_version"] and pre["hip_version"]
    m = pre["peer_access_matrix"]
    assert len(m) == pre["devices_visible"] and all(len(row) == len(m) and row[i] == 1 for i, row in enumerate(m))
    assert len(line["ms_per_step_of_every_rank"]) == 2 and max(line["ms_per_step_of_every_rank"]) == line["ms_per_step"]
    assert pre["peer_copies"]["status"] == 0 and pre["peer_copies"]["devices"] == [0, 0] and pre["peer_copies"]["peer_access"] == [-1, -1]
    assert line["shard_chain"]["chain_ms_per_link"] > 0 and len(line["shard_chain"]["links_ms"]) == 1
    assert line["end_to_end"]["calls_ms"]["n"] == 3 and line["end_to_end"]["phases_ms_of_the_median_call"]["jobs"] >= 1
    assert line["value_end_to_end"] == line["end_to_end"]["bz_encode_buffer_multi"]
    assert line["end_to_end"]["devices"] == [0, 0] and line["end_to_end"]["bz_encode_buffer_multi"] > 0
    assert line["extra"]["decode"]["round_trip_equals_input_on_every_rank"] is True
    assert all(line["checks"].values()), line["checks"]
    for k in ("stream_sha_equals_oracle_golden", "gpu_equals_oracle_on_cpu_sample", "decode_sharded_round_trip",
              "end_to_end_buffer_equals_device_stream"):
        assert k in line["checks"], k


def test_bench_preflight_failure_prints_a_line_and_leaves():
    """The transport self-test of the preflight fails on rank 1 (BZ_BENCH_PREFLIGHT_FAIL=1: that rank reports an error of
    its transport): rank 0 prints ONE line whose "preflight" names the rank, no step is timed, every rank leaves with 4."""
    import json
    outs = _bench_ranks(2, ["--steps", "1", "--warmup", "0", "--mib-per-gpu", "32", "--no-extras", "--no-cpu-baseline",
                            "--hang-timeout", "120"], {"BZ_BENCH_PREFLIGHT_FAIL": "1"}, 300)
    assert [o[0] for o in outs] == [4, 4], outs[0][2][-1500:] + outs[1][2][-1500:]
    lines0 = [x for x in outs[0][1].splitlines() if x.startswith("{")]
    assert len(lines0) == 1
    line = json.loads(lines0[0])
    assert line["value"] is None and line["n_gpus"] == 2
    assert line["preflight"]["status"] == "FAILED on rank(s) [1]"
    assert line["preflight"]["ranks"][1]["errors"] and line["preflight"]["ranks"][0]["errors"] == []


def test_bench_rank_death_does_not_hang_the_others():
    """Rank 1 dies in front of its second timed step (BZ_BENCH_DIE=1:1); rank 0 is then inside a collective
    whose peer is gone.  It must leave with a non-zero status well inside the hang timeout's reach instead of
    waiting for ever -- through the transport's own error or bench.py's watchdog (status 5)."""
    import time
    t0 = time.time()
    outs = _bench_ranks(2, ["--steps", "4", "--warmup", "1", "--mib-per-gpu", "32", "--no-extras", "--no-cpu-baseline",
                            "--hang-timeout", "45"], {"BZ_BENCH_DIE": "1:1"}, 300)
    took = time.time() - t0
    assert outs[1][0] == 9
    assert outs[0][0] not in (0, None) and "killed by the test" not in outs[0][1], outs[0][2][-1500:]
    assert took < 240
