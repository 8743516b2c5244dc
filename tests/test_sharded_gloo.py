"""N > 1 path on CPU: two/three processes over gloo drive the transport the multi-GPU encode and
decode use (rust-compression_amd/sharded.py: the C callbacks of bz_shard_comm / bz_allgather_fn over
torch.distributed) from the library's own C side (bz_shard_comm_selftest: the exchanges of
bz_gpu_encode_sharded with known patterns).  The data path itself needs a GPU:
tests/test_gpu_sharded.py runs it in two real processes."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _comm_worker(rank, world, port, q):
    """every rank: the library's transport self-test (patterns shaped like the exchanges of
    bz_gpu_encode_sharded) through sharded.TorchComm over gloo, buffers in host memory"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        pkg = importlib.import_module("rust-compression_amd")
        sharded = importlib.import_module("rust-compression_amd.sharded")
        comm = sharded.TorchComm(rank, world, torch.device("cpu"))
        rc = pkg.lib().bz_shard_comm_selftest(ctypes.byref(comm.struct), 1)
        q.put((rank, rc, list(comm.errors)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_comm_callbacks_over_gloo(world):
    """the four C callbacks of bz_shard_comm (all-gather, chain send/recv, variable-length gather),
    driven from the library's C side, carried by torch.distributed"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert got == [(r, 0, []) for r in range(world)]


def test_split_helpers():
    sharded = importlib.import_module("rust-compression_amd.sharded")
    n = 10 * 4096 + 5
    spans = [sharded.slab_tiles(n, r, 3) for r in range(3)]
    assert spans[0][0] == 0 and spans[-1][1] == 11
    assert all(spans[i][1] == spans[i + 1][0] for i in range(2))


def test_shard_skew_switch_moves_the_slab_edges():
    """BZ_SHARD_SKEW (read once per process): 0 = equal slabs; a larger value hands the ranks in front more tiles (they wait
    for nobody's cuts).  Whatever the value the slabs tile the input in rank order -- the bytes of a sharded stream do not
    depend on it (tests/test_gpu_sharded.py compares them with the oracle's)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import sys, importlib
sys.path.insert(0, %r)
sharded = importlib.import_module("rust-compression_amd.sharded")
n, world = 1 << 30, 8
spans = [sharded.slab_tiles(n, r, world) for r in range(world)]
assert spans[0][0] == 0 and spans[-1][1] == (n + 4095) // 4096
assert all(spans[i][1] == spans[i + 1][0] and spans[i][0] < spans[i][1] for i in range(world - 1))
print(" ".join(str(b - a) for a, b in spans))
""" % ROOT
    sizes = {}
    for skew in ("0", "0.05"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ_SHARD_SKEW=skew), capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        sizes[skew] = [int(x) for x in out.stdout.split()]
    assert max(sizes["0"]) - min(sizes["0"]) <= 1
    assert sizes["0.05"][0] > sizes["0.05"][-1] + 100 and sum(sizes["0.05"]) == sum(sizes["0"])


def test_shard_comm_parameter_checks():
    import ctypes
    pkg = importlib.import_module("rust-compression_amd")
    sharded = importlib.import_module("rust-compression_amd.sharded")
    L = pkg.lib()
    assert L.bz_shard_comm_selftest(None, 1) == pkg.BZ_E_PARAM
    bad = sharded.ShardComm(None, 3, 2)  # rank outside the world
    assert L.bz_shard_comm_selftest(ctypes.byref(bad), 1) == pkg.BZ_E_PARAM
    assert L.bz_gpu_encode_sharded(None, 9, None, 0, ctypes.byref(bad), None, 0, None, 0, None, 0, None) == pkg.BZ_E_PARAM


def _allgather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = importlib.import_module("rust-compression_amd.sharded")
        gather = sharded.allgather_bytes(rank, world, None)
        # the two exchanges of the sharded decode: per-candidate records, then per-rank summaries
        recs = bytes([rank + 1]) * 24 * 5
        got = gather(recs)
        ok = got == b"".join(bytes([r + 1]) * 24 * 5 for r in range(world))
        summ = (rank * 1000 + 7).to_bytes(8, "little") * 4
        got2 = gather(summ)
        ok = ok and got2 == b"".join((r * 1000 + 7).to_bytes(8, "little") * 4 for r in range(world))
        if rank == 0:
            q.put(ok)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_decode_allgather_adapter(world):
    """the collective behind bz_gpu_decode_device_sharded (sharded.allgather_bytes) over gloo"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allgather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
