"""N > 1 path on CPU: two/three processes over gloo run the same exchange code the GPU bench uses (rust-compression_amd/sharded.py).  The per-block bit strings come from the oracle
(this test's stand-in for the HIP engine) and rank 0's assembled stream must equal the oracle's
serial stream byte for byte."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, sample


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bits_of(stream: bytes):
    return int.from_bytes(stream, "big"), len(stream) * 8


def _block_strings(stream: bytes, stats):
    """Cut the oracle stream into per-block bit strings using the per-block bit counts."""
    v, total = _bits_of(stream)
    pos = 32
    out = []
    for st in stats:
        nb = st["bits"]
        chunk = (v >> (total - pos - nb)) & ((1 << nb) - 1)
        out.append((chunk, nb, st["block_crc"]))
        pos += nb
    return out


def _pack(blocks):
    """Bit strings -> logical 32-bit words (bit 31 first), the layout bz_gpu_encode_blocks produces."""
    words, woff = [], []
    for chunk, nb, _ in blocks:
        woff.append(len(words))
        nw = (nb + 31) // 32
        padded = chunk << (nw * 32 - nb)
        for k in range(nw):
            words.append((padded >> ((nw - 1 - k) * 32)) & 0xFFFFFFFF)
    t = torch.tensor(words if words else [0], dtype=torch.int64)
    return (t - ((t >> 31) << 32)).to(torch.int32), woff, len(words)


def _assemble(level, flat, woff, blen, crcs):
    """Pure-python restatement of bz_gpu_assemble (header, blocks, trailer, pad) for the CPU test."""
    v, nbits = 0, 0

    def put(x, n):
        nonlocal v, nbits
        v = (v << n) | (x & ((1 << n) - 1))
        nbits += n
    put(0x425A68, 24)
    put(0x30 + level, 8)
    comb = 0
    for off, nb, crc in zip(woff, blen, crcs):
        nw = (nb + 31) // 32
        chunk = 0
        for k in range(nw):
            chunk = (chunk << 32) | (int(flat[off + k]) & 0xFFFFFFFF)
        put(chunk >> (nw * 32 - nb), nb)
        comb = (((comb << 1) | (comb >> 31)) & 0xFFFFFFFF) ^ crc
    put(0x177245385090, 48)
    put(comb, 32)
    pad = (-nbits) % 8
    put(0, pad)
    return v.to_bytes(nbits // 8, "big")


def _worker(rank, world, port, level, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        sharded = importlib.import_module("rust-compression_amd.sharded")
        data = sample(2) + sample(1) + sample(4)
        stream, stats = oracle.encode(data, level, with_stats=True)
        blocks = _block_strings(stream, stats)
        nb = len(blocks)
        mine = sharded.split_contiguous(nb, rank, world)
        packed, woff, used = _pack([blocks[b] for b in mine])
        res = sharded.exchange(woff, [blocks[b][1] for b in mine], [blocks[b][2] for b in mine], packed, used,
                               rank, world, torch.device("cpu"))
        if rank == 0:
            buf, w_off, b_len, crcs = res
            out = _assemble(level, buf.view(-1).tolist(), w_off, b_len, crcs)
            q.put((nb, out == stream, len(out)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_round_robin_exchange_and_assembly(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    nb, same, n = q.get(timeout=10)
    assert nb >= 5 and same and n > 1000


def test_split_helpers():
    sharded = importlib.import_module("rust-compression_amd.sharded")
    assert sorted(sum((sharded.split_contiguous(11, r, 4) for r in range(4)), [])) == list(range(11))
    assert sharded.split_contiguous(3, 3, 4) == [2]
    n = 10 * 4096 + 5
    spans = [sharded.slab_tiles(n, r, 3) for r in range(3)]
    assert spans[0][0] == 0 and spans[-1][1] == 11
    assert all(spans[i][1] == spans[i + 1][0] for i in range(2))


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
