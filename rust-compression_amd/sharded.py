"""Multi-GPU orchestration of the block-encode path: one process per GPU.

The path shards naturally: once the input is split into blocks, every block is
independent.  Rank r owns a SLAB of the input (a contiguous range of 4 KiB tiles) and encodes the
blocks that END inside it; stream order is rank order.  What crosses ranks:

  1. all-gather of ONE int64 per rank (the last run start inside each slab) so that every rank knows
     the RLE1 phase at its left edge                                            -- 8 B per rank
  2. the cut chain: rank r-1 tells rank r where its first block starts          -- 8 B per hop
  3. all-gather of per-block (word offset, bit length, CRC)                     -- a few KB
  4. a variable-length gather of the packed bit strings to rank 0               -- ~0.2 x input bytes

and rank 0 assembles the stream (bit-granular concatenation + header/trailer).  With backend "nccl"
the collectives are RCCL over xGMI; the same code runs on CPU tensors with "gloo"
(tests/test_sharded_gloo.py).  There is no other collective on the data path.
"""
import torch
import torch.distributed as dist

TILE = 4096


def slab_tiles(n, rank, world):
    """Tile range [t0, t1) of `rank`: equal shares of the ceil(n / 4096) tiles."""
    ntiles = (n + TILE - 1) // TILE
    return ntiles * rank // world, ntiles * (rank + 1) // world


def split_contiguous(n_blocks, rank, world):
    """(CPU stand-in for the slab split) blocks of `rank` when blocks are dealt contiguously."""
    return list(range(n_blocks * rank // world, n_blocks * (rank + 1) // world))


def partition(eng, level, d_in, n, rank, world, device):
    """Slab-sharded split: returns this rank's block count (its blocks are then encoded with
    eng.encode_blocks(0, 1, nb, ...))."""
    t0, t1 = slab_tiles(n, rank, world)
    last = eng.slab_begin(level, d_in, n, t0, t1)
    if world > 1:
        mine = torch.tensor([last], dtype=torch.int64, device=device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        lasts = [int(p.item()) for p in parts]
    else:
        lasts = [last]
    eng.slab_count(max(lasts[:rank], default=-1))
    start = 0
    if rank > 0:
        buf = torch.zeros(1, dtype=torch.int64, device=device)
        dist.recv(buf, src=rank - 1)
        start = int(buf.item())
    nb, nxt, _tail = eng.slab_finish(start, rank == world - 1)
    if rank < world - 1:
        dist.send(torch.tensor([nxt], dtype=torch.int64, device=device), dst=rank + 1)
    return nb


def exchange(word_off, bit_len, crc, packed, words_used, rank, world, device, gather_buf=None):
    """Block bit strings of every rank -> rank 0, stream order = rank order.

    word_off/bit_len/crc: python lists for this rank's blocks; packed: int32 tensor holding their
    bit strings (words_used words are meaningful).
    Returns on rank 0: (all_packed [world, row] int32 tensor, lists woff, blen, crcs in stream order,
    woff indexing all_packed.view(-1)); on other ranks None."""
    k = len(word_off)
    head = torch.tensor([k, words_used], dtype=torch.int64, device=device)
    if world > 1:
        heads = [torch.empty_like(head) for _ in range(world)]
        dist.all_gather(heads, head)
        heads = torch.stack(heads).cpu()
    else:
        heads = head.unsqueeze(0).cpu()
    kmax = max(int(heads[:, 0].max().item()), 1)
    maxw = max(int(heads[:, 1].max().item()), 1)
    meta = torch.zeros((kmax, 3), dtype=torch.int64)
    if k:
        meta[:k, 0] = torch.tensor(word_off, dtype=torch.int64)
        meta[:k, 1] = torch.tensor(bit_len, dtype=torch.int64)
        meta[:k, 2] = torch.tensor(crc, dtype=torch.int64)
    meta = meta.to(device)
    if world > 1:
        parts = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(parts, meta)
        am = torch.stack(parts).cpu()
    else:
        am = meta.unsqueeze(0).cpu()
    if packed.numel() < maxw:  # every rank must contribute the same number of words
        grown = torch.zeros(maxw, dtype=packed.dtype, device=packed.device)
        grown[:packed.numel()] = packed
        packed = grown
    if rank == 0:
        if gather_buf is None or gather_buf.shape[0] < world or gather_buf.shape[1] < maxw:
            gather_buf = torch.empty((world, maxw), dtype=torch.int32, device=device)
        row = gather_buf.shape[1]
        if world > 1:
            dist.gather(packed[:maxw], [gather_buf[r, :maxw] for r in range(world)], dst=0)
        else:
            gather_buf[0, :maxw] = packed[:maxw]
        woff, blen, crcs = [], [], []
        for r in range(world):
            kr = int(heads[r, 0].item())
            woff += (am[r, :kr, 0] + r * row).tolist()
            blen += am[r, :kr, 1].tolist()
            crcs += am[r, :kr, 2].tolist()
        return gather_buf, woff, blen, crcs
    dist.gather(packed[:maxw], None, dst=0)
    return None


def allgather_bytes(rank, world, dev):
    """The one collective of the sharded decode (bz_gpu_decode_device_sharded): returns a function
    send: bytes -> concatenation of every rank's bytes in rank order, over torch.distributed (RCCL
    when the process group is "nccl": the few KB travel through a device tensor; gloo on CPU)."""
    import torch
    import torch.distributed as dist

    def gather(send):
        if world == 1:
            return bytes(send)
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        if dev is not None and dev.type == "cuda":
            t = t.to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return b"".join(bytes(o.cpu().numpy()) for o in outs)
    return gather
