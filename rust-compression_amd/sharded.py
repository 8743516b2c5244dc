"""Multi-GPU transport for the block-encode / decode paths: one process per GPU.

The sharding itself lives in the library (bz_gpu_encode_sharded, bz_gpu_decode_device_sharded,
csrc/engine.hip, csrc/dec_engine.hip): rank r owns a SLAB of the input (a contiguous range of 4 KiB
tiles), encodes the blocks that END inside it, stream order is rank order.  What crosses ranks goes
through four C callbacks (struct bz_shard_comm, include/bz2_mi355x.h):

  1. all-gather of 8 B per rank (the last run start inside each slab: the RLE1 phase at every
     slab's left edge)
  2. the cut chain: rank r-1 tells rank r where its first block starts          -- 16 B per hop
  3. all-gather of per-rank (blocks, words, status), then per-block (word offset, bit length, CRC)
  4. ONE variable-length gather of the packed bit strings to rank 0             -- ~0.2 x input bytes

This module is the torch.distributed implementation of those callbacks: with backend "nccl" the
collectives are RCCL over xGMI (device tensors); with "gloo" the same calls run on host tensors
(device buffers are staged through the host), which is how the path is tested with several processes
on one GPU and on CPU (tests/test_sharded_gloo.py, tests/test_gpu_sharded.py).
"""
import ctypes as C

import torch
import torch.distributed as dist

TILE = 4096

# the four callbacks of struct bz_shard_comm
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
GATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64),
                         C.POINTER(C.c_uint64))


class ShardComm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("allgather", ALLGATHER_FN),
                ("send", SEND_FN), ("recv", RECV_FN), ("gatherv", GATHERV_FN)]


def slab_tiles(n, rank, world):
    """Tile range [t0, t1) of `rank` (bz_shard_slab_tiles: the library's split -- nearly equal shares of the
    ceil(n / 4096) tiles, shrinking a little from rank to rank to make up for the cut chain, BZ_SHARD_SKEW)."""
    import importlib
    lib = importlib.import_module(__package__).lib()
    t0, t1 = C.c_uint64(0), C.c_uint64(0)
    rc = lib.bz_shard_slab_tiles(n, rank, world, C.byref(t0), C.byref(t1))
    if rc != 0:
        raise ValueError("bz_shard_slab_tiles: status %d" % rc)
    return t0.value, t1.value


class _DevBytes:
    """A raw device pointer as something torch.as_tensor understands."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """struct bz_shard_comm over torch.distributed.

    device: torch.device of this rank's GPU, or None / cpu when the "device" buffers of gatherv are host
    memory (CPU tests).  Buffers handed to the library as d_packed / d_gather should be registered
    (register(tensor)) so that gatherv finds the tensor behind a pointer; unregistered pointers are
    wrapped through __cuda_array_interface__ (cuda) or ctypes (cpu)."""

    def __init__(self, rank, world, device=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.device = device if device is not None else torch.device("cpu")
        backend = dist.get_backend(group) if world > 1 else "none"
        # RCCL moves device tensors; gloo moves host tensors
        self.wire = self.device if (backend == "nccl") else torch.device("cpu")
        self._tensors = {}
        self.errors = []
        self._cbs = (ALLGATHER_FN(self._allgather), SEND_FN(self._send), RECV_FN(self._recv), GATHERV_FN(self._gatherv))
        self.struct = ShardComm(None, rank, world, *self._cbs)

    def register(self, t):
        self._tensors[t.data_ptr()] = t
        return t

    # ---- helpers -----------------------------------------------------------------------------
    def _host_bytes(self, ptr, n):
        return torch.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=torch.uint8) if n else torch.empty(0, dtype=torch.uint8)

    def _buffer(self, ptr, nbytes):
        """uint8 view of `nbytes` bytes at `ptr` in the memory space of self.device"""
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        for base, t in self._tensors.items():
            size = t.numel() * t.element_size()
            if base <= ptr and ptr + nbytes <= base + size:
                flat = t.view(-1).view(torch.uint8)
                return flat[ptr - base:ptr - base + nbytes]
        if self.device.type == "cuda":
            return torch.as_tensor(_DevBytes(ptr, nbytes), device=self.device)
        return self._host_bytes(ptr, nbytes)

    def _settle(self):
        """With RCCL a request's wait() only orders torch's current stream behind the transfer; the library goes on
        on ITS stream (rank 0 assembles out of d_recv, the others write the next bit strings into d_send), so the
        transfer has to be over when the callback returns (csrc/rccl_comm.hip ends with a stream synchronise too)."""
        if self.wire.type == "cuda":
            torch.cuda.synchronize(self.wire)
        elif self.device.type == "cuda":
            # host wire (gloo), device buffers: rank 0's own part is an asynchronous device-to-device copy on
            # torch's stream, and with no other contributor nothing else would wait for it
            torch.cuda.synchronize(self.device)

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except Exception as e:  # the C side turns a non-zero return into BZ_E_UNEXPECTED
            self.errors.append(repr(e))
            return 1

    # ---- the callbacks ------------------------------------------------------------------------
    def _allgather(self, _ctx, send, nbytes, recv):
        def run():
            mine = self._host_bytes(send, nbytes).clone()
            parts = [torch.empty(nbytes, dtype=torch.uint8, device=self.wire) for _ in range(self.world)]
            dist.all_gather(parts, mine.to(self.wire), group=self.group)
            self._host_bytes(recv, self.world * nbytes).copy_(torch.cat(parts).cpu())
        return self._guard(run)

    def _send(self, _ctx, dst, buf, nbytes):
        def run():
            dist.send(self._host_bytes(buf, nbytes).clone().to(self.wire), dst=dst, group=self.group)
        return self._guard(run)

    def _recv(self, _ctx, src, buf, nbytes):
        def run():
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.wire)
            dist.recv(t, src=src, group=self.group)
            self._host_bytes(buf, nbytes).copy_(t.cpu())
        return self._guard(run)

    def _gatherv(self, _ctx, d_send, send_bytes, d_recv, recv_off, recv_bytes):
        def run():
            mine = self._buffer(d_send, send_bytes)
            if self.rank != 0:
                if send_bytes:
                    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.to(self.wire).contiguous(), 0, self.group)]):
                        req.wait()
                self._settle()
                return
            offs = [int(recv_off[r]) for r in range(self.world)]
            lens = [int(recv_bytes[r]) for r in range(self.world)]
            whole = self._buffer(d_recv, max(o + k for o, k in zip(offs, lens)))
            whole[offs[0]:offs[0] + lens[0]].copy_(mine)
            # all receives are posted together (one grouped RCCL call: the peers' bit strings arrive over their
            # own xGMI links side by side, not one after the other)
            ops, staged = [], []
            for r in range(1, self.world):
                if lens[r] == 0:
                    continue
                part = whole[offs[r]:offs[r] + lens[r]]
                if self.wire == part.device:
                    ops.append(dist.P2POp(dist.irecv, part, r, self.group))
                else:
                    t = torch.empty(lens[r], dtype=torch.uint8, device=self.wire)
                    ops.append(dist.P2POp(dist.irecv, t, r, self.group))
                    staged.append((part, t))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for part, t in staged:
                part.copy_(t)
            self._settle()
        return self._guard(run)


class ReplayComm(TorchComm):
    """One rank of a `world`-rank job played BY ITSELF, the ranks one after the other in one process (rank 0 first):
    what the ranks in front contributed to the exchanges is replayed from `history` (a dict shared by the ranks of
    one replay), the ranks behind contribute zeros.  The cuts, the block records and the timings of a rank are those of
    the real job -- they depend on the ranks in front only --, measured on a GPU that does nothing else (ranks that
    share one GPU as processes wait for each other's kernels, which says nothing about a link of the chain).  The
    stream rank 0 assembles in that pass holds rank 0's blocks only; the ranks' bit strings are kept, though, and rank
    0 played a SECOND time with the same history sees everything every rank contributed and assembles the job's
    stream (`replay_job`).  bench.py's `shard_link_replay`, tools/fuzz_sharded.py."""

    def __init__(self, rank, world, device, history):
        self.rank, self.world, self.group = rank, world, None
        self.device, self.wire = device, torch.device("cpu")
        self._tensors, self.errors, self.history, self._n_ag = {}, [], history, 0
        self._cbs = (ALLGATHER_FN(self._allgather), SEND_FN(self._send), RECV_FN(self._recv), GATHERV_FN(self._gatherv))
        self.struct = ShardComm(None, rank, world, *self._cbs)

    def _allgather(self, _ctx, send, nbytes, recv):
        def run():
            k, self._n_ag = self._n_ag, self._n_ag + 1
            mine = bytes(self._host_bytes(send, nbytes).numpy())
            seen = self.history.setdefault(("allgather", k), {})
            seen[self.rank] = mine
            # (the per-block records are padded to the largest block count a rank KNOWS of, which grows from rank to
            # rank in a replay: a recorded piece is cut or zero-padded to this call's size -- the padding is zeros anyway)
            out = b"".join(seen.get(r, b"")[:nbytes].ljust(nbytes, b"\0") for r in range(self.world))
            (C.c_uint8 * len(out)).from_address(recv)[:] = out
        return self._guard(run)

    def _send(self, _ctx, dst, buf, nbytes):
        def run():
            self.history[("hop", dst)] = bytes(self._host_bytes(buf, nbytes).numpy())
        return self._guard(run)

    def _recv(self, _ctx, src, buf, nbytes):
        def run():
            data = self.history[("hop", self.rank)]
            assert len(data) == nbytes
            (C.c_uint8 * nbytes).from_address(buf)[:] = data
        return self._guard(run)

    def _gatherv(self, _ctx, d_send, send_bytes, d_recv, recv_off, recv_bytes):
        def run():
            torch.cuda.synchronize(self.device)
            mine = self._buffer(d_send, send_bytes)
            if self.rank != 0:
                self.history[("packed", self.rank)] = mine.clone()
                return
            for r in range(self.world):
                part = mine if r == 0 else self.history.get(("packed", r))
                k = int(recv_bytes[r])
                if part is None or k == 0:
                    continue  # (first pass: the ranks behind have not run yet and announced nothing)
                assert part.numel() == k, (r, part.numel(), k)
                self._buffer(d_recv, int(recv_off[r]) + k)[int(recv_off[r]):].copy_(part)
            torch.cuda.synchronize(self.device)
        return self._guard(run)


def replay_job(eng, level, data_dev, n, world, d_out, cap, windows=True):
    """A `world`-rank bz_gpu_encode_sharded job on ONE engine and one GPU, rank by rank (ReplayComm), then rank 0 again
    with everything the ranks contributed: returns the length of the job's stream in d_out (a uint8 tensor of `cap`
    bytes) and the per-rank shard phases.  data_dev: the n input bytes on the device (16-byte aligned).  windows: each rank
    gets only its window of the input (bz_shard_window), copied to a buffer of its own."""
    import importlib
    pkg = importlib.import_module("rust-compression_amd")
    hist, phases = {}, []
    k = 0
    for rank in list(range(world)) + [0]:
        comm = ReplayComm(rank, world, data_dev.device, hist)
        if windows:
            off, nbytes = pkg.shard_window(level, n, rank, world)
            win = torch.empty(nbytes + 16, dtype=torch.uint8, device=data_dev.device)[:nbytes]
            win.copy_(data_dev[off:off + nbytes])
            k = eng.encode_sharded_window(level, win.data_ptr(), off, nbytes, n, comm, d_out.data_ptr(), cap if rank == 0 else 16)
        else:
            k = eng.encode_sharded(level, data_dev.data_ptr(), n, comm, d_out.data_ptr(), cap if rank == 0 else 16)
        assert not comm.errors, comm.errors
        phases.append(eng.shard_phases())
    return k, phases


def allgather_bytes(rank, world, dev):
    """The one collective of the sharded decode (bz_gpu_decode_device_sharded): returns a function
    send: bytes -> concatenation of every rank's bytes in rank order, over torch.distributed (RCCL
    when the process group is "nccl": the few KB travel through a device tensor; gloo on CPU)."""

    def gather(send):
        if world == 1:
            return bytes(send)
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        if dev is not None and dev.type == "cuda" and dist.get_backend() == "nccl":
            t = t.to(dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return b"".join(bytes(o.cpu().numpy()) for o in outs)
    return gather
