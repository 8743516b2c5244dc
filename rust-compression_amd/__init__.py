"""rust-compression_amd -- MI355X-native BZip2 block-encode path (and its inverse, the decode path).

Host-side mirror (Python flavour) of the reference's interface for this one path:

    reference (Rust)                                   here
    ------------------------------------------------   ---------------------------------------
    Action::{Run,Flush,Finish}   src/action.rs:8-13    Action.RUN / FLUSH / FINISH
    CompressionError             src/error.rs:10-15    CompressionError(kind)
    BZip2Encoder::new(level)     bzip2/encoder.rs:58   BZip2Encoder(level)   (ValueError = the panic)
    Encoder::next(iter, action)  traits/encoder.rs:81  BZip2Encoder.next(iter, action) -> int | None
    iter.encode(&mut enc, act)   traits/encoder.rs:12  encode(iterable, enc, action) -> iterator of ints
    BZip2Error                   bzip2/error.rs:5-11   BZip2Error(kind)  (a CompressionError, bzip2/error.rs:45-53)
    BZip2Decoder::new()          bzip2/decoder.rs:588  BZip2Decoder()
    Decoder::next(iter)          traits/decoder.rs:95  BZip2Decoder.next(iter) -> int | None, raises BZip2Error
    iter.decode(&mut dec)        traits/decoder.rs:15  decode(iterable, dec) -> iterator of ints

Everything below the iterator plumbing happens in the HIP library (csrc/, C ABI in
include/bz2_mi355x.h) loaded with ctypes.  There is no CPU implementation in this package: if the
library or a gfx950 device is missing, calls raise.  PyTorch is used only by GpuEngine's callers
for device memory (tensor.data_ptr()); nothing here imports torch.
"""
import ctypes as C
import enum
import os
import sys

from . import _build

__all__ = ["Action", "CompressionError", "BZip2Error", "BZip2Encoder", "BZip2Decoder", "encode", "decode",
           "compress", "decompress", "GpuEngine", "release_cached_resources", "last_call_phases",
           "build", "lib", "device_count", "encode_bound", "shard_window", "rccl_lib", "rccl_unique_id", "RcclComm"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BZ_OK, BZ_E_DATA, BZ_E_EOF, BZ_E_UNEXPECTED = 0, -1, -2, -3
BZ_E_MAGIC_FIRST, BZ_E_MAGIC = -4, -5
BZ_E_PARAM, BZ_E_NOGPU, BZ_E_NOMEM, BZ_E_CAPACITY = -6, -7, -8, -9


class Action(enum.IntEnum):
    """src/action.rs:8-13"""
    RUN = 0
    FLUSH = 1
    FINISH = 2


class CompressionError(Exception):
    """src/error.rs:10-15 (kinds DataError / UnexpectedEof / Unexpected) + this library's own."""
    KINDS = {BZ_E_DATA: "DataError", BZ_E_EOF: "UnexpectedEof", BZ_E_UNEXPECTED: "Unexpected",
             BZ_E_NOGPU: "NoGpu", BZ_E_NOMEM: "NoMemory", BZ_E_CAPACITY: "Capacity", BZ_E_PARAM: "Param"}

    def __init__(self, code):
        self.code = code
        self.kind = self.KINDS.get(code, "Unexpected")
        msg = lib().bz_strerror(code).decode() if _LIB is not None else str(code)
        super().__init__("%s (%d): %s" % (self.kind, code, msg))


class BZip2Error(CompressionError):
    """src/bzip2/error.rs:5-11.  `kind` is the BZip2Error variant; as a CompressionError the two
    magic variants are DataError (bzip2/error.rs:45-53)."""
    BZ_KINDS = {BZ_E_DATA: "DataError", BZ_E_MAGIC_FIRST: "DataErrorMagicFirst", BZ_E_MAGIC: "DataErrorMagic",
                BZ_E_EOF: "UnexpectedEof", BZ_E_UNEXPECTED: "Unexpected"}

    def __init__(self, code, partial=b""):
        super().__init__(code)
        self.bzip2_kind = self.BZ_KINDS.get(code, "Unexpected")
        if code in (BZ_E_MAGIC_FIRST, BZ_E_MAGIC):
            self.kind = "DataError"
        self.partial = partial  # bytes the iterator yielded before this Err


def build(force=False):
    """Compile csrc/*.hip for gfx950 into rust-compression_amd/libbz2_mi355x.so (in-tree)."""
    return _build.build(force=force)


EXPORTS = [
    "bz_strerror", "bz_version", "bz_device_count",
    "bz_enc_create", "bz_enc_write", "bz_enc_end", "bz_enc_read", "bz_enc_pending", "bz_enc_destroy",
    "bz_encode_buffer", "bz_free", "bz_enc_create_multi", "bz_enc_set_verify", "bz_enc_verify_stats", "bz_enc_phase_stats", "bz_encode_buffer_last_phases",
    "bz_gpu_engine_set_verify", "bz_gpu_verify_stats", "bz_encode_buffer_multi", "bz_release_cached_resources", "bz_peer_copy_selftest",
    "bz_gpu_engine_create", "bz_gpu_engine_destroy", "bz_gpu_engine_reserve", "bz_encode_bound", "bz_gpu_encode_device",
    "bz_gpu_partition", "bz_gpu_partition_slab_begin", "bz_gpu_partition_slab_count",
    "bz_gpu_partition_slab_finish", "bz_gpu_block_count", "bz_gpu_encode_blocks", "bz_gpu_assemble", "bz_gpu_encode_sharded",
    "bz_shard_comm_selftest", "bz_gpu_last_timings", "bz_shard_halo_bytes", "bz_shard_window", "bz_shard_slab_tiles", "bz_gpu_encode_sharded_window", "bz_gpu_last_shard_timings", "bz_gpu_last_shard_phases",
    "bz_gpu_last_bwt_stats", "bz_gpu_cut_stats", "bz_gpu_last_bwt_rounds", "bz_gpu_profile_enable", "bz_gpu_profile_kernels", "bz_gpu_profile_get",
    "bz_gpu_debug_bwt", "bz_gpu_debug_code_lengths", "bz_gpu_debug_block_stats", "bz_gpu_debug_block_sections",
    "bz_gpu_decode_device", "bz_gpu_decode_device_sharded", "bz_gpu_last_decode_timings", "bz_gpu_last_decode_stats", "bz_decode_buffer",
    "bz_dec_create", "bz_dec_write", "bz_dec_end", "bz_dec_read", "bz_dec_pending", "bz_dec_destroy",
    "df_encode_bound", "df_gpu_encode_device", "df_gpu_last_timings", "df_gpu_last_stats", "df_gpu_debug_codes",
    "df_gpu_debug_blocks", "df_encode_buffer", "df_gpu_encode_device_dict", "df_encode_buffer_dict", "df_enc_create_dict",
    "df_enc_create", "df_enc_write", "df_enc_end", "df_enc_read", "df_enc_pending", "df_enc_destroy", "df_enc_finished",
]


# int (*bz_allgather_fn)(void *ctx, const void *send, size_t bytes, void *recv)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one process do not
    work (the second sees no GPU), so when torch is installed its copy is loaded first and this
    library binds to it by soname.  torch itself is NOT imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the HIP library (never falls back to anything else)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libbz2_mi355x.so")
    if not os.path.exists(so):
        so = build()
    _share_hip_runtime_with_torch()
    L = C.CDLL(so)
    vp, sz, szp = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.bz_strerror.restype = C.c_char_p
    L.bz_strerror.argtypes = [C.c_int]
    L.bz_version.restype = C.c_char_p
    L.bz_device_count.restype = C.c_int
    L.bz_enc_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.bz_enc_write.argtypes = [vp, C.c_char_p, sz]
    L.bz_enc_end.argtypes = [vp, C.c_int]
    L.bz_enc_read.restype = C.c_long
    L.bz_enc_read.argtypes = [vp, u8p, sz]
    L.bz_enc_pending.restype = sz
    L.bz_enc_pending.argtypes = [vp]
    L.bz_enc_destroy.restype = None
    L.bz_enc_destroy.argtypes = [vp]
    L.bz_encode_buffer.argtypes = [C.c_int, C.c_int, C.c_char_p, sz, C.POINTER(u8p), szp]
    L.bz_enc_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_int]
    L.bz_encode_buffer_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_char_p, sz, C.POINTER(u8p), szp]
    L.bz_enc_phase_stats.argtypes = [vp, C.POINTER(C.c_double)]
    L.bz_encode_buffer_last_phases.argtypes = [C.POINTER(C.c_double)]
    L.bz_enc_set_verify.argtypes = [vp, C.c_int]
    L.bz_enc_verify_stats.argtypes = [vp, u64p]
    L.bz_gpu_engine_set_verify.argtypes = [vp, C.c_int]
    L.bz_gpu_verify_stats.argtypes = [vp, u64p]
    L.bz_release_cached_resources.restype = None
    L.bz_release_cached_resources.argtypes = []
    L.bz_peer_copy_selftest.argtypes = [C.POINTER(C.c_int), C.c_int, sz, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.bz_free.restype = None
    L.bz_free.argtypes = [vp]
    L.bz_gpu_engine_create.argtypes = [C.POINTER(vp), C.c_int, sz]
    L.bz_gpu_engine_reserve.argtypes = [vp, sz]
    L.bz_gpu_engine_destroy.restype = None
    L.bz_gpu_engine_destroy.argtypes = [vp]
    L.bz_encode_bound.restype = sz
    L.bz_encode_bound.argtypes = [sz]
    L.bz_gpu_encode_device.argtypes = [vp, C.c_int, vp, sz, vp, sz, szp]
    L.bz_gpu_partition.argtypes = [vp, C.c_int, vp, sz, C.c_int, szp, szp, C.POINTER(C.c_int)]
    L.bz_gpu_partition_slab_begin.argtypes = [vp, C.c_int, vp, sz, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64)]
    L.bz_gpu_partition_slab_count.argtypes = [vp, C.c_int64]
    L.bz_gpu_partition_slab_finish.argtypes = [vp, C.c_uint64, C.c_int, szp, u64p, C.POINTER(C.c_int)]
    L.bz_gpu_block_count.restype = sz
    L.bz_gpu_block_count.argtypes = [vp]
    L.bz_gpu_encode_blocks.argtypes = [vp, sz, sz, vp, sz, u64p, u64p, u32p, szp]
    L.bz_gpu_assemble.argtypes = [vp, C.c_int, sz, vp, u64p, u64p, u32p, C.c_int, C.c_int, C.c_int,
                                  C.c_uint, C.c_uint, C.c_uint32, u32p, vp, sz, szp,
                                  C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.bz_gpu_encode_sharded.argtypes = [vp, C.c_int, vp, sz, vp, vp, sz, vp, sz, vp, sz, szp]
    L.bz_shard_halo_bytes.restype = sz
    L.bz_shard_halo_bytes.argtypes = [C.c_int]
    L.bz_shard_window.argtypes = [C.c_int, sz, C.c_int, C.c_int, u64p, szp]
    L.bz_shard_slab_tiles.argtypes = [sz, C.c_int, C.c_int, u64p, u64p]
    L.bz_gpu_encode_sharded_window.argtypes = [vp, C.c_int, vp, C.c_uint64, sz, sz, vp, vp, sz, vp, sz, vp, sz, szp]
    L.bz_gpu_last_shard_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.bz_gpu_last_shard_phases.argtypes = [vp, C.POINTER(C.c_double)]
    L.bz_shard_comm_selftest.argtypes = [vp, C.c_int]
    L.bz_gpu_last_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.bz_gpu_last_bwt_stats.argtypes = [vp, u64p]
    L.bz_gpu_last_bwt_rounds.argtypes = [vp, u64p]
    L.bz_gpu_cut_stats.argtypes = [vp, u64p]
    L.bz_gpu_profile_enable.argtypes = [vp, C.c_int]
    L.bz_gpu_profile_kernels.argtypes = [vp]
    L.bz_gpu_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), u64p, C.POINTER(C.c_double), u64p]
    L.bz_gpu_debug_bwt.argtypes = [vp, C.c_char_p, sz, u32p]
    L.bz_gpu_debug_code_lengths.argtypes = [vp, u32p, sz, u8p, C.POINTER(C.c_int)]
    L.bz_gpu_debug_block_stats.argtypes = [vp, u32p, sz, szp]
    L.bz_gpu_debug_block_sections.argtypes = [vp, u32p, sz, szp]
    L.bz_gpu_decode_device.argtypes = [vp, vp, sz, vp, sz, szp]
    L.bz_gpu_decode_device_sharded.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.c_int, ALLGATHER_FN, vp, szp, szp, szp]
    L.bz_gpu_last_decode_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.bz_gpu_last_decode_stats.argtypes = [vp, u64p]
    L.bz_decode_buffer.argtypes = [C.c_int, C.c_char_p, sz, C.POINTER(u8p), szp]
    L.bz_dec_create.argtypes = [C.POINTER(vp), C.c_int]
    L.bz_dec_write.argtypes = [vp, C.c_char_p, sz]
    L.bz_dec_end.argtypes = [vp]
    L.bz_dec_read.restype = C.c_long
    L.bz_dec_read.argtypes = [vp, u8p, sz]
    L.bz_dec_pending.restype = sz
    L.bz_dec_pending.argtypes = [vp]
    L.bz_dec_destroy.restype = None
    L.bz_dec_destroy.argtypes = [vp]
    L.df_encode_bound.restype = sz
    L.df_encode_bound.argtypes = [sz]
    L.df_gpu_encode_device.argtypes = [vp, C.c_int, vp, sz, vp, sz, szp]
    L.df_gpu_last_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.df_gpu_last_stats.argtypes = [vp, u64p]
    L.df_gpu_debug_codes.argtypes = [vp, vp, sz, u32p, sz, szp]
    L.df_gpu_debug_blocks.argtypes = [vp, u64p, sz, szp]
    L.df_encode_buffer.argtypes = [C.c_int, C.c_int, C.c_char_p, sz, C.POINTER(u8p), szp]
    L.df_gpu_encode_device_dict.argtypes = [vp, C.c_int, vp, sz, C.c_char_p, sz, vp, sz, szp]
    L.df_encode_buffer_dict.argtypes = [C.c_int, C.c_int, C.c_char_p, sz, C.c_char_p, sz, C.POINTER(u8p), szp]
    L.df_enc_create_dict.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, sz]
    L.df_enc_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.df_enc_write.argtypes = [vp, C.c_char_p, sz]
    L.df_enc_end.argtypes = [vp, C.c_int]
    L.df_enc_finished.argtypes = [vp]
    L.df_enc_read.restype = C.c_long
    L.df_enc_read.argtypes = [vp, u8p, sz]
    L.df_enc_pending.restype = sz
    L.df_enc_pending.argtypes = [vp]
    L.df_enc_destroy.restype = None
    L.df_enc_destroy.argtypes = [vp]
    _LIB = L
    return L


def _check(rc):
    if rc != BZ_OK:
        raise CompressionError(rc)


def _settle():
    """Before a device-pointer call: an engine runs on its own HIP stream, the pointers handed in here usually
    belong to torch tensors whose contents (or whose memory, with torch's stream-ordered allocator) may still be
    in use by work queued on torch's streams.  Waits for that work (include/bz2_mi355x.h, "Stream ordering")."""
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
        t.cuda.synchronize()


_RCCL_LIB = None
RCCL_EXPORTS = ["bz_rccl_unique_id", "bz_rccl_comm_create", "bz_rccl_comm_destroy", "bz_rccl_comm_count"]


def rccl_lib():
    """The RCCL transport library (libbz2_mi355x_rccl.so, csrc/rccl_comm.hip): only it links librccl.  When
    torch is installed its bundled librccl is loaded first, so that the process has ONE RCCL."""
    global _RCCL_LIB
    if _RCCL_LIB is not None:
        return _RCCL_LIB
    lib()
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
    so = _build.build_rccl()  # (a no-op when the library is newer than its source)
    L = C.CDLL(so)
    L.bz_rccl_unique_id.argtypes = [C.c_char_p]
    L.bz_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.bz_rccl_comm_destroy.restype = None
    L.bz_rccl_comm_destroy.argtypes = [C.c_void_p]
    L.bz_rccl_comm_count.argtypes = [C.c_void_p]
    _RCCL_LIB = L
    return L


def rccl_unique_id():
    """rank 0: the 128 bytes every rank needs to join the communicator (ship them by any means)"""
    buf = C.create_string_buffer(128)
    _check(rccl_lib().bz_rccl_unique_id(buf))
    return buf.raw


class RcclComm:
    """struct bz_shard_comm over RCCL, implemented inside libbz2_mi355x_rccl.so (no Python in the data path).
    Creating it is collective (ncclCommInitRank): every rank calls it with the id rank 0 drew."""

    def __init__(self, unique_id, rank, world, device):
        self._p = C.c_void_p()
        _check(rccl_lib().bz_rccl_comm_create(C.byref(self._p), bytes(unique_id), rank, world, device))
        self.rank, self.world = rank, world
        self.errors = []

    @property
    def struct(self):
        """the bz_shard_comm the library filled in (GpuEngine.encode_sharded passes its address)"""
        from .sharded import ShardComm
        return ShardComm.from_address(self._p.value)

    def register(self, t):
        return t  # (device pointers are used as they are)

    def count(self):
        """ranks of the communicator as RCCL counts them (ncclCommCount)"""
        n = rccl_lib().bz_rccl_comm_count(self._p)
        if n < 0:
            raise CompressionError(n)
        return n

    def close(self):
        if getattr(self, "_p", None) and self._p.value and _RCCL_LIB is not None:
            _RCCL_LIB.bz_rccl_comm_destroy(self._p)
            self._p = C.c_void_p()

    __del__ = close


def shard_window(level, n, rank, world):
    """(offset, bytes) of the input a rank of bz_gpu_encode_sharded_window has to hold: its slab, the largest block's
    worth of input in front of it and one tile behind (bz_shard_window)."""
    off, nb = C.c_uint64(0), C.c_size_t(0)
    _check(lib().bz_shard_window(level, n, rank, world, C.byref(off), C.byref(nb)))
    return off.value, nb.value


def device_count():
    return lib().bz_device_count()


def encode_bound(n):
    return lib().bz_encode_bound(n)


class BZip2Encoder:
    """`BZip2Encoder` (src/bzip2/encoder.rs:40-159) over the C ABI's streaming context."""

    CHUNK = 1 << 20  # bytes pulled from the input iterator per refill

    def __init__(self, level=9, device=0, devices=None, verify=None):
        if level < 1 or level > 9:
            raise ValueError("invalid level")  # the reference panics (encoder.rs:59-61)
        self._h = C.c_void_p()
        if devices is None:
            _check(lib().bz_enc_create(C.byref(self._h), level, device))
        else:
            devs = (C.c_int * len(devices))(*devices)
            _check(lib().bz_enc_create_multi(C.byref(self._h), level, devs, len(devices)))
        if verify is not None:
            self.set_verify(verify)
        self._buf = (C.c_uint8 * 65536)()
        self._ready = b""
        self._pos = 0

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bz_enc_destroy(self._h)
            self._h = None

    VERIFY_STATS = ("blocks_checked", "jobs_redone", "jobs_failed_again", "nanoseconds")

    def set_verify(self, on=True):
        """Self-check (bz_enc_set_verify): every job's blocks are decoded on the device and compared with their input
        before their bytes can be read; a job that fails is encoded again without look-back passes."""
        _check(lib().bz_enc_set_verify(self._h, int(bool(on))))

    def verify_stats(self):
        s = (C.c_uint64 * 4)()
        _check(lib().bz_enc_verify_stats(self._h, s))
        return dict(zip(self.VERIFY_STATS, [int(x) for x in s]))

    def phase_stats(self):
        """where the time of this stream went so far (bz_enc_phase_stats)"""
        t = (C.c_double * 8)()
        _check(lib().bz_enc_phase_stats(self._h, t))
        return {k: (int(v) if k == "jobs" else round(v, 3)) for k, v in zip(PHASES, t)}

    @classmethod
    def with_devices(cls, level, devices):
        """The same encoder over several GPUs of this process (bz_enc_create_multi): chunks of the input go round
        two lanes per listed device; the stream is the one `BZip2Encoder(level)` writes."""
        return cls(level, devices=list(devices))

    def _refill(self):
        k = lib().bz_enc_read(self._h, self._buf, len(self._buf))
        if k < 0:
            raise CompressionError(k)
        self._ready = C.string_at(self._buf, k)
        self._pos = 0
        return k

    def next(self, it, action):
        """One `Encoder::next(iter, action)` call: an int byte, or None."""
        if self._pos >= len(self._ready) and self._refill() == 0:
            # pull the input (the reference pulls one byte per call; the bytes are the same)
            while True:
                chunk = bytearray()
                for b in it:
                    chunk.append(b)
                    if len(chunk) >= self.CHUNK:
                        break
                if chunk:
                    _check(lib().bz_enc_write(self._h, bytes(chunk), len(chunk)))
                    if lib().bz_enc_pending(self._h):
                        break
                if len(chunk) < self.CHUNK:
                    _check(lib().bz_enc_end(self._h, int(action)))
                    break
            if self._refill() == 0:
                return None
        b = self._ready[self._pos]
        self._pos += 1
        return b

    # bulk helpers (same semantics, fewer Python-level calls)
    def write(self, data):
        _check(lib().bz_enc_write(self._h, bytes(data), len(data)))

    def end(self, action):
        _check(lib().bz_enc_end(self._h, int(action)))

    def read_all(self):
        out = bytearray(self._ready[self._pos:])
        self._ready, self._pos = b"", 0
        while self._refill():
            out += self._ready
        self._ready, self._pos = b"", 0
        return bytes(out)

    read_available = read_all  # what is complete now (chunks are encoded while the caller goes on writing)

    def encode_all(self, data, action=Action.FINISH):
        """`data.encode(&mut self, action).collect()`"""
        self.write(data)
        self.end(action)
        return self.read_all()


def encode(iterable, encoder, action):
    """`EncodeExt::encode` / `EncodeIterator` (src/traits/encoder.rs:12-79)."""
    it = iter(iterable)
    while True:
        b = encoder.next(it, action)
        if b is None:
            return
        yield b


PHASES = ("caller_copy_in_ms", "caller_wait_staging_ms", "split_serial_ms", "encode_ms", "assemble_serial_ms", "download_ms",
          "jobs", "workers_wait_turn_ms")


def last_call_phases():
    """where the time of the last one-shot call (compress / bz_encode_buffer[_multi]) went (bz_encode_buffer_last_phases)"""
    t = (C.c_double * 8)()
    _check(lib().bz_encode_buffer_last_phases(t))
    return {k: (int(v) if k == "jobs" else round(v, 3)) for k, v in zip(PHASES, t)}


def release_cached_resources():
    """Frees what finished contexts parked for the next one (bz_release_cached_resources)."""
    lib().bz_release_cached_resources()


def compress(data, level=9, device=0, devices=None, verify=None):
    """One-shot over host buffers (bz_encode_buffer; `devices`: bz_encode_buffer_multi over that list of GPUs).
    verify=True: through a streaming context with the self-check on (the one-shot entry points take BZ_VERIFY=1
    from the environment); the bytes are the same."""
    if level < 1 or level > 9:
        raise ValueError("invalid level")
    if not isinstance(data, (bytes, bytearray)):
        data = bytes(data)
    if verify is not None:
        return BZip2Encoder(level, device, devices, verify=verify).encode_all(data)
    buf = data if isinstance(data, bytes) else (C.c_char * len(data)).from_buffer(data)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t(0)
    if devices is None:
        _check(lib().bz_encode_buffer(level, device, buf, len(data), C.byref(out), C.byref(n)))
    else:
        devs = (C.c_int * len(devices))(*devices)
        _check(lib().bz_encode_buffer_multi(level, devs, len(devices), buf, len(data), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        lib().bz_free(out)


# ---------------------------------------------------------------------------- Deflate / zlib / gzip
DEFLATE, ZLIB, GZIP = 0, 1, 2


class Inflater:
    """`Inflater` (src/deflate/encoder.rs:92-260) over the C ABI's streaming context: the reference's
    name for its Deflate ENCODER.  Action.RUN accumulates, Action.FLUSH writes the bytes so far as a
    byte-aligned segment (non-final block; the window and decompress_len carry over), Action.FINISH ends
    the stream.  ZlibEncoder / GZipEncoder end their container at the first None of the inner Inflater whatever
    the Action (zlib/encoder.rs:138-150): Run / Flush give header + what the Inflater yields + trailer, and the
    encoder then returns None without pulling its caller's iterator."""

    KIND = DEFLATE
    CHUNK = 1 << 20

    def __init__(self, device=0, dict_=b""):
        self._h = C.c_void_p()
        dict_ = bytes(dict_)
        _check(lib().df_enc_create_dict(C.byref(self._h), self.KIND, device, dict_, len(dict_)))
        self._buf = (C.c_uint8 * 65536)()
        self._ready = b""
        self._pos = 0

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.df_enc_destroy(self._h)
            self._h = None

    @classmethod
    def with_dict(cls, dict_, device=0):
        """`Inflater::with_dict` / `ZlibEncoder::with_dict`"""
        return cls(device, dict_)

    def _refill(self):
        k = lib().df_enc_read(self._h, self._buf, len(self._buf))
        if k < 0:
            raise CompressionError(k)
        self._ready = C.string_at(self._buf, k)
        self._pos = 0
        return k

    def next(self, it, action):
        """One `Encoder::next(iter, action)` call: an int byte, or None."""
        if self._pos >= len(self._ready) and self._refill() == 0:
            if self.KIND != DEFLATE and lib().df_enc_finished(self._h):
                return None  # the trailer is out: the wrappers do not touch the iterator any more (zlib/encoder.rs:130-136)
            while True:
                chunk = bytearray()
                for b in it:
                    chunk.append(b)
                    if len(chunk) >= self.CHUNK:
                        break
                if chunk:
                    _check(lib().df_enc_write(self._h, bytes(chunk), len(chunk)))
                if len(chunk) < self.CHUNK:
                    _check(lib().df_enc_end(self._h, int(action)))
                    break
            if self._refill() == 0:
                return None
        b = self._ready[self._pos]
        self._pos += 1
        return b

    def write(self, data):
        _check(lib().df_enc_write(self._h, bytes(data), len(data)))

    def end(self, action):
        _check(lib().df_enc_end(self._h, int(action)))

    def read_all(self):
        out = bytearray(self._ready[self._pos:])
        self._ready, self._pos = b"", 0
        while self._refill():
            out += self._ready
        self._ready, self._pos = b"", 0
        return bytes(out)

    def encode_all(self, data, action=Action.FINISH):
        self.write(data)
        self.end(action)
        return self.read_all()


class ZlibEncoder(Inflater):
    """`ZlibEncoder` (src/zlib/encoder.rs:55-157)."""
    KIND = ZLIB


class GZipEncoder(Inflater):
    """`GZipEncoder` (src/gzip/encoder.rs:50-135)."""
    KIND = GZIP


def deflate_compress(data, kind=DEFLATE, device=0, dict_=b""):
    """One-shot over host buffers (df_encode_buffer / df_encode_buffer_dict)."""
    data, dict_ = bytes(data), bytes(dict_)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t(0)
    _check(lib().df_encode_buffer_dict(kind, device, data, len(data), dict_, len(dict_), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        lib().bz_free(out)


def deflate_bound(n):
    return lib().df_encode_bound(n)


_DECODER_VERDICTS = (BZ_E_DATA, BZ_E_MAGIC_FIRST, BZ_E_MAGIC)


class BZip2Decoder:
    """`BZip2Decoder` (src/bzip2/decoder.rs:583-612) over the C ABI's streaming context.  Input is
    handed over in chunks; the library decodes whatever records are complete once BZ_DEC_CHUNK bytes
    have come in and at the end, so bytes come out before the input is exhausted for long inputs."""

    CHUNK = 1 << 20  # bytes pulled from the input iterator per refill

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().bz_dec_create(C.byref(self._h), device))
        self._buf = (C.c_uint8 * 65536)()
        self._ready = b""
        self._pos = 0
        self._ended = False

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bz_dec_destroy(self._h)
            self._h = None

    def _refill(self):
        """bytes fetched; 0 = nothing ready (or, after the end, the clean end); raises the decoder's Err item"""
        k = lib().bz_dec_read(self._h, self._buf, len(self._buf))
        if k < 0:
            raise BZip2Error(k) if k in _DECODER_VERDICTS else CompressionError(k)
        self._ready = C.string_at(self._buf, k)
        self._pos = 0
        return k

    def _end(self):
        if not self._ended:
            self._ended = True
            rc = lib().bz_dec_end(self._h)
            if rc != BZ_OK and rc not in _DECODER_VERDICTS:
                raise CompressionError(rc)  # infrastructure (no GPU, memory ...)

    def next(self, it):
        """One `Decoder::next(iter)` call: an int byte, None at the end, raises BZip2Error for Err."""
        while self._pos >= len(self._ready):
            if self._refill():
                break
            if self._ended:
                return None
            chunk = bytearray()
            for b in it:  # (the reference pulls bytes on demand; the bytes are the same)
                chunk.append(b)
                if len(chunk) >= self.CHUNK:
                    break
            if chunk:
                _check(lib().bz_dec_write(self._h, bytes(chunk), len(chunk)))
            if len(chunk) < self.CHUNK:
                self._end()
        b = self._ready[self._pos]
        self._pos += 1
        return b

    # bulk helpers (same semantics, fewer Python-level calls)
    def write(self, data):
        data = bytes(data)
        if data:
            _check(lib().bz_dec_write(self._h, data, len(data)))

    def read_available(self):
        """decoded bytes that are ready now (raises the Err item once the bytes in front of it are out)"""
        out = bytearray(self._ready[self._pos:])
        self._ready, self._pos = b"", 0
        while True:
            try:
                if not self._refill():
                    break
            except BZip2Error as e:
                e.partial = bytes(out)
                raise
            out += self._ready
            self._ready, self._pos = b"", 0
        return bytes(out)

    def decode_all(self, data):
        """`data.decode(&mut self).collect::<Result<Vec<_>, _>>()`; BZip2Error.partial holds the bytes
        yielded before an Err."""
        self.write(data)
        self._end()
        return self.read_available()


def decode(iterable, decoder):
    """`DecodeExt::decode` / `DecodeIterator` (src/traits/decoder.rs:15-86)."""
    it = iter(iterable)
    while True:
        b = decoder.next(it)
        if b is None:
            return
        yield b


def decompress(data, device=0):
    """One-shot over host buffers (bz_decode_buffer) -> (bytes yielded, verdict code)."""
    data = bytes(data)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t(0)
    rc = lib().bz_decode_buffer(device, data, len(data), C.byref(out), C.byref(n))
    if rc != BZ_OK and rc not in _DECODER_VERDICTS:
        raise CompressionError(rc)
    try:
        return C.string_at(out, n.value), rc
    finally:
        lib().bz_free(out)


class GpuEngine:
    """Device-resident engine (section 2 of the C ABI).  Pointers are plain ints
    (e.g. torch.Tensor.data_ptr())."""

    STAGES = ("rle1_crc_split", "bwt", "mtf_zle", "huffman", "emit_assemble", "total")

    def __init__(self, device=0, max_blocks_in_flight=64):
        self._h = C.c_void_p()
        _check(lib().bz_gpu_engine_create(C.byref(self._h), device, max_blocks_in_flight))

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bz_gpu_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def set_verify(self, on=True):
        """Self-check of the device-resident calls (bz_gpu_engine_set_verify)."""
        _check(lib().bz_gpu_engine_set_verify(self._h, int(bool(on))))

    def verify_stats(self):
        s = (C.c_uint64 * 4)()
        _check(lib().bz_gpu_verify_stats(self._h, s))
        return dict(zip(BZip2Encoder.VERIFY_STATS, [int(x) for x in s]))

    def encode_device(self, level, d_in, n, d_out, cap):
        _settle()
        out_len = C.c_size_t(0)
        _check(lib().bz_gpu_encode_device(self._h, level, d_in, n, d_out, cap, C.byref(out_len)))
        return out_len.value

    def partition(self, level, d_in, n, mode=Action.FINISH):
        _settle()
        nb, cons, tail = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        _check(lib().bz_gpu_partition(self._h, level, d_in, n, int(mode), C.byref(nb), C.byref(cons), C.byref(tail)))
        return nb.value, cons.value, tail.value

    TILE = 4096  # input bytes per split tile

    def slab_begin(self, level, d_in, n, tile0, tile1):
        _settle()
        last = C.c_int64(-1)
        _check(lib().bz_gpu_partition_slab_begin(self._h, level, d_in, n, tile0, tile1, C.byref(last)))
        return last.value

    def slab_count(self, carry_run):
        _check(lib().bz_gpu_partition_slab_count(self._h, carry_run))

    def slab_finish(self, start_in, is_last):
        nb, nxt, tail = C.c_size_t(0), C.c_uint64(0), C.c_int(0)
        _check(lib().bz_gpu_partition_slab_finish(self._h, start_in, int(is_last), C.byref(nb), C.byref(nxt), C.byref(tail)))
        return nb.value, nxt.value, tail.value

    def encode_blocks(self, first, stride, n_blocks, d_packed, cap_words):
        _settle()
        # the library writes one entry per block of ITS last partition: size the arrays from that count
        have = lib().bz_gpu_block_count(self._h)
        if n_blocks != have:
            raise ValueError("encode_blocks: %d blocks asked for, the last partition made %d" % (n_blocks, have))
        k = max(0, (n_blocks - first + stride - 1) // stride) if n_blocks > first else 0
        woff = (C.c_uint64 * max(k, 1))()
        blen = (C.c_uint64 * max(k, 1))()
        crc = (C.c_uint32 * max(k, 1))()
        used = C.c_size_t(0)
        _check(lib().bz_gpu_encode_blocks(self._h, first, stride, d_packed, cap_words, woff, blen, crc, C.byref(used)))
        return list(woff[:k]), list(blen[:k]), list(crc[:k]), used.value

    def assemble(self, level, d_packed, word_off, bit_len, crc, d_out, cap, header=True, trailer=True, pad=True,
                 carry_bits=0, carry_byte=0, combined_crc=0):
        _settle()
        k = len(word_off)
        woff = (C.c_uint64 * max(k, 1))(*word_off)
        blen = (C.c_uint64 * max(k, 1))(*bit_len)
        crcs = (C.c_uint32 * max(k, 1))(*crc)
        out_len, comb = C.c_size_t(0), C.c_uint32(0)
        ocb, ocy = C.c_uint(0), C.c_uint(0)
        _check(lib().bz_gpu_assemble(self._h, level, k, d_packed, woff, blen, crcs, int(header), int(trailer), int(pad),
                                     carry_bits, carry_byte, combined_crc, C.byref(comb), d_out, cap,
                                     C.byref(out_len), C.byref(ocb), C.byref(ocy)))
        return out_len.value, comb.value, ocb.value, ocy.value

    def encode_sharded(self, level, d_in, n, comm, d_out, cap, packed=None, gather=None):
        """One rank of a multi-GPU encode (bz_gpu_encode_sharded).  `comm` carries a `struct` attribute
        holding a bz_shard_comm (sharded.TorchComm).  packed / gather: optional (pointer, words)
        device buffers for this rank's bit strings / everybody's on rank 0.  -> stream bytes (0 on ranks > 0)."""
        _settle()
        out_len = C.c_size_t(0)
        pp, pw = packed if packed else (None, 0)
        gp, gw = gather if gather else (None, 0)
        _check(lib().bz_gpu_encode_sharded(self._h, level, d_in, n, C.byref(comm.struct), pp, pw, gp, gw, d_out, cap,
                                           C.byref(out_len)))
        return out_len.value

    def encode_sharded_window(self, level, d_window, window_off, window_bytes, n, comm, d_out, cap, packed=None, gather=None):
        """encode_sharded for a rank that holds only the input bytes [window_off, window_off + window_bytes) of the
        n-byte input at d_window (shard_window() says which bytes a rank needs)."""
        _settle()
        out_len = C.c_size_t(0)
        pp, pw = packed if packed else (None, 0)
        gp, gw = gather if gather else (None, 0)
        _check(lib().bz_gpu_encode_sharded_window(self._h, level, d_window, window_off, window_bytes, n, C.byref(comm.struct),
                                                  pp, pw, gp, gw, d_out, cap, C.byref(out_len)))
        return out_len.value

    def shard_timings(self):
        """ms of the last sharded encode on this rank: waiting for the cut, this rank's link of the chain, gather, assembly"""
        t = (C.c_double * 4)()
        _check(lib().bz_gpu_last_shard_timings(self._h, t))
        return dict(zip(("wait_for_cut_ms", "chain_link_ms", "gather_ms", "assemble_ms"), [round(x, 3) for x in t]))

    def shard_phases(self):
        """shard_timings() and: entry -> ready for the cut (nothing of it waits for another rank's cuts), the cuts alone
        once the hop is there, the whole call"""
        t = (C.c_double * 8)()
        _check(lib().bz_gpu_last_shard_phases(self._h, t))
        return dict(zip(("wait_for_cut_ms", "chain_link_ms", "gather_ms", "assemble_ms", "before_the_cut_ms", "cuts_ms", "_", "call_ms"),
                        [round(x, 3) for x in t]))

    DEC_STAGES = ("scan_huffman", "mtf", "inverse_bwt", "rle1_crc", "total")

    def decode_device(self, d_in, n, d_out, cap):
        """-> (bytes decoded, verdict).  d_out = None: sizes only."""
        _settle()
        out_len = C.c_size_t(0)
        rc = lib().bz_gpu_decode_device(self._h, d_in, n, d_out, cap, C.byref(out_len))
        if rc != BZ_OK and rc not in _DECODER_VERDICTS:
            raise CompressionError(rc)
        return out_len.value, rc

    def decode_device_sharded(self, d_in, n, d_out, cap, rank, world, allgather):
        """One rank of a multi-GPU decode.  `allgather(send: bytes) -> bytes` returns the concatenation of
        every rank's `send`, in rank order (sharded.allgather_bytes wraps torch.distributed).
        -> (bytes in this rank's slice, offset of the slice in the decoded file, total bytes, verdict)"""
        _settle()
        def cb(_ctx, send, nbytes, recv):
            try:
                got = allgather(C.string_at(send, nbytes))
                if len(got) != nbytes * world:
                    return 1
                C.memmove(recv, got, len(got))
                return 0
            except Exception:  # the C side turns this into BZ_E_UNEXPECTED
                return 1
        fn = ALLGATHER_FN(cb)
        out_len, off, tot = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = lib().bz_gpu_decode_device_sharded(self._h, d_in, n, d_out, cap, rank, world, fn, None,
                                                C.byref(out_len), C.byref(off), C.byref(tot))
        if rc != BZ_OK and rc not in _DECODER_VERDICTS:
            raise CompressionError(rc)
        return out_len.value, off.value, tot.value, rc

    DEFLATE_STAGES = ("hash_chains", "matches", "parse", "blocks_tables", "emit_checksums", "total")

    def deflate_encode_device(self, kind, d_in, n, d_out, cap, dict_=b""):
        _settle()
        out_len = C.c_size_t(0)
        dict_ = bytes(dict_)
        _check(lib().df_gpu_encode_device_dict(self._h, kind, d_in, n, dict_, len(dict_), d_out, cap, C.byref(out_len)))
        return out_len.value

    def deflate_timings(self):
        t = (C.c_double * 6)()
        _check(lib().df_gpu_last_timings(self._h, t))
        return dict(zip(self.DEFLATE_STAGES, t))

    def deflate_stats(self):
        s = (C.c_uint64 * 8)()
        _check(lib().df_gpu_last_stats(self._h, s))
        return dict(blocks=s[0], stored=s[1], fixed=s[2], dynamic=s[3], limited_tables=s[4], stream_bytes=s[5],
                    dynamic_without_distances=s[6])

    def deflate_debug_codes(self, d_in, n):
        """numpy uint32 [count, 2] of (len, pos); len 0: literal pos"""
        import numpy as np
        buf = np.zeros((n + 1, 2), dtype=np.uint32)
        cnt = C.c_size_t(0)
        _check(lib().df_gpu_debug_codes(self._h, d_in, n, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n + 1, C.byref(cnt)))
        return buf[:cnt.value]

    def deflate_debug_blocks(self):
        cnt = C.c_size_t(0)
        _check(lib().df_gpu_debug_blocks(self._h, None, 0, C.byref(cnt)))
        buf = (C.c_uint64 * (4 * max(cnt.value, 1)))()
        _check(lib().df_gpu_debug_blocks(self._h, buf, cnt.value, C.byref(cnt)))
        return [tuple(buf[4 * i + k] for k in range(4)) for i in range(cnt.value)]

    def decode_timings(self):
        t = (C.c_double * 5)()
        _check(lib().bz_gpu_last_decode_timings(self._h, t))
        return dict(zip(self.DEC_STAGES, t))

    def decode_stats(self):
        s = (C.c_uint64 * 4)()
        _check(lib().bz_gpu_last_decode_stats(self._h, s))
        return dict(zip(("candidates", "blocks", "streams", "forced_blocks"), [int(x) for x in s]))

    def timings(self):
        t = (C.c_double * 6)()
        _check(lib().bz_gpu_last_timings(self._h, t))
        return dict(zip(self.STAGES, t))

    def profile(self, on=True):
        _check(lib().bz_gpu_profile_enable(self._h, int(on)))

    def kernel_profile(self):
        """{kernel: {launches, seconds, bytes}} accumulated since profile(True)."""
        out = {}
        for i in range(lib().bz_gpu_profile_kernels(self._h)):
            name, n, s, b = C.c_char_p(), C.c_uint64(0), C.c_double(0), C.c_uint64(0)
            _check(lib().bz_gpu_profile_get(self._h, i, C.byref(name), C.byref(n), C.byref(s), C.byref(b)))
            out[name.value.decode()] = {"launches": n.value, "seconds": s.value, "bytes": b.value}
        return out

    def cut_stats(self):
        """Partitions of this engine whose block cuts came from the candidate tables / that fell back to the chain kernel."""
        s = (C.c_uint64 * 2)()
        _check(lib().bz_gpu_cut_stats(self._h, s))
        return {"from_tables": int(s[0]), "fell_back": int(s[1])}

    def bwt_stats(self):
        s = (C.c_uint64 * 4)()
        _check(lib().bz_gpu_last_bwt_stats(self._h, s))
        r = (C.c_uint64 * 64)()
        _check(lib().bz_gpu_last_bwt_rounds(self._h, r))
        unordered = [int(x) for x in r]
        while unordered and unordered[-1] == 0:
            unordered.pop()
        return {"rounds": s[0], "resorted_elements": s[1], "batches": s[2], "unordered_after_round": unordered,
                "fused_fallbacks": s[3]}  # sorts of this engine that fell back to the three-kernel passes (it stays on them)

    def block_stats(self):
        n = C.c_size_t(0)
        _check(lib().bz_gpu_debug_block_stats(self._h, None, 0, C.byref(n)))
        buf = (C.c_uint32 * (8 * max(n.value, 1)))()
        _check(lib().bz_gpu_debug_block_stats(self._h, buf, n.value, C.byref(n)))
        keys = ("nblock", "block_crc", "orig_ptr", "mtf_count", "in_use_count", "group_num", "n_selectors")
        out = []
        for i in range(n.value):
            d = dict(zip(keys, buf[i * 8:i * 8 + 7]))
            d["max_len"] = buf[i * 8 + 7] & 0xFFFF
            d["lm_tables"] = buf[i * 8 + 7] >> 16
            out.append(d)
        return out

    def block_sections(self):
        """Per block of the last encode: the figures of the reference's debug lines "pass k: size is .., grp uses are .." and
        "bits: mapping .., selectors .., code lengths .., codes .." (src/bzip2/encoder.rs:483-498, :556-636)."""
        n = C.c_size_t(0)
        _check(lib().bz_gpu_debug_block_sections(self._h, None, 0, C.byref(n)))
        buf = (C.c_uint32 * (32 * max(n.value, 1)))()
        _check(lib().bz_gpu_debug_block_sections(self._h, buf, n.value, C.byref(n)))
        out = []
        for i in range(n.value):
            w = buf[i * 32:(i + 1) * 32]
            out.append({"pass_size": list(w[0:4]), "fave": [list(w[4 + 6 * k:10 + 6 * k]) for k in range(4)],
                        "bits_mapping": w[28], "bits_selectors": w[29], "bits_lengths": w[30], "bits_codes": w[31]})
        return out

    # stage probes
    def debug_bwt(self, block):
        block = bytes(block)
        sa = (C.c_uint32 * len(block))()
        _check(lib().bz_gpu_debug_bwt(self._h, block, len(block), sa))
        return list(sa)

    def debug_code_lengths(self, freq):
        n = len(freq)
        f = (C.c_uint32 * n)(*freq)
        out = (C.c_uint8 * n)()
        lm = C.c_int(0)
        _check(lib().bz_gpu_debug_code_lengths(self._h, f, n, out, C.byref(lm)))
        return list(out), bool(lm.value)
