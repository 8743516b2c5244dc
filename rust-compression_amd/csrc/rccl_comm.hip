// rccl_comm.hip -- a ready-made transport for bz_gpu_encode_sharded / bz_gpu_decode_device_sharded: the four
// callbacks of struct bz_shard_comm (include/bz2_mi355x.h) over RCCL, i.e. over xGMI inside a node.
// Built as its own shared library (libbz2_mi355x_rccl.so, links librccl) so that the codec library does not
// depend on RCCL; a host that brings its own transport (MPI, torch.distributed) never loads it.
//
//   rank 0:      bz_rccl_unique_id(id)            -> ship the 128 bytes to the other ranks by any means
//   every rank:  bz_rccl_comm_create(&comm, id, rank, world, device)   (collective: ncclCommInitRank)
//                bz_gpu_encode_sharded(engine, level, d_in, n, comm, ...)
//                bz_rccl_comm_destroy(comm)
//
// The small HOST-memory exchanges (8-24 bytes per rank or block) travel through a pinned host buffer and a
// device staging buffer (RCCL moves device memory); the one large exchange -- every rank's packed bit strings
// to rank 0 -- is a single group of ncclSend / ncclRecv straight between the callers' device buffers, so the
// peers' data arrive over their own xGMI links side by side.
#include "../../include/bz2_mi355x.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

struct RcclComm {
    bz_shard_comm api; // first member: a bz_shard_comm* is the RcclComm*
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    int device = 0;
    unsigned char *h_pin = nullptr; // pinned staging for the host-byte exchanges
    unsigned char *d_small = nullptr;
    size_t small_cap = 0;
};

bool ok_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    fprintf(stderr, "bz2_mi355x_rccl: HIP error %s in %s\n", hipGetErrorString(e), what);
    return false;
}
bool ok_nccl(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return true;
    fprintf(stderr, "bz2_mi355x_rccl: RCCL error %s in %s\n", ncclGetErrorString(r), what);
    return false;
}

// Every exchange ends with a wait for the transport's stream.  A peer that died (or never arrives) would leave
// hipStreamSynchronize waiting for ever, so the wait polls: the stream, the communicator's asynchronous error
// state, and a deadline (BZ_RCCL_TIMEOUT_S, default 300 s).  On error or timeout the communicator is aborted
// (ncclCommAbort), this callback and every later one fail, and bz_gpu_encode_sharded returns BZ_E_UNEXPECTED
// instead of hanging.
double timeout_seconds()
{
    static const double v = [] {
        const char *s = getenv("BZ_RCCL_TIMEOUT_S");
        const double t = s ? atof(s) : 300.0;
        return t > 0 ? t : 300.0;
    }();
    return v;
}

bool wait_stream(RcclComm *c, const char *what)
{
    using clock = std::chrono::steady_clock;
    const clock::time_point t0 = clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(c->st);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) return ok_hip(q, what);
        (void)hipGetLastError(); // (hipErrorNotReady is sticky for hipGetLastError otherwise)
        if (spins < 20000) continue; // the small exchanges of the cut chain take microseconds: spin first
        ncclResult_t ar = ncclSuccess;
        const bool broken = ncclCommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress;
        const double waited = std::chrono::duration<double>(clock::now() - t0).count();
        if (broken || waited > timeout_seconds()) {
            fprintf(stderr, "bz2_mi355x_rccl: rank %d: %s did not complete (%s after %.1f s): communicator aborted\n", c->api.rank, what,
                    broken ? ncclGetErrorString(ar) : "timeout", waited);
            (void)ncclCommAbort(c->comm);
            c->comm = nullptr;
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

// room for `bytes` bytes per rank, send + receive side
int ensure_small(RcclComm *c, size_t bytes)
{
    const size_t want = bytes * (size_t)(c->api.world + 1) + 256;
    if (want <= c->small_cap) return 0;
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_small) (void)hipFree(c->d_small);
    c->h_pin = nullptr;
    c->d_small = nullptr;
    c->small_cap = 0;
    const size_t cap = want * 2;
    if (!ok_hip(hipHostMalloc((void **)&c->h_pin, cap, hipHostMallocDefault), "hipHostMalloc")) return 1;
    if (!ok_hip(hipMalloc((void **)&c->d_small, cap), "hipMalloc")) return 1;
    c->small_cap = cap;
    return 0;
}

int cb_allgather(void *ctx, const void *send, size_t bytes, void *recv)
{
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (!c->comm || !ok_hip(hipSetDevice(c->device), "hipSetDevice") || ensure_small(c, bytes)) return 1;
    const size_t world = (size_t)c->api.world;
    memcpy(c->h_pin, send, bytes);
    unsigned char *d_send = c->d_small, *d_recv = c->d_small + ((bytes + 255) & ~(size_t)255);
    if (!ok_hip(hipMemcpyAsync(d_send, c->h_pin, bytes, hipMemcpyHostToDevice, c->st), "H2D")) return 1;
    if (!ok_nccl(ncclAllGather(d_send, d_recv, bytes, ncclUint8, c->comm, c->st), "ncclAllGather")) return 1;
    if (!ok_hip(hipMemcpyAsync(c->h_pin, d_recv, bytes * world, hipMemcpyDeviceToHost, c->st), "D2H")) return 1;
    if (!wait_stream(c, "all-gather")) return 1;
    memcpy(recv, c->h_pin, bytes * world);
    return 0;
}

int cb_send(void *ctx, int dst, const void *buf, size_t bytes)
{
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (!c->comm || !ok_hip(hipSetDevice(c->device), "hipSetDevice") || ensure_small(c, bytes)) return 1;
    memcpy(c->h_pin, buf, bytes);
    if (!ok_hip(hipMemcpyAsync(c->d_small, c->h_pin, bytes, hipMemcpyHostToDevice, c->st), "H2D")) return 1;
    if (!ok_nccl(ncclSend(c->d_small, bytes, ncclUint8, dst, c->comm, c->st), "ncclSend")) return 1;
    return wait_stream(c, "send") ? 0 : 1;
}

int cb_recv(void *ctx, int src, void *buf, size_t bytes)
{
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (!c->comm || !ok_hip(hipSetDevice(c->device), "hipSetDevice") || ensure_small(c, bytes)) return 1;
    if (!ok_nccl(ncclRecv(c->d_small, bytes, ncclUint8, src, c->comm, c->st), "ncclRecv")) return 1;
    if (!ok_hip(hipMemcpyAsync(c->h_pin, c->d_small, bytes, hipMemcpyDeviceToHost, c->st), "D2H")) return 1;
    if (!wait_stream(c, "receive")) return 1;
    memcpy(buf, c->h_pin, bytes);
    return 0;
}

int cb_gatherv(void *ctx, const void *d_send, size_t send_bytes, void *d_recv, const uint64_t *recv_off,
               const uint64_t *recv_bytes)
{
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (!c->comm || !ok_hip(hipSetDevice(c->device), "hipSetDevice")) return 1;
    const int rank = c->api.rank, world = c->api.world;
    if (rank == 0 && send_bytes &&
        !ok_hip(hipMemcpyAsync(static_cast<unsigned char *>(d_recv) + recv_off[0], d_send, send_bytes, hipMemcpyDeviceToDevice,
                               c->st),
                "D2D"))
        return 1;
    bool fine = ok_nccl(ncclGroupStart(), "ncclGroupStart");
    if (rank == 0) {
        for (int r = 1; r < world && fine; ++r)
            if (recv_bytes[r])
                fine = ok_nccl(ncclRecv(static_cast<unsigned char *>(d_recv) + recv_off[r], recv_bytes[r], ncclUint8, r, c->comm, c->st),
                               "ncclRecv");
    } else if (send_bytes) {
        fine = fine && ok_nccl(ncclSend(d_send, send_bytes, ncclUint8, 0, c->comm, c->st), "ncclSend");
    }
    fine = ok_nccl(ncclGroupEnd(), "ncclGroupEnd") && fine;
    return (fine && wait_stream(c, "gather of the bit strings")) ? 0 : 1;
}

} // namespace

extern "C" int bz_rccl_unique_id(uint8_t id[BZ_RCCL_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= BZ_RCCL_ID_BYTES, "id buffer");
    if (!id) return BZ_E_PARAM;
    ncclUniqueId u;
    if (!ok_nccl(ncclGetUniqueId(&u), "ncclGetUniqueId")) return BZ_E_UNEXPECTED;
    memset(id, 0, BZ_RCCL_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return BZ_OK;
}

extern "C" int bz_rccl_comm_create(bz_shard_comm **out, const uint8_t id[BZ_RCCL_ID_BYTES], int rank, int world, int device)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return BZ_E_PARAM;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BZ_E_NOGPU;
    if (device < 0 || device >= ndev) return BZ_E_PARAM;
    if (!ok_hip(hipSetDevice(device), "hipSetDevice")) return BZ_E_UNEXPECTED;
    RcclComm *c = new RcclComm();
    c->device = device;
    c->api.ctx = c;
    c->api.rank = rank;
    c->api.world = world;
    c->api.allgather = cb_allgather;
    c->api.send = cb_send;
    c->api.recv = cb_recv;
    c->api.gatherv = cb_gatherv;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    if (!ok_hip(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking), "hipStreamCreate") ||
        !ok_nccl(ncclCommInitRank(&c->comm, world, u, rank), "ncclCommInitRank") || ensure_small(c, 4096)) {
        bz_rccl_comm_destroy(&c->api);
        return BZ_E_UNEXPECTED;
    }
    *out = &c->api;
    return BZ_OK;
}

extern "C" int bz_rccl_comm_count(const bz_shard_comm *comm)
{
    if (!comm) return BZ_E_PARAM;
    const RcclComm *c = static_cast<const RcclComm *>(comm->ctx);
    int n = 0;
    if (!c->comm || !ok_nccl(ncclCommCount(c->comm, &n), "ncclCommCount")) return BZ_E_UNEXPECTED;
    return n;
}

extern "C" void bz_rccl_comm_destroy(bz_shard_comm *comm)
{
    if (!comm) return;
    RcclComm *c = static_cast<RcclComm *>(comm->ctx);
    (void)hipSetDevice(c->device);
    if (c->st && c->comm) (void)hipStreamSynchronize(c->st); // (an aborted communicator's stream may never drain)
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}
