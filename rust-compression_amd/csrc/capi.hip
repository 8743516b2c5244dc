// capi.hip -- section 1 of the C ABI: the streaming encoder context (== BZip2Encoder) and the
// one-shot host-buffer call, built on the device engine (engine.hip).
//
// The context replays the control flow of the reference byte iterator
// (src/bzip2/encoder.rs:74-159 BZip2Encoder::{next_bits,next}, :718-739 EncoderInner::{flush,
// finish}, :224-291 write_block) at bulk granularity:
//   * bz_enc_write  == the input iterator yields bytes        (:80-85)
//   * bz_enc_end(a) == the input iterator returns None, and `Encoder::next` is polled until it
//                      returns None                            (:86-110, :129-146)
// State carried between calls is exactly the reference's: pending input (the reference keeps
// it as rle_buffer/rle_count + block_buf, here: the raw bytes since the last block cut), the
// combined CRC, "has a block been written" (block_no == 1, :179,245), the BitWriter carry
// (src/bitio/writer.rs:165-169) and the finished / bit_finished toggles (:45-48).
// Quirks of the reference are kept on purpose (they are observable in the bytes): Flush zero
// pads mid-stream, an empty write_block still rotates the combined CRC (:237-238) and repeats
// the "BZh" header while block_no == 1 (:245).
#include "../../include/bz2_mi355x.h"
#ifdef BZ_HOST_PIPELINE_TEST
// tests/host_stub: this file alone, compiled by a host compiler under ThreadSanitizer / AddressSanitizer against a stub
// engine, with the HIP calls below served by a host shim whose streams are threads (tests/test_host_pipeline_sanitize.py)
#include "hip_shim.h"
namespace bzgpu {
typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
inline bool env_verify() { return false; } // (bzgpu.h: BZ_VERIFY; the stub engine has no self-check)
} // namespace bzgpu
#else
#include "bzgpu.h"
void dec_release_cached(); // dec_engine.hip: the engines bz_decode_buffer keeps between calls
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

using namespace bzgpu;

extern "C" const char *bz_strerror(int code)
{
    switch (code) {
    case BZ_OK: return "ok";
    case BZ_E_DATA: return "data integrity error in data";   // src/error.rs:37
    case BZ_E_EOF: return "file ends unexpectedly";          // src/error.rs:38
    case BZ_E_UNEXPECTED: return "unexpected error";         // src/error.rs:39
    case BZ_E_MAGIC_FIRST: return "bad magic number (first block)";
    case BZ_E_MAGIC: return "bad magic number";
    case BZ_E_PARAM: return "invalid parameter (level must be 1..=9)";
    case BZ_E_NOGPU: return "no usable gfx950 (MI355X) device: this library has no CPU path";
    case BZ_E_NOMEM: return "out of device or host memory";
    case BZ_E_CAPACITY: return "output buffer too small";
    default: return "unknown status";
    }
}

extern "C" const char *bz_version(void) { return "bz2_mi355x 0.1 (gfx950)"; }

// ---- the pipeline behind a context -------------------------------------------------------------
// Host bytes reach the GPUs in CHUNKS (BZ_ENC_CHUNK_MIB, default 192 MiB since round 4 -- half the device and pinned
// memory of 384 MiB for one per cent of the warm rate, and a first call that allocates half as much --, the first one of
// a stream 64 MiB: a
// large batch keeps the latency-bound stages -- one workgroup per block in the Huffman stage -- a small share
// of a job; measured on 1 GiB with one lane: 64 MiB chunks 4.6 GB/s host to host, 128 MiB 5.9, 256 MiB 6.5)
// through two pinned staging buffers: bz_enc_write copies the caller's bytes into the pinned buffer being
// filled (the only CPU copy; pieces of 4 MiB or more are split over threads, four per device) and, when it is
// full, starts its upload (hipMemcpyAsync on the lane's copy stream) into the device staging buffer of the LANE
// the chunk's job will run on and hands the chunk to that lane's worker thread.
//
// A context drives one or several devices (bz_enc_create_multi): every entry of the caller's device list gets
// TWO lanes (BZ_ENC_LANES; an engine with its own streams and buffers each), lane l lives on devices[l mod n_devices],
// and job q runs on lane q mod (2 n_devices) -- consecutive jobs go to different devices, and the latency-bound tail of
// one job's encode (Huffman: one workgroup per block) runs beside the bandwidth-bound sort of the job that
// shares its device.  A job puts the unconsumed tail of the previous job's input (it lies in the previous lane's
// buffer: a device-to-device copy, across xGMI with hipMemcpyPeerAsync when the lanes sit on different devices)
// and its chunk side by side in device memory, replays the reference's write_block calls for them on its engine,
// downloads the stream bytes through a pinned buffer and hands them to the drainer thread, which appends them
// to the output queue in job order.  So the upload of chunk k+1 (DMA) and the copy of chunk k+2 (CPU, caller's
// thread) run beside the encode of chunk k; bz_enc_read hands out whatever is complete.  bz_enc_end submits the
// partial chunk with the caller's Action and waits for that job, so that everything the reference's iterator
// would have yielded by then is readable.  One-shot calls (bz_encode_buffer[_multi]) run the same pipeline;
// engines and buffers of destroyed contexts are kept for the next one (a cache keyed by the device list; see
// bz_release_cached_resources), so a call does not pay hipMalloc / hipHostMalloc again.
struct Lane {
    int device = 0;
    bz_gpu_engine *g = nullptr;  // created when the lane gets its first job
    void *d_stage = nullptr;     // the chunk as uploaded
    size_t stage_cap = 0;
    void *d_buf = nullptr;       // composed input = tail of the previous job's input + the chunk
    size_t d_buf_cap = 0;
    void *d_out = nullptr;
    size_t d_out_cap = 0;
    void *d_packed = nullptr;
    size_t d_packed_cap = 0;
    u8 *h_out = nullptr;         // pinned: stream bytes on their way to the output queue
    size_t h_out_cap = 0;
    hipStream_t st_up = nullptr, st_io = nullptr;
    hipEvent_t ev_up = nullptr;  // the lane's last upload
};

struct EncResources {
    std::vector<int> devices;    // as the caller listed them (a device may appear more than once: more lanes on it)
    std::vector<Lane> lanes;     // 2 per entry of `devices`; lane l -> devices[l % devices.size()]
    size_t engine_blocks = 0;
    size_t chunk = 0;            // bytes per chunk (BZ_ENC_CHUNK_MIB)
    u8 *h_in[2] = {nullptr, nullptr};   // pinned (portable), sized by the chunks they have held (h_cap)
    size_t h_cap[2] = {0, 0};
    int h_lane[2] = {-1, -1};    // the lane a pinned buffer was uploaded to last (its ev_up tells when the buffer is free)
};

static bool enc_trace()
{
    static const bool v = getenv("BZ_ENC_TRACE") != nullptr;
    return v;
}
static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// The calling thread's current HIP device is the caller's business: the entry points below switch devices (lanes of a
// multi-device context live on different GPUs) and put the caller's device back when they return.  Worker threads
// keep their own.
struct CallerDevice {
    int dev = -1;
    CallerDevice() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~CallerDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
    CallerDevice(const CallerDevice &) = delete;
    CallerDevice &operator=(const CallerDevice &) = delete;
};

static std::mutex g_cache_mu;
static std::vector<EncResources *> g_cache; // resources of destroyed contexts, ready for reuse

static size_t enc_chunk_bytes()
{
#ifdef BZ_HOST_PIPELINE_TEST
    if (const char *b = getenv("BZ_ENC_CHUNK_BYTES")) // (sanitizer runs: small chunks, many jobs; read at every call)
        return (size_t)std::max(4096L, atol(b));
#endif
    static const size_t v = [] {
        const char *s = getenv("BZ_ENC_CHUNK_MIB");
        long mib = s ? atol(s) : 192;
        if (mib < 1) mib = 1;
        if (mib > 1024) mib = 1024;
        return (size_t)mib << 20;
    }();
    return v;
}

// the first chunk of a stream is small: the GPU has work sooner, and a short stream takes little pinned memory (16 ... 192 MiB
// measured in round 4, two- and three-step ramps and geometric job sizes in round 6: flat, profiles/r06_e2e_schedule.md)
static size_t enc_first_chunk_bytes() { return (size_t)64 << 20; }

// lanes per entry of a context's device list (BZ_ENC_LANES, 2 .. 8; consecutive jobs must run on different lanes: the
// tail of a job's input is handed from lane to lane).  Default 2: the latency-bound tail of one job beside the sort of
// another.  Measured in round 4 on 1 GiB, one device, warm bz_encode_buffer calls: 2 lanes 103.3 ms, 3 lanes 98.1 ms
// in a process of its own and 102.6 ms inside bench.py (the split sections queue behind two other lanes' sorts: 62 ms
// summed over the jobs instead of 20), 4 lanes 101.3 ms -- and a third lane is a third workspace: the first call of a
// process took 1.26 s with three lanes against 0.36 s with two.
static size_t enc_lanes_per_device()
{
    static const size_t v = [] {
        const char *s = getenv("BZ_ENC_LANES");
        const long k = s ? atol(s) : 2;
        return (size_t)(k < 2 ? 2 : (k > 8 ? 8 : k));
    }();
    return v;
}

static void resources_free(EncResources *r)
{
    if (!r) return;
    for (Lane &l : r->lanes) {
        (void)hipSetDevice(l.device);
        if (l.d_stage) (void)hipFree(l.d_stage);
        if (l.d_buf) (void)hipFree(l.d_buf);
        if (l.ev_up) (void)hipEventDestroy(l.ev_up);
        if (l.h_out) (void)hipHostFree(l.h_out);
        if (l.d_out) (void)hipFree(l.d_out);
        if (l.d_packed) (void)hipFree(l.d_packed);
        if (l.st_io) (void)hipStreamDestroy(l.st_io);
        if (l.st_up) (void)hipStreamDestroy(l.st_up);
        if (l.g) bz_gpu_engine_destroy(l.g);
    }
    for (int i = 0; i < 2; ++i)
        if (r->h_in[i]) (void)hipHostFree(r->h_in[i]);
    delete r;
}

static int resources_get(const std::vector<int> &devices, EncResources **out)
{
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size(); ++i)
            if (g_cache[i]->devices == devices && g_cache[i]->chunk == enc_chunk_bytes()) {
                *out = g_cache[i];
                g_cache.erase(g_cache.begin() + (ptrdiff_t)i);
                return BZ_OK;
            }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BZ_E_NOGPU;
    for (int d : devices)
        if (d < 0 || d >= ndev) return BZ_E_PARAM;
    EncResources *r = new EncResources();
    r->devices = devices;
    r->chunk = enc_chunk_bytes();
    // blocks in flight: a chunk of level-9 text is chunk / 0.9 MB blocks (lower levels and run-heavy
    // inputs take several batches)
    r->engine_blocks = r->chunk / 800000 + 16;
    r->lanes.resize(enc_lanes_per_device() * devices.size());
    bool ok = true;
    for (size_t l = 0; l < r->lanes.size() && ok; ++l) { // (engines and buffers come with the jobs: job_split, grow)
        Lane &ln = r->lanes[l];
        ln.device = devices[l % devices.size()];
        ok = ok && hipSetDevice(ln.device) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&ln.ev_up, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&ln.st_up, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&ln.st_io, hipStreamNonBlocking) == hipSuccess;
    }
    // the tail of a job's input crosses to the next lane's device: direct access where the fabric offers it
    // (hipMemcpyPeerAsync works without, through the host)
    for (size_t a = 0; a < devices.size() && ok; ++a)
        for (size_t b = 0; b < devices.size(); ++b) {
            int can = 0;
            if (devices[a] == devices[b] || hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) continue;
            if (hipSetDevice(devices[a]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[b], 0);
            (void)hipGetLastError(); // (already enabled: not an error worth keeping)
        }
    if (!ok) {
        resources_free(r);
        return BZ_E_NOMEM;
    }
    *out = r;
    return BZ_OK;
}

static void resources_put(EncResources *r)
{
    if (!r) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (g_cache.size() < 2) g_cache.push_back(r);
    else resources_free(r);
}

// What a finished context leaves behind for the next one -- engines with their batch workspace (about 31.5 MB per
// block of the largest job seen), device staging buffers, 2 x chunk of pinned host memory -- is released here.
// ---- diagnostic for hosts with several devices -------------------------------------------------------------------------
// What a context over `devices` does between consecutive lanes when a job's unconsumed tail changes device
// (job_split: hipMemcpyPeerAsync, across xGMI where peer access can be enabled, through the host where not), done once
// per neighbour pair devices[i] -> devices[i + 1 mod n] with a known pattern, there and back, and compared on the host.
// peer_access[i]: 1 direct access enabled, 0 not offered (the copy goes through the host), -1 both lanes on one device
// (nothing copied); out_ms[i]: the round trip's wall time, -1 if the pair failed.  BZ_OK when every pair's bytes came
// back intact; the first HIP error is printed with the pair it belongs to.  No reference counterpart (the reference has
// no devices); bench.py's preflight and INTEGRATION.md use it to make a first contact with a multi-GPU host diagnosable.
extern "C" int bz_peer_copy_selftest(const int *devices, int n_devices, size_t bytes, int *peer_access, double *out_ms)
{
    if (!devices || n_devices < 1 || bytes == 0 || bytes > ((size_t)1 << 30)) return BZ_E_PARAM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BZ_E_NOGPU;
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return BZ_E_PARAM;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    std::vector<uint8_t> pat(bytes), back(bytes);
    for (size_t i = 0; i < bytes; ++i) pat[i] = (uint8_t)(i * 131u + (i >> 8) * 7u + 5u);
    int rc = BZ_OK;
    for (int i = 0; i < n_devices; ++i) {
        const int a = devices[i], b = devices[(i + 1) % n_devices];
        if (peer_access) peer_access[i] = -1;
        if (out_ms) out_ms[i] = 0.0;
        if (a == b) continue;
        int can_ab = 0, can_ba = 0;
        (void)hipDeviceCanAccessPeer(&can_ab, a, b);
        (void)hipDeviceCanAccessPeer(&can_ba, b, a);
        if (can_ab && hipSetDevice(a) == hipSuccess) (void)hipDeviceEnablePeerAccess(b, 0);
        if (can_ba && hipSetDevice(b) == hipSuccess) (void)hipDeviceEnablePeerAccess(a, 0);
        (void)hipGetLastError(); // (already enabled: not an error worth keeping)
        if (peer_access) peer_access[i] = (can_ab && can_ba) ? 1 : 0;
        void *da = nullptr, *db = nullptr, *da2 = nullptr;
        hipStream_t sa = nullptr, sb = nullptr;
        hipError_t he = hipSuccess;
        const char *what = "hipSetDevice";
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        do {
            if ((he = hipSetDevice(a)) != hipSuccess) break;
            what = "hipMalloc";
            if ((he = hipMalloc(&da, bytes)) != hipSuccess || (he = hipMalloc(&da2, bytes)) != hipSuccess) break;
            what = "hipStreamCreate";
            if ((he = hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)) != hipSuccess) break;
            what = "hipMemcpy (host to device)";
            if ((he = hipMemcpy(da, pat.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess) break;
            if ((he = hipMemset(da2, 0, bytes)) != hipSuccess) break;
            what = "hipSetDevice";
            if ((he = hipSetDevice(b)) != hipSuccess) break;
            what = "hipMalloc";
            if ((he = hipMalloc(&db, bytes)) != hipSuccess) break;
            what = "hipStreamCreate";
            if ((he = hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)) != hipSuccess) break;
            // there: on a stream of the RECEIVING device, as job_split queues it (the lane that takes the tail owns the copy)
            what = "hipMemcpyPeerAsync (there)";
            if ((he = hipMemcpyPeerAsync(db, b, da, a, bytes, sb)) != hipSuccess) break;
            if ((he = hipStreamSynchronize(sb)) != hipSuccess) break;
            what = "hipMemcpyPeerAsync (back)";
            if ((he = hipSetDevice(a)) != hipSuccess) break;
            if ((he = hipMemcpyPeerAsync(da2, a, db, b, bytes, sa)) != hipSuccess) break;
            if ((he = hipStreamSynchronize(sa)) != hipSuccess) break;
            what = "hipMemcpy (device to host)";
            if ((he = hipMemcpy(back.data(), da2, bytes, hipMemcpyDeviceToHost)) != hipSuccess) break;
        } while (false);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        bool ok = he == hipSuccess;
        if (!ok) {
            fprintf(stderr, "bz2_mi355x: peer copy self-test, devices %d -> %d: %s failed: %s\n", a, b, what, hipGetErrorString(he));
            (void)hipGetLastError();
        } else if (memcmp(back.data(), pat.data(), bytes) != 0) {
            fprintf(stderr, "bz2_mi355x: peer copy self-test, devices %d -> %d -> %d: the bytes came back changed\n", a, b, a);
            ok = false;
        }
        if (sa && hipSetDevice(a) == hipSuccess) (void)hipStreamDestroy(sa);
        if (hipSetDevice(a) == hipSuccess) {
            if (da) (void)hipFree(da);
            if (da2) (void)hipFree(da2);
        }
        if (hipSetDevice(b) == hipSuccess) {
            if (sb) (void)hipStreamDestroy(sb);
            if (db) (void)hipFree(db);
        }
        if (out_ms) out_ms[i] = ok ? ms : -1.0;
        if (!ok) rc = BZ_E_UNEXPECTED;
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return rc;
}

extern "C" void bz_release_cached_resources(void)
{
    std::vector<EncResources *> all;
    const CallerDevice caller_device;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        all.swap(g_cache);
    }
    for (EncResources *r : all) resources_free(r);
#ifndef BZ_HOST_PIPELINE_TEST
    dec_release_cached(); // (the engines bz_decode_buffer keeps)
#endif
}

struct EncJob {
    u64 seq;         // number of the job in the stream; its lane is seq mod (number of lanes)
    int slot;        // pinned buffer that held the chunk (-1: no data, only the Action)
    size_t n;        // bytes of the chunk (uploaded to the lane's staging buffer)
    int mode;        // BZ_ACTION_*
    size_t tail_run; // length of the run of equal bytes at the chunk's end (n: the chunk is one run)
};

struct Drain {
    int lane;
    size_t bytes;
};

struct bz_enc {
    int level = 9;
    std::vector<int> devices;
    EncResources *r = nullptr;
    size_t n_lanes = 0;
    // output queue: a malloc'ed byte buffer (the drainer thread appends, bz_enc_read takes from the front;
    // realloc grows it -- the kernel remaps large blocks instead of copying them -- and the one-shot call
    // hands the buffer itself to its caller)
    std::mutex out_mu;
    u8 *out = nullptr;
    size_t out_len = 0, out_cap = 0, out_head = 0;
    // stream bytes leave a device through its lane's pinned buffer: the worker downloads a job's bytes (outside
    // the serial assembly section: the next job assembles meanwhile) and posts them; the drainer appends the
    // posts in job order
    std::thread drainer;
    std::map<u64, Drain> drains; // job number -> (lane, bytes); every job posts one (0 bytes: nothing came out)
    u64 drained = 0;             // jobs whose bytes are in the output queue (== the next job number the drainer takes)
    std::vector<char> drain_busy; // per lane: its pinned output buffer is waiting for the drainer
    // reference state (owned by the worker while jobs are in flight)
    bool finished = false;       // BZip2Encoder.finished      (encoder.rs:45)
    bool bit_finished = false;   // BZip2Encoder.bit_finished  (encoder.rs:48)
    bool inner_finished = false; // EncoderInner.finished      (encoder.rs:164), set by the worker that assembles Finish
    bool finish_submitted = false; // ... and the caller's copy: a Finish job has been handed over
    bool oneshot = false;          // bz_encode_buffer: the output queue is read only at the end
    bool reserve_stop = false;     // (out_mu) tells reserve_output to give up
    std::atomic<int> out_waiters{0}; // threads that want out_mu for an append or a read: reserve_output stands back for them
    bool any_block = false;      // block_no > 1               (encoder.rs:168,179)
    u32 combined_crc = 0;        // encoder.rs:167
    unsigned carry_bits = 0;     // BitWriter.counter          (writer.rs:167)
    unsigned carry_byte = 0;     // BitWriter.buf              (writer.rs:166)
    // input side (caller's thread)
    int fill_slot = 0;
    size_t fill = 0;
    u64 chunks_filled = 0;       // chunks handed over so far
    size_t chunk_bytes = 0;      // 0: the resources' chunk size; a one-shot call knows its length and balances them
    size_t reserve_blocks = 0;   // a one-shot call whose lanes will each see several jobs: their workspace is made once, at that size
    // Chain state 1 (handed from a job's SPLIT phase to the next job's): the unconsumed input lies in
    // lanes[tail_lane].d_buf, `tail_len` bytes at offset `tail_off`; finish_seen: a Finish job has been split
    int tail_lane = 0;
    size_t tail_off = 0, tail_len = 0;
    bool finish_seen = false;
    u64 split_done = 0; // jobs whose split phase is over
    u64 asm_done = 0;   // jobs whose assembly phase is over (chain state 2: carry bits, combined CRC, any_block)
    // workers: one per lane
    std::vector<std::thread> worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<EncJob> jobs;
    std::vector<u64> lane_uploads, lane_composed; // per lane: chunks uploaded to its staging buffer / released by its worker
    u64 queued = 0;              // jobs queued
    bool stop = false, drain_stop = false;
    int err = BZ_OK;             // sticky
    // where the time of a stream went, in ms (mu): [0] caller's copies into pinned memory, [1] caller waiting for a free
    // staging buffer, [2] SPLIT sections (serial from job to job), [3] ENCODE (side by side on the lanes), [4] ASSEMBLE
    // sections (serial), [5] downloads, [6] jobs, [7] workers waiting for their turn in the two serial sections
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool verify = false;         // bz_enc_set_verify / BZ_VERIFY=1: every job's blocks are decoded and compared before they leave
    u64 vstats[4] = {0, 0, 0, 0}; // (mu) blocks checked, jobs redone, redone jobs that failed again, ns spent checking
};

static int grow(void **p, size_t *cap, size_t want)
{
    if (want <= *cap) return BZ_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t sz = want + want / 4 + 4096;
    if (hipMalloc(p, sz) != hipSuccess) return BZ_E_NOMEM;
    *cap = sz;
    return BZ_OK;
}
static int grow_pinned(u8 **p, size_t *cap, size_t want)
{
    if (want <= *cap) return BZ_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t sz = want + want / 4 + 4096;
    if (hipHostMalloc((void **)p, sz, hipHostMallocPortable) != hipSuccess) return BZ_E_NOMEM;
    *cap = sz;
    return BZ_OK;
}

static inline u32 rotl1(u32 v) { return (v << 1) | (v >> 31); }

// Start of the pending chunk of the device input d[0..n): the run that is still open at its end, cut
// every 255 bytes from its start (encoder.rs:676-690).  `tail_run` bytes at the end are known to be
// equal (the caller scanned its chunk); only when that covers the whole chunk does the run reach
// back into bytes that exist on the device alone, which are then fetched in windows from the end.
static int pending_chunk_start(const u8 *d, size_t n, size_t chunk_n, size_t tail_run, size_t *start)
{
    *start = 0;
    if (n == 0) return BZ_OK;
    size_t run = tail_run < chunk_n ? tail_run : chunk_n;
    if (run == 0) run = 1;
    if (run >= chunk_n && n > run) {
        // the whole chunk (or no chunk at all) is one run: look at the bytes in front of it
        u8 b = 0;
        if (hipMemcpy(&b, d + n - 1, 1, hipMemcpyDeviceToHost) != hipSuccess) return BZ_E_UNEXPECTED;
        size_t pos = n - run; // bytes [pos, n) are equal to b
        std::vector<u8> w;
        size_t win = 65536;
        while (pos > 0) {
            const size_t k = pos < win ? pos : win;
            w.resize(k);
            if (hipMemcpy(w.data(), d + pos - k, k, hipMemcpyDeviceToHost) != hipSuccess) return BZ_E_UNEXPECTED;
            size_t i = k;
            while (i > 0 && w[i - 1] == b) --i;
            pos -= k - i;
            if (i > 0) break;
            win *= 4;
        }
        run = n - pos;
    }
    const size_t rs = n - run;
    const size_t q = n - 1 - rs;
    *start = n - 1 - (q % 255);
    return BZ_OK;
}

static void copy_in(u8 *dst, const u8 *src, size_t n, size_t n_devices);

// appends to the output queue (out_mu held by the caller)
static int out_append_locked(bz_enc *e, const u8 *p, size_t n)
{
    if (e->out_head && e->out_head == e->out_len) e->out_head = e->out_len = 0;
    if (e->out_len + n > e->out_cap) {
        if (e->out_head) { // make room by dropping what has been read
            memmove(e->out, e->out + e->out_head, e->out_len - e->out_head);
            e->out_len -= e->out_head;
            e->out_head = 0;
        }
        if (e->out_len + n > e->out_cap) {
            const size_t want = std::max(e->out_len + n, e->out_cap + e->out_cap / 2 + 4096);
            u8 *q = (u8 *)realloc(e->out, want);
            if (!q) return BZ_E_NOMEM;
            e->out = q;
            e->out_cap = want;
        }
    }
    copy_in(e->out + e->out_len, p, n, e->devices.size()); // (large pieces on several threads)
    e->out_len += n;
    return BZ_OK;
}

// One-shot call: nobody reads the output queue before the end, and the bytes of the last jobs are copied with
// nothing left to hide them behind.  First-touch page faults cost more than the copy itself (226 MB: ~40 ms
// against ~10), so room for the expected result is reserved and touched by a helper thread while the GPU
// works (in slices, under the queue's lock: an append may move the buffer in between).
static void reserve_output(bz_enc *e, size_t expect)
{
    // (1 MiB under the lock at a time, and none while the drainer waits for it: a std::mutex is not fair, and a helper
    // that takes it again at once kept the drainer -- and with it the lanes' downloads -- out until ALL of the room
    // was touched: 40 ms of a 1 GiB call, 120 ms of a 4 GiB one)
    constexpr size_t kSlice = (size_t)1 << 20;
    for (size_t pos = 0; pos < expect; pos += kSlice) {
        while (e->out_waiters.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
        std::lock_guard<std::mutex> lk(e->out_mu);
        if (e->reserve_stop) return;
        if (e->out_cap < expect) {
            u8 *q = (u8 *)realloc(e->out, expect);
            if (!q) return;
            e->out = q;
            e->out_cap = expect;
        }
        const size_t a = std::max(pos, e->out_len), b = std::min(pos + kSlice, e->out_cap);
        volatile u8 *t = e->out;
        for (size_t i = a; i < b; i += 4096) t[i] = 0;
    }
}

static void drainer_main(bz_enc *e)
{
    for (;;) {
        Drain d;
        {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [&] { return e->drain_stop || e->drains.count(e->drained) != 0; });
            auto it = e->drains.find(e->drained);
            if (it == e->drains.end()) return; // stop (the workers have gone: nothing more will be posted)
            d = it->second;
            e->drains.erase(it);
        }
        int rc = BZ_OK;
        if (d.bytes) {
            e->out_waiters.fetch_add(1, std::memory_order_acq_rel);
            std::lock_guard<std::mutex> lk(e->out_mu);
            e->out_waiters.fetch_sub(1, std::memory_order_acq_rel);
            rc = out_append_locked(e, e->r->lanes[(size_t)d.lane].h_out, d.bytes);
        }
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (rc != BZ_OK && e->err == BZ_OK) e->err = rc;
            if (d.bytes) e->drain_busy[(size_t)d.lane] = 0;
            e->drained += 1;
        }
        e->cv.notify_all();
    }
}

// ---- one job = (optional chunk of input) + the Action it arrived under, in three phases -----------
// A job replays in bulk the write_block calls the reference would have made for its Action over the
// input that is pending (src/bzip2/encoder.rs:671-739, :224-291):
//   SPLIT     compose the device input (unconsumed tail of the previous job + the chunk), RLE1 + block
//             split on the lane's engine.  Serial from job to job: it needs the previous job's tail and
//             hands its own on (chain state 1).
//   ENCODE    the blocks (rotation sort, MTF, Huffman, bit emission).  Independent of every other job:
//             as many jobs as there are lanes are in this phase at the same time.
//   ASSEMBLE  concatenate the block bit strings behind the stream so far (BitWriter carry, combined
//             CRC, "has a block been written": chain state 2).  Serial from job to job.
//   DOWNLOAD  device -> the lane's pinned buffer -> (drainer thread, in job order) the output queue.
struct JobState {
    const u8 *d = nullptr;
    size_t n_all = 0, n_eff = 0, n_blocks = 0, consumed = 0;
    size_t out_len = 0;
    int tail = 0;
    int mode = BZ_ACTION_RUN; // the job's Action as EncoderInner sees it: behind a Finish, flush() and finish() do nothing any more
    std::vector<uint64_t> woff, blen;
    std::vector<uint32_t> crc;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
};

static void chunk_composed(bz_enc *e, int lane)
{
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->lane_composed[(size_t)lane] += 1; // the lane's staging buffer may take the next upload
    }
    e->cv.notify_all();
}

static int job_split(bz_enc *e, const EncJob &j, int lane, JobState &js)
{
    EncResources *r = e->r;
    Lane &ln = r->lanes[(size_t)lane];
    // Input that arrives after Finish goes on being collected as the reference does it: EncoderInner::next has no
    // `finished` test (encoder.rs:671-697), so a block is written -- BEHIND the trailer, without a header unless no block
    // has been written yet -- whenever another 100000 * level - 19 bytes have come together, while flush() and finish()
    // do nothing any more (:718-739): every later job is a Run job, whatever the caller's Action.  (Such output is no
    // .bz2 stream; it is what the reference yields, tests/test_gpu_parity.py::test_streaming_write_after_finish.)
    js.mode = e->finish_seen ? BZ_ACTION_RUN : j.mode;
    if (!ln.g) { // a lane's engine is created when the lane gets its first job
        const int rc = bz_gpu_engine_create(&ln.g, ln.device, r->engine_blocks);
        if (rc != BZ_OK) return rc;
    }
    if (e->reserve_blocks && j.seq < (u64)e->n_lanes) { // (the lane's first job of a one-shot call that has more for it)
        const int rc = bz_gpu_engine_reserve(ln.g, e->reserve_blocks);
        if (rc != BZ_OK) return rc;
    }
    js.t0 = now_ms();
    js.n_all = e->tail_len + j.n;
    if (js.n_all) {
        // compose: the unconsumed tail of the previous job's input, then the new chunk
        int rc = grow(&ln.d_buf, &ln.d_buf_cap, js.n_all + 64);
        if (rc != BZ_OK) return rc;
        hipStream_t st = ln.st_io;
        // (consecutive jobs run on different lanes, so the tail always comes from another lane's buffer; that job
        // may still be encoding -- it only reads its RLE1 image by then)
        if (e->tail_len && e->tail_lane == lane) return BZ_E_UNEXPECTED;
        if (e->tail_len) {
            const Lane &from = r->lanes[(size_t)e->tail_lane];
            const u8 *src = (const u8 *)from.d_buf + e->tail_off;
            const hipError_t ce = from.device == ln.device
                                      ? hipMemcpyAsync(ln.d_buf, src, e->tail_len, hipMemcpyDeviceToDevice, st)
                                      : hipMemcpyPeerAsync(ln.d_buf, ln.device, src, from.device, e->tail_len, st); // xGMI
            if (ce != hipSuccess) return BZ_E_UNEXPECTED;
        }
        if (j.n) {
            if (hipStreamWaitEvent(st, ln.ev_up, 0) != hipSuccess) return BZ_E_UNEXPECTED;
            if (hipMemcpyAsync((u8 *)ln.d_buf + e->tail_len, ln.d_stage, j.n, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return BZ_E_UNEXPECTED;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return BZ_E_UNEXPECTED;
        if (j.n) chunk_composed(e, lane);
        js.d = (const u8 *)ln.d_buf;
    }
    js.n_eff = js.n_all;
    int rc;
    if (js.mode != BZ_ACTION_FINISH && (rc = pending_chunk_start(js.d, js.n_all, j.n, j.tail_run, &js.n_eff)) != BZ_OK)
        return rc;
    if (js.n_eff > 0) {
        rc = bz_gpu_partition(ln.g, e->level, js.d, js.n_eff, js.mode, &js.n_blocks, &js.consumed, &js.tail);
        if (rc != BZ_OK) return rc;
    }
    js.t1 = now_ms();
    // the bytes that went into blocks; what is left is the next job's tail
    size_t drop;
    if (js.mode == BZ_ACTION_RUN) drop = js.n_blocks ? js.consumed : 0;
    else if (js.mode == BZ_ACTION_FLUSH) drop = js.n_eff;
    else drop = js.n_all;
    if (js.n_all) { // (a job without any input leaves the tail where it is)
        e->tail_lane = lane;
        e->tail_off = drop;
        e->tail_len = js.n_all - drop;
    }
    if (js.mode == BZ_ACTION_FINISH) e->finish_seen = true;
    return BZ_OK;
}

static int job_encode(bz_enc *e, int lane, JobState &js)
{
    Lane &ln = e->r->lanes[(size_t)lane];
    if (js.n_blocks == 0) {
        js.t2 = now_ms();
        return BZ_OK;
    }
    js.woff.resize(js.n_blocks);
    js.blen.resize(js.n_blocks);
    js.crc.resize(js.n_blocks);
    const size_t cap_words = bz_encode_bound(js.n_eff) / 4 + 2 * js.n_blocks + 16;
    int rc = grow(&ln.d_packed, &ln.d_packed_cap, cap_words * 4);
    if (rc != BZ_OK) return rc;
    size_t used = 0;
    uint64_t v0[4] = {0, 0, 0, 0}, v1[4] = {0, 0, 0, 0};
    (void)bz_gpu_engine_set_verify(ln.g, e->verify ? 1 : 0); // (a lane's engine may come from the cache: the switch is the context's)
    (void)bz_gpu_verify_stats(ln.g, v0);
    rc = bz_gpu_encode_blocks(ln.g, 0, 1, ln.d_packed, cap_words, js.woff.data(), js.blen.data(), js.crc.data(), &used);
    (void)bz_gpu_verify_stats(ln.g, v1);
    if (e->verify) {
        std::lock_guard<std::mutex> lk(e->mu);
        for (int i = 0; i < 4; ++i) e->vstats[i] += v1[i] - v0[i];
    }
    js.t2 = now_ms();
    return rc;
}

static int job_assemble(bz_enc *e, const EncJob &j, int lane, JobState &js)
{
    Lane &ln = e->r->lanes[(size_t)lane];
    (void)j;
    const int mode = js.mode;
    const size_t n_blocks = js.n_blocks;
    if (mode == BZ_ACTION_RUN && n_blocks == 0) return BZ_OK; // no write_block call happened

    // The flush()/finish() call itself sees an empty block_buf when every byte went into
    // blocks closed by a cut (or there was nothing at all).
    const bool final_call_empty =
        (mode == BZ_ACTION_FLUSH && !(n_blocks > 0 && js.tail)) || (mode == BZ_ACTION_FINISH && n_blocks == 0);
    u32 comb = e->combined_crc;
    const int write_header = e->any_block ? 0 : 1; // block_no == 1 (encoder.rs:245)
    if (final_call_empty && n_blocks == 0) comb = rotl1(comb) ^ 0u; // encoder.rs:237-238, crc of nothing = 0
    const int trailer = (mode == BZ_ACTION_FINISH) ? 1 : 0;

    size_t bits_bound = 0;
    for (size_t k = 0; k < n_blocks; ++k) bits_bound += (size_t)js.blen[k];
    const size_t out_cap = bits_bound / 8 + 64;
    int rc;
    if (!ln.g && (rc = bz_gpu_engine_create(&ln.g, ln.device, e->r->engine_blocks)) != BZ_OK) return rc; // (an Action-only first job)
    if ((rc = grow(&ln.d_out, &ln.d_out_cap, out_cap)) != BZ_OK) return rc;
    unsigned ocb = 0, ocy = 0;
    u32 comb_out = comb;
    rc = bz_gpu_assemble(ln.g, e->level, n_blocks, ln.d_packed, js.woff.data(), js.blen.data(), js.crc.data(), write_header,
                         trailer, 0, e->carry_bits, e->carry_byte, comb, &comb_out, ln.d_out, ln.d_out_cap, &js.out_len, &ocb,
                         &ocy);
    if (rc != BZ_OK) return rc;
    e->carry_bits = ocb;
    e->carry_byte = ocy;
    if (final_call_empty && n_blocks > 0) comb_out = rotl1(comb_out); // the extra, empty write_block(false)
    e->combined_crc = comb_out;
    if (n_blocks > 0) e->any_block = true;
    if (mode == BZ_ACTION_FINISH) e->inner_finished = true;
    return BZ_OK;
}

// device -> the lane's pinned buffer (free again once the drainer has appended the lane's previous bytes)
static int job_download(bz_enc *e, int lane, JobState &js)
{
    Lane &ln = e->r->lanes[(size_t)lane];
    if (js.out_len == 0) return BZ_OK;
    {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return !e->drain_busy[(size_t)lane]; });
    }
    int rc;
    if ((rc = grow_pinned(&ln.h_out, &ln.h_out_cap, js.out_len)) != BZ_OK) return rc;
    if (hipMemcpyAsync(ln.h_out, ln.d_out, js.out_len, hipMemcpyDeviceToHost, ln.st_io) != hipSuccess ||
        hipStreamSynchronize(ln.st_io) != hipSuccess)
        return BZ_E_UNEXPECTED;
    return BZ_OK;
}

// worker of one lane: takes the jobs whose number is the lane's modulo the number of lanes
static void worker_main(bz_enc *e, int lane)
{
    const u64 nl = (u64)e->n_lanes;
    for (;;) {
        EncJob j;
        {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [&] { return e->stop || (!e->jobs.empty() && (int)(e->jobs.front().seq % nl) == lane); });
            if (e->jobs.empty() || (int)(e->jobs.front().seq % nl) != lane) return; // stop
            j = e->jobs.front();
            e->jobs.pop_front();
        }
        e->cv.notify_all(); // (another lane may find its job at the front now)
        (void)hipSetDevice(e->r->lanes[(size_t)lane].device);
        JobState js;
        int rc;
        // SPLIT, in job order
        const double tw0 = now_ms();
        {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [&] { return e->split_done == j.seq; });
            rc = e->err;
        }
        const double tw1 = now_ms();
        if (rc == BZ_OK) rc = job_split(e, j, lane, js);
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (rc != BZ_OK && e->err == BZ_OK) e->err = rc;
            // nobody waits for a staging buffer of a failed context
            if (rc != BZ_OK && j.n) e->lane_composed[(size_t)lane] = e->lane_uploads[(size_t)lane];
            e->split_done += 1;
        }
        e->cv.notify_all();
        // ENCODE, beside the other lanes'
        if (rc == BZ_OK) rc = job_encode(e, lane, js);
        // ASSEMBLE, in job order
        const double tw2 = now_ms();
        {
            std::unique_lock<std::mutex> lk(e->mu);
            if (rc != BZ_OK && e->err == BZ_OK) e->err = rc;
            e->cv.wait(lk, [&] { return e->asm_done == j.seq; });
            if (rc == BZ_OK) rc = e->err;
        }
        const double tw3 = now_ms();
        js.t2 = tw3; // (the assembly section begins when its turn has come)
        if (rc == BZ_OK) rc = job_assemble(e, j, lane, js);
        js.t3 = now_ms();
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (rc != BZ_OK && e->err == BZ_OK) e->err = rc;
            e->asm_done += 1;
        }
        e->cv.notify_all();
        // DOWNLOAD beside the next job's assembly; the drainer appends in job order
        if (rc == BZ_OK) rc = job_download(e, lane, js);
        const double t4 = now_ms();
        if (enc_trace())
            fprintf(stderr, "bz_enc job %llu (lane %d, device %d): mode %d, %zu bytes, %zu blocks: split %.2f ms, encode %.2f ms, "
                            "assemble %.2f ms, download %.2f ms (at %.1f)\n",
                    (unsigned long long)j.seq, lane, e->r->lanes[(size_t)lane].device, j.mode, js.n_all, js.n_blocks, js.t1 - js.t0,
                    tw2 - js.t1, js.t3 - tw3, t4 - js.t3, now_ms());
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (rc != BZ_OK && e->err == BZ_OK) e->err = rc;
            if (js.t1 > 0) {
                e->phase_ms[2] += js.t1 - js.t0;
                e->phase_ms[3] += tw2 - js.t1;
                e->phase_ms[4] += js.t3 - tw3;
                e->phase_ms[5] += t4 - js.t3;
                e->phase_ms[7] += (tw1 - tw0) + (tw3 - tw2);
            }
            e->phase_ms[6] += 1;
            Drain d;
            d.lane = lane;
            d.bytes = rc == BZ_OK ? js.out_len : 0;
            if (d.bytes) e->drain_busy[(size_t)lane] = 1;
            e->drains[j.seq] = d;
        }
        e->cv.notify_all();
    }
}

static int ensure_started(bz_enc *e)
{
    if (e->r) return BZ_OK;
    int rc = resources_get(e->devices, &e->r);
    if (rc != BZ_OK) return rc;
    e->n_lanes = e->r->lanes.size();
    e->drain_busy.assign(e->n_lanes, 0);
    e->lane_uploads.assign(e->n_lanes, 0);
    e->lane_composed.assign(e->n_lanes, 0);
    for (size_t l = 0; l < e->n_lanes; ++l) e->worker.emplace_back(worker_main, e, (int)l);
    e->drainer = std::thread(drainer_main, e);
    return BZ_OK;
}

// hands the chunk being filled (possibly empty) to its lane's worker together with `mode`
static int submit(bz_enc *e, int mode, bool wait)
{
    EncResources *r = e->r;
    EncJob j;
    j.seq = e->queued; // (only the caller's thread queues jobs) jobs are numbered in stream order
    j.slot = -1;
    j.n = e->fill;
    j.mode = mode;
    j.tail_run = 0;
    const int lane = (int)(j.seq % (u64)e->n_lanes);
    if (e->fill) {
        Lane &ln = r->lanes[(size_t)lane];
        const int s = e->fill_slot;
        const u8 *h = r->h_in[s];
        size_t run = 1;
        while (run < e->fill && h[e->fill - 1 - run] == h[e->fill - 1]) ++run;
        j.tail_run = run;
        j.slot = s;
        {
            // the lane's device staging buffer is free once its worker has composed the chunk that used it last
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [&] { return e->lane_composed[(size_t)lane] >= e->lane_uploads[(size_t)lane] || e->err != BZ_OK; });
            if (e->err != BZ_OK) return e->err;
        }
        if (enc_trace())
            fprintf(stderr, "bz_enc upload: %zu bytes from pinned buffer %d to lane %d (device %d) (at %.1f)\n", e->fill, s, lane,
                    ln.device, now_ms());
        if (hipSetDevice(ln.device) != hipSuccess) return BZ_E_UNEXPECTED;
        {
            const int grc = grow(&ln.d_stage, &ln.stage_cap, e->fill + 64);
            if (grc != BZ_OK) return grc;
        }
        if (hipMemcpyAsync(ln.d_stage, h, e->fill, hipMemcpyHostToDevice, ln.st_up) != hipSuccess ||
            hipEventRecord(ln.ev_up, ln.st_up) != hipSuccess)
            return BZ_E_UNEXPECTED;
        r->h_lane[s] = lane;
        e->fill_slot ^= 1;
        e->fill = 0;
        e->chunks_filled += 1;
    }
    u64 ticket;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (j.n) e->lane_uploads[(size_t)lane] += 1;
        e->jobs.push_back(j);
        ticket = ++e->queued;
    }
    e->cv.notify_all();
    if (wait) { // until the job is done and its bytes are in the output queue
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->drained >= ticket; });
        return e->err;
    }
    return BZ_OK;
}

extern "C" int bz_enc_create_multi(bz_enc **out, int level, const int *devices, int n_devices)
{
    if (!out) return BZ_E_PARAM;
    *out = nullptr;
    if (level < 1 || level > 9) return BZ_E_PARAM; // the reference panics (encoder.rs:59-61)
    if (!devices || n_devices < 1 || n_devices > 64) return BZ_E_PARAM;
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0) return BZ_E_PARAM;
    bz_enc *e = new bz_enc();
    e->level = level;
    e->devices.assign(devices, devices + n_devices);
    e->verify = bzgpu::env_verify();
    *out = e;
    return BZ_OK;
}

extern "C" int bz_enc_set_verify(bz_enc *e, int on)
{
    if (!e) return BZ_E_PARAM;
    std::lock_guard<std::mutex> lk(e->mu);
    e->verify = on != 0; // (jobs already handed to a lane keep the setting they started with)
    return BZ_OK;
}

extern "C" int bz_enc_verify_stats(bz_enc *e, uint64_t out[4])
{
    if (!e || !out) return BZ_E_PARAM;
    std::lock_guard<std::mutex> lk(e->mu);
    for (int i = 0; i < 4; ++i) out[i] = e->vstats[i];
    return BZ_OK;
}

extern "C" int bz_enc_create(bz_enc **out, int level, int device) { return bz_enc_create_multi(out, level, &device, 1); }

extern "C" void bz_enc_destroy(bz_enc *e)
{
    if (!e) return;
    if (e->r) {
        const CallerDevice caller_device;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            e->stop = true;
        }
        e->cv.notify_all();
        for (auto &w : e->worker)
            if (w.joinable()) w.join();
        {
            std::lock_guard<std::mutex> lk(e->mu);
            e->drain_stop = true;
        }
        e->cv.notify_all();
        if (e->drainer.joinable()) e->drainer.join();
        for (Lane &l : e->r->lanes) {
            (void)hipSetDevice(l.device);
            (void)hipStreamSynchronize(l.st_up);
            (void)hipStreamSynchronize(l.st_io);
        }
        if (e->err == BZ_OK) resources_put(e->r);
        else resources_free(e->r);
    }
    free(e->out);
    delete e;
}

static void copy_in(u8 *dst, const u8 *src, size_t n, size_t n_devices)
{
    // the one CPU copy of the input: caller's memory -> pinned staging; large pieces on several threads (four per
    // device of the context: the copy has to keep every device's upload fed)
    constexpr size_t kPar = (size_t)4 << 20;
    if (n < kPar) {
        memcpy(dst, src, n);
        return;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw >= 8 ? 4 : (hw >= 4 ? 2 : 1);
    if (n_devices > 1) nt = std::min<size_t>(nt * n_devices, std::max<size_t>(hw / 2, 1));
    nt = std::min(nt, n / ((size_t)1 << 20));
    if (nt <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n / nt + 4095) & ~(size_t)4095;
    for (size_t t = 1; t < nt; ++t) {
        const size_t a = t * per, b = (t + 1 == nt) ? n : (t + 1) * per;
        if (a < n) th.emplace_back([=] { memcpy(dst + a, src + a, (b < n ? b : n) - a); });
    }
    memcpy(dst, src, per < n ? per : n);
    for (auto &t : th) t.join();
}

extern "C" int bz_enc_write(bz_enc *e, const uint8_t *in, size_t n)
{
    if (!e || (!in && n)) return BZ_E_PARAM;
    if (n == 0) return e->err;
    const CallerDevice caller_device;
    int rc = ensure_started(e);
    if (rc != BZ_OK) return rc;
    EncResources *r = e->r;
    double t_copy = 0, t_wait = 0;
    while (n) {
        // (the first chunk of a stream is small -- 64 MiB at most: the GPU has work sooner, and a short stream
        // takes little pinned memory; the staging buffers grow with the chunks they hold)
        const size_t later = e->chunk_bytes ? e->chunk_bytes : r->chunk;
        const size_t cap = e->chunks_filled == 0 ? std::min(later, enc_first_chunk_bytes()) : later;
        if (e->fill == cap) {
            // A full chunk goes to the worker when MORE input arrives (its complete blocks are encoded
            // while the caller goes on writing); the last chunk of a stream is left for bz_enc_end, which
            // sends it together with the caller's Action instead of paying a job for the tail block alone.
            const double ts0 = now_ms();
            if ((rc = submit(e, BZ_ACTION_RUN, false)) != BZ_OK) return rc;
            t_wait += now_ms() - ts0;
        }
        if (e->fill == 0 && r->h_lane[e->fill_slot] >= 0) {
            // the pinned buffer of this slot is free once its last upload has completed
            const double ts0 = now_ms();
            const Lane &ul = r->lanes[(size_t)r->h_lane[e->fill_slot]];
#ifdef BZ_HOST_PIPELINE_TEST
            // (the sanitizer run's own check: BZ_TEST_HOST_RACE=1 drops this wait, and ThreadSanitizer must then report
            // the caller's copy into a pinned buffer racing with the upload that still reads it)
            if (getenv("BZ_TEST_HOST_RACE")) r->h_lane[e->fill_slot] = -1;
            else
#endif
            if (hipSetDevice(ul.device) != hipSuccess || hipEventSynchronize(ul.ev_up) != hipSuccess) return BZ_E_UNEXPECTED;
            r->h_lane[e->fill_slot] = -1;
            t_wait += now_ms() - ts0;
        }
        if (e->fill + std::min(n, cap - e->fill) > r->h_cap[e->fill_slot]) {
            // room for what this call adds (short streams stay small), for a whole chunk once it is half full
            size_t want = e->fill + std::min(n, cap - e->fill);
            want = std::max(want, 2 * r->h_cap[e->fill_slot]); // (geometric: many small writes, few reallocations)
            // (a one-shot call's balanced chunks are shorter than the resources' chunk size; the buffers still get that
            // size, or the next context, which fills whole chunks, would have to replace them: 2 x 384 MiB of pinned
            // memory cost 1.5 s)
            if (want > cap / 2) want = e->chunk_bytes ? std::max(cap, r->chunk) : cap;
            u8 *nh = nullptr;
            if (hipSetDevice(r->lanes[0].device) != hipSuccess ||
                hipHostMalloc((void **)&nh, want, hipHostMallocPortable) != hipSuccess)
                return BZ_E_NOMEM;
            if (e->fill) memcpy(nh, r->h_in[e->fill_slot], e->fill);
            if (r->h_in[e->fill_slot]) (void)hipHostFree(r->h_in[e->fill_slot]);
            r->h_in[e->fill_slot] = nh;
            r->h_cap[e->fill_slot] = want;
        }
        const size_t k = std::min(n, cap - e->fill);
        const double tc0 = now_ms();
        copy_in(r->h_in[e->fill_slot] + e->fill, in, k, e->devices.size());
        t_copy += now_ms() - tc0;
        e->fill += k;
        in += k;
        n -= k;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    e->phase_ms[0] += t_copy;
    e->phase_ms[1] += t_wait;
    return e->err;
}

extern "C" int bz_enc_phase_stats(bz_enc *e, double out_ms[8])
{
    if (!e || !out_ms) return BZ_E_PARAM;
    std::lock_guard<std::mutex> lk(e->mu);
    for (int i = 0; i < 8; ++i) out_ms[i] = e->phase_ms[i];
    return BZ_OK;
}

static std::mutex g_last_mu;
static double g_last_phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int bz_encode_buffer_last_phases(double out_ms[8])
{
    if (!out_ms) return BZ_E_PARAM;
    std::lock_guard<std::mutex> lk(g_last_mu);
    for (int i = 0; i < 8; ++i) out_ms[i] = g_last_phases[i];
    return BZ_OK;
}

extern "C" int bz_enc_end(bz_enc *e, int action)
{
    if (!e || action < BZ_ACTION_RUN || action > BZ_ACTION_FINISH) return BZ_E_PARAM;
    const CallerDevice caller_device;
    int rc = ensure_started(e);
    if (rc != BZ_OK) return rc;
    // (The chunk still being filled goes to the workers with the caller's Action WITHOUT waiting for the
    // jobs in flight first -- its upload and split run beside them; `finish_submitted` is the caller's
    // copy of EncoderInner.finished, which a worker sets only when the Finish job is assembled.  A job
    // that is waited for has every earlier job in front of it done, too: assembly is in job order.)
    // blocks the reference would already have emitted while it was consuming the input
    // (behind a Finish every Action is a Run for EncoderInner: flush() and finish() do nothing, but the bytes the iterator
    // yielded have gone through EncoderInner::next -- encoder.rs:80-85 -- and the blocks they completed come out)
    if (action == BZ_ACTION_RUN ? !e->finish_submitted || e->fill : e->finish_submitted && e->fill) {
        if ((rc = submit(e, BZ_ACTION_RUN, true)) != BZ_OK) return rc;
    }
    for (;;) {
        // next_bits: queue empty and the iterator is exhausted (encoder.rs:86-110)
        if (!e->finished) {
            if (action == BZ_ACTION_FLUSH && !e->finish_submitted) {       // :718-727
                if ((rc = submit(e, BZ_ACTION_FLUSH, true)) != BZ_OK) return rc;
            } else if (action == BZ_ACTION_FINISH && !e->finish_submitted) { // :729-739
                e->finish_submitted = true;
                if ((rc = submit(e, BZ_ACTION_FINISH, true)) != BZ_OK) return rc;
            } else {
                // nothing to encode: the state below must still be the workers' last word
                std::unique_lock<std::mutex> lk(e->mu);
                e->cv.wait(lk, [&] { return e->drained >= e->queued; });
                if (e->err != BZ_OK) return e->err;
            }
            e->finished = true;
        }
        e->finished = false; // ... and the next poll returns None (:87-89)
        // Encoder::next saw None from next_bits (:129-146)
        if (e->bit_finished) {
            e->bit_finished = false;
            break;
        }
        if (action == BZ_ACTION_RUN) break;
        e->bit_finished = true;
        if (e->carry_bits == 0) break; // writer.flush() -> None (writer.rs:226-242)
        {
            std::lock_guard<std::mutex> lk(e->out_mu);
            const u8 b = (u8)e->carry_byte;
            if ((rc = out_append_locked(e, &b, 1)) != BZ_OK) return rc;
        }
        e->carry_bits = 0;
        e->carry_byte = 0;
    }
    return BZ_OK;
}

extern "C" size_t bz_enc_pending(const bz_enc *e)
{
    if (!e) return 0;
    bz_enc *m = const_cast<bz_enc *>(e);
    std::lock_guard<std::mutex> lk(m->out_mu);
    return e->out_len - e->out_head;
}

extern "C" long bz_enc_read(bz_enc *e, uint8_t *out, size_t cap)
{
    if (!e || (!out && cap)) return BZ_E_PARAM;
    {
        std::lock_guard<std::mutex> lk(e->out_mu);
        const size_t avail = e->out_len - e->out_head;
        const size_t k = avail < cap ? avail : cap;
        if (k) memcpy(out, e->out + e->out_head, k);
        e->out_head += k;
        if (e->out_head == e->out_len) e->out_head = e->out_len = 0;
        if (k) return (long)k;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    return e->err != BZ_OK ? (long)e->err : 0;
}

extern "C" void bz_free(void *p) { free(p); }

extern "C" int bz_encode_buffer_multi(int level, const int *devices, int n_devices, const uint8_t *in, size_t in_len,
                                      uint8_t **out, size_t *out_len)
{
    if (!out || !out_len || (!in && in_len)) return BZ_E_PARAM;
    *out = nullptr;
    *out_len = 0;
    if (level < 1 || level > 9) return BZ_E_PARAM;
    bz_enc *e = nullptr;
    int rc = bz_enc_create_multi(&e, level, devices, n_devices);
    if (rc != BZ_OK) return rc;
    e->oneshot = true;
    std::thread reserver;
    if (in_len >= ((size_t)8 << 20)) reserver = std::thread(reserve_output, e, in_len / 3 + ((size_t)1 << 20));
    // the same pipeline as the streaming context: chunks are uploaded and encoded while the rest of the
    // input is still being copied to the pinned staging buffers
    {
        // Chunks of equal size, and a whole number of rounds over the lanes behind the first one: jobs go round the
        // lanes, and a last chunk that is shorter than the others (or one too many) runs alone at the end with the rest
        // of the GPUs idle (1 GiB on one device in 64 + 384 + 384 + 192 MiB: the last 22 ms).  With several devices the
        // chunks get smaller rather than leave lanes without a job, down to 32 MiB (a job of 40 blocks still fills a
        // device's 256 CUs in the sort, and the per-job latencies of the tail stages run side by side).
        const size_t most = enc_chunk_bytes(), first = std::min(most, enc_first_chunk_bytes());
        const size_t lanes = enc_lanes_per_device() * (size_t)n_devices, least = std::min(most, (size_t)32 << 20);
        if (in_len > first + most || (n_devices > 1 && in_len > first + 2 * least)) {
            const size_t rest = in_len - first;
            size_t k = (rest + most - 1) / most;
            const size_t per = enc_lanes_per_device();
            k = (k + per - 1) / per * per; // (a whole number of rounds over one device's lanes)
            if (n_devices > 1) {
                const size_t k0 = k;
                k = (k + lanes - 1) / lanes * lanes;
                while (k > k0 && rest / k < least) k -= per;
            }
            e->chunk_bytes = (((rest + k - 1) / k) + 4095) & ~(size_t)4095;
            // every lane gets a job of that size sooner or later (the first lane's first job is the small first chunk):
            // its workspace is made for it at once -- level-9 text is ~0.9 MB a block; run-heavy or low-level inputs make
            // more blocks and grow the workspace as before
            if (k + 1 > lanes) e->reserve_blocks = e->chunk_bytes / 880000 + 4;
        }
    }
    if (in_len) rc = bz_enc_write(e, in, in_len);
    if (rc == BZ_OK) rc = bz_enc_end(e, BZ_ACTION_FINISH);
    if (reserver.joinable()) {
        {
            std::lock_guard<std::mutex> lk(e->out_mu);
            e->reserve_stop = true;
        }
        reserver.join();
    }
    if (rc == BZ_OK) {
        std::lock_guard<std::mutex> lk(e->out_mu);
        // the output queue's buffer itself goes to the caller (nothing has been read from it)
        if (!e->out) e->out = (u8 *)malloc(1);
        if (!e->out) {
            rc = BZ_E_NOMEM;
        } else {
            if (e->out_head) memmove(e->out, e->out + e->out_head, e->out_len - e->out_head);
            *out = e->out;
            *out_len = e->out_len - e->out_head;
            e->out = nullptr;
            e->out_len = e->out_cap = e->out_head = 0;
        }
    }
    {
        double ph[8];
        (void)bz_enc_phase_stats(e, ph);
        std::lock_guard<std::mutex> lk(g_last_mu);
        for (int i = 0; i < 8; ++i) g_last_phases[i] = ph[i];
    }
    bz_enc_destroy(e);
    return rc;
}

extern "C" int bz_encode_buffer(int level, int device, const uint8_t *in, size_t in_len, uint8_t **out, size_t *out_len)
{
    return bz_encode_buffer_multi(level, &device, 1, in, in_len, out, out_len);
}
