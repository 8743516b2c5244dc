// hip_shim.h -- the slice of the HIP runtime API that csrc/capi.hip uses, on the HOST, so that the host pipeline
// (lanes, worker threads, drainer, staging buffers, resource cache) can run under ThreadSanitizer / AddressSanitizer
// on a box without a GPU (tests/test_host_pipeline_sanitize.py; SURVEY.md section 5, "ASan build of host code").
//
// Streams are real: every stream is a thread with a queue, hipMemcpyAsync / hipMemcpyPeerAsync / hipEventRecord /
// hipStreamWaitEvent are queued and run in stream order, hipStreamSynchronize / hipEventSynchronize wait.  So an
// ordering the code under test forgot -- reading a pinned buffer before the copy into it has finished, freeing a
// buffer a queued copy still reads, a clear on the wrong stream -- is a data race the sanitizer SEES, not a matter of
// GPU timing.  "Device memory" is host memory (malloc); four devices are reported so that the multi-device paths
// (peer copies, per-device lanes) run.  Test infrastructure only: nothing under rust-compression_amd/ includes it
// unless BZ_HOST_PIPELINE_TEST is defined by the test's own build line.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocPortable = 1, hipHostMallocDefault = 0 };

namespace hipshim {
constexpr int kDevices = 4;
inline int &current_device()
{
    static thread_local int d = 0;
    return d;
}
struct Event {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0; // records queued / records that have run
};
struct Stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread th;
    Stream()
    {
        th = std::thread([this] {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    f = std::move(q.front());
                    q.pop_front();
                    busy = true;
                }
                f();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    busy = false;
                }
                cv.notify_all();
            }
        });
    }
    ~Stream()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void push(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.empty() && !busy; });
    }
};
} // namespace hipshim

typedef hipshim::Stream *hipStream_t;
typedef hipshim::Event *hipEvent_t;

inline hipError_t hipGetDeviceCount(int *n) { *n = hipshim::kDevices; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= hipshim::kDevices) return hipErrorInvalidValue; hipshim::current_device() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = hipshim::current_device(); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "hip shim error"; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int a, int b) { *can = a != b; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipshim::Stream(); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { if (s) s->drain(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipshim::Event(); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memmove(dst, src, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s)
{
    if (!s) return hipMemcpy(dst, src, n, k);
    s->push([=] { memmove(dst, src, n); });
    return hipSuccess;
}
inline hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t n, hipStream_t s)
{
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s);
}
inline hipError_t hipMemset(void *dst, int v, size_t n) { memset(dst, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s)
{
    if (!s) { memset(dst, v, n); return hipSuccess; }
    s->push([=] { memset(dst, v, n); });
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    uint64_t ticket;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        ticket = ++e->recorded;
    }
    auto done = [e, ticket] {
        {
            std::lock_guard<std::mutex> lk(e->mu);
            if (e->completed < ticket) e->completed = ticket;
        }
        e->cv.notify_all();
    };
    if (s) s->push(done);
    else done();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lk(e->mu);
    const uint64_t want = e->recorded; // the last record queued so far
    e->cv.wait(lk, [&] { return e->completed >= want; });
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    uint64_t want;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        want = e->recorded;
    }
    auto wait = [e, want] {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->completed >= want; });
    };
    if (s) s->push(wait);
    else wait();
    return hipSuccess;
}
