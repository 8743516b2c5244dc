// copy_pool.h -- copies between PAGEABLE caller memory and the device BESIDE the kernels.
//
// The entry points over host buffers (bz_decode_buffer, bz_dec_*, df_encode_buffer) hand the library plain malloc'ed
// memory, and a hipMemcpy of such memory occupies the thread that calls it.  What the copies cost on an MI355X host
// (tools/ubench/hostcopy.hip, profiles/r05_host_copies.md):
//   * host -> device from touched pageable memory: 50 GB/s from ONE thread (1 GiB in 21 ms); slices on several threads
//     are SLOWER (4 threads: 35 ms, 8 threads: 46-55 ms -- the runtime serialises them and adds its overhead per call);
//   * device -> host into touched pageable memory: 19 ms per GiB from one thread (as fast as into pinned memory);
//     into FRESH memory 43-51 ms (huge pages) -- the kernel zeroes every page at its first touch, 39 ms per GiB on one
//     thread, but 6 ms on eight; slices on several threads: 80-120 ms.
// So: ONE thread per direction issues the copies, in order, slice by slice (a slice is what a caller can wait for), and
// a copy into fresh memory is preceded by a first touch of its pages on eight threads.  The thread
// that launches kernels goes on with the next batch or part meanwhile; a job is waited for by ticket before its device
// buffer is reused, everything before a call returns.  Threads start with the first job and end with the pool.
#pragma once
#include "engine_state.h"

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct CopyPool {
    struct Job {
        void *dst;
        const void *src;
        size_t n;
        hipMemcpyKind kind;
        bool fresh; // (device -> host) the destination's pages have not been touched yet
        size_t ticket;
        std::function<void(bool)> on_done; // (optional) called on the copying thread when the bytes have landed (or the copy failed)
    };
    int device = 0;
    std::thread io[2]; // [0] host -> device, [1] device -> host
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q[2];
    std::vector<char> done; // per ticket - base
    size_t base = 0;        // tickets below it are done and forgotten
    bool stop = false, failed_ = false;

    explicit CopyPool(int dev) : device(dev) {}
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : io)
            if (t.joinable()) t.join();
    }
    static unsigned touch_threads()
    {
        return 8u; // (eight threads touch a gigabyte of fresh pages in 6 ms; more are not faster: profiles/r05_host_copies.md)
    }
    static size_t slice_bytes() { return (size_t)32 << 20; }

    // first touch of [p, p + n): one write per 4 KiB page (a huge page is faulted in by its first one), on several threads
    static void touch(void *p, size_t n)
    {
        const unsigned nt = n >= ((size_t)8 << 20) ? touch_threads() : 1u;
        const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
        auto body = [=](unsigned t) {
            volatile char *c = static_cast<volatile char *>(p);
            const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
            for (size_t o = lo; o < hi; o += 4096) c[o] = 0;
            if (hi > lo) c[hi - 1] = 0;
        };
        if (nt == 1) return body(0);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(body, t);
        body(0);
        for (auto &x : th) x.join();
    }

    void run(int dir)
    {
        hipStream_t st = nullptr;
        bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q[dir].empty(); });
                if (q[dir].empty()) break; // stop, and nothing left
                j = std::move(q[dir].front());
                q[dir].pop_front();
            }
            if (j.fresh) touch(j.dst, j.n);
            const bool good = ok && (j.n == 0 || (hipMemcpyAsync(j.dst, j.src, j.n, j.kind, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess));
            if (j.on_done) j.on_done(good);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!good) failed_ = true;
                done[j.ticket - base] = 1;
            }
            cv.notify_all();
        }
        if (st) (void)hipStreamDestroy(st);
    }

    // queues dst[0, n) <- src[0, n) as ONE job (callers that want to wait for parts of a copy submit the parts); returns its ticket
    size_t submit(void *dst, const void *src, size_t n, hipMemcpyKind kind, bool fresh = false, std::function<void(bool)> on_done = nullptr)
    {
        const int dir = kind == hipMemcpyHostToDevice ? 0 : 1;
        size_t ticket;
        {
            std::lock_guard<std::mutex> lk(mu);
            ticket = base + done.size();
            done.push_back(0);
            Job j;
            j.dst = dst;
            j.src = src;
            j.n = n;
            j.kind = kind;
            j.fresh = fresh && dir == 1;
            j.ticket = ticket;
            j.on_done = std::move(on_done);
            q[dir].push_back(std::move(j));
            if (!io[dir].joinable()) io[dir] = std::thread([this, dir] { run(dir); });
        }
        cv.notify_all();
        return ticket;
    }
    void wait(size_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ticket < base || ticket - base >= done.size() || done[ticket - base]; });
    }
    void wait_all()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {
            for (char v : done)
                if (!v) return false;
            return true;
        });
        base += done.size(); // (a long-lived pool does not grow a list for ever)
        done.clear();
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(mu);
        return failed_;
    }
};
