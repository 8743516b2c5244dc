// copy_pool.h -- copies between PAGEABLE caller memory and the device on several threads, beside the kernels.
//
// The entry points over host buffers (bz_decode_buffer, bz_dec_*, df_encode_buffer) hand the library plain malloc'ed
// memory.  One hipMemcpy of such memory runs on the calling thread: the runtime pins the pages (fresh output pages are
// first faulted in and zeroed by the kernel -- a GiB of them costs more than its decode) and the caller waits.  Here a
// copy is cut into slices that a few threads take from a queue, each with a stream of its own: the page work of the
// slices runs side by side, and the thread that launches kernels goes on with the next batch meanwhile (a job is
// waited for by ticket before its device buffer is reused; everything is waited for before a call returns).
// BZ_COPY_THREADS in the environment (default 4, 1 ... 16).  Threads start with the first job and end with the pool.
#pragma once
#include "engine_state.h"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct CopyPool {
    struct Slice {
        void *dst;
        const void *src;
        size_t n;
        hipMemcpyKind kind;
        size_t ticket;
    };
    int device = 0;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Slice> q;
    std::vector<size_t> left; // per ticket - base: slices not yet done
    size_t base = 0;          // tickets below it are done and forgotten
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
        for (auto &t : threads) t.join();
    }
    static unsigned thread_count()
    {
        static const unsigned n = [] {
            const char *e = getenv("BZ_COPY_THREADS");
            long v = e ? atol(e) : 4;
            if (v < 1) v = 1;
            if (v > 16) v = 16;
            return (unsigned)v;
        }();
        return n;
    }
    static size_t slice_bytes() { return (size_t)8 << 20; }

    void run()
    {
        hipStream_t st = nullptr;
        bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (;;) {
            Slice s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) break; // stop, and nothing left
                s = q.front();
                q.pop_front();
            }
            const bool done = ok && hipMemcpyAsync(s.dst, s.src, s.n, s.kind, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!done) failed_ = true;
                left[s.ticket - base] -= 1;
            }
            cv.notify_all();
        }
        if (st) (void)hipStreamDestroy(st);
    }

    // queues dst[0, n) <- src[0, n); returns the job's ticket
    size_t submit(void *dst, const void *src, size_t n, hipMemcpyKind kind)
    {
        size_t ticket;
        {
            std::lock_guard<std::mutex> lk(mu);
            ticket = base + left.size();
            const size_t per = slice_bytes();
            const size_t k = (n + per - 1) / per;
            left.push_back(k);
            for (size_t i = 0; i < k; ++i) {
                Slice s;
                s.dst = static_cast<u8 *>(dst) + i * per;
                s.src = static_cast<const u8 *>(src) + i * per;
                s.n = (i + 1 == k) ? n - i * per : per;
                s.kind = kind;
                s.ticket = ticket;
                q.push_back(s);
            }
            if (threads.empty() && k) {
                const unsigned nt = (unsigned)std::min<size_t>(thread_count(), 16);
                for (unsigned t = 0; t < nt; ++t) threads.emplace_back([this] { run(); });
            }
        }
        cv.notify_all();
        return ticket;
    }
    void wait(size_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ticket < base || ticket - base >= left.size() || left[ticket - base] == 0; });
    }
    void wait_all()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {
            for (size_t v : left)
                if (v) return false;
            return true;
        });
        base += left.size(); // (a long-lived pool does not grow a list for ever)
        left.clear();
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(mu);
        return failed_;
    }
};
