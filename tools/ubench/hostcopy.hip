// hostcopy.hip -- what the copies of a host-buffer call cost on this box (pageable caller memory <-> device):
// one hipMemcpy against slices on several threads, fresh pages against touched ones, pinned staging + CPU copies.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/hostcopy tools/ubench/hostcopy.hip -lpthread ; run on the GPU box
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void *fresh(size_t n, bool huge)
{
    void *q = nullptr;
    if (posix_memalign(&q, (size_t)2 << 20, n) != 0) return nullptr;
    if (huge) madvise(q, n, MADV_HUGEPAGE);
    return q;
}
static void sliced(void *dst, const void *src, size_t n, hipMemcpyKind kind, int threads, size_t slice)
{
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    const size_t k = (n + slice - 1) / slice;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&] {
            hipStream_t st;
            hipSetDevice(0);
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= k) break;
                const size_t off = i * slice, len = std::min(slice, n - off);
                hipMemcpyAsync((char *)dst + off, (const char *)src + off, len, kind, st);
                hipStreamSynchronize(st);
            }
            hipStreamDestroy(st);
        });
    for (auto &t : th) t.join();
}
static void cpu_copy(void *dst, const void *src, size_t n, int threads)
{
    std::vector<std::thread> th;
    const size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] {
            const size_t off = (size_t)t * per;
            if (off < n) memcpy((char *)dst + off, (const char *)src + off, std::min(per, n - off));
        });
    for (auto &t : th) t.join();
}
int main()
{
    const size_t N = (size_t)1 << 30, M = (size_t)226 << 20;
    void *d = nullptr, *pin = nullptr;
    hipSetDevice(0);
    hipMalloc(&d, N);
    hipMemset(d, 7, N);
    hipHostMalloc(&pin, N, hipHostMallocPortable);
    memset(pin, 1, N);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        double t0;
        void *h;
        h = fresh(N, true); t0 = now(); hipMemcpy(h, d, N, hipMemcpyDeviceToHost); printf("D2H 1 GiB, one hipMemcpy, fresh huge pages: %.1f ms\n", now() - t0);
        t0 = now(); hipMemcpy(h, d, N, hipMemcpyDeviceToHost); printf("D2H 1 GiB, one hipMemcpy, touched pages:   %.1f ms\n", now() - t0); free(h);
        h = fresh(N, false); t0 = now(); hipMemcpy(h, d, N, hipMemcpyDeviceToHost); printf("D2H 1 GiB, one hipMemcpy, fresh 4K pages:    %.1f ms\n", now() - t0); free(h);
        t0 = now(); hipMemcpy(pin, d, N, hipMemcpyDeviceToHost); printf("D2H 1 GiB into pinned memory:               %.1f ms\n", now() - t0);
        for (int th : {2, 4, 8, 16}) {
            h = fresh(N, true); t0 = now(); sliced(h, d, N, hipMemcpyDeviceToHost, th, (size_t)8 << 20); printf("D2H 1 GiB, %2d threads x 8 MiB slices, fresh huge pages: %.1f ms\n", th, now() - t0);
            t0 = now(); sliced(h, d, N, hipMemcpyDeviceToHost, th, (size_t)8 << 20); printf("D2H 1 GiB, %2d threads x 8 MiB slices, touched pages:    %.1f ms\n", th, now() - t0); free(h);
        }
        for (int th : {1, 2, 4, 8, 16}) {
            h = fresh(N, true); t0 = now(); cpu_copy(h, pin, N, th); printf("memcpy pinned -> fresh huge pages, %2d threads: %.1f ms\n", th, now() - t0);
            t0 = now(); cpu_copy(h, pin, N, th); printf("memcpy pinned -> touched pages,    %2d threads: %.1f ms\n", th, now() - t0); free(h);
        }
        for (int th : {1, 4, 8}) {
            h = fresh(N, true); t0 = now();
            { std::vector<std::thread> v; const size_t per = N / th; for (int t = 0; t < th; ++t) v.emplace_back([=] { for (size_t o = 0; o < per; o += 4096) ((volatile char *)h)[(size_t)t * per + o] = 0; }); for (auto &x : v) x.join(); }
            printf("first touch of 1 GiB of huge-page memory, %d threads: %.1f ms\n", th, now() - t0); free(h);
        }
        h = malloc(M); memset(h, 3, M);
        t0 = now(); hipMemcpy(d, h, M, hipMemcpyHostToDevice); printf("H2D 226 MiB pageable, one hipMemcpy: %.1f ms\n", now() - t0);
        for (int th : {2, 4, 8}) { t0 = now(); sliced(d, h, M, hipMemcpyHostToDevice, th, (size_t)8 << 20); printf("H2D 226 MiB pageable, %d threads x 8 MiB: %.1f ms\n", th, now() - t0); }
        t0 = now(); hipMemcpy(d, pin, M, hipMemcpyHostToDevice); printf("H2D 226 MiB pinned: %.1f ms\n", now() - t0);
        free(h);
        h = malloc(N); memset(h, 3, N);
        t0 = now(); hipMemcpy(d, h, N, hipMemcpyHostToDevice); printf("H2D 1 GiB pageable, one hipMemcpy: %.1f ms\n", now() - t0);
        for (int th : {4, 8}) { t0 = now(); sliced(d, h, N, hipMemcpyHostToDevice, th, (size_t)8 << 20); printf("H2D 1 GiB pageable, %d threads x 8 MiB: %.1f ms\n", th, now() - t0); }
        free(h);
        printf("--\n");
    }
    return 0;
}
