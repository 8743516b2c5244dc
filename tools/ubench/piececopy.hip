// piececopy.hip -- a streaming caller's pieces: what one bz_dec_write / bz_dec_read of k bytes costs as a memcpy through a
// host buffer against a hipMemcpy straight to / from the device (pageable caller memory, one thread).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/piececopy tools/ubench/piececopy.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)256 << 20;
    void *d = nullptr, *pin = nullptr;
    hipSetDevice(0);
    hipMalloc(&d, N);
    hipMemset(d, 7, N);
    hipHostMalloc(&pin, N, hipHostMallocPortable);
    memset(pin, 1, N);
    char *src = (char *)malloc(N); // the caller's compressed file (touched)
    memset(src, 3, N);
    char *seg = nullptr; // a landed segment (touched)
    posix_memalign((void **)&seg, (size_t)2 << 20, N);
    madvise(seg, N, MADV_HUGEPAGE);
    memset(seg, 5, N);
    char *vec = (char *)malloc(N); // the context's chunk buffer (touched: recycled)
    memset(vec, 0, N);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t piece : {(size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20}) {
            char *sink = (char *)malloc(piece); // the caller's read buffer, reused
            memset(sink, 0, piece);
            double t0 = now();
            for (size_t o = 0; o < N; o += piece) memcpy(vec + o, src + o, piece);
            const double w_cpu = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) hipMemcpy((char *)d + o, src + o, piece, hipMemcpyHostToDevice);
            const double w_dev = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) { hipMemcpyAsync((char *)d + o, src + o, piece, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
            const double w_dev_st = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) memcpy(sink, seg + o, piece);
            const double r_cpu = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) memcpy(sink, (char *)pin + o, piece);
            const double r_pin = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) hipMemcpy(sink, (char *)d + o, piece, hipMemcpyDeviceToHost);
            const double r_dev = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += piece) { hipMemcpyAsync(sink, (char *)d + o, piece, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
            const double r_dev_st = now() - t0;
            printf("piece %6zu KiB, 256 MiB: write memcpy %.1f ms, hipMemcpy H2D %.1f, async+sync %.1f | read memcpy from segment %.1f, from pinned %.1f, hipMemcpy D2H %.1f, async+sync %.1f\n",
                   piece >> 10, w_cpu, w_dev, w_dev_st, r_cpu, r_pin, r_dev, r_dev_st);
            free(sink);
        }
        printf("--\n");
    }
    return 0;
}
