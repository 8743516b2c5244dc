// roundtrip.hip -- what one "kernel -> a few bytes to the host -> next kernel" round trip costs, four ways (one block alone
// is a chain of ~25 of them):
//   a: hipMemcpyAsync to pageable memory + hipStreamSynchronize (what the engine does)
//   b: the same into pinned memory
//   c: a one-lane kernel copies the bytes into host-mapped pinned memory and raises a sequence word; the host polls the word
//   d: the producing kernel itself writes the mailbox and the word (no second launch)
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/roundtrip tools/ubench/roundtrip.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
__global__ void k_work(u64 *d, u32 it) // a short kernel whose result the host needs
{
    if (threadIdx.x == 0 && blockIdx.x == 0) d[0] = d[0] + it + 1;
}
__global__ void k_mail(volatile u64 *mail, volatile u32 *seq, const u64 *d, u32 s)
{
    mail[0] = d[0];
    __threadfence_system();
    *seq = s;
}
__global__ void k_work_mail(u64 *d, u32 it, volatile u64 *mail, volatile u32 *seq, u32 s)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        d[0] = d[0] + it + 1;
        mail[0] = d[0];
        __threadfence_system();
        *seq = s;
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st;
    hipStreamCreate(&st);
    u64 *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    u64 *pin;
    hipHostMalloc(&pin, 64, hipHostMallocDefault);
    u64 *mail;
    u32 *seq;
    hipHostMalloc(&mail, 64, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc(&seq, 64, hipHostMallocMapped | hipHostMallocCoherent);
    *seq = 0;
    u64 *dmail;
    u32 *dseq;
    hipHostGetDevicePointer((void **)&dmail, mail, 0);
    hipHostGetDevicePointer((void **)&dseq, seq, 0);
    const int N = 2000;
    u64 pageable = 0;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (u32)i);
            hipMemcpyAsync(&pageable, d, 8, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
        }
        double ta = (now() - t0) / N * 1e6;
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (u32)i);
            hipMemcpyAsync(pin, d, 8, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
        }
        double tb = (now() - t0) / N * 1e6;
        u32 s = *seq;
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (u32)i);
            ++s;
            hipLaunchKernelGGL(k_mail, dim3(1), dim3(1), 0, st, dmail, dseq, d, s);
            while (*(volatile u32 *)seq != s) {}
        }
        double tc = (now() - t0) / N * 1e6;
        t0 = now();
        for (int i = 0; i < N; ++i) {
            ++s;
            hipLaunchKernelGGL(k_work_mail, dim3(64), dim3(256), 0, st, d, (u32)i, dmail, dseq, s);
            while (*(volatile u32 *)seq != s) {}
        }
        double td = (now() - t0) / N * 1e6;
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (u32)i);
            hipStreamSynchronize(st);
        }
        double te = (now() - t0) / N * 1e6;
        t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (u32)i);
        hipStreamSynchronize(st);
        double tf = (now() - t0) / N * 1e6;
        printf("rep %d: a pageable copy + sync %.1f us | b pinned copy + sync %.1f | c mail kernel + poll %.1f | d own mail + poll %.1f | e kernel + sync (no data) %.1f | f back-to-back launches %.1f  (mail %llu)\n",
               rep, ta, tb, tc, td, te, tf, (unsigned long long)mail[0]);
    }
    return 0;
}
