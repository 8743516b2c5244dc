// micro-benchmark: cost of one "hop" (off += len[off]) for a lone wave, several formulations
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;

template <int MODE> __global__ __launch_bounds__(64) void k(const u32 *lens, u64 *out, u32 iters)
{
    __shared__ u32 s_len[64];
    const u32 l = threadIdx.x;
    u32 my = lens[(blockIdx.x * 64 + l) & 1023];
    s_len[l] = my;
    __syncthreads();
    u64 chain = 0; u32 off = 0; u32 total = 0;
    const u64 p0 = __ballot(my & 1), p1 = __ballot(my & 2), p2 = __ballot(my & 4), p3 = __ballot(my & 8);
    u64 t0 = clock64();
    for (u32 it = 0; it < iters; ++it) {
        off = 0;
        if (MODE == 0) {
            do { chain |= 1ull << off; off += (u32)__builtin_amdgcn_readlane((int)my, (int)off); } while (off < 64);
        } else if (MODE == 1) { // bit planes, scalar only
            do {
                chain |= 1ull << off;
                const u32 len = (u32)((p0 >> off) & 1) | ((u32)((p1 >> off) & 1) << 1) | ((u32)((p2 >> off) & 1) << 2) | ((u32)((p3 >> off) & 1) << 3);
                off += len;
            } while (off < 64);
        } else if (MODE == 2) { // LDS, lane 0 only
            if (l == 0) { do { chain |= 1ull << off; off += s_len[off]; } while (off < 64); }
        } else if (MODE == 3) { // LDS read by all lanes (uniform address), readfirstlane
            do { chain |= 1ull << off; off += (u32)__builtin_amdgcn_readfirstlane((int)s_len[off]); } while (off < 64);
        }
        total += (u32)__popcll(chain);
        chain = (chain >> 63);
    }
    u64 t1 = clock64();
    if (l == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = total; }
}

int main(int argc, char **argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 1024;
    const u32 iters = argc > 2 ? atoi(argv[2]) : 20000;
    u32 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 3 + (i * 7 % 3); // lens 3..5
    u32 *d; u64 *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, nblk * 16); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    u64 *ho = (u64 *)malloc(nblk * 16);
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(64), 0, 0, d, o, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(64), 0, 0, d, o, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(64), 0, 0, d, o, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(64), 0, 0, d, o, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(ho, o, nblk * 16, hipMemcpyDeviceToHost);
        const double hops = (double)ho[1];
        printf("mode %d blocks %d: %.3f ms, %.1f ns/hop, clock64 ticks/hop %.1f (hops/iter %.1f)\n", mode, nblk, ms, ms * 1e6 / hops, (double)ho[0] / hops, hops / iters);
    }
    return 0;
}
