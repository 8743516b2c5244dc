// micro-benchmark: random 4-byte scatter / gather rate into XCD-local windows of different sizes (how much of
// the 4 MiB L2 of an XCD a randomly accessed array may take before the rate drops)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE> __global__ __launch_bounds__(512) void k(u32 *buf, u32 nwin, u32 tiles, u32 win_elems, u32 *sink)
{
    const u32 lid = blockIdx.x;
    const u32 x = lid & 7u, slot = lid >> 3;
    const u32 win = (slot / tiles) * 8u + x, tile = slot % tiles;
    if (win >= nwin) return;
    u32 *w = buf + (size_t)win * win_elems;
    u32 acc = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = tile * 8192u + r * 512u + threadIdx.x;
        const u32 j = mix(i * 2654435761u + win) % win_elems;
        if (MODE == 0) w[j] = i; else acc += w[j];
    }
    if (MODE && acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    u32 *buf, *sink;
    const size_t total = (size_t)1024 * 901120;
    hipMalloc(&buf, total * 4); hipMalloc(&sink, 64);
    hipMemset(buf, 0, total * 4);
    const u32 sizes[] = {131072, 262144, 450560, 524288, 675840, 786432, 901120, 1048576, 1802240};
    for (u32 we : sizes) {
        const u32 tiles = we / 8192, nwin = (u32)(total / we) & ~7u;
        const u32 grid = nwin * tiles;
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, we, sink);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, we, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("window %.2f MB %s: %.3f ms, %.1f G ops/s\n", we * 4.0 / 1e6, mode ? "gather " : "scatter", best, (double)grid * 8192.0 / best / 1e6);
        }
    }
    return 0;
}
