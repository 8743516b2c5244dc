// micro-benchmark: random 4-byte scatter / gather rate into per-XCD 3.6 MB windows (like the rank array R)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;
constexpr u32 kWin = 901120; // elements per window (3.6 MB)

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: scatter, window chosen by (blockIdx & 7) + 8*(blockIdx/ (8*tiles))  (XCD-local, like xcd_remap)
// mode 1: scatter, window = blockIdx / tiles (blocks of one window spread over all XCDs)
// mode 2/3: gather with the same two mappings
template <int MODE> __global__ __launch_bounds__(512) void k(u32 *buf, u32 nwin, u32 tiles, u32 *sink)
{
    const u32 lid = blockIdx.x;
    u32 win, tile;
    if (MODE == 0 || MODE == 2) { const u32 x = lid & 7u, slot = lid >> 3; win = (slot / tiles) * 8u + x; tile = slot % tiles; }
    else { win = lid / tiles; tile = lid % tiles; }
    if (win >= nwin) return;
    u32 *w = buf + (size_t)win * kWin;
    u32 acc = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = tile * 8192u + r * 512u + threadIdx.x;
        const u32 j = mix(i * 2654435761u + win) % kWin;
        if (MODE < 2) w[j] = i; else acc += w[j];
    }
    if (MODE >= 2 && acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const u32 nwin = 1024, tiles = 110;
    u32 *buf, *sink; hipMalloc(&buf, (size_t)nwin * kWin * 4); hipMalloc(&sink, 64);
    hipMemset(buf, 0, (size_t)nwin * kWin * 4);
    const u32 grid = nwin * tiles;
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, sink);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, sink);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, sink);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, buf, nwin, tiles, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops = (double)grid * 8192.0;
        printf("mode %d (%s, %s): %.3f ms, %.1f G ops/s\n", mode, mode < 2 ? "scatter" : "gather", (mode & 1) ? "spread" : "xcd-local", ms, ops / ms / 1e6);
    }
    return 0;
}
