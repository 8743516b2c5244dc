// chase.hip -- pointer chases of many walkers inside one table per XCD: how fast is a step when the table fits the XCD's L2
// (4 MB) and when it does not?  (the decoder's inverse BWT: one random 4-byte load per byte, T = 3.6 MB per level-9 block)
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/chase tools/ubench/chase.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
typedef unsigned int u32;
// entry bytes: 4 = one u32 per slot; 3 = three bytes per slot (unaligned dword load, low 24 bits)
template <int EB>
__global__ __launch_bounds__(256) void k_chase(const unsigned char *__restrict__ tabs, size_t tab_stride, u32 slots, u32 steps, u32 walkers_per_xcd,
                                                unsigned char *__restrict__ out, u32 *__restrict__ sink)
{
    const u32 xcd = (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    const unsigned char *t = tabs + (size_t)xcd * tab_stride;
    // walker id inside the XCD: workgroups go round the XCDs
    const u32 wid = (blockIdx.x / 8u) * blockDim.x + threadIdx.x;
    if (wid >= walkers_per_xcd) return;
    u32 p = (u32)(((unsigned long long)wid * 2654435761ull) % slots);
    unsigned char *o = out + ((size_t)xcd * walkers_per_xcd + wid) * steps;
    u32 acc = 0;
    for (u32 s = 0; s < steps; ++s) {
        u32 v;
        if (EB == 4) v = reinterpret_cast<const u32 *>(t)[p];
        else {
            __builtin_memcpy(&v, t + (size_t)p * 3u, 4);
            v &= 0xFFFFFFu;
        }
        acc += v;
        __builtin_nontemporal_store((unsigned char)v, o + s);
        p = v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    hipSetDevice(0);
    const u32 steps = 128;
    std::mt19937 rng(7);
    for (int eb : {4, 3}) {
        for (u32 slots : {300000u, 500000u, 700000u, 900000u, 1200000u}) {
            // one random cyclic permutation per XCD
            std::vector<u32> perm(slots), nxt(slots);
            std::iota(perm.begin(), perm.end(), 0u);
            std::shuffle(perm.begin(), perm.end(), rng);
            for (u32 i = 0; i < slots; ++i) nxt[perm[i]] = perm[(i + 1) % slots];
            const size_t stride = ((size_t)slots * eb + 4 + 255) & ~(size_t)255;
            std::vector<unsigned char> host(stride * 8, 0);
            for (int x = 0; x < 8; ++x)
                for (u32 i = 0; i < slots; ++i) {
                    unsigned char *q = host.data() + x * stride + (size_t)i * eb;
                    q[0] = nxt[i] & 255, q[1] = (nxt[i] >> 8) & 255, q[2] = (nxt[i] >> 16) & 255;
                    if (eb == 4) q[3] = 0;
                }
            unsigned char *d_t, *d_o;
            u32 *d_s;
            hipMalloc(&d_t, host.size());
            hipMemcpy(d_t, host.data(), host.size(), hipMemcpyHostToDevice);
            hipMalloc(&d_s, 64);
            for (u32 wgs_per_xcd : {16u, 32u, 64u}) {
                const u32 walkers = wgs_per_xcd * 256u;
                hipMalloc(&d_o, (size_t)8 * walkers * steps);
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (eb == 4) hipLaunchKernelGGL(k_chase<4>, dim3(wgs_per_xcd * 8u), dim3(256), 0, 0, d_t, stride, slots, steps, walkers, d_o, d_s);
                    else hipLaunchKernelGGL(k_chase<3>, dim3(wgs_per_xcd * 8u), dim3(256), 0, 0, d_t, stride, slots, steps, walkers, d_o, d_s);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep) best = std::min(best, ms);
                }
                const double gsteps = 8.0 * walkers * steps / (best * 1e-3) / 1e9;
                printf("entry %d B, table %.2f MB per XCD, %5u walkers per XCD: %.3f ms, %.1f G steps/s (1.07 G steps = %.1f ms)\n", eb, slots * (double)eb / 1e6, walkers, best,
                       gsteps, 1.07 / gsteps * 1e3);
                hipFree(d_o);
            }
            hipFree(d_t);
            hipFree(d_s);
        }
    }
    return 0;
}
