// micro-benchmark + property check: does a returning LDS atomic add (ds_add_rtn_u32) hand the lanes of ONE
// wave instruction that hit the same address their old values in ascending lane order?  The radix passes
// could then rank a row of 64 elements with one LDS atomic per row instead of the "match any" ballots.
// Prints the number of (row, lane) results that differ from the lane-ordered expectation, per digit pattern,
// and the time of both ranking methods.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32; typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int BITS> __device__ __forceinline__ u64 wave_match_digit(u32 dg, bool ok)
{
    u32 diff_lo = 0, diff_hi = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const u32 mine = (u32)((int)(dg << (31 - b)) >> 31);
        const u64 m = __ballot((int)mine < 0);
        diff_lo = __builtin_amdgcn_bitop3_b32(diff_lo, (u32)m, mine, 0xF6);
        diff_hi = __builtin_amdgcn_bitop3_b32(diff_hi, (u32)(m >> 32), mine, 0xF6);
    }
    const u64 diff = ((u64)diff_hi << 32) | diff_lo;
    return __ballot(ok) & ~diff;
}

// pattern p: how the digit of element i is drawn
__device__ __forceinline__ u32 digit_of(u32 i, u32 pattern, u32 seed)
{
    const u32 h = mix(i * 2654435761u + seed);
    switch (pattern) {
    case 0: return h & 1023u;                       // uniform over 1024
    case 1: return h & 63u;                         // uniform over 64
    case 2: return h & 7u;                          // uniform over 8
    case 3: return h & 1u;                          // two values
    case 4: return 5u;                              // all equal
    case 5: return (h & 255u) < 200u ? 17u : (h >> 8) & 1023u; // one heavy digit
    case 6: return (i & 63u) >> 1;                  // neighbouring lane pairs
    case 7: return (i & 31u);                       // lane l and l+32 collide (same bank, two halves)
    case 8: return ((i & 63u) * 32u) & 1023u;       // same bank, different addresses
    case 9: return (h % 3u) * 341u;                 // three values far apart
    default: return (h & 1023u) & ~(h >> 10 & 1023u);
    }
}

// MODE 0: check atomics against match-any; MODE 1: time atomics; MODE 2: time match-any
template <int MODE> __global__ __launch_bounds__(512) void k(u32 pattern, u32 seed, u32 iters, u32 *bad, u32 *sink)
{
    __shared__ u32 s_cnt[8][1024];
    __shared__ u32 s_ref[8][1024];
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    u32 nbad = 0, acc = 0;
    for (u32 it = 0; it < iters; ++it) {
        for (u32 i = threadIdx.x; i < 8 * 1024; i += 512) { (&s_cnt[0][0])[i] = 0; (&s_ref[0][0])[i] = 0; }
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 16; ++r) {
            const u32 i = ((blockIdx.x * iters + it) * 16u + r) * 512u + threadIdx.x;
            const u32 dg = digit_of(i, pattern, seed);
            const bool ok = (MODE != 0) || (mix(i + 77u) % 11u != 0u); // some lanes sit out in the check
            u32 got = 0, want = 0;
            if (MODE == 0 || MODE == 1) {
                if (ok) got = __hip_atomic_fetch_add(&s_cnt[w][dg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (MODE == 0 || MODE == 2) {
                const u64 peers = wave_match_digit<10>(dg, ok);
                if (ok) {
                    const u32 before = __popcll(peers & lt_mask);
                    const u32 c0 = s_ref[w][dg];
                    want = c0 + before;
                    if ((peers >> l) == 1ull) s_ref[w][dg] = c0 + before + 1u;
                }
            }
            if (MODE == 0 && ok && got != want) ++nbad;
            acc += got + want;
        }
        __syncthreads();
    }
    if (MODE == 0 && nbad) atomicAdd(bad, nbad);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    u32 *bad, *sink;
    hipMalloc(&bad, 4); hipMalloc(&sink, 64);
    const u32 grid = 2048, iters = 8;
    for (u32 pattern = 0; pattern <= 10; ++pattern) {
        u32 total_bad = 0;
        for (u32 seed = 1; seed <= 4; ++seed) {
            hipMemset(bad, 0, 4);
            hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, pattern, seed * 7919u, iters, bad, sink);
            u32 h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            total_bad += h;
        }
        float t[2] = {0, 0};
        for (int mode = 1; mode <= 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, pattern, 1u, iters, bad, sink);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, pattern, 1u, iters, bad, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            t[mode - 1] = best;
        }
        const double elems = (double)grid * iters * 8192.0;
        printf("pattern %2u: out-of-lane-order results %u of %.0f; atomic %.3f ms (%.1f G/s), match-any %.3f ms (%.1f G/s)\n",
               pattern, total_bad, elems * 4, t[0], elems / t[0] / 1e6, t[1], elems / t[1] / 1e6);
    }
    return 0;
}
