// k_deflate.hip -- Deflate encode: the reference's LzssEncoder + Inflater,
// re-derived as data-parallel stages.  Everything the reference decides serially is a pure
// function of the input, which is what makes the stream reproducible bit for bit:
//
//   * SlideDict pushes EVERY position into a 16-bit hash of its 3 bytes (lzss/slidedict.rs:
//     192-214, 80-115), so the chain it walks from position p (search_dic :216-267) is "the earlier
//     positions with the same hash, nearest first, while the summed distance is <= 0x8000, at most
//     255 of them, stopping behind the first one that matches to the limit".  Sorting positions by
//     hash (stable, two counting passes per 512 Ki-position chunk: k_df_shist / k_df_sscan /
//     k_df_sscatter) puts every chain in one run: the candidates of an entry are the entries in front of it.
//   * The candidate kept is the first one with the greatest length (the comparison closure of
//     deflate/encoder.rs:34-51 can only prefer a new candidate that is strictly longer, because a
//     later candidate is always farther): k_df_match2 reads the candidates off the sorted order, a lane per
//     entry, 16 text bytes per step out of LDS.  (Rounds 1 and 2 walked the chains: k_df_prev turned the runs into
//     the reference's `pos` array, k_df_match walked it with a 32 KiB window staged in LDS; BZ_DF_MATCH=walk.)
//   * LzssEncoder::encode (lzss/encoder.rs:132-184) looks at the matches of p, p+1, p+2 and
//     advances by `len + lazy_index`: adv(p) is a function of p, the parse is the orbit of 0
//     under p -> p + adv(p).  Orbits are found per 4096-position tile for every possible entry
//     offset (an entry is < 260 past the tile start), composed 64 tiles at a time, resolved top
//     down, and marked: k_df_tile_orbit / k_df_compose / k_df_resolve / k_df_mark2 (every entry is followed
//     until it lands on the tile's canonical orbit; rounds 1 and 2 doubled pointers twelve times:
//     k_df_tile_tab / k_df_mark, BZ_DF_PARSE=doubling).
//   * InflaterInner::next (deflate/encoder.rs:577-636) closes a block when the next code would
//     take it past 0xFFFF bytes: the next block starts at the last code start <= start + 0xFFFF
//     (k_df_cuts).  write_block (:454-547) then chooses stored / fixed / dynamic from the block's
//     symbol counts: k_df_block replays make_table (huffman/cano_huff_table.rs, the serial heap
//     procedure -- in the registers of a wave -- and the package-merge fallback) and the code-length run
//     coding (:318-452) exactly; k_df_emit writes the bits LSB first (bitio/writer.rs, Right).
#include <cstdlib>
#include <cstring>

#include "bzgpu.h"
#include "k_deflate.h"

namespace dfgpu {
using namespace bzgpu;
typedef int i32;

// ---------------------------------------------------------------------------------- hash + chains
// slidedict.rs:80-87 on a 64-bit usize: the three bytes fold to a 24-bit value, times HASH_FRAC, top 16 bits
__device__ __forceinline__ u32 hash16(u32 b0, u32 b1, u32 b2)
{
    const u64 h = (u64)((b0 << 16) | (b1 << 8) | b2);
    return (u32)((h * 0x7A7C4F9F7A7C4F9Full) >> 48);
}

// The chain sort.  Trigram positions are handled in chunks of kChunk, every chunk together with the 32 KiB of
// positions in front of it (they are the chain candidates of its first positions): entry j of chunk c is
// position c * kChunk - kWin + j, j < kChunkStride; entries in front of position 0 or behind the last trigram
// do not exist.  Two stable counting passes (low byte of the hash, then high byte) order a chunk by (hash,
// position); the hash is recomputed from the text each time (the chunk's 1 MB of text sits in L2), so only
// positions are moved.  Same three-kernel shape as the decoder's T-vector sort (k_dec.hip).
// hash of the trigram at pos: one unaligned dword load (legal on gfx950) except at the very end of the text
__device__ __forceinline__ u32 df_hash_at(const u8 *__restrict__ in, u32 pos, u64 ntri)
{
    u32 w;
    if ((u64)pos + 2 <= ntri) { // pos + 3 < n: the fourth byte exists
        typedef u32 __attribute__((aligned(1))) u32u;
        w = *reinterpret_cast<const u32u *>(in + pos);
    } else w = (u32)in[pos] | ((u32)in[pos + 1] << 8) | ((u32)in[pos + 2] << 16);
    return hash16(w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu);
}

__device__ __forceinline__ u32 df_chunk_count(u32 c, u64 ntri, u32 &j0)
{
    j0 = c == 0 ? kWin : 0u;
    const u64 lim = ntri - (u64)c * kChunk + kWin; // entries j with position < ntri
    return (u32)(lim < kChunkStride ? lim : kChunkStride) - j0;
}

// element `idx` of the pass input of chunk c.  PASS 0: the chunk's positions in order, digit = low byte of the
// hash; what it hands on is the position relative to the chunk (21 bits) with the HIGH byte of the hash on top,
// so that PASS 1 (digit = that byte) does not touch the text again.
template <int PASS>
__device__ __forceinline__ bool df_pass_elem(const u8 *__restrict__ in, u64 ntri, const u32 *__restrict__ src, u32 c, u32 idx,
                                             u32 count, u32 j0, u32 &val, u32 &dg)
{
    if (idx >= count) { val = 0; dg = 0; return false; }
    if (PASS == 0) {
        const u32 rel = j0 + idx;
        const u32 h = df_hash_at(in, c * kChunk - kWin + rel, ntri);
        dg = h & 0xFFu;
        val = rel | ((h >> 8) << 24);
    } else {
        const u32 e = src[(size_t)c * kChunkStride + idx];
        dg = e >> 24;
        val = c * kChunk - kWin + (e & 0xFFFFFFu); // the position itself goes out
    }
    return true;
}

template <int PASS>
__global__ __launch_bounds__(kSortThreads) void k_df_shist(const u8 *__restrict__ in, u64 ntri, const u32 *__restrict__ src,
                                                           u32 *__restrict__ hist)
{
    __shared__ u32 s_hist[4][256];
    const u32 c = blockIdx.x / kChunkTiles, tile = blockIdx.x % kChunkTiles;
    u32 j0;
    const u32 count = df_chunk_count(c, ntri, j0);
    const u32 start = tile * kSortTile;
    for (u32 i = threadIdx.x; i < 4u * 256u; i += kSortThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    if (start < count) {
        u32 *mine = s_hist[threadIdx.x & 3u];
#pragma unroll 4
        for (u32 r = 0; r < 16; ++r) {
            u32 pos, dg;
            if (df_pass_elem<PASS>(in, ntri, src, c, start + r * kSortThreads + threadIdx.x, count, j0, pos, dg)) atomicAdd(&mine[dg], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256)
        hist[(size_t)blockIdx.x * 256u + threadIdx.x] =
            s_hist[0][threadIdx.x] + s_hist[1][threadIdx.x] + s_hist[2][threadIdx.x] + s_hist[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_df_sscan(u32 *__restrict__ hist, u32 *__restrict__ tbase)
{
    __shared__ u32 s_w[4];
    const u32 c = blockIdx.x, d = threadIdx.x;
    u32 *h = hist + (size_t)c * kChunkTiles * 256u;
    u32 run = 0;
    for (u32 t = 0; t < kChunkTiles; ++t) {
        const u32 v = h[t * 256u + d];
        h[t * 256u + d] = run;
        run += v;
    }
    const u32 inc = wave_incl_sum(run);
    if ((d & 63u) == 63u) s_w[d >> 6] = inc;
    __syncthreads();
    u32 carry = 0;
    for (u32 k = 0; k < (d >> 6); ++k) carry += s_w[k];
    tbase[(size_t)c * 256u + d] = carry + inc - run;
}

template <int PASS>
__global__ __launch_bounds__(kSortThreads) void k_df_sscatter(const u8 *__restrict__ in, u64 ntri, const u32 *__restrict__ src,
                                                              const u32 *__restrict__ hist, const u32 *__restrict__ tbase,
                                                              u32 *__restrict__ dst, const u32 *__restrict__ tbase0,
                                                              u16 *__restrict__ hash_out)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ u32 s_buf[kSortTile];
    __shared__ u8 s_dg[kSortTile];
    __shared__ u8 s_lo[PASS ? kSortTile : 4];
    __shared__ u32 s_b0[PASS ? 256 : 4]; // PASS 1: where the low-byte buckets of pass 0 begin in this chunk
    __shared__ u32 s_base[256];
    __shared__ u16 s_tpre[256];
    __shared__ u16 s_cnt[NW][256];
    __shared__ u32 s_wsum[4];
    const u32 c = blockIdx.x / kChunkTiles, tile = blockIdx.x % kChunkTiles;
    u32 j0;
    const u32 count = df_chunk_count(c, ntri, j0);
    const u32 start = tile * kSortTile;
    if (start >= count) return;
    for (u32 i = threadIdx.x; i < NW * 256u / 2u; i += kSortThreads) reinterpret_cast<u32 *>(&s_cnt[0][0])[i] = 0;
    if (threadIdx.x < 256) s_base[threadIdx.x] = hist[(size_t)blockIdx.x * 256u + threadIdx.x] + tbase[(size_t)c * 256u + threadIdx.x];
    if (PASS && threadIdx.x < 256) s_b0[threadIdx.x] = tbase0[(size_t)c * 256u + threadIdx.x];
    __syncthreads();
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    u16 *my_cnt = s_cnt[w];
    u32 posv[16], dgv[16], rnk[16];
    u32 okmask = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r)
        okmask |= (df_pass_elem<PASS>(in, ntri, src, c, start + w * 1024u + r * 64u + l, count, j0, posv[r], dgv[r]) ? 1u : 0u) << r;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const bool ok = (okmask >> r) & 1u;
        const u32 dg = dgv[r];
        const u64 peers = wave_match_digit<8>(dg, ok);
        rnk[r] = 0xFFFFFFFFu;
        if (ok) {
            const u32 before = __popcll(peers & lt_mask);
            const u32 c0 = my_cnt[dg];
            rnk[r] = c0 + before;
            if ((peers >> l) == 1ull) my_cnt[dg] = (u16)(c0 + before + 1u);
        }
    }
    __syncthreads();
    u32 tot = 0;
    if (threadIdx.x < 256) {
        for (u32 k = 0; k < NW; ++k) {
            const u32 cc = s_cnt[k][threadIdx.x];
            s_cnt[k][threadIdx.x] = (u16)tot;
            tot += cc;
        }
        const u32 inc = wave_incl_sum(tot);
        if (l == 63) s_wsum[w] = inc;
        s_tpre[threadIdx.x] = (u16)(inc - tot); // wave-local for now
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        u32 carry = 0;
        for (u32 k = 0; k < w; ++k) carry += s_wsum[k];
        s_tpre[threadIdx.x] = (u16)(s_tpre[threadIdx.x] + carry);
    }
    __syncthreads();
    const u32 cnt_tile = (count - start < kSortTile) ? count - start : kSortTile;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (rnk[r] != 0xFFFFFFFFu) {
            const u32 lpos = (u32)s_tpre[dgv[r]] + (u32)my_cnt[dgv[r]] + rnk[r];
            s_buf[lpos] = posv[r]; // digit order inside the tile, so that consecutive lanes store consecutive words
            s_dg[lpos] = (u8)dgv[r];
            if (PASS && hash_out) {
                // the input of pass 1 is ordered by the low byte: the bucket an input index falls into IS that byte
                const u32 idx = start + w * 1024u + r * 64u + l;
                u32 lo = 0;
#pragma unroll
                for (u32 s = 128; s; s >>= 1)
                    if (s_b0[lo + s] <= idx) lo += s;
                s_lo[lpos] = (u8)lo;
            }
        }
    }
    __syncthreads();
    u32 *out = dst + (size_t)c * kChunkStride;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        if (i < cnt_tile) {
            const u32 dg = s_dg[i];
            const u32 o = s_base[dg] + (i - (u32)s_tpre[dg]);
            out[o] = s_buf[i];
            if (PASS && hash_out) hash_out[(size_t)c * kChunkStride + o] = (u16)((dg << 8) | s_lo[i]);
        }
    }
}

// a chunk ordered by (hash, position): distance to the previous position with the same hash (0 = none in the
// window), and the length of the chain search_dic would walk from it (how many earlier positions with this
// hash lie within the window, at most 255): the match kernel groups positions of similar chain length.
// The entries in front of a chunk's first position only serve as predecessors.
__global__ __launch_bounds__(256) void k_df_prev(u64 ntri, const u32 *__restrict__ S, const u16 *__restrict__ H,
                                                 u32 *__restrict__ pe)
{
    // the 255 entries in front of the workgroup's 256 and its own, staged once: the search runs in LDS
    __shared__ u32 s_k[256 + 256];
    __shared__ u32 s_v[256 + 256];
    // workgroups are dealt round-robin to the 8 XCDs: the r-th workgroup of XCD x takes the r-th slice of the
    // x-th, (x+8)-th, ... chunk, so that the 4 MB a chunk scatters into stay in one L2
    u32 slice = blockIdx.x;
    {
        const u32 x = blockIdx.x & 7u, r = blockIdx.x >> 3;
        const u32 cand = ((r / kPrevSpan) * 8 + x) * kPrevSpan + (r % kPrevSpan);
        const u32 full = gridDim.x / (8 * kPrevSpan) * (8 * kPrevSpan);
        if (blockIdx.x < full) slice = cand; // the ragged tail keeps the plain order
    }
    const u32 c = slice / kPrevSpan, i0 = (slice % kPrevSpan) * 256;
    u32 j0;
    const u32 count = df_chunk_count(c, ntri, j0);
    if (i0 >= count) return;
    const u32 *Sc = S + (size_t)c * kChunkStride;
    const u16 *Hc = H + (size_t)c * kChunkStride;
    for (u32 j = threadIdx.x; j < 512; j += 256) {
        const i64 g = (i64)i0 - 256 + (i64)j;
        const bool ok = g >= 0 && (u64)g < count;
        s_v[j] = ok ? __builtin_nontemporal_load(Sc + g) : 0u; // streamed: must not evict the scatter lines
        s_k[j] = ok ? (u32)__builtin_nontemporal_load(Hc + g) : 0xFFFFFFFFu;
    }
    __syncthreads();
    const u32 i = i0 + threadIdx.x;
    if (i >= count) return;
    const u32 li = 256 + threadIdx.x;
    const u32 p = s_v[li];
    const u32 h = s_k[li];
    if ((u64)p < (u64)c * kChunk) return; // history entry of this chunk: written by the chunk that owns it
    u32 d = 0;
    if (s_k[li - 1] == h) {
        const u32 dd = p - s_v[li - 1];
        if (dd <= kWin) d = dd;
    }
    u32 e = 0;
    if (d) { // smallest j in [i - 255, i) with the same hash and S[j] + window >= p (monotone in j)
        u32 lo = li - kChain, hi = li - 1; // hi qualifies; entries before the chunk start carry key ~0
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (s_k[mid] == h && s_v[mid] + kWin >= p) hi = mid; else lo = mid + 1;
        }
        e = li - lo;
    }
    pe[p] = d | (e << 16); // one scattered store per position: distance (16 bits) and chain length
}

// ---------------------------------------------------------------------------------- matches
#ifndef DF_HOPS
#define DF_HOPS 4
#endif
constexpr u32 kHops = DF_HOPS, kQSlots = 3 + DF_HOPS;
constexpr u32 kMDataBytes = kWin + kMTile + 272;

__global__ __launch_bounds__(kMThreads) void k_df_match(const u8 *__restrict__ in, const u32 *__restrict__ pe, u64 n,
                                                        u32 *__restrict__ M)
{
    __shared__ u32 s_w[kMDataBytes / 4];
    __shared__ u16 s_prev[kWin + kMTile];
    __shared__ u16 s_order[kMTile];
    __shared__ u16 s_q[kQSlots][kMThreads]; // per lane: summed distances of the candidates waiting to be measured
    u8 *s_est = reinterpret_cast<u8 *>(&s_q[0][0]); // chain lengths of the tile: only needed while the positions are ordered
    __shared__ u32 s_hist[256];
    __shared__ u32 s_next;
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * kMTile;
    const i64 base = (i64)t0 - (i64)kWin; // multiple of 4
    const u32 count = (n - t0) < (u64)kMTile ? (u32)(n - t0) : kMTile;
    if (tid < 256) s_hist[tid] = 0;
    if (tid == 0) s_next = 0;
    for (u32 w = tid; w < kMDataBytes / 4; w += kMThreads) {
        const i64 g = base + 4 * (i64)w;
        u32 v = 0;
        if (g >= 0 && (u64)g + 4 <= n) v = *reinterpret_cast<const u32 *>(in + g);
        else if (g >= 0 && (u64)g < n)
            for (u32 b = 0; b < 4 && (u64)g + b < n; ++b) v |= (u32)in[g + b] << (8 * b);
        s_w[w] = v;
    }
    for (u32 k4 = tid; k4 < (kWin + kMTile) / 4; k4 += kMThreads) { // four positions per load
        const i64 g = base + 4 * (i64)k4;
        u32 v[4] = {0, 0, 0, 0};
        if (g >= 0 && (u64)g + 4 <= n) {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(pe + g));
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else if (g >= 0)
            for (u32 j = 0; j < 4 && (u64)g + j < n; ++j) v[j] = pe[g + j];
        reinterpret_cast<u32 *>(s_prev)[2 * k4] = (v[0] & 0xFFFFu) | (v[1] << 16);
        reinterpret_cast<u32 *>(s_prev)[2 * k4 + 1] = (v[2] & 0xFFFFu) | (v[3] << 16);
        if (4 * k4 >= kWin) // the tile's own positions: their chain lengths
            reinterpret_cast<u32 *>(s_est)[k4 - kWin / 4] =
                ((v[0] >> 16) & 0xFFu) | (((v[1] >> 16) & 0xFFu) << 8) | (((v[2] >> 16) & 0xFFu) << 16) | (((v[3] >> 16) & 0xFFu) << 24);
    }
    __syncthreads();
    // positions of the tile ordered by chain length, longest first (a counting sort on 256 lengths): the 64
    // lanes of a wave then walk chains of about the same length
    u32 my_est[kMTile / kMThreads];
    for (u32 j = 0; j < kMTile / kMThreads; ++j) {
        const u32 k = j * kMThreads + tid;
        my_est[j] = k < count ? 255u - s_est[k] : 0u;
        if (k < count) atomicAdd(&s_hist[my_est[j]], 1u);
    }
    __syncthreads();
    if (tid < 64) { // exclusive prefix over the 256 bins
        u32 v[4], s = 0;
        for (u32 j = 0; j < 4; ++j) { v[j] = s_hist[tid * 4 + j]; s += v[j]; }
        u32 inc = s;
        for (u32 dlt = 1; dlt < 64; dlt <<= 1) {
            const u32 o = __shfl_up(inc, dlt);
            if (tid >= dlt) inc += o;
        }
        u32 run = inc - s;
        for (u32 j = 0; j < 4; ++j) { s_hist[tid * 4 + j] = run; run += v[j]; }
    }
    __syncthreads();
    for (u32 j = 0; j < kMTile / kMThreads; ++j) {
        const u32 k = j * kMThreads + tid;
        if (k < count) s_order[atomicAdd(&s_hist[my_est[j]], 1u)] = (u16)k;
    }
    __syncthreads();
    // four bytes at any offset from two aligned dwords (an unaligned ds_read_b32 is legal on gfx950 but was
    // measured slower here)
    auto ld4 = [&](u32 byte) -> u32 { return __builtin_amdgcn_alignbyte(s_w[(byte >> 2) + 1], s_w[byte >> 2], byte & 3u); };
    const u32 lane = tid & 63u;
    const u32 nchunks = (count + 63) / 64;
    for (;;) {
        u32 c = 0;
        if (lane == 0) c = atomicAdd(&s_next, 1u);
        c = (u32)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= nchunks) break;
        const u32 idx = c * 64 + lane;
        const bool valid = idx < count;
        const u32 k = valid ? s_order[idx] : 0u;
        const u32 lp = kWin + k;
        const u64 p = t0 + k;
        const u32 limit = (n - p) < (u64)kMaxMatch ? (u32)(n - p) : kMaxMatch; // search_dic :228
        u32 cum = valid ? s_prev[lp] : 0u;                                       // (a first hop is never longer than the window)
        bool active = cum != 0;
        u32 best_len = 0, best_dist = 0, cnt = kChain;
        // bytes p .. p+7
        const u32 pw0 = s_w[lp >> 2], pw1 = s_w[(lp >> 2) + 1], pw2 = s_w[(lp >> 2) + 2];
        const u32 a0 = __builtin_amdgcn_alignbyte(pw1, pw0, lp & 3u), a1 = __builtin_amdgcn_alignbyte(pw2, pw1, lp & 3u);
        // The walk (search_dic :232-265).  A candidate replaces the best one only if it is strictly longer
        // (deflate/encoder.rs:34-51 with a farther candidate), so it must agree with p on bytes 0 .. best_len,
        // in particular on the last four of them (fewer while best_len < 3): that test is all the walk does.
        // Candidates that pass are queued (four per lane) and measured later, for all lanes at once: the
        // result is the first candidate of the greatest length either way, a stale best_len only lets a few
        // more candidates into the queue, and a candidate that matches to the limit (where search_dic stops)
        // is the first of the greatest length too -- walking past it until the next measuring changes nothing.
        // (a match shorter than 3 is no match, lzss/encoder.rs:135-141: the walk starts as if 2 bytes were beaten)
        best_len = 2;
        u32 foff = 0, fmask = 0xFFFFFFu, fq = a0 & 0xFFFFFFu; // offset of the tested dword, its mask, p's bytes there
        u32 qn = 0;                                   // candidates in this lane's queue (s_q[..][tid])
        auto measure = [&]() {
            bool hit = false;
            for (u32 s = 0; s < kQSlots; ++s) {
                if (!__ballot(s < qn)) break;
                if (s < qn && !hit) {
                    const u32 ccum = s_q[s][tid];
                    const u32 lcq = lp - ccum;
                    const u32 cw0 = s_w[lcq >> 2], cw1 = s_w[(lcq >> 2) + 1], cw2 = s_w[(lcq >> 2) + 2];
                    const u32 x0 = a0 ^ __builtin_amdgcn_alignbyte(cw1, cw0, lcq & 3u);
                    const u32 x1 = a1 ^ __builtin_amdgcn_alignbyte(cw2, cw1, lcq & 3u);
                    const u64 xx = ((u64)x1 << 32) | x0; // first differing byte of the first eight, no branches
                    u32 l = xx ? ((u32)__builtin_ctzll(xx) >> 3) : 8u;
                    if (l == 8) // check_match :175-188, eight bytes per trip
                        while (l < limit) {
                            const u32 pa = lp + l, ca = lcq + l;
                            const u32 p0 = s_w[pa >> 2], p1 = s_w[(pa >> 2) + 1], p2 = s_w[(pa >> 2) + 2];
                            const u32 c0 = s_w[ca >> 2], c1 = s_w[(ca >> 2) + 1], c2 = s_w[(ca >> 2) + 2];
                            const u32 y0 = __builtin_amdgcn_alignbyte(p1, p0, pa & 3u) ^ __builtin_amdgcn_alignbyte(c1, c0, ca & 3u);
                            const u32 y1 = __builtin_amdgcn_alignbyte(p2, p1, pa & 3u) ^ __builtin_amdgcn_alignbyte(c2, c1, ca & 3u);
                            const u64 yy = ((u64)y1 << 32) | y0;
                            if (yy) { l += (u32)__builtin_ctzll(yy) >> 3; break; }
                            l += 8;
                        }
                    l = l < limit ? l : limit;
                    if (l > best_len) { best_len = l; best_dist = ccum; }
                    hit = l == limit; // :258-259: nothing behind this candidate counts
                }
            }
            qn = 0;
            foff = best_len >= 3 ? best_len - 3 : 0u;
            fmask = best_len >= 3 ? 0xFFFFFFFFu : ((1u << (8 * (best_len + 1))) - 1);
            fq = best_len < limit ? (ld4(lp + foff) & fmask) : 0u;
            if (hit) active = false;
        };
        auto hop = [&]() { // one candidate: the byte test, the queue, the next link
            const u32 lc = lp - cum;
            const u32 fa = lc + foff;
            const u32 fw0 = s_w[fa >> 2], fw1 = s_w[(fa >> 2) + 1];
            const u32 d = s_prev[lc]; // (read together with the bytes: one LDS round trip per candidate)
            const bool pass = active && ((__builtin_amdgcn_alignbyte(fw1, fw0, fa & 3u) ^ fq) & fmask) == 0;
            if (pass) s_q[qn][tid] = (u16)cum;
            qn += pass ? 1u : 0u;
            cnt -= 1;
            const u32 ncum = cum + d;
            const bool cont = active && cnt != 0 && d != 0 && ncum <= kWin; // :262, :234
            cum = cont ? ncum : cum;
            active = cont;
        };
        while (__ballot(active)) { // kHops candidates per trip (the queue holds 3 + kHops, measuring starts at four)
#pragma unroll
            for (u32 hh = 0; hh < kHops; ++hh) hop();
            if (__ballot(qn >= 4)) measure();
        }
        if (__ballot(qn != 0)) measure();
        if (valid) M[p] = best_len >= kMinMatch ? (best_len | ((best_dist - 1) << 9)) : 0u;
    }
}

// ---------------------------------------------------------------------------------- matches, from the sorted order
// The same search (search_dic, lzss/slidedict.rs:218-266) without a walk.  In a chunk ordered by (hash, position)
// the candidates of the entry at index i are the entries i-1, i-2, ... while the hash is the same, the distance
// is within the window and k <= 255: the chain IS the run in front of the entry.  A lane takes one entry, a wave
// 64 neighbouring ones (chains of about the same length without any ordering step), and in step k every lane
// compares the first 16 bytes of its own text with those of entry i-k -- staged in LDS once per workgroup, read
// back with one ds_read_b128 at consecutive addresses -- by four XORs, four find-first-bit and two three-way
// minima: no chain links, no dependent loads, no byte tests.  What is kept is the first candidate of the
// greatest length (deflate/encoder.rs:34-51: a farther candidate replaces the kept one only if it is strictly
// longer).  Candidates that agree on all 16 bytes (1.2 % of the pairs on text) are queued per lane and measured in
// batches against the text in global memory (the chunk's 544 KB sit in its XCD's L2), after the four-byte test
// at the kept length that the walking kernel applies to every candidate; measuring late changes nothing (see
// there), and a candidate that reaches the limit ends the lane's search (:258-259).
constexpr u32 kM2Threads = DF_M2_THREADS; // entries per workgroup
constexpr u32 kM2Span = kChunkStride / kM2Threads;
constexpr u32 kM2Hist = 256;              // entries staged in front of them (255 are needed)
static_assert(kChunkStride % kM2Threads == 0, "workgroups tile a sort chunk");

__device__ __forceinline__ u32 ffbl_raw(u32 x) // index of the lowest set bit, 0xFFFFFFFF for 0
{
    u32 r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

typedef u32 df_u32u __attribute__((aligned(1)));
typedef u64 df_u64u __attribute__((aligned(1)));
typedef u32 df_u32x4 __attribute__((ext_vector_type(4)));
typedef df_u32x4 __attribute__((aligned(1))) df_u32x4u;

// eight text bytes at q; `avail` bytes exist there
__device__ __forceinline__ u64 df_ld8(const u8 *__restrict__ in, u64 q, u64 avail)
{
    if (avail >= 8) return *reinterpret_cast<const df_u64u *>(in + q);
    u64 v = 0;
    for (u32 b = 0; b < 8 && b < avail; ++b) v |= (u64)in[q + b] << (8 * b);
    return v;
}

// the first 16 text bytes at pos, zeros behind the end of the text
__device__ __forceinline__ df_u32x4 df_snippet(const u8 *__restrict__ in, u32 pos, u64 n)
{
    if ((u64)pos + 16 <= n) return *reinterpret_cast<const df_u32x4u *>(in + pos);
    u32 w[4] = {0, 0, 0, 0};
    for (u32 b = 0; b < 16 && (u64)pos + b < n; ++b) w[b >> 2] |= (u32)in[pos + b] << (8 * (b & 3u));
    df_u32x4 sn;
    sn.x = w[0]; sn.y = w[1]; sn.z = w[2]; sn.w = w[3];
    return sn;
}

// first differing bit of two 16-byte pieces, 128 if none
__device__ __forceinline__ u32 df_diff16(df_u32x4 x, df_u32x4 y)
{
    const u32 f0 = ffbl_raw(x.x ^ y.x), f1 = ffbl_raw(x.y ^ y.y) | 32u, f2 = ffbl_raw(x.z ^ y.z) | 64u, f3 = ffbl_raw(x.w ^ y.w) | 96u;
    u32 r = f0 < f1 ? f0 : f1;
    r = r < f2 ? r : f2;
    u32 r2 = f3 < 128u ? f3 : 128u;
    return r < r2 ? r : r2;
}

__global__ __launch_bounds__(kM2Threads) void k_df_match2(const u8 *__restrict__ in, u64 n, u64 ntri, const u32 *__restrict__ S,
                                                          u32 *__restrict__ M)
{
    typedef df_u32x4 u32x4;
    __shared__ u32x4 s_snip[kM2Hist + kM2Threads];
    __shared__ u32 s_pos[kM2Hist + kM2Threads];
    __shared__ u32 s_key[kM2Hist + kM2Threads];
    // workgroups are dealt round-robin to the 8 XCDs (as in k_df_prev): one chunk's text, and the 2 MB of match
    // words its entries scatter into, stay in one L2
    u32 slice = blockIdx.x;
    {
        const u32 x = blockIdx.x & 7u, r = blockIdx.x >> 3;
        const u32 cand = ((r / kM2Span) * 8 + x) * kM2Span + (r % kM2Span);
        const u32 full = gridDim.x / (8 * kM2Span) * (8 * kM2Span);
        if (blockIdx.x < full) slice = cand;
    }
    const u32 c = slice / kM2Span, i0 = (slice % kM2Span) * kM2Threads;
    u32 j0;
    const u32 count = df_chunk_count(c, ntri, j0);
    if (i0 >= count) return;
    const u32 *Sc = S + (size_t)c * kChunkStride;
    const u32 tid = threadIdx.x;
    // the workgroup's entries and the 256 in front of them: position, hash, the first 16 text bytes (positions fit
    // 32 bits: a part of a long stream is at most 1 GiB + history)
    const u32 n32 = (u32)n;
    static_assert(kM2Threads == kM2Hist, "one entry in front and one own entry per thread");
    {
        const i32 g0 = (i32)i0 - (i32)kM2Hist + (i32)tid, g1 = (i32)(i0 + tid);
        const bool ok0 = g0 >= 0, ok1 = (u32)g1 < count; // (g0 < i0 < count)
        u32 pos0 = 0, pos1 = 0;
        if (ok0) pos0 = __builtin_nontemporal_load(Sc + g0);
        if (ok1) pos1 = __builtin_nontemporal_load(Sc + g1);
        u32x4 sn0 = {0, 0, 0, 0}, sn1 = {0, 0, 0, 0};
        const bool in0 = ok0 && pos0 + 16 <= n32, in1 = ok1 && pos1 + 16 <= n32;
        if (in0) sn0 = *reinterpret_cast<const df_u32x4u *>(in + pos0);
        if (in1) sn1 = *reinterpret_cast<const df_u32x4u *>(in + pos1);
        if (ok0 && !in0) sn0 = df_snippet(in, pos0, n); // (the text ends within the 16 bytes)
        if (ok1 && !in1) sn1 = df_snippet(in, pos1, n);
        // (every entry has its three bytes: the hash is that of its trigram)
        s_pos[tid] = pos0;
        s_key[tid] = ok0 ? hash16(sn0.x & 0xFFu, (sn0.x >> 8) & 0xFFu, (sn0.x >> 16) & 0xFFu) : 0xFFFFFFFFu;
        s_snip[tid] = sn0;
        s_pos[kM2Hist + tid] = pos1;
        s_key[kM2Hist + tid] = ok1 ? hash16(sn1.x & 0xFFu, (sn1.x >> 8) & 0xFFu, (sn1.x >> 16) & 0xFFu) : 0xFFFFFFFFu;
        s_snip[kM2Hist + tid] = sn1;
    }
    __syncthreads();
    const u32 li = kM2Hist + tid;
    const u32 p = s_pos[li], h = s_key[li];
    const u32x4 a = s_snip[li];
    const bool own = i0 + tid < count && (u64)p >= (u64)c * kChunk; // a history entry is written by the chunk that owns it
    // chain length: the entries in front with the same hash and within the window form one run (monotone in j)
    u32 e = 0;
    if (own && s_key[li - 1] == h && p - s_pos[li - 1] <= kWin) {
        u32 lo = li - kChain, hi = li - 1;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (s_key[mid] == h && s_pos[mid] + kWin >= p) hi = mid; else lo = mid + 1;
        }
        e = li - lo;
    }
    // Where the chain ends because the entries in front have another hash, running on does no harm (another hash is
    // another trigram: fewer than three bytes agree, which never beats "no match"); only a chain cut by the window, by
    // the 255 candidates or by the start of the chunk's entries has to be masked.  em: the step a lane is masked from.
    const u32 em = (own && !(s_key[li - e - 1] != 0xFFFFFFFFu && s_key[li - e - 1] != h)) ? e : 0xFFFFFFFFu;
    const u32 limit = (n32 - p) < kMaxMatch ? n32 - p : kMaxMatch; // search_dic :228
    // A step gives r = first differing bit of the 16 bytes, at most `cap` (the limit where it lies inside them, else
    // 128: "all 16 agree, the rest is to be measured").  What a lane keeps is the greatest key
    //     (r | 7) << 24 | (255 - k) << 16
    // -- the longest, and among equally long ones the nearest --; bit 31 of a step's key says "to be measured", and
    // the last 32 of those bits wait in `qm` (one v_alignbit per step) until the wave measures them together.
    u32 cap = limit < 16 ? limit * 8 : 128u;
    asm volatile("" : "+v"(cap)); // (keeps the select out of the loop)
    u32 best = ((2u * 8 + 7) << 24) | (255u << 16); // a match shorter than 3 is none (lzss/encoder.rs:135-141)
    u32 qm = 0;
    u32 mkey = 0;          // the longest measured candidate: length << 8 | 255 - k
    // Measuring the candidates whose bits are set in qm (bit j belongs to step kcur - j): every lane offers its oldest
    // one -- so that a later candidate only counts if it is strictly longer --, and the WAVE measures the offered
    // candidates one after the other, four in flight: lane i compares the dwords at byte 16 + 4 i of the two texts
    // (two coalesced 256-byte reads instead of 64 lanes reading 16 bytes each at unrelated places, trip after trip: a
    // match to the limit costs the same two reads as one of 17 bytes), a ballot finds the first difference.  Text
    // offsets fit 32 bits (a part is at most 1 GiB + history).
    const u32 lane = tid & 63u;
    auto measure = [&](u32 kcur) {
        for (;;) {
            bool has = qm != 0 && e != 0;
            if (!has) qm = 0;
            if (!__ballot(has)) break;
            u32 k_item = 0, cp_item = 0;
            if (has) {
                const u32 j = 31u - (u32)__clz(qm);
                qm &= ~(1u << j);
                k_item = (kcur < em ? kcur : em) - j; // (a masked lane's bits stop moving when its chain ends)
                cp_item = s_pos[li - k_item];
                // to count, it must be longer than the longest one measured so far: the four bytes that end at that
                // length have to agree (the walking kernel's test; one small read per lane instead of a measurement)
                const u32 mlen = mkey >> 8;
                if (mlen) has = *reinterpret_cast<const df_u32u *>(in + (cp_item + mlen - 3)) == *reinterpret_cast<const df_u32u *>(in + (p + mlen - 3));
            }
            u64 B = __ballot(has);
            while (B) { // up to four of them in flight
                u32 sl[4], ps[4], cs[4], lim[4], xa[4], xb[4];
                bool use[4];
#pragma unroll
                for (u32 i = 0; i < 4; ++i) {
                    use[i] = B != 0;
                    if (use[i]) {
                        sl[i] = (u32)__builtin_ctzll(B);
                        B &= B - 1;
                        ps[i] = (u32)__builtin_amdgcn_readlane((int)p, (int)sl[i]);
                        cs[i] = (u32)__builtin_amdgcn_readlane((int)cp_item, (int)sl[i]);
                        lim[i] = (u32)__builtin_amdgcn_readlane((int)limit, (int)sl[i]);
                        if ((u64)ps[i] + 16 + 256 <= n) {
                            xa[i] = *reinterpret_cast<const df_u32u *>(in + (ps[i] + 16) + 4 * lane);
                            xb[i] = *reinterpret_cast<const df_u32u *>(in + (cs[i] + 16) + 4 * lane);
                        } else { // the text ends within the 256 bytes: byte by byte, zeros behind the end on both sides
                            xa[i] = xb[i] = 0;
                            for (u32 b = 0; b < 4; ++b) {
                                const u64 qa = (u64)ps[i] + 16 + 4 * lane + b, qb = (u64)cs[i] + 16 + 4 * lane + b;
                                if (qa < n) { xa[i] |= (u32)in[qa] << (8 * b); xb[i] |= (u32)in[qb] << (8 * b); }
                            }
                        }
                    }
                }
#pragma unroll
                for (u32 i = 0; i < 4; ++i) {
                    if (use[i]) {
                        const u32 x = xa[i] ^ xb[i];
                        const u64 nz = __ballot(x != 0) & ((1ull << 61) - 1); // 61 dwords reach byte 260
                        u32 l = 16 + 61 * 4;
                        if (nz) {
                            const u32 f = (u32)__builtin_ctzll(nz);
                            const u32 xv = (u32)__builtin_amdgcn_readlane((int)x, (int)f);
                            l = 16 + 4 * f + ((u32)__builtin_ctz(xv) >> 3);
                        }
                        l = l < lim[i] ? l : lim[i];
                        if (lane == sl[i]) {
                            if (l > (mkey >> 8)) mkey = (l << 8) | (255u - k_item);
                            if (l == lim[i]) e = 0; // :258-259: nothing behind this candidate counts
                        }
                    }
                }
            }
        }
    };
    // One step, as the 16 or 17 vector instructions it takes (the compiler's rendering of the same C++ needs 21 and a
    // branch): where a chain is cut, its lane drops out of EXEC (v_cmpx); four XORs, four find-first-bit, the word
    // offsets, two three-way minima (the second one also applies the cap), the key, its top bit into qm, the maximum.
#define DF_M2_STEP(D, J)                                                                                                      \
    "s_add_u32 %[k], %[k0], " #J "\n"                                                                                         \
    "s_sub_u32 %[kc], 0x7ff, %[k]\n"                                                                                          \
    "s_lshl_b32 %[kc], %[kc], 16\n"                                                                                           \
    "s_cmp_eq_u32 %[nm], 0\n"                                                                                                 \
    "s_cbranch_scc1 1f\n"                                                                                                     \
    "v_cmpx_le_u32_e32 vcc, %[k], %[e]\n"                                                                                     \
    "1:\n"                                                                                                                    \
    "v_xor_b32_e32 %[t0], %[a0], %[" #D "0]\n"                                                                                \
    "v_xor_b32_e32 %[t1], %[a1], %[" #D "1]\n"                                                                                \
    "v_xor_b32_e32 %[t2], %[a2], %[" #D "2]\n"                                                                                \
    "v_xor_b32_e32 %[t3], %[a3], %[" #D "3]\n"                                                                                \
    "v_ffbl_b32_e32 %[t0], %[t0]\n"                                                                                           \
    "v_ffbl_b32_e32 %[t1], %[t1]\n"                                                                                           \
    "v_ffbl_b32_e32 %[t2], %[t2]\n"                                                                                           \
    "v_ffbl_b32_e32 %[t3], %[t3]\n"                                                                                           \
    "v_or_b32_e32 %[t1], 32, %[t1]\n"                                                                                         \
    "v_or_b32_e32 %[t2], 64, %[t2]\n"                                                                                         \
    "v_or_b32_e32 %[t3], 0x60, %[t3]\n"                                                                                       \
    "v_min3_u32 %[t0], %[t0], %[t1], %[t2]\n"                                                                                 \
    "v_min3_u32 %[t0], %[t0], %[t3], %[cap]\n"                                                                                \
    "v_lshl_or_b32 %[t0], %[t0], 24, %[kc]\n"                                                                                 \
    "v_alignbit_b32 %[qm], %[qm], %[t0], 31\n"                                                                                \
    "v_max_u32_e32 %[best], %[best], %[t0]\n"
#define DF_M2_PAIR(CA, CB, JA, JB)                                                                                            \
    asm volatile("s_mov_b64 %[sv], exec\n" DF_M2_STEP(da, JA) DF_M2_STEP(db, JB) "s_mov_b64 exec, %[sv]\n"                    \
                 : [best] "+v"(best), [qm] "+v"(qm), [k] "=&s"(ks), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),           \
                   [t3] "=&v"(t3), [sv] "=&s"(sv), [kc] "=&s"(kc)                                                             \
                 : [k0] "s"(k0s), [nm] "s"(needmask), [a0] "v"(a.x), [a1] "v"(a.y), [a2] "v"(a.z), [a3] "v"(a.w),              \
                   [e] "v"(em), [cap] "v"(cap), [da0] "v"(CA.x), [da1] "v"(CA.y), [da2] "v"(CA.z), [da3] "v"(CA.w),           \
                   [db0] "v"(CB.x), [db1] "v"(CB.y), [db2] "v"(CB.z), [db3] "v"(CB.w)                                         \
                 : "vcc", "scc")
    u32 roff = li * 16; // byte offset of the lane's own row of s_snip
    u32 k0 = 0;
    while (__ballot(k0 < e)) {
        // up to 32 steps, then the measuring (its changes to e and mkey stay out of the inner loop's registers)
        for (u32 q = 0; q < 4; ++q) { // steps k0 + 1 .. k0 + 8: eight reads in flight, then the steps
            if (q && !__ballot(k0 < e)) break;
            roff -= 8 * 16;
            asm volatile("" : "+v"(roff)); // (so that the reads below take immediate offsets)
            const u32x4 *row = reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(s_snip) + roff);
            u32 t0, t1, t2, t3, kc, ks;
            u64 sv;
            const u32x4 c1 = row[7], c2 = row[6], c3 = row[5], c4 = row[4], c5 = row[3], c6 = row[2], c7 = row[1], c8 = row[0];
            // some lane's chain is cut inside these eight steps: then (and only then) the steps carry their v_cmpx
            const u64 cut = __ballot(em < k0 + 8);
            const u32 needmask = (u32)cut | (u32)(cut >> 32);
            const u32 k0s = (u32)__builtin_amdgcn_readfirstlane((int)k0);
            DF_M2_PAIR(c1, c2, 1, 2);
            DF_M2_PAIR(c3, c4, 3, 4);
            DF_M2_PAIR(c5, c6, 5, 6);
            DF_M2_PAIR(c7, c8, 7, 8);
            k0 = k0s + 8;
        }
        if (__ballot(qm != 0)) measure(k0);
    }
#undef DF_M2_PAIR
#undef DF_M2_STEP
    if (own) {
        u32 best_len = best >> 27, best_k = 255u - ((best >> 16) & 0xFFu);
        if (mkey) { best_len = mkey >> 8; best_k = 255u - (mkey & 0xFFu); }
        M[p] = best_len >= kMinMatch ? (best_len | ((p - s_pos[li - best_k] - 1) << 9)) : 0u;
    }
}

// ---------------------------------------------------------------------------------- parse
// lzss/encoder.rs:132-184: step[p] = advance | lazy_index << 9 if a code sequence started at p
__device__ __forceinline__ u32 df_step_word(u32 m0, u32 m1, u32 m2)
{
    u32 out_len = m0 & 511u, out_pos = m0 >> 9, li = 0;
    if (out_len < kMinMatch) return 1u;
    if (out_len < kMaxMatch) {
        const u32 il = m1 & 511u, ip = m1 >> 9;
        if (il > kMinMatch && (il << 3) + ip > (out_len << 3) + out_pos) { out_len = il; out_pos = ip; li = 1; }
    }
    if (out_len < kMaxMatch) {
        const u32 il = m2 & 511u, ip = m2 >> 9;
        if (il > kMinMatch && (il << 3) + ip > (out_len << 3) + out_pos) { out_len = il; out_pos = ip; li = 2; }
    }
    return (out_len + li) | (li << 9);
}

// four positions per thread: two aligned 16-byte loads of match words, one 8-byte store of step words
__global__ __launch_bounds__(256) void k_df_adv(const u32 *__restrict__ M, u64 n, u16 *__restrict__ step)
{
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    const u64 p0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= n) return;
    u32 m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool al = (reinterpret_cast<uintptr_t>(M) & 15u) == 0 && (reinterpret_cast<uintptr_t>(step) & 7u) == 0;
    if (al && p0 + 8 <= n) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(M + p0), b = *reinterpret_cast<const u32x4 *>(M + p0 + 4);
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    } else
        for (u32 j = 0; j < 8 && p0 + j < n; ++j) m[j] = M[p0 + j]; // (a match word behind the end is "no match")
    u32 s[4];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) s[j] = df_step_word(m[j], m[j + 1], m[j + 2]);
    if (al && p0 + 4 <= n) {
        u32x2 o;
        o.x = s[0] | (s[1] << 16); o.y = s[2] | (s[3] << 16);
        *reinterpret_cast<u32x2 *>(step + p0) = o;
    } else
        for (u32 j = 0; j < 4 && p0 + j < n; ++j) step[p0 + j] = (u16)s[j];
}

#ifndef DF_TAB_THREADS
#define DF_TAB_THREADS 512
#endif
constexpr u32 kTabThreads = DF_TAB_THREADS;
__global__ __launch_bounds__(kTabThreads) void k_df_tile_tab(const u16 *__restrict__ step, u64 n, u16 *__restrict__ tab)
{
    __shared__ u16 s_nx[kPTile];
    const u64 t0 = (u64)blockIdx.x * kPTile;
    for (u32 k = threadIdx.x; k < kPTile; k += kTabThreads) {
        const u64 p = t0 + k;
        s_nx[k] = (u16)(k + (p < n ? (step[p] & 511u) : 1u));
    }
    for (u32 r = 0; r < 12; ++r) {
        __syncthreads();
        for (u32 k = threadIdx.x; k < kPTile; k += kTabThreads) {
            const u32 v = s_nx[k];
            if (v < kPTile) s_nx[k] = s_nx[v]; // in place: a fresher value is only farther along the same path
        }
    }
    __syncthreads();
    for (u32 e = threadIdx.x; e < kEntries; e += kTabThreads) tab[(u64)blockIdx.x * kEntries + e] = (u16)(s_nx[e] - kPTile);
}

__global__ __launch_bounds__(320) void k_df_compose(const u16 *__restrict__ tin, u32 nin, u16 *__restrict__ tout)
{
    const u32 e = threadIdx.x;
    if (e >= kEntries) return;
    const u32 c0 = blockIdx.x * kFan, c1 = (c0 + kFan < nin) ? c0 + kFan : nin;
    u32 v = e;
    for (u32 c = c0; c < c1; ++c) v = tin[(u64)c * kEntries + v];
    tout[(u64)blockIdx.x * kEntries + e] = (u16)v;
}

__global__ __launch_bounds__(64) void k_df_resolve(const u16 *__restrict__ tab_child, u32 nchild,
                                                   const u16 *__restrict__ ent_parent, u32 nparent,
                                                   u16 *__restrict__ ent_child)
{
    const u32 g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nparent) return;
    u32 v = ent_parent[g];
    const u32 c0 = g * kFan, c1 = (c0 + kFan < nchild) ? c0 + kFan : nchild;
    for (u32 c = c0; c < c1; ++c) {
        ent_child[c] = (u16)v;
        v = tab_child[(u64)c * kEntries + v];
    }
}

// marks the code starts of the steps that begin in this tile and writes code[] for every position those
// steps cover (the ranges of consecutive tiles meet exactly)
#ifndef DF_MARK_THREADS
#define DF_MARK_THREADS 512
#endif
constexpr u32 kMarkThreads = DF_MARK_THREADS;
__global__ __launch_bounds__(kMarkThreads) void k_df_mark(const u16 *__restrict__ step, const u32 *__restrict__ M,
                                                 const u16 *__restrict__ ent, u64 n, u32 *__restrict__ code,
                                                 u64 *__restrict__ bm)
{
    __shared__ u16 s_j[2][kPTile];
    __shared__ u16 s_step[kPTile];
    __shared__ u8 s_mark[kPTile];
    __shared__ u8 s_type[kPTile + 264];
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * kPTile;
    const u32 entry = ent[blockIdx.x];
    for (u32 k = tid; k < kPTile; k += kMarkThreads) {
        const u64 p = t0 + k;
        const u32 s = p < n ? step[p] : 1u;
        s_step[k] = (u16)s;
        s_j[0][k] = (u16)(k + (s & 511u));
        s_mark[k] = (k == entry) ? 1 : 0;
    }
    for (u32 k = tid; k < kPTile + 264; k += kMarkThreads) s_type[k] = 0;
    for (u32 r = 0; r < 12; ++r) {
        __syncthreads();
        const u32 cur = r & 1u;
        for (u32 k = tid; k < kPTile; k += kMarkThreads) {
            const u32 v = s_j[cur][k];
            if (s_mark[k] && v < kPTile) s_mark[v] = 1;
            s_j[cur ^ 1u][k] = (u16)(v < kPTile ? s_j[cur][v] : v);
        }
    }
    __syncthreads();
    // after 12 exact doublings every entry has left the tile: the exit of `entry`
    const u32 exitp = (entry < kPTile) ? (u32)s_j[0][entry] : entry;
    for (u32 k = tid; k < kPTile; k += kMarkThreads) {
        if (!s_mark[k]) continue;
        const u32 s = s_step[k], adv = s & 511u, li = s >> 9;
        // (4: a step of the parse starts here -- with li > 0 the step's reference is a code start that is not one)
        if (adv == 1) s_type[k] = 1 | 4;
        else {
            for (u32 i = 0; i < li; ++i) s_type[k + i] = 1;
            s_type[k + li] = 2;
            s_type[k] |= 4;
        }
    }
    __syncthreads();
    // code words, and one bit per position for the block cuts (the 64 positions of a wave straddle two words)
    for (u32 k0 = entry + (tid & ~63u); k0 < exitp; k0 += kMarkThreads) {
        const u32 k = k0 + (tid & 63u);
        const u64 q = t0 + k;
        const bool ok = k < exitp && q < n;
        const u32 ty = ok ? s_type[k] : 0u;
        if (ok) code[q] = ty == 0 ? 0u : (((ty & 3u) == 1 ? F_CODE : (F_CODE | F_REF | M[q])) | ((ty & 4u) ? F_STEP : 0u));
        const u64 bal = __ballot(ty != 0);
        if ((tid & 63u) == 0 && bal) {
            const u64 q0 = t0 + k0;
            const u32 sh = (u32)(q0 & 63u);
            atomicOr(reinterpret_cast<unsigned long long *>(bm + (q0 >> 6)), (unsigned long long)(bal << sh));
            if (sh && (bal >> (64 - sh))) atomicOr(reinterpret_cast<unsigned long long *>(bm + (q0 >> 6) + 1), (unsigned long long)(bal >> (64 - sh)));
        }
    }
}

// ---------------------------------------------------------------------------------- parse, without doubling (round 3)
// Orbits of p -> p + adv(p) that start at different places fall into step after a few codes (a literal moves by one,
// so any orbit soon lands on a position another one visits), and from there on they ARE the same orbit.  So a tile
// needs one orbit in full -- the one entered at its first position, the "canonical" one -- and every other entry
// only has to be followed until it lands on a position of that orbit:
//   k_df_tile_orbit  the canonical orbit of a tile as a bitmap (kept in global memory for k_df_mark2) and the exit of
//                    every entry offset (the table k_df_tile_tab made with twelve doubling rounds over 4096 positions).
//                    The canonical orbit itself is put together from 64 speculative walks, one lane per 64
//                    positions starting at the sub-tile's first position: where the orbit enters a sub-tile it
//                    either lands on the lane's path at once (then the lane's path from there and the lane's exit
//                    are the orbit's) or is followed until it does.
//   k_df_mark2       the orbit of the tile's true entry = its steps up to the first position on the canonical orbit +
//                    the canonical orbit from there; then the code words as in k_df_mark.
// Every step taken is checked against the bitmap, nothing is assumed: an input whose orbits never meet (none is
// known with short steps) only makes the walks longer.
#ifndef DF_ORB_THREADS
#define DF_ORB_THREADS 128
#endif
#ifndef DF_MARK2_THREADS
#define DF_MARK2_THREADS 256
#endif
constexpr u32 kOrbThreads = DF_ORB_THREADS, kMark2Threads = DF_MARK2_THREADS;
static_assert(kOrbThreads >= 64 && kMark2Threads >= 64 && kPTile == 64 * 64, "64 sub-tiles of 64 positions, one lane each");

__device__ __forceinline__ void df_stage_steps(const u16 *__restrict__ step, u64 n, u64 t0, u32 tid, u32 nthreads, u16 *s_step)
{
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const bool al = (reinterpret_cast<uintptr_t>(step) & 15u) == 0;
    for (u32 c = tid; c < kPTile / 8; c += nthreads) {
        const u64 p = t0 + (u64)c * 8;
        if (al && p + 8 <= n) *reinterpret_cast<u32x4 *>(s_step + c * 8) = *reinterpret_cast<const u32x4 *>(step + p);
        else
            for (u32 j = 0; j < 8; ++j) s_step[c * 8 + j] = p + j < n ? step[p + j] : (u16)1;
    }
}
// advance of a step word (never 0: a walk must move)
__device__ __forceinline__ u32 df_adv_of(u32 s) { const u32 a = s & 511u; return a ? a : 1u; }

__global__ __launch_bounds__(kOrbThreads) void k_df_tile_orbit(const u16 *__restrict__ step, u64 n, u16 *__restrict__ tab,
                                                               u64 *__restrict__ canon)
{
    __shared__ __attribute__((aligned(16))) u16 s_step[kPTile];
    __shared__ u64 s_m[64], s_C[64];
    __shared__ u32 s_x[64];
    __shared__ u32 s_exit;
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * kPTile;
    df_stage_steps(step, n, t0, tid, kOrbThreads, s_step);
    if (tid < 64) s_C[tid] = 0;
    __syncthreads();
    if (tid < 64) { // one lane per sub-tile: the path from its first position, and where it leaves
        u32 pos = tid * 64;
        const u32 end = pos + 64;
        u64 m = 0;
        do {
            m |= 1ull << (pos & 63u);
            pos += df_adv_of(s_step[pos]);
        } while (pos < end);
        s_m[tid] = m;
        s_x[tid] = pos;
    }
    __syncthreads();
    if (tid == 0) { // the canonical orbit, sub-tile by sub-tile
        u32 cur = 0;
        while (cur < kPTile) {
            const u32 sb = cur >> 6, off = cur & 63u;
            const u64 m = s_m[sb];
            if ((m >> off) & 1ull) { s_C[sb] |= m & (~0ull << off); cur = s_x[sb]; }
            else { s_C[sb] |= 1ull << off; cur += df_adv_of(s_step[cur]); }
        }
        s_exit = cur;
    }
    __syncthreads();
    const u32 cexit = s_exit;
    for (u32 e = tid; e < kEntries; e += kOrbThreads) {
        u32 cur = e;
        while (cur < kPTile && !((s_C[cur >> 6] >> (cur & 63u)) & 1ull)) cur += df_adv_of(s_step[cur]);
        tab[(u64)blockIdx.x * kEntries + e] = (u16)((cur < kPTile ? cexit : cur) - kPTile);
    }
    if (tid < 64) canon[(u64)blockIdx.x * 64 + tid] = s_C[tid];
}

__global__ __launch_bounds__(kMark2Threads) void k_df_mark2(const u16 *__restrict__ step, const u32 *__restrict__ M,
                                                           const u16 *__restrict__ ent, const u16 *__restrict__ tab,
                                                           const u64 *__restrict__ canon, u64 n, u32 *__restrict__ code,
                                                           u64 *__restrict__ bm, u32 tile0)
{
    __shared__ __attribute__((aligned(16))) u16 s_step[kPTile];
    __shared__ u64 s_C[64], s_mk[64];
    __shared__ u8 s_type[kPTile + 264];
    __shared__ u32 s_exitp, s_mw;
    const u32 tid = threadIdx.x;
    const u32 tile = tile0 + blockIdx.x; // (the tiles are marked in pieces: the block chain runs beside the next piece)
    const u64 t0 = (u64)tile * kPTile;
    const u32 entry = ent[tile];
    df_stage_steps(step, n, t0, tid, kMark2Threads, s_step);
    if (tid < 64) { s_C[tid] = canon[(u64)tile * 64 + tid]; s_mk[tid] = 0; }
    for (u32 k = tid; k < kPTile + 264; k += kMark2Threads) s_type[k] = 0;
    __syncthreads();
    if (tid == 0) { // from the entry to the first position on the canonical orbit (or out of the tile)
        u32 cur = entry, mw = 64;
        while (cur < kPTile) {
            const u32 sb = cur >> 6, off = cur & 63u;
            if ((s_C[sb] >> off) & 1ull) { s_mk[sb] |= s_C[sb] & (~0ull << off); mw = sb; break; }
            s_mk[sb] |= 1ull << off;
            cur += df_adv_of(s_step[cur]);
        }
        s_mw = mw;
        s_exitp = mw < 64 ? (u32)tab[(u64)tile * kEntries] + kPTile : cur; // (entry offset 0 is on the canonical orbit)
    }
    __syncthreads();
    if (tid < 64 && tid > s_mw) s_mk[tid] = s_C[tid]; // behind the meeting point: the canonical orbit
    __syncthreads();
    const u32 exitp = s_exitp;
    for (u32 k = tid; k < kPTile; k += kMark2Threads) {
        if (!((s_mk[k >> 6] >> (k & 63u)) & 1ull)) continue;
        const u32 s = s_step[k], adv = s & 511u, li = s >> 9;
        // (4: a step of the parse starts here -- with li > 0 the step's reference is a code start that is not one)
        if (adv <= 1) s_type[k] = 1 | 4;
        else {
            for (u32 i = 0; i < li; ++i) s_type[k + i] = 1;
            s_type[k + li] = 2;
            s_type[k] |= 4;
        }
    }
    __syncthreads();
    // code words, and one bit per position for the block cuts (the 64 positions of a wave straddle two words)
    for (u32 k0 = entry + (tid & ~63u); k0 < exitp; k0 += kMark2Threads) {
        const u32 k = k0 + (tid & 63u);
        const u64 q = t0 + k;
        const bool ok = k < exitp && q < n;
        const u32 ty = ok ? s_type[k] : 0u;
        if (ok) code[q] = ty == 0 ? 0u : (((ty & 3u) == 1 ? F_CODE : (F_CODE | F_REF | M[q])) | ((ty & 4u) ? F_STEP : 0u));
        const u64 bal = __ballot(ty != 0);
        if ((tid & 63u) == 0 && bal) {
            const u64 q0 = t0 + k0;
            const u32 sh = (u32)(q0 & 63u);
            atomicOr(reinterpret_cast<unsigned long long *>(bm + (q0 >> 6)), (unsigned long long)(bal << sh));
            if (sh && (bal >> (64 - sh))) atomicOr(reinterpret_cast<unsigned long long *>(bm + (q0 >> 6) + 1), (unsigned long long)(bal >> (64 - sh)));
        }
    }
}

// block starts: b' = the last code start <= b + 0xFFFF (InflaterInner::next :585-593).  A serial chain of
// n / 65535 hops; a hop loses at most 257 bytes against b + 0xFFFF, so the bits that the next kCutGroup hops
// can look at are known in advance: they are loaded together, the hops then run out of LDS.
constexpr u32 kCutGroup = 32;
constexpr u32 kCutWords = (257 * kCutGroup + 64) / 64 + 3;
// dl0: InflaterInner.decompress_len at the segment's first code (not 0 behind an Action::Flush, which closes a
// block without resetting it, :638-647 against :585-593): the first block then counts as having started dl0
// bytes in front of the segment -- the chain starts at the (negative) position -dl0.
// first: the first block starts at this code start (not 0 when a part of a long stream begins with the literals of a
// step whose reference opens its first block; they went out with the part before).
// The chain can be taken in pieces (round 3: beside the marking kernel, which runs over the tiles in pieces too):
// a call goes on while the window's end b + 0xFFFF lies in front of `limit` -- the code-start bits in front of
// `limit` are final --, keeps where it stands in state[0..1] (b, blocks so far) and picks up there when `resume`
// is set; the call with limit >= n ends the chain.
__global__ __launch_bounds__(256) void k_df_cuts(const u64 *__restrict__ bm, u64 n, u64 *__restrict__ bstart,
                                                 u32 *__restrict__ nb_out, u32 cap, u32 dl0, u32 first, u64 limit,
                                                 u32 resume, u64 *__restrict__ state, u32 *__restrict__ kdone)
{
    __shared__ u64 s_bm[kCutGroup][kCutWords];
    __shared__ i64 s_b;
    __shared__ u32 s_k, s_done;
    const u32 tid = threadIdx.x;
    const u64 nwords = (n + 63) / 64;
    if (tid == 0) {
        if (resume) { s_b = (i64)state[0]; s_k = (u32)state[1]; s_done = (u32)(state[1] >> 32); }
        else { s_b = (i64)first - (i64)dl0; s_k = 0; s_done = 0; if (cap) bstart[0] = first; }
    }
    __syncthreads();
    const u64 lim = limit < n ? limit : n;
    for (;;) {
        const i64 b = s_b;
        const u64 x = (u64)(b + (i64)kBlockMax); // >= 0: dl0 <= 0xFFFF
        if (x >= lim || s_done) break;
        // every region padded to kCutWords words ending at its highest word: all loads of a group are in
        // flight together (one memory round trip per kCutGroup hops)
        constexpr u32 kPer = (kCutGroup * kCutWords + 255) / 256;
        u64 v[kPer];
#pragma unroll
        for (u32 i = 0; i < kPer; ++i) {
            const u32 f = i * 256 + tid;
            const u32 j = f / kCutWords, w = f % kCutWords;
            const u64 whi = (x + (u64)j * kBlockMax) >> 6;
            const i64 word = (i64)whi - (i64)(kCutWords - 1) + (i64)w;
            v[i] = (j < kCutGroup && word >= 0 && (u64)word < nwords) ? bm[word] : 0ull;
        }
#pragma unroll
        for (u32 i = 0; i < kPer; ++i) {
            const u32 f = i * 256 + tid;
            if (f < kCutGroup * kCutWords) s_bm[f / kCutWords][f % kCutWords] = v[i];
        }
        __syncthreads();
        if (tid == 0) {
            i64 cur = b;
            u32 k = s_k;
            for (u32 j = 0; j < kCutGroup; ++j) {
                const u64 xj = (u64)(cur + (i64)kBlockMax);
                if (xj >= lim) break;
                const i64 wlo = (i64)((x + (u64)j * kBlockMax) >> 6) - (i64)(kCutWords - 1); // word of s_bm[j][0]
                i64 w = (i64)(xj >> 6) - wlo;
                u64 m = s_bm[j][w] & (~0ull >> (63 - (u32)(xj & 63)));
                while (!m && w > 0) { --w; m = s_bm[j][w]; }
                if (!m) { s_done = 2; break; } // a code is at most 258 bytes long: cannot happen
                cur = (i64)(((u64)(wlo + w) << 6) + 63 - (u64)__builtin_clzll(m));
                ++k;
                if (k < cap) bstart[k] = (u64)cur;
            }
            s_b = cur;
            s_k = k;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (limit < n) { // not the last piece: remember where the chain stands
            // (a piece that found the chain broken -- s_done == 2 -- says so in the state: the pieces behind it leave at
            // once and the last one reports 0xFFFFFFFF; the blocks this piece had closed before stay with k_df_block)
            state[0] = (u64)s_b;
            state[1] = (u64)s_k | ((u64)s_done << 32);
            if (kdone) *kdone = s_k; // blocks 0 .. s_k - 1 have both their ends: k_df_block can take them
        } else {
            const u32 k = s_k + 1; // blocks
            if (s_done == 2) *nb_out = 0xFFFFFFFFu;
            else {
                if (k < cap + 1) bstart[k] = n;
                *nb_out = k;
            }
            if (kdone) *kdone = s_done == 2 ? 0xFFFFFFFFu : k;
        }
    }
}

// ---------------------------------------------------------------------------------- tables
// deflate/mod.rs:76-125 without the tables: code, extra bits, extra value of len-3 / dist-1
__device__ __forceinline__ void len_code(u32 l3, u32 &code, u32 &eb, u32 &ev)
{
    if (l3 < 8) { code = l3; eb = 0; ev = 0; }
    else if (l3 == 255) { code = 28; eb = 0; ev = 0; }
    else {
        const u32 hb = 31u - (u32)__builtin_clz(l3), nb = hb - 2;
        code = ((nb + 1) << 2) | ((l3 >> nb) & 3u);
        eb = nb;
        ev = l3 & ((1u << nb) - 1);
    }
}
__device__ __forceinline__ void dist_code(u32 d0, u32 &code, u32 &eb, u32 &ev)
{
    if (d0 < 4) { code = d0; eb = 0; ev = 0; }
    else {
        const u32 hb = 31u - (u32)__builtin_clz(d0), nb = hb - 1;
        code = ((nb + 1) << 1) | ((d0 >> nb) & 1u);
        eb = nb;
        ev = d0 & ((1u << nb) - 1);
    }
}
__device__ __forceinline__ u32 len_ext_bits(u32 c) { return (c < 8 || c == 28) ? 0u : (c >> 2) - 1; }
__device__ __forceinline__ u32 dist_ext_bits(u32 c) { return c < 4 ? 0u : (c >> 1) - 1; }

// cano_huff_table.rs:14-31
__device__ void df_down_heap(u32 *buf, u32 nn, u32 len)
{
    const u32 tmp = buf[nn];
    u32 leaf = (nn << 1) + 1;
    while (leaf < len) {
        if (leaf + 1 < len && buf[buf[leaf]] > buf[buf[leaf + 1]]) leaf += 1;
        if (buf[tmp] < buf[buf[leaf]]) break;
        buf[nn] = buf[leaf];
        nn = leaf;
        leaf = (nn << 1) + 1;
    }
    buf[nn] = tmp;
}

// cano_huff_table.rs:58-151 ("reverse package merge"), weights x + y.  freq: k non-zero weights, out: k lengths.
// scr: 3*k + 64 + 2*lim*row words, row >= 2k + 4.
__device__ __forceinline__ void df_gen_code_lm(const u32 *freq, u32 n, u32 lim, u32 *scr, u8 *out)
{
    const u32 row = 2 * n + 4;
    u32 *map = scr, *sfreq = map + n, *c = sfreq + n, *misc = c + n;
    u32 *max_elem = misc, *b = misc + 20, *cur = misc + 40;
    u32 *val = misc + 64, *ty = val + lim * row;
    for (u32 i = 0; i < n; ++i) { // stable, descending (:64-70)
        const u32 f = freq[i];
        u32 p = i;
        while (p > 0 && sfreq[p - 1] < f) { sfreq[p] = sfreq[p - 1]; map[p] = map[p - 1]; --p; }
        sfreq[p] = f;
        map[p] = i;
    }
    for (u32 j = 0; j < lim; ++j) { max_elem[j] = 0; b[j] = 0; cur[j] = 0; }
    u32 excess = (1u << lim) - n;
    const u32 half = 1u << (lim - 1);
    max_elem[lim - 1] = n;
    for (u32 j = 0; j < lim; ++j) {
        if (excess >= half) { b[j] = 1; excess -= half; }
        excess <<= 1;
        if (lim >= 2 + j) max_elem[lim - 2 - j] = max_elem[lim - 1 - j] / 2 + n;
    }
    max_elem[0] = b[0];
    for (u32 j = 1; j < lim; ++j)
        if (max_elem[j] > 2 * max_elem[j - 1] + b[j]) max_elem[j] = 2 * max_elem[j - 1] + b[j];
    for (u32 j = 0; j < lim; ++j)
        for (u32 t = 0; t < max_elem[j]; ++t) { val[j * row + t] = 0; ty[j * row + t] = 0; }
    for (u32 i = 0; i < n; ++i) c[i] = lim;
    for (u32 t = 0; t < n && t < max_elem[lim - 1]; ++t) { val[(lim - 1) * row + t] = sfreq[t]; ty[(lim - 1) * row + t] = t; }
    if (b[lim - 1] == 1) { c[0] -= 1; cur[lim - 1] += 1; }
    u32 j = lim - 1;
    while (j > 0) {
        u32 i = 0, next = cur[j];
        for (u32 t = 0; t < max_elem[j - 1]; ++t) {
            const u32 weight = (next + 1 < max_elem[j]) ? val[j * row + next] + val[j * row + next + 1] : 0u;
            if (weight > sfreq[i]) { val[(j - 1) * row + t] = weight; ty[(j - 1) * row + t] = n; next += 2; }
            else {
                val[(j - 1) * row + t] = sfreq[i];
                ty[(j - 1) * row + t] = i;
                i += 1;
                if (i >= n) break;
            }
        }
        j -= 1;
        cur[j] = 0;
        if (b[j] == 1) { // take_package (:40-55) with an explicit stack
            u32 lvl[20], ph[20];
            int sp = 0;
            lvl[0] = j; ph[0] = 0;
            while (sp >= 0) {
                const u32 li = lvl[sp];
                if (ph[sp] == 0) {
                    const u32 x = ty[li * row + cur[li]];
                    if (x == n) { ph[sp] = 1; ++sp; lvl[sp] = li + 1; ph[sp] = 0; }
                    else { c[x] -= 1; cur[li] += 1; --sp; }
                } else if (ph[sp] == 1) { ph[sp] = 2; ++sp; lvl[sp] = li + 1; ph[sp] = 0; }
                else { cur[li] += 1; --sp; }
            }
        }
    }
    for (u32 i = 0; i < n; ++i) out[map[i]] = (u8)c[i];
}

// make_table (cano_huff_table.rs:198-230): lengths for the non-zero counts, 0 elsewhere; returns the
// length of the reference's vector (one past the last non-zero count).  buf: 2*nsym words; w: nsym words;
// lm_scr: scratch of df_gen_code_lm; sets *lm when the limited path ran.
__device__ u32 df_make_table(const u32 *freq, u32 nsym, u32 lim, u8 *out, u32 *buf, u32 *w, u8 *tmp, u32 *lm_scr, u32 *lm)
{
    u32 k = 0, last = 0;
    for (u32 i = 0; i < nsym; ++i) {
        out[i] = 0;
        if (freq[i]) { w[k++] = freq[i]; last = i + 1; }
    }
    if (k == 0) return 0;
    if (k == 1) tmp[0] = 1; // gen_code :158-160
    else {
        const u32 n = k;
        for (u32 i = 0; i < n; ++i) { buf[i] = n + i; buf[n + i] = w[i]; }
        for (u32 i = n >> 1; i-- > 0;) df_down_heap(buf, i, n); // create_heap :33-38 (len = 2n, s = n)
        for (u32 i = n - 1; i >= 1; --i) {
            const u32 m1 = buf[0];
            buf[0] = buf[i];
            df_down_heap(buf, 0, i);
            const u32 m2 = buf[0];
            buf[i] = buf[m1] + buf[m2];
            buf[0] = i;
            buf[m1] = i;
            buf[m2] = i;
            df_down_heap(buf, 0, i);
        }
        buf[1] = 0;
        for (u32 i = 2; i < n; ++i) buf[i] = buf[buf[i]] + 1;
        bool too_long = false;
        for (u32 i = 0; i < n; ++i) {
            const u32 l = buf[buf[i + n]] + 1;
            tmp[i] = (u8)l;
            if (l > lim) too_long = true;
        }
        if (too_long) {
            df_gen_code_lm(w, n, lim, lm_scr, tmp);
            if (lm) *lm += 1;
        }
    }
    k = 0;
    for (u32 i = 0; i < nsym; ++i)
        if (freq[i]) out[i] = tmp[k++];
    return last;
}

// The same, by a whole wave.  The heap procedure is a chain of dependent accesses to one small array (two sift-downs
// of six or seven levels per merge, half a dozen reads per level): out of LDS that is 0.4 ms for the hundred symbols
// of a literal/length table, and it is what a block's workgroup spends its time on.  Here the array lives in
// registers -- element e in lane e & 63 of register e >> 6 -- and every access is a v_readlane / v_writelane at a
// wave-uniform index: the control flow runs on the scalar unit, an access costs a few cycles instead of an LDS round
// trip.  Same procedure, same order of every comparison, same lengths.  NR: registers (2 * nsym <= 64 * NR).
template <u32 NR>
struct DfWaveArr {
    typedef u32 vec_t __attribute__((ext_vector_type(NR <= 1 ? 1 : (NR <= 4 ? 4 : 16))));
    vec_t r;
    // (a vector indexed with a wave-uniform value: the compiler addresses the register through M0 -- one move and
    // one v_readlane per access instead of a chain of compares over the registers)
    __device__ __forceinline__ u32 get(u32 x) const
    {
        const u32 word = NR <= 1 ? r[0] : r[x >> 6];
        return (u32)__builtin_amdgcn_readlane((int)word, (int)(x & 63u));
    }
    __device__ __forceinline__ void set(u32 x, u32 val)
    {
        // (a compare and a select stand in for v_writelane: this compiler has no builtin for it)
        const u32 lane = (u32)__builtin_amdgcn_mbcnt_hi(~0u, (u32)__builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (NR <= 1) r[0] = lane == (x & 63u) ? val : r[0];
        else {
            const u32 word = r[x >> 6];
            r[x >> 6] = lane == (x & 63u) ? val : word;
        }
    }
};

template <u32 NR>
__device__ __forceinline__ void df_down_heap_wave(DfWaveArr<NR> &a, u32 nn, u32 len)
{
    const u32 tmp = a.get(nn);
    const u32 wt = a.get(tmp);
    u32 leaf = (nn << 1) + 1;
    while (leaf < len) {
        u32 c = a.get(leaf), wc = a.get(c);
        if (leaf + 1 < len) {
            const u32 c2 = a.get(leaf + 1), wc2 = a.get(c2);
            if (wc > wc2) { leaf += 1; c = c2; wc = wc2; }
        }
        if (wt < wc) break;
        a.set(nn, c);
        nn = leaf;
        leaf = (nn << 1) + 1;
    }
    a.set(nn, tmp);
}

// all 64 lanes of a wave call this with the same arguments; the arrays are in LDS
template <u32 NR>
__device__ __forceinline__ u32 df_make_table_wave(const u32 *freq, u32 nsym, u32 lim, u8 *out, u32 *buf, u32 *w, u8 *tmp, u32 *lm_scr, u32 *lm,
                                  u32 lane)
{
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    u32 k = 0, last = 0;
    for (u32 i0 = 0; i0 < nsym; i0 += 64) { // the non-zero counts, in symbol order
        const u32 i = i0 + lane;
        const u32 f = i < nsym ? freq[i] : 0u;
        if (i < nsym) out[i] = 0;
        const u64 nz = __ballot(f != 0);
        if (f) w[k + (u32)__popcll(nz & lt)] = f;
        if (nz) last = i0 + 64 - (u32)__builtin_clzll(nz);
        k += (u32)__popcll(nz);
    }
    __builtin_amdgcn_wave_barrier();
    if (k == 0) return 0;
    if (k == 1) { if (lane == 0) tmp[0] = 1; } // gen_code :158-160
    else {
        const u32 n = k;
        DfWaveArr<NR> a;
#pragma unroll
        for (u32 q = 0; q < NR; ++q) {
            const u32 e = q * 64 + lane;
            a.r[q] = e < n ? n + e : (e < 2 * n ? w[e - n] : 0u);
        }
#pragma unroll
        for (u32 q = NR; q < (NR <= 1 ? 1u : (NR <= 4 ? 4u : 16u)); ++q) a.r[q] = 0;
        for (u32 i = n >> 1; i-- > 0;) df_down_heap_wave(a, i, n); // create_heap :33-38 (len = 2n, s = n)
        for (u32 i = n - 1; i >= 1; --i) {
            const u32 m1 = a.get(0);
            a.set(0, a.get(i));
            df_down_heap_wave(a, 0, i);
            const u32 m2 = a.get(0);
            a.set(i, a.get(m1) + a.get(m2));
            a.set(0, i);
            a.set(m1, i);
            a.set(m2, i);
            df_down_heap_wave(a, 0, i);
        }
        a.set(1, 0);
        for (u32 i = 2; i < n; ++i) a.set(i, a.get(a.get(i)) + 1);
#pragma unroll
        for (u32 q = 0; q < NR; ++q) {
            const u32 e = q * 64 + lane;
            if (e < 2 * n) buf[e] = a.r[q];
        }
        __builtin_amdgcn_wave_barrier();
        bool too_long = false;
        for (u32 i0 = 0; i0 < n; i0 += 64) {
            const u32 i = i0 + lane;
            u32 l = 0;
            if (i < n) { l = buf[buf[i + n]] + 1; tmp[i] = (u8)l; }
            if (__ballot(l > lim)) too_long = true;
        }
        __builtin_amdgcn_wave_barrier();
        if (too_long) {
            if (lane == 0) {
                df_gen_code_lm(w, n, lim, lm_scr, tmp);
                if (lm) atomicAdd(lm, 1u);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    k = 0;
    for (u32 i0 = 0; i0 < nsym; i0 += 64) {
        const u32 i = i0 + lane;
        const u32 f = i < nsym ? freq[i] : 0u;
        const u64 nz = __ballot(f != 0);
        if (f) out[i] = tmp[k + (u32)__popcll(nz & lt)];
        k += (u32)__popcll(nz);
    }
    __builtin_amdgcn_wave_barrier();
    return last;
}

// canonical codes, bit-reversed (huffman/mod.rs:16-63 with is_reverse)
__device__ __forceinline__ void df_make_codes(const u8 *len, u32 n, u16 *code)
{
    u32 cur = 0, last = 0;
    for (u32 l = 1; l <= 15; ++l)
        for (u32 s = 0; s < n; ++s)
            if (len[s] == l) {
                cur <<= (last < l ? l - last : 0);
                last = l;
                code[s] = (u16)(__brev(cur) >> (32 - l));
                cur += 1;
            }
}

struct LsbSink { // single lane, LSB first into 32-bit words
    u32 *w;
    u64 acc;
    u32 nacc, widx;
    __device__ __forceinline__ void put(u32 v, u32 nbits)
    {
        if (!nbits) return;
        acc |= (u64)v << nacc;
        nacc += nbits;
        if (nacc >= 32) { w[widx++] = (u32)acc; acc >>= 32; nacc -= 32; }
    }
    __device__ __forceinline__ u32 bits() const { return widx * 32 + nacc; }
    __device__ __forceinline__ void finish() { if (nacc) w[widx] = (u32)acc; }
};

// enc_tab_to_freq (deflate/encoder.rs:318-376): run coding of one length table
__device__ __forceinline__ u32 df_tab_runs(const u8 *tab, u32 n, u8 *ls, u8 *le, u32 *freq)
{
    u32 k = 0, old = 255, len = 0;
    for (u32 i = 0; i <= n; ++i) {
        const u32 d = i < n ? tab[i] : 255u;
        if (old != d) {
            if (old == 0) {
                if (len >= 11) { freq[18] += 1; ls[k] = 18; le[k++] = (u8)(len - 11); }
                else if (len >= 3) { freq[17] += 1; ls[k] = 17; le[k++] = (u8)(len - 3); }
                else { for (u32 t = 0; t < len; ++t) { ls[k] = 0; le[k++] = 0; } freq[0] += len; }
            } else if (len >= 3) { freq[16] += 1; ls[k] = 16; le[k++] = (u8)(len - 3); }
            else if (len > 0) { for (u32 t = 0; t < len; ++t) { ls[k] = (u8)old; le[k++] = 0; } freq[old] += len; }
            if (d != 0 && d != 255) { ls[k] = (u8)d; le[k++] = 0; freq[d] += 1; len = 0; }
            else len = 1;
            old = d;
        } else {
            len += 1;
            if (old == 0 && len == 138) { freq[18] += 1; ls[k] = 18; le[k++] = 127; len = 0; }
            else if (old != 0 && len == 6) { freq[16] += 1; ls[k] = 16; le[k++] = 3; len = 0; }
        }
    }
    return k;
}

// one workgroup per Deflate block: symbol counts, the three tables, the header, the choice of block type
__global__ __launch_bounds__(kBThreads) void k_df_block(const u8 *__restrict__ in, const u32 *__restrict__ code,
                                                        const u64 *__restrict__ bstart, const u32 *__restrict__ nb_p,
                                                        DfBlock *__restrict__ blocks, u8 *__restrict__ lens,
                                                        u32 *__restrict__ hdr, u32 *__restrict__ lm_scratch, u32 dl0,
                                                        u32 last_is_final, const u32 *__restrict__ kr, u32 piece_last)
{
    __shared__ u32 s_sf[288], s_of[32];
    __shared__ u32 s_buf[2][2 * 288], s_w[2][288];
    __shared__ u8 s_tmp[2][288], s_sl[288], s_ol[32];
    __shared__ u32 s_n[2], s_lm;
    __shared__ u8 s_ls[2][320], s_le[2][320];
    __shared__ u32 s_hdr[kHdrWords];
    __shared__ u32 s_lm7[3 * 19 + 64 + 2 * 7 * (2 * 19 + 4)];
    __shared__ u32 s_btype;
    __shared__ u32 s_lenfreq[19], s_rk[3], s_lm2;
    __shared__ u8 s_lenenc[19], s_lenmap[19], s_lentab[19];
    __shared__ u16 s_lcode[19];
    static_assert(kBThreads >= 128, "a wave for each of the block's two tables");
    const u32 tid = threadIdx.x, k = blockIdx.x;
    // kr: the blocks of this launch are kr[0] .. kr[1] - 1 (the chain of block starts is still running for the blocks
    // behind them; only the last piece knows the number of blocks)
    u32 nblocks;
    if (kr) {
        const u32 lo = kr[0], hi = kr[1];
        if (hi == 0xFFFFFFFFu || k < lo || k >= hi) return;
        nblocks = piece_last ? *nb_p : 0xFFFFFFFEu;
    } else {
        nblocks = *nb_p;
        if (nblocks == 0xFFFFFFFFu || k >= nblocks) return;
    }
    const u64 b0 = bstart[k], b1 = bstart[k + 1];
    const bool is_final = (k + 1 == nblocks) && last_is_final; // (a flushed segment ends with a non-final block)
    for (u32 i = tid; i < 288; i += kBThreads) s_sf[i] = 0;
    if (tid < 32) s_of[tid] = 0;
    if (tid == 0) s_lm = 0;
    __syncthreads();
    for (u64 q0 = b0; q0 < b1; q0 += kBThreads * 16) { // sixteen independent loads in flight per thread
        u32 c[16], lit[16];
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            const u64 q = q0 + (u64)j * kBThreads + tid;
            c[j] = q < b1 ? code[q] : 0u;
            lit[j] = q < b1 ? in[q] : 0u; // (beside the code word, not behind it: one round trip per batch)
        }
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            if (!(c[j] & F_CODE)) continue;
            if (c[j] & F_REF) {
                u32 lc, eb, ev, dc;
                len_code((c[j] & 511u) - 3, lc, eb, ev);
                dist_code((c[j] >> 9) & 32767u, dc, eb, ev);
                atomicAdd(&s_sf[257 + lc], 1u);
                atomicAdd(&s_of[dc], 1u);
            } else atomicAdd(&s_sf[lit[j]], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) s_sf[256] += 1; // init_block :274-279
    __syncthreads();
    u32 *lm_scr = lm_scratch + (size_t)k * kDfLmWords;
    // the two tables side by side, a wave each (the heap lives in the wave's registers: df_make_table_wave)
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63u;
    if (wave == 0) {
        const u32 r = df_make_table_wave<9>(s_sf, 286, 15, s_sl, s_buf[0], s_w[0], s_tmp[0], lm_scr, &s_lm, lane);
        if (lane == 0) s_n[0] = r;
    } else if (wave == 1) {
        const u32 r = df_make_table_wave<1>(s_of, 30, 15, s_ol, s_buf[1], s_w[1], s_tmp[1], lm_scr + kDfLmTable, &s_lm, lane);
        if (lane == 0) s_n[1] = r;
    }
    __syncthreads();
    // create_custom_huffman_table :395-452: the run coding of both tables (one lane), its code (the wave), the header
    if (tid < 19) { s_lenfreq[tid] = 0; s_lenenc[tid] = 0; }
    if (tid == 0) { s_lm2 = 0; }
    __syncthreads();
    if (tid == 0) {
        s_rk[0] = df_tab_runs(s_sl, s_n[0], s_ls[0], s_le[0], s_lenfreq);
        s_rk[1] = df_tab_runs(s_ol, s_n[1], s_ls[1], s_le[1], s_lenfreq);
    }
    __syncthreads();
    if (wave == 0) {
        const u32 r = df_make_table_wave<1>(s_lenfreq, 19, 7, s_lenenc, s_buf[0], s_w[0], s_tmp[0], s_lm7, &s_lm2, lane);
        if (lane == 0) s_rk[2] = r;
    }
    __syncthreads();
    if (tid == 0) {
        const u32 sym_n = s_n[0], off_n = s_n[1];
        const u32 symk = s_rk[0], offk = s_rk[1];
        const u8 *len_enc = s_lenenc;
        const u32 lm2 = s_lm2;
        const u32 len_enc_n = s_rk[2];
        // (small arrays with computed indices live in LDS: on the stack they cost the kernel a private segment)
        u8 *len_map = s_lenmap, *len_tab = s_lentab;
        { // {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2}
            const u64 m0 = 0x0507090B0D0F1103ull, m1 = 0x12100E0C0A080604ull;
            for (u32 i = 0; i < 8; ++i) { len_map[i] = (u8)(m0 >> (8 * i)); len_map[8 + i] = (u8)(m1 >> (8 * i)); }
            len_map[16] = 0; len_map[17] = 1; len_map[18] = 2;
        }
        for (u32 i = 0; i < 19; ++i) len_tab[i] = 0;
        u32 len_count = 3;
        for (u32 i = 0; i < len_enc_n; ++i)
            if (len_enc[i]) { len_tab[len_map[i]] = len_enc[i]; if (len_map[i] > len_count) len_count = len_map[i]; }
        u32 hlit = 0, hdist = 0;
        for (u32 i = 0; i < sym_n; ++i) if (s_sl[i]) hlit = i;
        hlit -= 256;
        for (u32 i = 0; i < off_n; ++i) if (s_ol[i]) hdist = i;
        u16 *lcode = s_lcode;
        for (u32 i = 0; i < 19; ++i) lcode[i] = 0;
        df_make_codes(len_enc, len_enc_n, lcode);
        LsbSink sk;
        sk.w = s_hdr; sk.acc = 0; sk.nacc = 0; sk.widx = 0;
        sk.put(is_final ? 1u : 0u, 1);
        sk.put(2, 2);
        sk.put(hlit, 5);
        sk.put(hdist, 5);
        sk.put(len_count - 3, 4);
        for (u32 i = 0; i <= len_count; ++i) sk.put(len_tab[i], 3);
        for (u32 which = 0; which < 2; ++which) {
            const u32 cnt = which ? offk : symk;
            for (u32 i = 0; i < cnt; ++i) {
                const u32 s = s_ls[which][i];
                sk.put(lcode[s], len_enc[s]);
                if (s == 16) sk.put(s_le[which][i], 2);
                else if (s == 17) sk.put(s_le[which][i], 3);
                else if (s == 18) sk.put(s_le[which][i], 7);
            }
        }
        sk.finish();
        const u32 hdr_bits = sk.bits(); // BFINAL + the header the reference counts (which includes BTYPE)
        // cals_comp_len :549-575 for the custom and the fixed tables
        u64 custom = hdr_bits - 1, fixed = 2;
        for (u32 i = 0; i < 286; ++i) {
            const u32 f = s_sf[i];
            if (!f) continue;
            const u32 ext = i >= 257 ? len_ext_bits(i - 257) : 0u;
            if (i < sym_n) custom += (u64)f * (s_sl[i] + ext);
            const u32 fl = i < 144 ? 8u : (i < 256 ? 9u : (i < 280 ? 7u : 8u));
            fixed += (u64)f * (fl + ext);
        }
        for (u32 i = 0; i < 30; ++i) {
            const u32 f = s_of[i];
            if (!f) continue;
            const u32 ext = dist_ext_bits(i);
            if (i < off_n) custom += (u64)f * (s_ol[i] + ext);
            fixed += (u64)f * (5u + ext);
        }
        // decompress_len: the bytes of this block's codes -- and, for the first block of a segment behind a
        // flush, the dl0 bytes the counter still holds (a stored block then repeats them, :488-501)
        const u64 dlen = b1 - b0 + (k == 0 ? dl0 : 0u);
        const u64 original = (dlen << 3) + 2 + 16 + 16;
        DfBlock o;
        o.bytes = (u32)dlen;
        o.lm = s_lm + lm2;
        if (original <= custom && original <= fixed) { o.btype = 0; o.hdr_bits = 3; o.bits = 0; s_hdr[0] = is_final ? 1u : 0u; }
        else if (fixed <= custom) { o.btype = 1; o.hdr_bits = 3; o.bits = 1 + fixed; s_hdr[0] = (is_final ? 1u : 0u) | 2u; }
        else { o.btype = 2; o.hdr_bits = hdr_bits; o.bits = 1 + custom; if (off_n == 0) o.lm |= 0x100u; } // no distance code at all
        blocks[k] = o;
        s_btype = o.btype;
    }
    __syncthreads();
    const u32 btype = s_btype;
    for (u32 i = tid; i < 320; i += kBThreads) { // the lengths the emission codes with: 288 literal/length + 32 distance
        u8 v = 0;
        if (i < 288) {
            if (btype == 1) v = (u8)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8))); // deflate/mod.rs:15-21
            else if (btype == 2 && i < 286) v = s_sl[i];
        } else {
            if (btype == 1) v = 5;                                                       // :23-25
            else if (btype == 2 && i - 288 < 30) v = s_ol[i - 288];
        }
        lens[(size_t)k * 320 + i] = v;
    }
    for (u32 i = tid; i < kHdrWords; i += kBThreads) hdr[(size_t)k * kHdrWords + i] = s_hdr[i];
}

// bit offset of every block; a stored block is padded to a byte behind its 3 header bits (:488-501)
__global__ __launch_bounds__(256) void k_df_offsets(DfBlock *__restrict__ blocks, const u32 *__restrict__ nb_p,
                                                    u64 *__restrict__ total_bits, u32 bit0)
{
    __shared__ u64 s_bits[1024]; // bits, or ~bytes for a stored block
    __shared__ u64 s_off[1024];
    __shared__ u64 s_wv[4];
    __shared__ u64 s_run;
    const u32 nb = *nb_p;
    if (nb == 0xFFFFFFFFu) return;
    if (threadIdx.x == 0) s_run = bit0; // (a later part of a long stream goes on inside the byte its predecessor ended in)
    for (u32 k0 = 0; k0 < nb; k0 += 1024) {
        __syncthreads();
        for (u32 j = threadIdx.x; j < 1024 && k0 + j < nb; j += 256) {
            const DfBlock b = blocks[k0 + j];
            s_bits[j] = b.btype == 0 ? ~(u64)b.bytes : b.bits;
        }
        __syncthreads();
        // a batch without a stored block is a plain prefix sum (stored blocks are padded to a byte: serial)
        bool stored = false;
        for (u32 j = threadIdx.x; j < 1024 && k0 + j < nb; j += 256) stored = stored || (i64)s_bits[j] < 0;
        if (!__syncthreads_or(stored)) {
            const u32 j0 = threadIdx.x * 4;
            u64 v[4], sum = 0;
            for (u32 j = 0; j < 4; ++j) { v[j] = (k0 + j0 + j < nb) ? s_bits[j0 + j] : 0ull; sum += v[j]; }
            u64 inc = sum;
            for (u32 dlt = 1; dlt < 64; dlt <<= 1) {
                const u64 o = __shfl_up(inc, dlt);
                if ((threadIdx.x & 63u) >= dlt) inc += o;
            }
            if ((threadIdx.x & 63u) == 63u) s_wv[threadIdx.x >> 6] = inc;
            __syncthreads();
            u64 carry = s_run;
            for (u32 w = 0; w < (threadIdx.x >> 6); ++w) carry += s_wv[w];
            u64 off = carry + inc - sum;
            for (u32 j = 0; j < 4; ++j) { s_off[j0 + j] = off; off += v[j]; }
            __syncthreads();
            if (threadIdx.x == 255) s_run = off;
        } else if (threadIdx.x == 0) {
            u64 off = s_run;
            for (u32 j = 0; j < 1024 && k0 + j < nb; ++j) {
                s_off[j] = off;
                const u64 v = s_bits[j];
                if ((i64)v < 0) off = ((off + 3 + 7) & ~7ull) + 32 + 8ull * ~v;
                else off += v;
            }
            s_run = off;
        }
        __syncthreads();
        for (u32 j = threadIdx.x; j < 1024 && k0 + j < nb; j += 256) blocks[k0 + j].bit_off = s_off[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) *total_bits = s_run;
}

__device__ __forceinline__ void or_bits(u32 *out, u64 bit, u64 v, u32 nbits)
{
    if (!nbits) return;
    const u64 w = bit >> 5;
    const u32 sh = (u32)bit & 31u;
    const u64 lo = v << sh;
    const u32 hi = sh ? (u32)(v >> (64 - sh)) : 0u;
    if ((u32)lo) atomicOr(&out[w], (u32)lo);
    if ((u32)(lo >> 32)) atomicOr(&out[w + 1], (u32)(lo >> 32));
    if (hi) atomicOr(&out[w + 2], hi);
}

__global__ __launch_bounds__(kEThreads) void k_df_emit(const u8 *__restrict__ in, const u32 *__restrict__ code,
                                                       const u64 *__restrict__ bstart, const u32 *__restrict__ nb_p,
                                                       const DfBlock *__restrict__ blocks, const u8 *__restrict__ lens,
                                                       const u32 *__restrict__ hdr, u32 *__restrict__ out)
{
    __shared__ u16 s_sc[288], s_oc[32];
    __shared__ u8 s_sl[288], s_ol[32];
    __shared__ u32 s_wsum[kEThreads / 64];
    __shared__ u64 s_base;
    const u32 tid = threadIdx.x, k = blockIdx.x;
    const u32 nblocks = *nb_p;
    if (nblocks == 0xFFFFFFFFu || k >= nblocks) return;
    const DfBlock bi = blocks[k];
    const u64 b0 = bstart[k], b1 = bstart[k + 1];
    u8 *out8 = reinterpret_cast<u8 *>(out);
    if (bi.btype == 0) {
        if (tid == 0) or_bits(out, bi.bit_off, hdr[(size_t)k * kHdrWords] & 7u, 3);
        const u64 B = (bi.bit_off + 3 + 7) >> 3;
        if (tid == 0) {
            const u32 l = bi.bytes & 0xFFFFu, nl = l ^ 0xFFFFu;
            out8[B] = (u8)l; out8[B + 1] = (u8)(l >> 8); out8[B + 2] = (u8)nl; out8[B + 3] = (u8)(nl >> 8);
        }
        // the last `bytes` bytes up to the block's end (nocomp_buf[decompress_len - i], :496-499): behind a flush
        // that reaches back in front of the block, into bytes an earlier block has carried already
        const u8 *src = in + b1 - bi.bytes;
        for (u64 i = tid; i < bi.bytes; i += kEThreads) out8[B + 4 + i] = src[i];
        return;
    }
    for (u32 i = tid; i < 288; i += kEThreads) { s_sl[i] = lens[(size_t)k * 320 + i]; s_sc[i] = 0; }
    if (tid < 32) { s_ol[tid] = lens[(size_t)k * 320 + 288 + tid]; s_oc[tid] = 0; }
    __syncthreads();
    if (tid == 0) df_make_codes(s_sl, 288, s_sc);
    if (tid == kEThreads / 2) df_make_codes(s_ol, 32, s_oc); // (another wave where there is one)
    // header bits (BFINAL first)
    for (u32 i = tid; i * 32 < bi.hdr_bits; i += kEThreads) {
        const u32 nbits = bi.hdr_bits - i * 32 < 32 ? bi.hdr_bits - i * 32 : 32;
        or_bits(out, bi.bit_off + (u64)i * 32, hdr[(size_t)k * kHdrWords + i], nbits);
    }
    if (tid == 0) s_base = bi.bit_off + bi.hdr_bits;
    __syncthreads();
    const u32 lane = tid & 63u, wave = tid >> 6;
    // 4 consecutive positions per thread: their code words and text bytes are loaded together (16 + 4 bytes, one
    // round trip), and the next chunk's are on their way while this one is coded
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((aligned(4))) u32x4u;
    auto fetch = [&](u64 q0, u32 (&cw)[4], u32 &tx) {
        cw[0] = cw[1] = cw[2] = cw[3] = 0; tx = 0;
        if (q0 + 4 <= b1) {
            const u32x4 w = *reinterpret_cast<const u32x4u *>(code + q0);
            cw[0] = w.x; cw[1] = w.y; cw[2] = w.z; cw[3] = w.w;
            tx = *reinterpret_cast<const df_u32u *>(in + q0);
        } else
            for (u32 j = 0; j < 4 && q0 + j < b1; ++j) { cw[j] = code[q0 + j]; tx |= (u32)in[q0 + j] << (8 * j); }
    };
    u32 ncw[4], ntx;
    fetch(b0 + (u64)tid * 4, ncw, ntx);
    for (u64 c0 = b0; c0 < b1; c0 += kEThreads * 4) {
        u64 v[4];
        u32 nb[4];
        u32 tot = 0;
        u32 cw[4] = {ncw[0], ncw[1], ncw[2], ncw[3]};
        const u32 tx = ntx;
        if (c0 + kEThreads * 4 < b1) fetch(c0 + kEThreads * 4 + (u64)tid * 4, ncw, ntx);
        for (u32 j = 0; j < 4; ++j) {
            v[j] = 0; nb[j] = 0;
            const u32 c = cw[j];
            if (!(c & F_CODE)) continue; // (also every position at or behind b1: its word was not loaded)
            if (c & F_REF) {
                u32 lc, leb, lev, dc, deb, dev;
                len_code((c & 511u) - 3, lc, leb, lev);
                dist_code((c >> 9) & 32767u, dc, deb, dev);
                u64 x = s_sc[257 + lc];
                u32 nn = s_sl[257 + lc];
                x |= (u64)lev << nn; nn += leb;
                x |= (u64)s_oc[dc] << nn; nn += s_ol[dc];
                x |= (u64)dev << nn; nn += deb;
                v[j] = x; nb[j] = nn;
            } else {
                const u32 sy = (tx >> (8 * j)) & 0xFFu;
                v[j] = s_sc[sy]; nb[j] = s_sl[sy];
            }
            tot += nb[j];
        }
        // exclusive scan of tot over the workgroup
        u32 inc = tot;
        for (u32 d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        u32 wbase = 0, all = 0;
        for (u32 w = 0; w < kEThreads / 64; ++w) {
            const u32 t = s_wsum[w];
            if (w < wave) wbase += t;
            all += t;
        }
        u64 bit = s_base + wbase + (inc - tot);
        for (u32 j = 0; j < 4; ++j) {
            or_bits(out, bit, v[j], nb[j]);
            bit += nb[j];
        }
        __syncthreads();
        if (tid == 0) s_base += all;
        __syncthreads();
    }
    if (tid == 0) or_bits(out, s_base, s_sc[256], s_sl[256]); // end of block
}

// ---------------------------------------------------------------------------------- checksums (f-3)
// per 64 KiB piece: sum of bytes and sum of (len - i) * byte (Adler-32, adler32.rs:20-66), and the
// reflected CRC-32 of the piece with a zero register (crc32.rs:40-55, 74-78); the host combines
__global__ __launch_bounds__(256) void k_df_sums(const u8 *__restrict__ in, u64 n, u64 *__restrict__ asum,
                                                 u64 *__restrict__ bsum, u32 *__restrict__ crc, u32 *__restrict__ last_sub,
                                                 DfCrcShifts xk)
{
    __shared__ u32 s_tab[256];
    __shared__ u64 s_a[256], s_b[256];
    __shared__ u32 s_c[256];
    __shared__ u32 s_piece[kSumPiece / 4 + 256];
    const u32 tid = threadIdx.x;
    {
        u32 c = tid;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        s_tab[tid] = c;
    }
    __syncthreads();
    const u64 p0 = (u64)blockIdx.x * kSumPiece;
    const u64 plen = (n - p0) < (u64)kSumPiece ? (n - p0) : (u64)kSumPiece;
    // the piece is staged with coalesced loads; a thread's 256 bytes are 64 words apart from its neighbour's, plus
    // one word per thread so that the threads of a wave read different banks
    for (u32 w = tid; w < kSumPiece / 4; w += 256) {
        const u64 g = p0 + 4ull * w;
        u32 v = 0;
        if ((reinterpret_cast<uintptr_t>(in) & 3u) == 0 && g + 4 <= n) v = *reinterpret_cast<const u32 *>(in + g);
        else
            for (u32 b = 0; b < 4 && g + b < n; ++b) v |= (u32)in[g + b] << (8 * b);
        s_piece[w + (w >> 6)] = v;
    }
    __syncthreads();
    // thread t: bytes [t*256, t*256+256) of the piece
    const u64 s0 = (u64)tid * 256;
    u64 a = 0, b = 0;
    u32 c = 0;
    for (u32 wq = 0; wq < 64; ++wq) {
        const u32 v = s_piece[tid * 65 + wq];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const u64 i = s0 + wq * 4 + k;
            if (i < plen) {
                const u32 d = (v >> (8 * k)) & 0xFFu;
                a += d;
                b += (plen - i) * d;
                c = s_tab[(c ^ d) & 0xFFu] ^ (c >> 8);
            }
        }
    }
    s_a[tid] = a; s_b[tid] = b; s_c[tid] = c;
    __syncthreads();
    if (tid == 0) {
        u64 sa = 0, sb = 0;
        for (u32 t = 0; t < 256; ++t) { sa += s_a[t]; sb += s_b[t]; }
        asum[blockIdx.x] = sa;
        bsum[blockIdx.x] = sb;
    }
    // a full piece folds its 256 sub-piece registers here: crc(A || B) = crc(A) * x^(8 |B|) ^ crc(B) for registers
    // that start at zero, eight levels with x^(8 * 256 * 2^level) from the host.  The last, partial piece hands
    // its sub-piece registers to the host (their lengths differ).
    if (plen == kSumPiece) {
        for (u32 lv = 0; lv < 8; ++lv) {
            const u32 stride = 1u << lv;
            if ((tid & (2 * stride - 1)) == 0) {
                u32 a0 = s_c[tid], bb = xk.x[lv], pr = 0; // a0 * bb in GF(2)[x] / P, reflected (bit 31 = x^0)
                for (u32 m = 1u << 31; m != 0 && a0 != 0; m >>= 1) {
                    if (a0 & m) { pr ^= bb; a0 &= ~m; }
                    bb = (bb & 1u) ? (bb >> 1) ^ 0xEDB88320u : bb >> 1;
                }
                s_c[tid] = pr ^ s_c[tid + stride];
            }
            __syncthreads();
        }
        if (tid == 0) crc[blockIdx.x] = s_c[0];
    } else last_sub[tid] = s_c[tid];
}

// ---------------------------------------------------------------------------------- launchers
#define DFCHK(x) do { if ((x) != hipSuccess) return -1; } while (0)

u32 df_chunks(u64 n)
{
    const u64 ntri = n >= 3 ? n - 2 : 0;
    return ntri ? (u32)((ntri + kChunk - 1) / kChunk) : 0u;
}

// v0, s: df_chunks(n) * kChunkStride words each, hs as many u16; hist: df_chunks(n) * kChunkTiles * 256 words;
// tbase: 2 * df_chunks(n) * 256 words (the digit bases of both passes)
int df_launch_chains(hipStream_t st, const u8 *in, u64 n, u32 *v0, u32 *s, u16 *hs, u32 *hist, u32 *tbase, u32 *pe)
{
    const u64 ntri = n >= 3 ? n - 2 : 0;
    // every trigram position is written by the chunk that owns it; the last two positions have no trigram
    if (pe) DFCHK(hipMemsetAsync(pe + ntri, 0, (n - ntri + 8) * sizeof(u32), st));
    if (!ntri) return 0;
    const u32 nchunks = df_chunks(n);
    const dim3 tiles(nchunks * kChunkTiles);
    hipLaunchKernelGGL((k_df_shist<0>), tiles, dim3(kSortThreads), 0, st, in, ntri, (const u32 *)nullptr, hist);
    hipLaunchKernelGGL(k_df_sscan, dim3(nchunks), dim3(256), 0, st, hist, tbase);
    u32 *tbase1 = tbase + (size_t)nchunks * 256;
    hipLaunchKernelGGL((k_df_sscatter<0>), tiles, dim3(kSortThreads), 0, st, in, ntri, (const u32 *)nullptr, hist, tbase, v0,
                       (const u32 *)nullptr, (u16 *)nullptr);
    hipLaunchKernelGGL((k_df_shist<1>), tiles, dim3(kSortThreads), 0, st, in, ntri, v0, hist);
    hipLaunchKernelGGL(k_df_sscan, dim3(nchunks), dim3(256), 0, st, hist, tbase1);
    // (the hashes of the sorted positions are only kept for k_df_prev: k_df_match2 has the trigrams in hand)
    hipLaunchKernelGGL((k_df_sscatter<1>), tiles, dim3(kSortThreads), 0, st, in, ntri, v0, hist, tbase1, s, tbase, pe ? hs : (u16 *)nullptr);
    if (pe) hipLaunchKernelGGL(k_df_prev, dim3(nchunks * kPrevSpan), dim3(256), 0, st, ntri, s, hs, pe);
    return 0;
}

// matches straight from the sorted chunks (s of df_launch_chains with pe == nullptr)
int df_launch_match2(hipStream_t st, const u8 *in, u64 n, const u32 *s, u32 *M)
{
    if (!n) return 0;
    const u64 ntri = n >= 3 ? n - 2 : 0;
    DFCHK(hipMemsetAsync(M + ntri, 0, (n - ntri + 8) * sizeof(u32), st)); // the last two positions have no trigram
    if (!ntri) return 0;
    hipLaunchKernelGGL(k_df_match2, dim3(df_chunks(n) * kM2Span), dim3(kM2Threads), 0, st, in, n, ntri, s, M);
    return 0;
}

int df_launch_match(hipStream_t st, const u8 *in, const u32 *pe, u64 n, u32 *M)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_df_match, dim3((u32)((n + kMTile - 1) / kMTile)), dim3(kMThreads), 0, st, in, pe, n, M);
    return 0;
}

// canon: counts[0] * 64 words for the tiles' canonical orbits (k_df_tile_orbit / k_df_mark2), or nullptr for the
// doubling kernels of rounds 1 and 2 (k_df_tile_tab / k_df_mark)
int df_launch_parse(hipStream_t st, const u32 *M, u64 n, u16 *step, u16 *const *tabs, u16 *const *ents, const u32 *counts,
                    u32 nlevels, u32 *code, u64 *bm, u64 *canon, const DfPiecewiseCuts *pc)
{
    if (!n) return 0;
    DFCHK(hipMemsetAsync(bm, 0, ((n + 63) / 64 + 2) * sizeof(u64), st));
    hipLaunchKernelGGL(k_df_adv, dim3((u32)((n + 1023) / 1024)), dim3(256), 0, st, M, n, step);
    if (canon) hipLaunchKernelGGL(k_df_tile_orbit, dim3(counts[0]), dim3(kOrbThreads), 0, st, step, n, tabs[0], canon);
    else hipLaunchKernelGGL(k_df_tile_tab, dim3(counts[0]), dim3(kTabThreads), 0, st, step, n, tabs[0]);
    for (u32 l = 1; l + 1 < nlevels; ++l)
        hipLaunchKernelGGL(k_df_compose, dim3(counts[l]), dim3(320), 0, st, tabs[l - 1], counts[l - 1], tabs[l]);
    DFCHK(hipMemsetAsync(ents[nlevels - 1], 0, sizeof(u16), st)); // the single top group is entered at 0
    for (u32 l = nlevels - 1; l >= 1; --l)
        hipLaunchKernelGGL(k_df_resolve, dim3((counts[l] + 63) / 64), dim3(64), 0, st, tabs[l - 1], counts[l - 1], ents[l],
                           counts[l], ents[l - 1]);
    if (!canon) hipLaunchKernelGGL(k_df_mark, dim3(counts[0]), dim3(kMarkThreads), 0, st, step, M, ents[0], n, code, bm);
    else if (!pc) hipLaunchKernelGGL(k_df_mark2, dim3(counts[0]), dim3(kMark2Threads), 0, st, step, M, ents[0], tabs[0], canon, n, code, bm, 0u);
    else {
        // The tiles are marked in pieces, and the chain of block starts -- one workgroup, 16 385 dependent hops per GiB,
        // 4 ms on its own -- runs on a second stream beside the marking of the NEXT piece: a piece of the chain goes
        // as far as the bits of the pieces marked so far are final (k_df_cuts: limit).  Plain stream dependencies,
        // nothing waits on the device for anything.
        const u32 ntiles = counts[0], np = df_cut_pieces(ntiles);
        for (u32 i = 0; i < np; ++i) {
            const u32 lo = (u32)((u64)ntiles * i / np), hi = (u32)((u64)ntiles * (i + 1) / np);
            hipLaunchKernelGGL(k_df_mark2, dim3(hi - lo), dim3(kMark2Threads), 0, st, step, M, ents[0], tabs[0], canon, n, code, bm, lo);
            DFCHK(hipEventRecord(pc->ev[i], st));
            DFCHK(hipStreamWaitEvent(pc->st2, pc->ev[i], 0));
            const u64 limit = i + 1 == np ? ~0ull : (u64)hi * kPTile;
            hipLaunchKernelGGL(k_df_cuts, dim3(1), dim3(256), 0, pc->st2, bm, n, pc->bstart, pc->nb, pc->cap, pc->dl0, pc->first, limit,
                               i ? 1u : 0u, pc->state, pc->kdone ? pc->kdone + i + 1 : (u32 *)nullptr);
            if (pc->kdone) DFCHK(hipEventRecord(pc->evc[i], pc->st2)); // (the caller lets the blocks of this piece follow)
        }
        if (!pc->kdone) {
            DFCHK(hipEventRecord(pc->ev[kCutPieces], pc->st2));
            DFCHK(hipStreamWaitEvent(st, pc->ev[kCutPieces], 0));
        }
    }
    return 0;
}

int df_launch_cuts(hipStream_t st, u64 n, u64 *bm, u64 *bstart, u32 *nb, u32 cap, u32 dl0, u32 first)
{
    hipLaunchKernelGGL(k_df_cuts, dim3(1), dim3(256), 0, st, bm, n, bstart, nb, cap, dl0, first, ~0ull, 0u, (u64 *)nullptr, (u32 *)nullptr);
    return 0;
}

u32 df_cut_pieces(u32 ntiles) { return ntiles < kCutPieces * 64 ? 1u : kCutPieces; }

// the blocks kr[0] .. kr[1] - 1 (their ends are known: a piece of the chain has passed them)
int df_launch_blocks_piece(hipStream_t st, const u8 *in, const u32 *code, u64 *bstart, u32 *nb, u32 cap, DfBlock *blocks, u8 *lens,
                           u32 *hdr, u32 *lm_scratch, u32 dl0, u32 last_is_final, const u32 *kr, u32 piece_last)
{
    hipLaunchKernelGGL(k_df_block, dim3(cap), dim3(kBThreads), 0, st, in, code, bstart, nb, blocks, lens, hdr, lm_scratch, dl0,
                       last_is_final, kr, piece_last);
    return 0;
}
int df_launch_block_offsets(hipStream_t st, DfBlock *blocks, const u32 *nb, u64 *total_bits, u32 bit0)
{
    hipLaunchKernelGGL(k_df_offsets, dim3(1), dim3(256), 0, st, blocks, nb, total_bits, bit0);
    return 0;
}

int df_launch_blocks(hipStream_t st, const u8 *in, const u32 *code, u64 *bstart, u32 *nb, u32 cap,
                     DfBlock *blocks, u8 *lens, u32 *hdr, u32 *lm_scratch, u64 *total_bits, u32 dl0, u32 last_is_final, u32 bit0)
{
    hipLaunchKernelGGL(k_df_block, dim3(cap), dim3(kBThreads), 0, st, in, code, bstart, nb, blocks, lens, hdr, lm_scratch, dl0,
                       last_is_final, (const u32 *)nullptr, 1u);
    hipLaunchKernelGGL(k_df_offsets, dim3(1), dim3(256), 0, st, blocks, nb, total_bits, bit0);
    return 0;
}

int df_launch_emit(hipStream_t st, const u8 *in, const u32 *code, const u64 *bstart, const u32 *nb, u32 cap,
                   const DfBlock *blocks, const u8 *lens, const u32 *hdr, u32 *out)
{
    hipLaunchKernelGGL(k_df_emit, dim3(cap), dim3(kEThreads), 0, st, in, code, bstart, nb, blocks, lens, hdr, out);
    return 0;
}

int df_launch_sums(hipStream_t st, const u8 *in, u64 n, u64 *asum, u64 *bsum, u32 *crc, u32 *last_sub, DfCrcShifts xk)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_df_sums, dim3((u32)((n + kSumPiece - 1) / kSumPiece)), dim3(256), 0, st, in, n, asum, bsum, crc, last_sub, xk);
    return 0;
}

// ---- what the host used to look at between the stages of a PART of a long stream (deflate_engine.hip) ---------------------
// A part that is not the last one keeps the blocks that START at or before n - guard and hands the next part the STEP of
// the parse that holds the first block left out.  Rounds 1-4 read the block starts and three code words back in the middle
// of the part's kernels (four round trips with the GPU idle; behind the copying thread of a host-buffer call each of them
// took a fraction of a millisecond); the decision is three comparisons.
__global__ __launch_bounds__(256) void k_df_part_keep(const u64 *__restrict__ bstart, u32 *__restrict__ nb, u32 bcap,
                                                        const u32 *__restrict__ code, u64 n, u64 guard, DfPartRes *__restrict__ res)
{
    __shared__ u32 s_keep;
    const u32 nb_all = *nb;
    if (threadIdx.x == 0) s_keep = 0;
    __syncthreads();
    const bool bad = nb_all == 0xFFFFFFFFu || nb_all > bcap || n <= guard;
    if (!bad) {
        u32 best = 0; // (block starts ascend: the blocks kept are a prefix)
        for (u32 k = threadIdx.x; k < nb_all; k += blockDim.x)
            if (bstart[k] + guard <= n) best = k + 1u;
        if (best) atomicMax(&s_keep, best);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const u32 keep = s_keep;
    u32 err = bad ? 1u : 0u;
    if (!err && (keep == 0u || keep >= nb_all)) err = 2u; // (a part is much longer than the guard)
    u64 bcut = 0;
    u32 skip = 0;
    if (!err) {
        bcut = bstart[keep];
        if (bcut < 2) err = 3u;
    }
    if (!err) { // the step of the parse that emitted the code at bcut: at most two literals in front of it
        const u32 c0 = code[bcut - 2], c1 = code[bcut - 1], c2 = code[bcut];
        skip = (c2 & F_STEP) ? 0u : ((c1 & F_STEP) ? 1u : 2u);
        const u32 cs = skip == 0u ? c2 : (skip == 1u ? c1 : c0);
        if (!(cs & F_STEP) || !(c2 & F_CODE)) err = 4u;
    }
    res->nb_all = nb_all;
    res->err = err;
    res->skip = skip;
    res->consumed = bcut - skip;
    *nb = err ? 0u : keep; // (an error: the kernels behind this one find nothing to do, the host reports it)
}

// behind the emission: the partly filled byte a part hands on, where the last block starts, what kinds of blocks there were
__global__ __launch_bounds__(256) void k_df_part_tail(const DfBlock *__restrict__ blocks, const u64 *__restrict__ bstart, const u32 *__restrict__ nb,
                                                        const u64 *__restrict__ total_bits, const u8 *__restrict__ stream, DfPartRes *__restrict__ res)
{
    __shared__ u32 s_st[6];
    if (threadIdx.x < 6) s_st[threadIdx.x] = 0;
    __syncthreads();
    const u32 n = *nb;
    if (n != 0xFFFFFFFFu) {
        u32 c[6] = {0, 0, 0, 0, 0, 0};
        for (u32 k = threadIdx.x; k < n; k += blockDim.x) {
            const DfBlock b = blocks[k];
            c[b.btype < 3u ? b.btype : 0u] += 1u;
            c[3] += b.lm & 0xFFu;
            c[4] += (b.lm >> 8) & 1u;
        }
#pragma unroll
        for (u32 i = 0; i < 5; ++i)
            if (c[i]) atomicAdd(&s_st[i], c[i]);
    }
    __syncthreads();
    if (threadIdx.x < 5) res->st[threadIdx.x] = s_st[threadIdx.x];
    if (threadIdx.x == 0) {
        res->last_bstart = (n && n != 0xFFFFFFFFu) ? bstart[n - 1u] : 0ull;
        res->end_byte = stream[*total_bits >> 3];
    }
}

int df_launch_part_keep(hipStream_t st, const u64 *bstart, u32 *nb, u32 bcap, const u32 *code, u64 n, u64 guard, DfPartRes *res)
{
    hipLaunchKernelGGL(k_df_part_keep, dim3(1), dim3(256), 0, st, bstart, nb, bcap, code, n, guard, res);
    return 0;
}
int df_launch_part_tail(hipStream_t st, const DfBlock *blocks, const u64 *bstart, const u32 *nb, const u64 *total_bits, const u8 *stream, DfPartRes *res)
{
    hipLaunchKernelGGL(k_df_part_tail, dim3(1), dim3(256), 0, st, blocks, bstart, nb, total_bits, stream, res);
    return 0;
}

} // namespace dfgpu
