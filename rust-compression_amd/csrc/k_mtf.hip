// k_mtf.hip -- move-to-front + zero-run coding of the BWT last column.
//
// Reference being replaced: MtfPosition::pop (src/bzip2/mtf.rs:16-39), the symbol
// loop and zle_write of write_blockdata (src/bzip2/encoder.rs:324-358, 653-669).
//
// MTF is a serial recurrence on a <=256-entry list, made parallel the classic way:
//   M1  every 256-symbol chunk reports its "recency list" (distinct symbols, most
//       recent first) -- that is exactly the head of the MTF list after the chunk;
//   M2  one wave per block composes the chunk reports left to right and stores the
//       list each chunk starts from (3516 chunks/block, 256 B each);
//   M3  one lane per chunk replays its 256 symbols against its start list held in
//       LDS (65-dword stride => bank = lane + entry/4), 4 list entries per LDS
//       access, and writes the rank bytes.
// The list works on raw byte values, initialised with the in-use bytes in
// increasing order, which is the same thing as the reference's unseq2seq mapping
// (encoder.rs:304-318) because that mapping is monotone.
//   Z1-Z3  zero runs -> RUNA/RUNB digits (bijective base 2, LSB first), rank r>0 ->
//       symbol r+1, EOB appended, symbol histogram (LDS atomics).
#include "bzgpu.h"
#include <cstdlib>

namespace bzgpu {

constexpr u32 kChunkWGs = (kMaxMtfChunks + 255) / 256;
constexpr u32 kMtfSmallAlpha = 96;                     // blocks with at most this many symbols use k_mtf_ranks_small // 14 workgroups of 256 chunks

__device__ __forceinline__ u32 popc8(const u32 *b)
{
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c += __popc(b[i]);
    return c;
}

// ---- M1: recency list of each chunk ------------------------------------------------
__global__ __launch_bounds__(256) void k_mtf_summaries(MtfArgs a)
{
    __shared__ u32 s_seen[8 * 256]; // [word][thread]: conflict-free
    const u32 lb = blockIdx.y;
    const u32 n = a.blocks[lb].n;
    const u32 chunk = blockIdx.x * 256u + threadIdx.x;
    const u32 beg = chunk * kMtfChunk;
    if (beg >= n) return;
    const u32 end = (beg + kMtfChunk < n) ? beg + kMtfChunk : n;
    const u32 alpha = popc8(a.inuse_bits + lb * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) s_seen[i * 256 + threadIdx.x] = 0;
    const u8 *L = a.L + (size_t)lb * kSlot;
    u8 *out = a.summ + ((size_t)lb * kMaxMtfChunks + chunk) * 256u;
    u32 cnt = 0;
    // walk backwards, 64 bytes per visit (four 16-byte loads issued together: a line is fetched once and used
    // while it is there; chunk starts are 256-byte aligned)
    for (int v4 = (int)(kMtfChunk / 64u) - 1; v4 >= 0 && cnt < alpha; --v4) {
        if (beg + (u32)v4 * 64u >= end) continue;
        uint4 q4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 pl = beg + (u32)v4 * 64u + (u32)u * 16u;
            q4[u] = (pl < end) ? *reinterpret_cast<const uint4 *>(L + pl) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const u32 p0 = beg + (u32)v4 * 64u + (u32)u * 16u;
            if (p0 >= end || cnt >= alpha) continue;
            const u32 wv[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
#pragma unroll
            for (int k = 15; k >= 0; --k) {
                if (p0 + (u32)k < end) {
                    const u32 c = (wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                    const u32 bit = 1u << (c & 31u);
                    const u32 wd = s_seen[(c >> 5) * 256 + threadIdx.x];
                    if (!(wd & bit)) {
                        s_seen[(c >> 5) * 256 + threadIdx.x] = wd | bit;
                        out[cnt++] = (u8)c;
                    }
                }
            }
        }
    }
    a.summ_len[(size_t)lb * kMaxMtfChunks + chunk] = (u16)cnt;
}


// ---- M2 in three short steps instead of one chain of 1758 links per block ---------------------------------------
// A chunk's report R acts on the list as  list -> R ++ (list \ R)  (its symbols, most recent first, move to the
// front).  Two chunks in a row act as ONE report: R2 ++ (R1 \ R2) -- the same operation applied to the list R1.  So
// the reports of a GROUP of 32 chunks are merged into running prefixes (32 links, all groups side by side, in place
// over the reports), one wave per block takes the list through the 55 group totals (55 links), and every chunk's start
// list is its group's start list under the prefix in front of the chunk (one link, all chunks side by side).
// 1758 dependent links of four barriers each (1.48 ms per GiB) become 32 + 55 + 8.
constexpr u32 kMtfGroup = 32;
constexpr u32 kMtfGroups = (kMaxMtfChunks + kMtfGroup - 1) / kMtfGroup;

// list (s_list[0..len), one wave, lane l holds entries 4l..4l+3) -> R ++ (list \ R); R = m bytes, lane l's four in w.
// Returns the new length.  s_new / s_mark: 256 bytes each.  The caller's workgroup may hold several waves (each with
// its own arrays) as long as all of them call this the same number of times.
__device__ __forceinline__ u32 mtf_apply_report(u8 *s_list, u32 len, u32 m, u32 w, u8 *s_new, u8 *s_mark, u32 l)
{
    reinterpret_cast<u32 *>(s_mark)[l] = 0;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 i = l * 4u + k;
        if (i < m) {
            const u8 v = (u8)(w >> (8u * k));
            s_mark[v] = 1;
            s_new[i] = v;
        }
    }
    __syncthreads();
    u32 keep[4], nk = 0;
    u8 ev[4];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 e = l * 4 + k;
        ev[k] = s_list[e];
        keep[k] = (e < len && !s_mark[ev[k]]) ? 1u : 0u;
        nk += keep[k];
    }
    const u32 inc = wave_incl_sum(nk);
    u32 pos = m + inc - nk;
#pragma unroll
    for (u32 k = 0; k < 4; ++k)
        if (keep[k]) s_new[pos++] = ev[k];
    const u32 kept = __shfl(inc, 63, 64);
    __syncthreads();
    reinterpret_cast<u32 *>(s_list)[l] = reinterpret_cast<u32 *>(s_new)[l];
    __syncthreads();
    return m + kept;
}

// step 1: running merges of the reports inside every group (in place: summ[c] becomes the report of chunks g0..c)
__global__ __launch_bounds__(64) void k_mtf_merge_groups(MtfArgs a)
{
    __shared__ u8 s_cur[256], s_new[256], s_mark[256];
    const u32 lb = blockIdx.y, l = threadIdx.x;
    const u32 n = a.blocks[lb].n;
    const u32 nchunks = (n + kMtfChunk - 1) / kMtfChunk;
    const u32 c0 = blockIdx.x * kMtfGroup;
    if (c0 + 2u > nchunks) return; // (the last chunk's report is never applied)
    const size_t rep0 = (size_t)lb * kMaxMtfChunks;
    // (a list holds at most `alpha` bytes: only the lanes in front of that touch memory -- 36 bytes of a 256-byte
    // slot for text)
    const bool act = l * 4u < popc8(a.inuse_bits + lb * 8);
    u32 len = a.summ_len[rep0 + c0];
    reinterpret_cast<u32 *>(s_cur)[l] = act ? reinterpret_cast<const u32 *>(a.summ + (rep0 + c0) * 256u)[l] : 0u;
    __syncthreads();
    u32 c1 = c0 + kMtfGroup;
    if (c1 > nchunks - 1u) c1 = nchunks - 1u;
    u32 m_next = c0 + 1u < c1 ? (u32)a.summ_len[rep0 + c0 + 1u] : 0u;
    u32 w_next = (c0 + 1u < c1 && act) ? reinterpret_cast<const u32 *>(a.summ + (rep0 + c0 + 1u) * 256u)[l] : 0u;
    for (u32 c = c0 + 1u; c < c1; ++c) {
        const u32 m = m_next, w = w_next;
        if (c + 1u < c1) { // (the next report is on its way while this one is folded in)
            m_next = a.summ_len[rep0 + c + 1u];
            w_next = act ? reinterpret_cast<const u32 *>(a.summ + (rep0 + c + 1u) * 256u)[l] : 0u;
        }
        len = mtf_apply_report(s_cur, len, m, w, s_new, s_mark, l);
        if (act) reinterpret_cast<u32 *>(a.summ + (rep0 + c) * 256u)[l] = reinterpret_cast<u32 *>(s_cur)[l];
        if (l == 0) a.summ_len[rep0 + c] = (u16)len;
    }
}

// step 2: the list at the start of every group (one wave per block, the group totals one after the other)
__global__ __launch_bounds__(64) void k_mtf_group_starts(MtfArgs a)
{
    __shared__ u8 s_state[256], s_new[256], s_mark[256];
    const u32 lb = blockIdx.x, l = threadIdx.x;
    const u32 n = a.blocks[lb].n;
    const u32 nchunks = (n + kMtfChunk - 1) / kMtfChunk;
    const u32 *bits = a.inuse_bits + lb * 8;
    { // identity list: in-use byte values ascending (as in k_mtf_compose)
        u32 before = 0;
        for (u32 q = 0; q < (l * 4u) / 32u; ++q) before += __popc(bits[q]);
        const u32 wd = bits[(l * 4u) >> 5];
        const u32 sh = (l * 4u) & 31u;
        before += __popc(wd & ((1u << sh) - 1u));
        reinterpret_cast<u32 *>(s_state)[l] = 0;
        __syncthreads();
        u32 pos = before;
        for (u32 k = 0; k < 4; ++k)
            if ((wd >> (sh + k)) & 1u) s_state[pos++] = (u8)(l * 4u + k);
    }
    __syncthreads();
    const u32 alpha = popc8(bits);
    const size_t rep0 = (size_t)lb * kMaxMtfChunks;
    const bool act = l * 4u < alpha;
    for (u32 c0 = 0; c0 < nchunks; c0 += kMtfGroup) {
        if (act) reinterpret_cast<u32 *>(a.init_state + (rep0 + c0) * 256u)[l] = reinterpret_cast<u32 *>(s_state)[l];
        if (c0 + 2u > nchunks) break;
        u32 e = c0 + kMtfGroup - 1u; // the group's last report that is applied: its total
        if (e > nchunks - 2u) e = nchunks - 2u;
        const u32 m = a.summ_len[rep0 + e];
        const u32 w = act ? reinterpret_cast<const u32 *>(a.summ + (rep0 + e) * 256u)[l] : 0u;
        (void)mtf_apply_report(s_state, alpha, m, w, s_new, s_mark, l);
    }
}

// step 3: the list every other chunk starts from = its group's start list under the merged reports in front of it
__global__ __launch_bounds__(256) void k_mtf_chunk_starts(MtfArgs a)
{
    __shared__ u8 s_state[4][256], s_new[4][256], s_mark[4][256];
    const u32 lb = blockIdx.y, l = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 n = a.blocks[lb].n;
    const u32 nchunks = (n + kMtfChunk - 1) / kMtfChunk;
    const u32 c0 = blockIdx.x * kMtfGroup;
    if (c0 + 1u >= nchunks) return;
    const u32 alpha = popc8(a.inuse_bits + lb * 8);
    const size_t rep0 = (size_t)lb * kMaxMtfChunks;
    const bool act = l * 4u < alpha;
    const u32 g32 = act ? reinterpret_cast<const u32 *>(a.init_state + (rep0 + c0) * 256u)[l] : 0u;
    for (u32 k = 0; k < kMtfGroup / 4u; ++k) { // (uniform trip count: the barriers inside are the workgroup's)
        const u32 c = c0 + 1u + k * 4u + wv;
        const bool live = c < nchunks && c < c0 + kMtfGroup;
        reinterpret_cast<u32 *>(s_state[wv])[l] = g32;
        const u32 m = live ? (u32)a.summ_len[rep0 + c - 1u] : 0u;
        const u32 w = (live && act) ? reinterpret_cast<const u32 *>(a.summ + (rep0 + c - 1u) * 256u)[l] : 0u;
        (void)mtf_apply_report(s_state[wv], alpha, m, w, s_new[wv], s_mark[wv], l);
        if (live && act) reinterpret_cast<u32 *>(a.init_state + (rep0 + c) * 256u)[l] = reinterpret_cast<u32 *>(s_state[wv])[l];
    }
}


// ---- M3': the same for small alphabets, without the data-dependent list walk ------------------
// rank(i) = number of symbols used more recently than symbol(i) = #{c : last[c] > last[sym(i)]},
// where last[c] is the time of c's latest occurrence (start list: -1 for the front, -2, ...).
// Every lane does alpha/2 packed compares per symbol whatever the data, so the 64 lanes of a wave
// stay in step (the list walk above costs each step the LARGEST rank among its lanes).  The table
// lives in LDS as [pair of symbols][lane] 2 x i16: conflict-free, 48 KiB for <= 96 symbols.
constexpr u32 kMtfSmallPairs = kMtfSmallAlpha / 2;
constexpr u32 kMtfSubBlocks = 16; // batches of up to this many blocks: SUB = 8 (k_mtf_ranks_small)
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ short2_t as_short2(u32 v)
{
    short2_t r;
    __builtin_memcpy(&r, &v, 4);
    return r;
}
// Round 4: the same for LARGE alphabets (97 .. 256 symbols in use) with 128 lanes per workgroup -- 128 pairs x 128
// lanes x 4 B = 64 KB of LDS.  The list walk of k_mtf_ranks is a chain of dependent LDS read-modify-writes, as long as
// the symbol's rank and, in a wave, as long as the LARGEST rank among its lanes: 19.5 ms per 256 MiB of random bytes,
// 3.4 ms for ONE 98 KB block of the reference's sample1 (193 lanes with 512 symbols each, 6.7 us per symbol).  The
// compares are independent reads that the LDS pipelines: its bandwidth, not its latency, is what they cost.
// LANES chunks per workgroup, at most 2 * PAIRS symbols; the instance takes the blocks with ALPHA_LO < alpha <= 2 * PAIRS.
// SUB (round 6: a few blocks alone -- one 98 KB block of 256 symbols is 193 chunks, 0.6 ms of 512 dependent symbols on
// four waves): SUB neighbouring lanes share a chunk, each compares every SUB-th pair of the table and the partial counts
// are added across the lanes (the lanes of a chunk sit in one wave: they run in lockstep, the LDS serves them in order).
template <u32 PAIRS, u32 LANES, u32 ALPHA_LO, u32 SUB = 1>
__global__ __launch_bounds__(LANES) void k_mtf_ranks_small(MtfArgs a)
{
    constexpr u32 CL = LANES / SUB; // chunks per workgroup = columns of the table
    __shared__ u32 s_last[PAIRS * CL];
    __shared__ u8 s_code[256];
    const u32 lb = blockIdx.y;
    const u32 n = a.blocks[lb].n;
    const u32 chunk0 = blockIdx.x * CL;
    const u32 sub = threadIdx.x % SUB;
    if (chunk0 * kMtfChunk >= n) return;
    const u32 *bits = a.inuse_bits + lb * 8;
    const u32 alpha = popc8(bits);
    if (alpha > 2u * PAIRS || alpha <= ALPHA_LO) return;
    for (u32 v = threadIdx.x; v < 256u; v += LANES) { // byte value -> code (rank among the bytes in use)
        u32 before = 0;
        for (u32 q = 0; q < (v >> 5); ++q) before += __popc(bits[q]);
        before += __popc(bits[v >> 5] & ((1u << (v & 31u)) - 1u));
        s_code[v] = (u8)before;
    }
    __syncthreads();
    const u32 chunk = chunk0 + threadIdx.x / SUB;
    const u32 beg = chunk * kMtfChunk;
    if (beg >= n) return;
    const u32 end = (beg + kMtfChunk < n) ? beg + kMtfChunk : n;
    const u32 npairs = (alpha + 1u) >> 1;
    u32 *my = s_last + threadIdx.x / SUB;
    // (the compare loop below runs over blocks of 16 pairs: the rows behind the alphabet's last pair hold "never seen" too)
    constexpr u32 kPB = 32; // pairs per block of the pipelined compare loop (two blocks per trip)
    static_assert(PAIRS % (2u * kPB) == 0 || PAIRS <= kMtfSmallPairs, "the compare loop takes the table two blocks at a time");
    const u32 npairs16 = ((npairs + 2u * kPB - 1u) / (2u * kPB) * (2u * kPB)) < PAIRS ? ((npairs + 2u * kPB - 1u) / (2u * kPB) * (2u * kPB)) : PAIRS;
    for (u32 q = sub; q < (SUB > 1 ? PAIRS : npairs16); q += SUB) my[q * CL] = 0x80008000u; // both halves: -32768 = never seen
    if (SUB > 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (sub == 0) {
        // start list -> times -1, -2, ...
        const u8 *st = a.init_state + ((size_t)lb * kMaxMtfChunks + chunk) * 256u;
        for (u32 q = 0; q < alpha; ++q) {
            const u32 c = s_code[st[q]];
            const u32 t = (u32)(0xFFFFu - q) & 0xFFFFu; // (i16)(-1 - q)
            u32 wd = my[(c >> 1) * CL];
            wd = (c & 1u) ? ((wd & 0x0000FFFFu) | (t << 16)) : ((wd & 0xFFFF0000u) | t);
            my[(c >> 1) * CL] = wd;
        }
    }
    if (SUB > 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    const u8 *L = a.L + (size_t)lb * kSlot;
    u8 *R8 = a.rank8 + (size_t)lb * kSlot;
    // A lane's chunk is 512 consecutive bytes: it is read 64 bytes at a time (four 16-byte loads issued
    // together, the ranks stored the same way), so a line is fetched once and used while it is there;
    // 16 bytes per visit meant eight visits per 128-byte line with 64 other lanes' lines in between
    // (9x the column's bytes in HBM traffic).
    for (u32 v4 = 0; v4 < kMtfChunk / 64u; ++v4) {
        if (beg + v4 * 64u >= end) break;
        uint4 q4[4];
#pragma unroll
        for (u32 u = 0; u < 4; ++u) {
            const u32 pl = beg + v4 * 64u + u * 16u;
            q4[u] = (pl < end) ? *reinterpret_cast<const uint4 *>(L + pl) : make_uint4(0, 0, 0, 0); // (slots are padded)
        }
#pragma unroll
      for (u32 u = 0; u < 4; ++u) {
        const u32 v = v4 * 4u + u;
        const u32 p0 = beg + v * 16u;
        if (p0 >= end) break;
        const u32 wv[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
        u32 ov[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            u32 rank = 0;
            if (p0 + k < end) {
                const u32 c = s_code[(wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
                const u32 wc = my[(c >> 1) * CL];
                const u32 lsu = (c & 1u) ? (wc >> 16) : (wc & 0xFFFFu);
                // both halves of a table word against the symbol's own time in packed 16-bit arithmetic:
                // ls - t saturates (a never-seen -32768 must not wrap), its sign says t > ls
                const short2_t ls2 = as_short2(lsu | (lsu << 16));
                short2_t acc = {0, 0};
              if (SUB == 1 && PAIRS > kMtfSmallPairs) { // (<= 96 symbols: three workgroups per CU hide the wait already; measured slower with it: 2.02 -> 2.27 ms per 256 MiB of text)
                // Round 5: the table rows of the NEXT eight pairs are on their way while the eight at hand are compared -- the
                // compiler's loop asked for eight rows, waited for them and compared them, one wave per SIMD with nothing to hide
                // the wait behind (64 KB of LDS per 128 lanes): 7.3 ms per 256 MiB of random bytes, 200 cycles per eight pairs of
                // which 96 were arithmetic.
                u32 ra[kPB], rb[kPB];
#pragma unroll
                for (u32 e = 0; e < kPB; ++e) ra[e] = my[e * CL];
                for (u32 q = 0; q < npairs16; q += 2u * kPB) {
#pragma unroll
                    for (u32 e = 0; e < kPB; ++e) rb[e] = my[(q + kPB + e) * CL];
#pragma unroll
                    for (u32 e = 0; e < kPB; ++e) acc -= __builtin_elementwise_sub_sat(ls2, as_short2(ra[e])) >> (short2_t){15, 15};
                    // (the last trip asks for the table's first rows again: never used)
                    const u32 qn = (q + 2u * kPB < npairs16) ? q + 2u * kPB : 0u;
#pragma unroll
                    for (u32 e = 0; e < kPB; ++e) ra[e] = my[(qn + e) * CL];
#pragma unroll
                    for (u32 e = 0; e < kPB; ++e) acc -= __builtin_elementwise_sub_sat(ls2, as_short2(rb[e])) >> (short2_t){15, 15};
                }
              } else if (SUB > 1) { // this lane's PAIRS / SUB rows, all asked for at once (a loop that waits for each row is 820 us
                // for sample1's block, the one-lane form 600)
                constexpr u32 PER = PAIRS / SUB;
                u32 ra[PER];
#pragma unroll
                for (u32 e = 0; e < PER; ++e) ra[e] = my[(sub * PER + e) * CL];
#pragma unroll
                for (u32 e = 0; e < PER; ++e) acc -= __builtin_elementwise_sub_sat(ls2, as_short2(ra[e])) >> (short2_t){15, 15};
              } else
                for (u32 q = 0; q < npairs; ++q) {
                    const short2_t wd = as_short2(my[q * CL]);
                    const short2_t neg = __builtin_elementwise_sub_sat(ls2, wd) >> (short2_t){15, 15}; // -1 where t > ls
                    acc -= neg;
                }
                rank = (u32)(int)acc.x + (u32)(int)acc.y;
                if (SUB > 1) { // the chunk's lanes add up what each counted
#pragma unroll
                    for (u32 d = 1; d < SUB; d <<= 1) rank += (u32)__shfl_xor((int)rank, (int)d, 64);
                }
                const u32 t = (v * 16u + k) & 0xFFFFu; // time inside the chunk, 0..kMtfChunk-1
                if (sub == 0) my[(c >> 1) * CL] = (c & 1u) ? ((wc & 0x0000FFFFu) | (t << 16)) : ((wc & 0xFFFF0000u) | t);
            }
            ov[k >> 2] |= rank << ((k & 3) * 8);
        }
        if (sub == 0) *reinterpret_cast<uint4 *>(R8 + p0) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
      }
    }
}

// ---- ZLE helpers ----------------------------------------------------------------------------
struct ZSeg {
    u8 r[16];
    u32 valid;
    u32 p0;
    int next_nonzero; // 1 if the position after the segment holds a non-zero rank or is past the end
};

__device__ __forceinline__ void zload(const u8 *__restrict__ R8, u32 n, u32 tile, ZSeg &s)
{
    s.p0 = tile * kSortTile + threadIdx.x * 16u;
    s.valid = s.p0 < n ? ((n - s.p0) < 16u ? (n - s.p0) : 16u) : 0u;
    if (s.valid) {
        const uint4 q = *reinterpret_cast<const uint4 *>(R8 + s.p0);
        const u32 wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) s.r[k] = (u8)(wv[k >> 2] >> ((k & 3) * 8));
        const u32 nx = s.p0 + s.valid;
        s.next_nonzero = (nx >= n) ? 1 : (R8[nx] != 0);
    } else {
        s.next_nonzero = 1;
    }
}

__device__ __forceinline__ int block_excl_max_int(int v, int *sh /*[16]*/)
{
    const u32 l = lane_id(), w = threadIdx.x >> 6;
    const int inc = wave_incl_max32(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    int carry = -1;
    for (u32 k = 0; k < w; ++k) carry = sh[k] > carry ? sh[k] : carry;
    const int prev = __shfl_up(inc, 1, 64);
    const int ex = (l == 0) ? -1 : prev;
    __syncthreads();
    return ex > carry ? ex : carry;
}

__device__ __forceinline__ u32 block_excl_sum1024(u32 v, u32 *sh /*[16]*/, u32 &total)
{
    const u32 l = lane_id(), w = threadIdx.x >> 6;
    const u32 inc = wave_incl_sum(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    u32 carry = 0, tot = 0;
    for (u32 k = 0; k < kSortThreads / 64; ++k) {
        if (k < w) carry += sh[k];
        tot += sh[k];
    }
    total = tot;
    __syncthreads();
    return carry + inc - v;
}

// Z1: last non-zero rank position per tile
__global__ __launch_bounds__(kSortThreads) void k_zle_last(MtfArgs a)
{
    __shared__ int s_last;
    const u32 lb = blockIdx.y, tile = blockIdx.x;
    const u32 n = a.blocks[lb].n;
    if (tile * kSortTile >= n) return;
    if (threadIdx.x == 0) s_last = -1;
    __syncthreads();
    ZSeg s;
    zload(a.rank8 + (size_t)lb * kSlot, n, tile, s);
    int last = -1;
#pragma unroll
    for (u32 k = 0; k < 16; ++k)
        if (k < s.valid && s.r[k]) last = (int)(s.p0 + k);
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) {
        const int o = __shfl_xor(last, dd, 64);
        last = o > last ? o : last;
    }
    if (lane_id() == 0 && last >= 0) atomicMax(&s_last, last);
    __syncthreads();
    if (threadIdx.x == 0) a.ztile_last[lb * kTilesPerBlock + tile] = s_last;
}

// number of symbols a zero run of length z becomes: digits of z in bijective base 2
__device__ __forceinline__ u32 run_digits(u32 z) { return 31u - (u32)__clz(z + 1u); }

// Z2 (WRITE=false): count output symbols per tile.  Z3 (WRITE=true): write them.
template <bool WRITE>
__global__ __launch_bounds__(kSortThreads) void k_zle_emit(MtfArgs a)
{
    __shared__ int s_mi[16];
    __shared__ u32 s_su[16];
    __shared__ u32 s_freq[kMaxAlpha + 2];
    __shared__ int s_carry;
    __shared__ u32 s_base;
    // the tile's symbols, staged so that they leave as whole rows (a tile emits at most one symbol per position, plus
    // the digits of a zero run that began in an earlier tile).  Measured: no gain and no loss -- the kernel's 2.3 ms
    // per GiB are the serial walk of a thread over its 16 positions, not its stores, the counters or the offsets.
    __shared__ u16 s_out[WRITE ? kSortTile + 64 : 1];
    const u32 lb = blockIdx.y, tile = blockIdx.x;
    const u32 n = a.blocks[lb].n;
    if (tile * kSortTile >= n) return;
    if (threadIdx.x == 0) {
        int c = -1;
        for (int t = (int)tile - 1; t >= 0 && c < 0; --t) c = a.ztile_last[lb * kTilesPerBlock + t];
        s_carry = c;
        if (WRITE) {
            u32 b = 0;
            for (u32 t = 0; t < tile; ++t) b += a.ztile_cnt[lb * kTilesPerBlock + t];
            s_base = b;
        }
    }
    if (WRITE)
        for (u32 i = threadIdx.x; i < kMaxAlpha + 2; i += kSortThreads) s_freq[i] = 0;
    __syncthreads();
    ZSeg s;
    zload(a.rank8 + (size_t)lb * kSlot, n, tile, s);
    int last = -1;
#pragma unroll
    for (u32 k = 0; k < 16; ++k)
        if (k < s.valid && s.r[k]) last = (int)(s.p0 + k);
    int lnz = block_excl_max_int(last, s_mi); // last non-zero position before this segment
    lnz = lnz > s_carry ? lnz : s_carry;
    // pass 1: count
    u32 cnt = 0;
    {
        int ln = lnz;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (k < s.valid) {
                const u32 p = s.p0 + k;
                if (s.r[k]) {
                    ln = (int)p;
                    cnt += 1;
                } else {
                    const bool run_end = (k + 1 < s.valid) ? (s.r[(k + 1) & 15] != 0) : (s.next_nonzero != 0);
                    if (run_end) cnt += run_digits((u32)((int)p - ln));
                }
            }
        }
    }
    u32 total;
    u32 off = block_excl_sum1024(cnt, s_su, total);
    if (!WRITE) {
        if (threadIdx.x == 0) a.ztile_cnt[lb * kTilesPerBlock + tile] = total;
        return;
    }
    u16 *out = s_out + off;
    // RUNA, RUNB and the two smallest ranks are most of the symbols: a thread counts them in
    // registers and adds once (single adds to four hot LDS words serialise the workgroup)
    u32 hot0 = 0, hot1 = 0, hot2 = 0, hot3 = 0;
    {
        int ln = lnz;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (k < s.valid) {
                const u32 p = s.p0 + k;
                if (s.r[k]) {
                    ln = (int)p;
                    const u32 sym = (u32)s.r[k] + 1u; // encoder.rs:340,349
                    *out++ = (u16)sym;
                    if (sym == 2u) ++hot2;
                    else if (sym == 3u) ++hot3;
                    else atomicAdd(&s_freq[sym], 1u);
                } else {
                    const bool run_end = (k + 1 < s.valid) ? (s.r[(k + 1) & 15] != 0) : (s.next_nonzero != 0);
                    if (run_end) {
                        u32 zc = (u32)((int)p - ln) + 1u; // encoder.rs:659-667
                        while (zc > 1u) {
                            const u32 run = zc & 1u;
                            *out++ = (u16)run;
                            hot0 += run ^ 1u;
                            hot1 += run;
                            zc >>= 1;
                        }
                    }
                }
            }
        }
    }
    hot0 = wave_sum(hot0);
    hot1 = wave_sum(hot1);
    hot2 = wave_sum(hot2);
    hot3 = wave_sum(hot3);
    if ((threadIdx.x & 63u) == 0) {
        if (hot0) atomicAdd(&s_freq[0], hot0);
        if (hot1) atomicAdd(&s_freq[1], hot1);
        if (hot2) atomicAdd(&s_freq[2], hot2);
        if (hot3) atomicAdd(&s_freq[3], hot3);
    }
    __syncthreads();
    {
        u16 *dst = a.mtf + (size_t)lb * kMtfStride + s_base;
        for (u32 i = threadIdx.x; i < total; i += kSortThreads) dst[i] = s_out[i];
    }
    const bool last_tile = (tile + 1) * kSortTile >= n;
    if (last_tile && threadIdx.x == 0) {
        const u32 alpha_in = popc8(a.inuse_bits + lb * 8);
        const u32 eob = alpha_in + 1u; // encoder.rs:316
        a.mtf[(size_t)lb * kMtfStride + s_base + total] = (u16)eob;
        atomicAdd(&s_freq[eob], 1u);
        a.out[lb].mtf_count = s_base + total + 1u;
        a.out[lb].in_use_count = alpha_in;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < kMaxAlpha; i += kSortThreads)
        if (s_freq[i]) atomicAdd(&a.mtf_freq[(size_t)lb * kMaxAlpha + i], s_freq[i]);
}

// Z1 + Z2 + Z3 in ONE launch.  What a tile needs from the tiles in front of it -- the last position with a non-zero
// rank (a zero run is written by the tile it ends in, with the digits of its whole length) and the number of symbols
// they write -- comes by decoupled look-back, as in the fused radix passes: tiles take tickets from the counter of
// the XCD they run on (all tiles of a block on one XCD, the words travel through that XCD's L2), a tile publishes what
// it holds itself (AGG), looks back until its predecessors' sums are settled, publishes the settled value (INCL).
// Two words of one 16-byte granule per tile: x = last non-zero position (0xFFFFF: none), y = symbols written; both
// cleared before the launch (flag 0 = not there yet).  The look-back for the symbol count runs after the tile has
// filled its staging buffer, so only the copy-out waits for it.  ztick[0..7]: tickets, ztick[8]: a look-back gave up
// -- the host checks both after the batch and redoes the stage with the three kernels if they are not as expected.
__global__ __launch_bounds__(kSortThreads) void k_zle_fused(MtfArgs a)
{
    __shared__ int s_mi[16];
    __shared__ u32 s_su[16];
    __shared__ u32 s_freq[kMaxAlpha + 2];
    __shared__ int s_carry;
    __shared__ u32 s_base, s_ticket;
    __shared__ u16 s_out[kSortTile + 64];
    const u32 xcd = (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; // XCC_ID
    const u32 my_tiles = a.tiles * ((a.nb + 7u - xcd) / 8u); // (a.tiles: tiles per block the launch covers)
    if (threadIdx.x == 0) s_ticket = atomicAdd(&a.ztick[xcd], 1u);
    __syncthreads();
    const u32 slot = s_ticket;
    if (slot >= my_tiles) return;
    const u32 b8 = slot / a.tiles;
    const u32 tile = slot - b8 * a.tiles;
    const u32 lb = b8 * 8u + xcd;
    const u32 n = a.blocks[lb].n;
    if (tile * kSortTile >= n) return; // (tiles beyond the block publish nothing; nobody looks back at them)
    for (u32 i = threadIdx.x; i < kMaxAlpha + 2; i += kSortThreads) s_freq[i] = 0;
    ZSeg s;
    zload(a.rank8 + (size_t)lb * kSlot, n, tile, s);
    int last = -1;
#pragma unroll
    for (u32 k = 0; k < 16; ++k)
        if (k < s.valid && s.r[k]) last = (int)(s.p0 + k);
    int lnz = block_excl_max_int(last, s_mi); // last non-zero position before this segment, inside the tile
    u32 *mystate = a.zstate + ((size_t)lb * kTilesPerBlock + tile) * 4u;
    u32 w_last = 0; // (thread 0) this tile's settled x word
    if (threadIdx.x == 0) {
        int tl = -1; // the tile's own last non-zero position (s_mi holds the waves' inclusive maxima)
        for (u32 k = 0; k < kSortThreads / 64; ++k) tl = s_mi[k] > tl ? s_mi[k] : tl;
        const u32 vl = tl >= 0 ? (u32)tl : kLbValMask;
        st_sc1_x4(mystate, make_uint4((tile ? kLbAgg : kLbIncl) | vl, 0u, 0u, 0u));
        int c = -1;
        if (tile) {
            u32 spins = 0;
            for (u32 p = tile; p > 0 && c < 0;) {
                --p;
                const u32 *src = a.zstate + ((size_t)lb * kTilesPerBlock + p) * 4u;
                uint4 v = ld_sc1_x4(src);
                while ((v.x & kLbFlagMask) == 0u) {
                    if (lb_give_up(spins, a.ztick + 8, kLbSpinMax >> 3)) {
                        atomicExch(a.ztick + 8, 1u);
                        v = make_uint4(kLbIncl, 0u, 0u, 0u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    v = ld_sc1_x4(src);
                }
                const u32 x = v.x & kLbValMask;
                if (x != kLbValMask) c = (int)x;
                else if ((v.x & kLbFlagMask) == kLbIncl) break; // nothing non-zero in front at all
            }
        }
        w_last = kLbIncl | (tl >= 0 ? (u32)tl : (c >= 0 ? (u32)c : kLbValMask));
        if (tile) st_sc1_x4(mystate, make_uint4(w_last, 0u, 0u, 0u));
        s_carry = c;
    }
    __syncthreads();
    lnz = lnz > s_carry ? lnz : s_carry;
    // pass 1: count
    u32 cnt = 0;
    {
        int ln = lnz;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (k < s.valid) {
                const u32 p = s.p0 + k;
                if (s.r[k]) {
                    ln = (int)p;
                    cnt += 1;
                } else {
                    const bool run_end = (k + 1 < s.valid) ? (s.r[(k + 1) & 15] != 0) : (s.next_nonzero != 0);
                    if (run_end) cnt += run_digits((u32)((int)p - ln));
                }
            }
        }
    }
    u32 total;
    const u32 off = block_excl_sum1024(cnt, s_su, total);
    if (threadIdx.x == 0) st_sc1_x4(mystate, make_uint4(w_last, (tile ? kLbAgg : kLbIncl) | total, 0u, 0u));
    // pass 2: the symbols into the staging buffer (as k_zle_emit<true>)
    u16 *out = s_out + off;
    u32 hot0 = 0, hot1 = 0, hot2 = 0, hot3 = 0;
    {
        int ln = lnz;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (k < s.valid) {
                const u32 p = s.p0 + k;
                if (s.r[k]) {
                    ln = (int)p;
                    const u32 sym = (u32)s.r[k] + 1u; // encoder.rs:340,349
                    *out++ = (u16)sym;
                    if (sym == 2u) ++hot2;
                    else if (sym == 3u) ++hot3;
                    else atomicAdd(&s_freq[sym], 1u);
                } else {
                    const bool run_end = (k + 1 < s.valid) ? (s.r[(k + 1) & 15] != 0) : (s.next_nonzero != 0);
                    if (run_end) {
                        u32 zc = (u32)((int)p - ln) + 1u; // encoder.rs:659-667
                        while (zc > 1u) {
                            const u32 run = zc & 1u;
                            *out++ = (u16)run;
                            hot0 += run ^ 1u;
                            hot1 += run;
                            zc >>= 1;
                        }
                    }
                }
            }
        }
    }
    hot0 = wave_sum(hot0);
    hot1 = wave_sum(hot1);
    hot2 = wave_sum(hot2);
    hot3 = wave_sum(hot3);
    if ((threadIdx.x & 63u) == 0) {
        if (hot0) atomicAdd(&s_freq[0], hot0);
        if (hot1) atomicAdd(&s_freq[1], hot1);
        if (hot2) atomicAdd(&s_freq[2], hot2);
        if (hot3) atomicAdd(&s_freq[3], hot3);
    }
    // where the tile's symbols go: the symbols of the tiles in front
    if (threadIdx.x == 0) {
        u32 base = 0;
        if (tile) {
            u32 spins = 0;
            for (u32 p = tile; p > 0;) {
                --p;
                const u32 *src = a.zstate + ((size_t)lb * kTilesPerBlock + p) * 4u;
                uint4 v = ld_sc1_x4(src);
                while ((v.y & kLbFlagMask) == 0u) {
                    if (lb_give_up(spins, a.ztick + 8, kLbSpinMax >> 3)) {
                        atomicExch(a.ztick + 8, 1u);
                        v = make_uint4(0u, kLbIncl, 0u, 0u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    v = ld_sc1_x4(src);
                }
                base += v.y & kLbValMask;
                if ((v.y & kLbFlagMask) == kLbIncl) break;
            }
            st_sc1_x4(mystate, make_uint4(w_last, kLbIncl | ((base + total) & kLbValMask), 0u, 0u));
        }
        s_base = base;
    }
    __syncthreads();
    {
        u16 *dst = a.mtf + (size_t)lb * kMtfStride + s_base;
        for (u32 i = threadIdx.x; i < total; i += kSortThreads) dst[i] = s_out[i];
    }
    const bool last_tile = (tile + 1) * kSortTile >= n;
    if (last_tile && threadIdx.x == 0) {
        const u32 alpha_in = popc8(a.inuse_bits + lb * 8);
        const u32 eob = alpha_in + 1u; // encoder.rs:316
        a.mtf[(size_t)lb * kMtfStride + s_base + total] = (u16)eob;
        atomicAdd(&s_freq[eob], 1u);
        a.out[lb].mtf_count = s_base + total + 1u;
        a.out[lb].in_use_count = alpha_in;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < kMaxAlpha; i += kSortThreads)
        if (s_freq[i]) atomicAdd(&a.mtf_freq[(size_t)lb * kMaxAlpha + i], s_freq[i]);
}

void launch_mtf(hipStream_t st, const MtfArgs &a)
{
    (void)hipMemsetAsync(a.mtf_freq, 0, (size_t)a.nb * kMaxAlpha * sizeof(u32), st);
    hipLaunchKernelGGL(k_mtf_summaries, dim3(kChunkWGs, a.nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_mtf_merge_groups, dim3(kMtfGroups, a.nb), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_mtf_group_starts, dim3(a.nb), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_mtf_chunk_starts, dim3(kMtfGroups, a.nb), dim3(256), 0, st, a);
    // ranks: the compare form for every alphabet (<= 96 symbols: 256 chunks per workgroup; more: 128 -- the list walk of
    // rounds 1-3 took 21.0 ms per 256 MiB of random bytes against 8.9: DESIGN.md section 4)
    if (a.nb <= kMtfSubBlocks) { // a few blocks alone: eight lanes per chunk (one 98 KB block of 256 symbols: 600 -> 224 us; SUB = 4: 263)
        hipLaunchKernelGGL((k_mtf_ranks_small<kMtfSmallPairs, 256, 0, 8>), dim3((kMaxMtfChunks + 31) / 32, a.nb), dim3(256), 0, st, a);
        hipLaunchKernelGGL((k_mtf_ranks_small<128, 128, kMtfSmallAlpha, 8>), dim3((kMaxMtfChunks + 15) / 16, a.nb), dim3(128), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_mtf_ranks_small<kMtfSmallPairs, 256, 0>), dim3(kChunkWGs, a.nb), dim3(256), 0, st, a);
        hipLaunchKernelGGL((k_mtf_ranks_small<128, 128, kMtfSmallAlpha>), dim3((kMaxMtfChunks + 127) / 128, a.nb), dim3(128), 0, st, a);
    }
    if (a.fused_zle) {
        (void)hipMemsetAsync(a.zstate, 0, (size_t)a.nb * kTilesPerBlock * 16, st);
        (void)hipMemsetAsync(a.ztick, 0, 64, st);
        hipLaunchKernelGGL(k_zle_fused, dim3(a.tiles, xcd_grid_y(a.nb)), dim3(kSortThreads), 0, st, a);
        return;
    }
    const dim3 grid(a.tiles, a.nb);
    hipLaunchKernelGGL(k_zle_last, grid, dim3(kSortThreads), 0, st, a);
    hipLaunchKernelGGL((k_zle_emit<false>), grid, dim3(kSortThreads), 0, st, a);
    hipLaunchKernelGGL((k_zle_emit<true>), grid, dim3(kSortThreads), 0, st, a);
}

} // namespace bzgpu
