// k_huff.hip -- Huffman stage of one block per workgroup: table count, initial tables,
// four refinement passes, code lengths, canonical codes, selector MTF and the whole
// block header; plus the payload emission kernel.
//
// Reference being replaced (src/bzip2/encoder.rs): table count :370-376, initial tables
// :379-426, refinement :433-509, create_huffman :641-651, selector MTF :511-517, codes
// :519-524, mapping table :527-554, selectors :567-574, coding tables :583-601, payload
// :609-629; src/huffman/cano_huff_table.rs:14-225 (code lengths; the serial heap procedure
// and the "reverse package merge" fallback are replayed step for step by ONE lane per
// table -- SURVEY.md F3/F5: bit-exactness hinges on the heap's tie-breaks);
// src/huffman/mod.rs:22-67 (canonical codes).
//
// Table order: the reference stores `len` reversed and always walks it with .rev()
// (encoder.rs:452-454, 504-508, 520-524, 585); here tables are kept in selector order
// t = 0..group_num-1, which is the order the stream carries them in.
#include "bzgpu.h"
#include <cstdlib>

namespace bzgpu {

constexpr u32 kHuffThreads = 512;

// encoder.rs:647-650
__device__ __forceinline__ u32 weight_add(u32 x, u32 y)
{
    const u32 dx = x & 0xFFu, dy = y & 0xFFu;
    return ((x & 0xFFFFFF00u) + (y & 0xFFFFFF00u)) | (1u + (dx > dy ? dx : dy));
}

// cano_huff_table.rs:14-31
__device__ void down_heap(u32 *buf, u32 nn, u32 len)
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

// cano_huff_table.rs:58-151 ("reverse package merge"), scratch in global memory.
// freq = the weights (all non-zero), n symbols, out = lengths in symbol order.
// `row` = words per package-merge row (>= 2n+4); scratch needs 4*kMaxAlpha + 64 + 2*lim*row words.
__device__ void gen_code_lm(const u32 *freq, u32 n, u32 *scr, u32 row, u8 *out)
{
    const u32 lim = kLim;
    u32 *map = scr;                 // [n]
    u32 *sfreq = map + kMaxAlpha;   // [n]
    u32 *c = sfreq + kMaxAlpha;     // [n]
    u32 *misc = c + kMaxAlpha;      // max_elem[17], b[17], cur[17]
    u32 *max_elem = misc, *b = misc + 20, *cur = misc + 40;
    u32 *val = misc + 64 + kMaxAlpha; // [lim][row]
    u32 *ty = val + lim * row;        // [lim][row]

    // stable sort by weight, descending (:64-70): insertion sort keeps equal keys in order
    for (u32 i = 0; i < n; ++i) {
        const u32 f = freq[i];
        u32 p = i;
        while (p > 0 && sfreq[p - 1] < f) {
            sfreq[p] = sfreq[p - 1];
            map[p] = map[p - 1];
            --p;
        }
        sfreq[p] = f;
        map[p] = i;
    }
    for (u32 j = 0; j < lim; ++j) { max_elem[j] = 0; b[j] = 0; cur[j] = 0; }
    u32 excess = (1u << lim) - n;       // :75
    const u32 half = 1u << (lim - 1);   // :76
    max_elem[lim - 1] = n;              // :77
    for (u32 j = 0; j < lim; ++j) {     // :79-88
        if (excess >= half) {
            b[j] = 1;
            excess -= half;
        }
        excess <<= 1;
        if (lim >= 2 + j) max_elem[lim - 2 - j] = max_elem[lim - 1 - j] / 2 + n;
    }
    max_elem[0] = b[0];                 // :90-95
    for (u32 j = 1; j < lim; ++j)
        if (max_elem[j] > 2 * max_elem[j - 1] + b[j]) max_elem[j] = 2 * max_elem[j - 1] + b[j];

    for (u32 j = 0; j < lim; ++j)       // :97-98 (zero initialised vectors)
        for (u32 t = 0; t < max_elem[j]; ++t) { val[j * row + t] = 0; ty[j * row + t] = 0; }
    for (u32 i = 0; i < n; ++i) c[i] = lim; // :99
    for (u32 t = 0; t < n && t < max_elem[lim - 1]; ++t) { // :101-104
        val[(lim - 1) * row + t] = sfreq[t];
        ty[(lim - 1) * row + t] = t;
    }
    if (b[lim - 1] == 1) {              // :107-110
        c[0] -= 1;
        cur[lim - 1] += 1;
    }
    u32 j = lim - 1;
    while (j > 0) {                     // :112-142
        u32 i = 0;
        u32 next = cur[j];
        for (u32 t = 0; t < max_elem[j - 1]; ++t) {
            const u32 weight = (next + 1 < max_elem[j])
                                   ? weight_add(val[j * row + next], val[j * row + next + 1])
                                   : 0u;
            if (weight > sfreq[i]) {
                val[(j - 1) * row + t] = weight;
                ty[(j - 1) * row + t] = n;
                next += 2;
            } else {
                val[(j - 1) * row + t] = sfreq[i];
                ty[(j - 1) * row + t] = i;
                i += 1;
                if (i >= n) break;
            }
        }
        j -= 1;
        cur[j] = 0;
        if (b[j] == 1) {
            // take_package(ty, c, cur, j) (:40-55) with an explicit stack -- kept in the scratch area (the words
            // between misc and val are free): a local array indexed at run time would live in scratch memory, i.e.
            // behind a global-memory round trip per access
            u32 *lvl = misc + 64, *ph = misc + 84;
            int sp = 0;
            lvl[0] = j;
            ph[0] = 0;
            while (sp >= 0) {
                const u32 li = lvl[sp];
                if (ph[sp] == 0) {
                    const u32 x = ty[li * row + cur[li]];
                    if (x == n) {
                        ph[sp] = 1;
                        ++sp;
                        lvl[sp] = li + 1;
                        ph[sp] = 0;
                    } else {
                        c[x] -= 1;
                        cur[li] += 1;
                        --sp;
                    }
                } else if (ph[sp] == 1) {
                    ph[sp] = 2;
                    ++sp;
                    lvl[sp] = li + 1;
                    ph[sp] = 0;
                } else {
                    cur[li] += 1;
                    --sp;
                }
            }
        }
    }
    for (u32 i = 0; i < n; ++i) out[map[i]] = (u8)c[i]; // :144-150
}

// cano_huff_table.rs:153-196 gen_code on the bzip2 weights (encoder.rs:641-651).
// rfreq: symbol counts of this table; buf: 2*alpha words of LDS; returns 1 if the
// length-limited path was taken.
__device__ int heap_code_lengths(const u32 *rfreq, u32 alpha, u32 *buf, u8 *out)
{
    const u32 n = alpha;
    if (n == 1) { // cannot happen on this path (alpha >= 3), kept for the probe entry
        out[0] = 1;
        return 0;
    }
    for (u32 i = 0; i < n; ++i) {
        buf[i] = n + i;
        const u32 f = rfreq[i];
        buf[n + i] = (f > 1u ? f : 1u) << 8; // encoder.rs:642-645
    }
    // create_heap, :33-38
    for (u32 i = n >> 1; i-- > 0;) down_heap(buf, i, n);
    for (u32 i = n - 1; i >= 1; --i) { // :168-178
        const u32 m1 = buf[0];
        buf[0] = buf[i];
        down_heap(buf, 0, i);
        const u32 m2 = buf[0];
        buf[i] = weight_add(buf[m1], buf[m2]);
        buf[0] = i;
        buf[m1] = i;
        buf[m2] = i;
        down_heap(buf, 0, i);
    }
    buf[1] = 0; // :181-184
    for (u32 i = 2; i < n; ++i) buf[i] = buf[buf[i]] + 1;
    int too_long = 0;
    for (u32 i = 0; i < n; ++i) { // :186-188
        const u32 l = buf[buf[i + n]] + 1;
        out[i] = (u8)l;
        if (l > kLim) too_long = 1;
    }
    return too_long; // 1: the caller must redo the table with gen_code_lm (:190-194)
}

// The same procedure with the WEIGHT kept beside the node in every heap entry (round 4).  A node's weight is fixed when
// the node is made and never changes while it is in the heap, so an entry can be the 64-bit word weight << 32 | node: a
// sift step then compares its two children after ONE LDS round trip (two adjacent 8-byte reads) instead of two (the
// children's node numbers, then their weights) -- and the procedure is nothing but dependent round trips.  The
// comparisons are the reference's, on the same values, in the same order (cano_huff_table.rs:14-38, :153-196): `>`
// between the children, `<` against the sifted entry.  What the reference keeps in the same array besides the heap --
// the parent of every node, later its depth -- lives in a u16 array (`par`); node weights need no home of their own (an
// extracted entry carries its).  arena: 3 n words, 8-byte aligned.  Same lengths as heap_code_lengths (the probe
// kernel runs both on every probed table).
__device__ __forceinline__ void down_heap_w(u64 *H, u32 nn, u32 len)
{
    const u64 tmp = H[nn];
    const u32 wt = (u32)(tmp >> 32);
    u32 leaf = (nn << 1) + 1;
    while (leaf < len) {
        u64 c = H[leaf];
        if (leaf + 1 < len) {
            const u64 c2 = H[leaf + 1];
            if ((u32)(c >> 32) > (u32)(c2 >> 32)) {
                leaf += 1;
                c = c2;
            }
        }
        if (wt < (u32)(c >> 32)) break;
        H[nn] = c;
        nn = leaf;
        leaf = (nn << 1) + 1;
    }
    H[nn] = tmp;
}
__device__ int heap_code_lengths_w(const u32 *rfreq, u32 alpha, u32 *arena, u8 *out)
{
    const u32 n = alpha;
    if (n == 1) {
        out[0] = 1;
        return 0;
    }
    u64 *H = reinterpret_cast<u64 *>(arena);          // [n]
    u16 *par = reinterpret_cast<u16 *>(arena + 2 * n); // [2 n]
    for (u32 i = 0; i < n; ++i) {
        const u32 f = rfreq[i];
        H[i] = ((u64)((f > 1u ? f : 1u) << 8) << 32) | (u64)(n + i); // encoder.rs:642-645
    }
    for (u32 i = n >> 1; i-- > 0;) down_heap_w(H, i, n); // create_heap, :33-38
    for (u32 i = n - 1; i >= 1; --i) { // :168-178
        const u64 m1 = H[0];
        H[0] = H[i];
        down_heap_w(H, 0, i);
        const u64 m2 = H[0];
        H[0] = ((u64)weight_add((u32)(m1 >> 32), (u32)(m2 >> 32)) << 32) | (u64)i;
        par[(u32)m1] = (u16)i;
        par[(u32)m2] = (u16)i;
        down_heap_w(H, 0, i);
    }
    par[1] = 0; // :181-184
    for (u32 i = 2; i < n; ++i) par[i] = (u16)(par[par[i]] + 1u);
    int too_long = 0;
    for (u32 i = 0; i < n; ++i) { // :186-188
        const u32 l = (u32)par[par[i + n]] + 1u;
        out[i] = (u8)l;
        if (l > kLim) too_long = 1;
    }
    return too_long;
}

// The same procedure PIPELINED over eight lanes (round 6).  The main loop is 2 (n - 1) sift-downs from the root, each
// up to log2(n) levels long, and each needs the root its predecessor left -- but only the root and the root's
// children: a sift-down that has made its step at level t never touches levels <= t again.  So the next sift-down
// starts two steps behind the one in front of it (its first step reads level 1, which the one in front finished one
// step earlier) and follows it down the tree two levels apart: what a step reads (the children, one level below the
// sift's place) and what the sifts in front of it write (their own places, two and more levels below) never meet.  A
// lane per sift-down in flight, every lane one level per step; a new sift-down every second step instead of every
// log2(n)-th.  The one other dependence: a "pop" takes the LAST entry of the heap for the root
// (cano_huff_table.rs:169-171), and a sift-down in flight whose place is an ancestor of that entry (or the entry) may
// yet move it: the pop waits a step (one start in five on a 258-symbol table).  create_heap (:33-38) runs level by
// level: the sift-downs of one level work in disjoint subtrees.  Same comparisons on the same values as
// heap_code_lengths_w, same array after every sift-down (the probe kernel runs all forms on every probed table).
//
// What a step costs is its INSTRUCTIONS, not its LDS round trip (a lone wave issues a vector instruction every four to
// eight cycles: the first form, eight lanes per table and six tables in one wave, took 1 000 cycles a step for 120
// instructions, as long as the one-lane form's three levels).  So: one WAVE per table -- what is the same for the
// eight lanes (the loop's counters, the two smallest, whether to start a sift-down) is then the same for the wave
// and lives in scalar registers behind scalar branches --, and the step's vector part is the level of a sift-down.
// All 64 lanes of a wave call it with the same arguments (`on`: the wave has a table); lanes 0-7 carry the sift-downs.
constexpr u32 kPipeLanes = 8;
__device__ __forceinline__ void heap_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ int heap_code_lengths_pipe(const u32 *rfreq, u32 n_, u32 *arena, u8 *out, u32 lane, bool on_)
{
    // (the same in the 64 lanes, and the compiler is to know: the loop's counters then live in scalar registers)
    const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)n_);
    const bool on = __builtin_amdgcn_readfirstlane(on_ ? 1 : 0) != 0;
    if (n == 1) {
        if (on && lane == 0) out[0] = 1;
        return 0;
    }
    u64 *H = reinterpret_cast<u64 *>(arena);           // [n]
    u16 *par = reinterpret_cast<u16 *>(arena + 2 * n); // [2 n]
#ifdef BZ_TAB_TIMING
    const unsigned long long tt0 = wall_clock64(), tc0 = clock64();
    u32 nsteps = 0;
#endif
    if (on)
        for (u32 i = lane; i < n; i += 64u) {
            const u32 f = rfreq[i];
            H[i] = ((u64)((f > 1u ? f : 1u) << 8) << 32) | (u64)(n + i); // encoder.rs:642-645
        }
    heap_wave_sync();
    const u32 half = n >> 1; // create_heap, :33-38: the nodes half - 1 .. 0, deepest level first
    for (int L = 31 - (int)__clz(half); L >= 0; --L) {
        const u32 first = (1u << L) - 1u;
        u32 last = (2u << L) - 2u;
        if (last > half - 1u) last = half - 1u;
        if (on)
            for (u32 i = first + lane; i <= last; i += 64u) down_heap_w(H, i, n);
        heap_wave_sync();
    }
    // A step = ONE LDS round trip: the children of this lane's sift-down, the root and the last entry are read side by
    // side (the root was written a step ago at the latest; the last entry is used only if no sift-down that was in
    // flight when the step began can reach it), then the step's write.  The lanes of a wave run in lockstep and the
    // LDS serves a wave's instructions in order: the scheduling barrier keeps the compiler from moving a step's reads
    // above the write of the step before.  An entry is the pair (node, weight): the u64 weight << 32 | node of
    // down_heap_w, read as two words.
    const uint2 *Hr = reinterpret_cast<const uint2 *>(arena);
    uint2 *Hw = reinterpret_cast<uint2 *>(arena);
    bool active = false;                              // per lane: the sift-down, its place and level, the sifted entry
    u32 nn = 0, lv = 0, len = 0, tw = 0, tn = 0;
    bool done = !on;                                  // per wave
    u32 i = n, typ = 0, since = 2, sidx = 0, m1w = 0, m1n = 0;
#ifdef BZ_TAB_TIMING
    const unsigned long long tt1 = wall_clock64();
#endif
    for (;;) {
#ifdef BZ_TAB_TIMING
        nsteps++;
#endif
        const u32 p = i - 1u; // the entry a pop would take (i >= 1)
        const bool start = !done && since >= 1u; // (scalar) this step may start a sift-down: it reads the root, and
        bool wait = false;                       // for a pop the last entry -- if no lane's place (level lv) is above it
        const u32 leaf = (nn << 1) + 1u;
        const bool has = active & (leaf < len), has2 = active & (leaf + 1u < len);
        uint2 c1 = Hr[has ? leaf : 0u], c2 = Hr[has2 ? leaf + 1u : 0u];
        u32 rn = 0, rw = 0, en = 0, ew = 0;
        if (start) {
            const uint2 root = Hr[0], lastv = Hr[p];
            if (typ == 0u) {
                const u32 lp = 31u - (u32)__clz(p + 1u);
                const bool mine = active & (lp >= lv) & (((p + 1u) >> ((lp - lv) & 31u)) == nn + 1u);
                wait = __ballot(mine) != 0ull;
            }
            rn = (u32)__builtin_amdgcn_readfirstlane((int)root.x);
            rw = (u32)__builtin_amdgcn_readfirstlane((int)root.y);
            en = (u32)__builtin_amdgcn_readfirstlane((int)lastv.x);
            ew = (u32)__builtin_amdgcn_readfirstlane((int)lastv.y);
        }
        asm volatile("" : "+v"(c1.x), "+v"(c1.y), "+v"(c2.x), "+v"(c2.y)); // (whole entries now, not the nodes in a second round trip)
        // one level of this lane's sift-down (down_heap_w's loop body, cano_huff_table.rs:18-27)
        const u32 w2 = has2 ? c2.y : 0xFFFFFFFFu;
        const bool pick2 = c1.y > w2;
        const u32 sw = pick2 ? c2.y : c1.y, sn = pick2 ? c2.x : c1.x;
        const bool stop = !has | (tw < sw);
        if (active) Hw[nn] = stop ? make_uint2(tn, tw) : make_uint2(sn, sw);
        nn = stop ? nn : leaf + (pick2 ? 1u : 0u);
        lv += stop ? 0u : 1u;
        active = active & !stop;
        // the next sift-down of the table (scalar)
        since += 1u;
        if (!done && since >= 2u) {
            const bool take = lane == (sidx & (kPipeLanes - 1u));
            if (typ == 0u) { // :169-171  m1 = the root; the last entry takes its place
                if (i < 2u) done = true;
                else if (!wait) {
                    i = p;
                    m1w = rw;
                    m1n = rn;
                    active = active | take;
                    nn = take ? 0u : nn;
                    lv = take ? 0u : lv;
                    len = take ? i : len;
                    tw = take ? ew : tw;
                    tn = take ? en : tn;
                    sidx += 1u;
                    since = 0;
                    typ = 1u;
                }
            } else { // :172-177  m2 = the root; the new node takes its place
                const u32 mw = weight_add(m1w, rw);
                active = active | take;
                nn = take ? 0u : nn;
                lv = take ? 0u : lv;
                len = take ? i : len;
                tw = take ? mw : tw;
                tn = take ? i : tn;
                if (lane == 0) {
                    par[m1n] = (u16)i;
                    par[rn] = (u16)i;
                }
                sidx += 1u;
                since = 0;
                typ = 0u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (done && __ballot(active) == 0ull) break;
    }
#ifdef BZ_TAB_TIMING
    const unsigned long long tt2 = wall_clock64();
#endif
    heap_wave_sync();
    // :181-184, the depth of every inner node (the root is node 1; a parent's number is smaller than its child's, so
    // the reference walks the nodes upwards once).  Here by pointer jumping: Q[k] = hops << 16 | the node `hops` parents
    // above k; a round adds what the node up there has reached so far (one word: read and written whole), and the
    // root points at itself with 0 hops.  ceil(log2(depth)) rounds of the 64 lanes instead of n dependent round trips
    // of one.  Q lies over the heap, which is dead.
    u32 *Q = arena;
    if (on)
        for (u32 k = 1u + lane; k < n; k += 64u) Q[k] = k == 1u ? 1u : ((1u << 16) | (u32)par[k]);
    heap_wave_sync();
    for (;;) {
        bool more = false;
        if (on)
            for (u32 k = 2u + lane; k < n; k += 64u) {
                const u32 q = Q[k], up = Q[q & 0xFFFFu];
                Q[k] = (((q >> 16) + (up >> 16)) << 16) | (up & 0xFFFFu);
                more = more | ((up & 0xFFFFu) != 1u);
            }
        heap_wave_sync();
        if (__ballot(more) == 0ull) break;
    }
    bool too_long = false;
    if (on)
        for (u32 k = lane; k < n; k += 64u) { // :186-188
            const u32 l = (Q[par[k + n]] >> 16) + 1u;
            out[k] = (u8)l;
            if (l > kLim) too_long = true;
        }
    const bool any_long = __ballot(too_long) != 0ull;
    heap_wave_sync();
#ifdef BZ_TAB_TIMING
    if (lane == 0 && blockIdx.x == 0) {
        const unsigned long long tt3 = wall_clock64(), tc3 = clock64();
        printf("heap n %u: create %llu, main %llu (%u steps), tail %llu  [x10 ns]; shader clocks %llu\n", n, tt1 - tt0, tt2 - tt1, nsteps, tt3 - tt2, tc3 - tc0);
    }
#endif
    return any_long ? 1 : 0;
}

// The same procedure by a whole WAVE with the heap in its registers (WaveArr, bzgpu.h): the procedure is a chain of
// dependent accesses -- one lane per table paid an LDS round trip for each (0.8 ms per pass for a 258-symbol table,
// 12 ms per 256 MiB of binary data in k_huff_tables).  Statement for statement heap_code_lengths above (the reference's
// cano_huff_table.rs:14-38, :153-196 on the weights of encoder.rs:641-651); `buf` (2 n words of LDS) takes the finished
// array for the last two loops, which are gathers.  All 64 lanes call it with the same arguments.
template <u32 NR>
__device__ __forceinline__ void down_heap_wave(WaveArr<NR> &a, u32 nn, u32 len)
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
template <u32 NR>
__device__ int heap_code_lengths_wave(const u32 *rfreq, u32 n, u32 *buf, u8 *out, u32 lane)
{
    WaveArr<NR> a;
#pragma unroll
    for (u32 q = 0; q < (NR <= 1 ? 1u : (NR <= 4 ? 4u : 16u)); ++q) {
        const u32 e = q * 64u + lane;
        u32 v = 0;
        if (e < n) v = n + e;
        else if (e < 2u * n) {
            const u32 f = rfreq[e - n];
            v = (f > 1u ? f : 1u) << 8; // encoder.rs:642-645
        }
        a.r[q] = v;
    }
    for (u32 i = n >> 1; i-- > 0;) down_heap_wave(a, i, n); // create_heap, :33-38
    for (u32 i = n - 1; i >= 1; --i) { // :168-178
        const u32 m1 = a.get(0);
        a.set(0, a.get(i));
        down_heap_wave(a, 0, i);
        const u32 m2 = a.get(0);
        a.set(i, weight_add(a.get(m1), a.get(m2)));
        a.set(0, i);
        a.set(m1, i);
        a.set(m2, i);
        down_heap_wave(a, 0, i);
    }
    a.set(1, 0); // :181-184
    for (u32 i = 2; i < n; ++i) a.set(i, a.get(a.get(i)) + 1);
#pragma unroll
    for (u32 q = 0; q < (NR <= 1 ? 1u : (NR <= 4 ? 4u : 16u)); ++q) {
        const u32 e = q * 64u + lane;
        if (e < 2u * n) buf[e] = a.r[q];
    }
    __builtin_amdgcn_wave_barrier();
    bool too_long = false;
    for (u32 i0 = 0; i0 < n; i0 += 64u) { // :186-188
        const u32 i = i0 + lane;
        u32 l = 0;
        if (i < n) {
            l = buf[buf[i + n]] + 1;
            out[i] = (u8)l;
        }
        if (__ballot(l > kLim)) too_long = true;
    }
    __builtin_amdgcn_wave_barrier();
    return too_long ? 1 : 0; // 1: the caller must redo the table with the package-merge (:190-194)
}

// length-limited redo from the weights; scr/row as for gen_code_lm, the weights go to scr's tail
__device__ void lm_code_lengths(const u32 *rfreq, u32 n, u32 *scr, u32 scr_words, u32 row, u8 *out)
{
    u32 *w = scr + scr_words - kMaxAlpha;
    for (u32 i = 0; i < n; ++i) {
        const u32 f = rfreq[i];
        w[i] = (f > 1u ? f : 1u) << 8;
    }
    gen_code_lm(w, n, scr, row, out);
}

// MSB-first bit sink writing logical 32-bit words (single lane)
struct BitSink {
    u32 *w;
    u64 acc;   // bits are left aligned in acc
    u32 nacc;  // number of valid bits in acc (< 32 between calls)
    u32 widx;
    __device__ void put(u32 v, u32 nbits)
    {
        if (nbits == 0) return;
        acc |= ((u64)v << (64 - nacc - nbits));
        nacc += nbits;
        if (nacc >= 32) {
            w[widx++] = (u32)(acc >> 32);
            acc <<= 32;
            nacc -= 32;
        }
    }
    __device__ void finish()
    {
        if (nacc) w[widx] = (u32)(acc >> 32); // zero padded partial word
    }
    __device__ u32 bits() const { return widx * 32 + nacc; }
};

// Cooperative, coalesced load of the symbols of groups [g0, g0 + blockDim) into LDS (two u16 per
// dword; a group starts on a 100-byte = dword boundary).
__device__ __forceinline__ void stage_symbols(u32 *s_sym, const u16 *__restrict__ mtf, u32 g0, u32 mtf_count)
{
    const u32 first = g0 * kGSize;                       // symbol index, even
    const u32 avail = mtf_count - first;                 // > 0
    const u32 want = blockDim.x * kGSize;
    const u32 nsym = avail < want ? avail : want;
    const u32 ndw = (nsym + 1u) >> 1;
    const u32 *src = reinterpret_cast<const u32 *>(mtf + first);
    // kGSize / 2 = 25 dwords per thread, fetched as batches of loads issued back to back: a thread
    // waits for memory three times per call, not 25 times (all 25 at once costs the registers that
    // let two workgroups share a CU)
    constexpr u32 kBatch = 9;
#pragma unroll
    for (u32 k0 = 0; k0 < kGSize / 2; k0 += kBatch) {
        u32 v[kBatch];
#pragma unroll
        for (u32 k = 0; k < kBatch; ++k) {
            const u32 i = threadIdx.x + (k0 + k) * blockDim.x;
            v[k] = (k0 + k < kGSize / 2 && i < ndw) ? src[i] : 0u;
        }
#pragma unroll
        for (u32 k = 0; k < kBatch; ++k) {
            const u32 i = threadIdx.x + (k0 + k) * blockDim.x;
            if (k0 + k < kGSize / 2 && i < ndw) s_sym[i] = v[k];
        }
    }
}

__global__ __launch_bounds__(kHuffThreads, 4) void k_huffman(HuffArgs a) // (4 waves per SIMD = two workgroups per CU)
{
    __shared__ u8 s_len[6][kMaxAlpha + 6];
    __shared__ u32 s_rfreq[6][kMaxAlpha];
    __shared__ unsigned long long s_pack[kMaxAlpha]; // 6 x 10-bit lengths per symbol
    __shared__ u32 s_buf[6][2 * kMaxAlpha + 4];
    __shared__ u32 s_lm[6];
    __shared__ u32 s_need[6];
    __shared__ u32 s_scan[kHuffThreads / 64];
    __shared__ u32 s_run;
    __shared__ u32 s_first[6][24], s_lcount[6][24];
    // 512 groups x 50 symbols staged per sweep step: coalesced global reads, and a lane's 50
    // symbols are 25 dwords at a 25-dword stride (odd => no LDS bank conflicts)
    __shared__ u32 s_sym[kHuffThreads * kGSize / 2];

    const u32 lb = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#define BZ_HT(i)
    const u16 *mtf = a.mtf + (size_t)lb * a.mtf_stride;
    const u32 *mtf_freq = a.mtf_freq + (size_t)lb * kMaxAlpha;
    BlockOut &bo = a.out[lb];
    const u32 mtf_count = bo.mtf_count;
    const u32 in_use_count = bo.in_use_count;
    const u32 alpha = in_use_count + 2; // encoder.rs:367
    u8 *selector = a.selector + (size_t)lb * kSelStride;

    u32 group_num; // encoder.rs:370-376
    if (mtf_count < 200) group_num = 2;
    else if (mtf_count < 600) group_num = 3;
    else if (mtf_count < 1200) group_num = 4;
    else if (mtf_count < 2400) group_num = 5;
    else group_num = 6;
    const u32 n_selectors = (mtf_count + kGSize - 1) / kGSize;

    // lanes 0 .. group_num-1 of wave 0 each own one coding table in the single-lane sections
    const bool tab_lane = (wave == 0 && lane < group_num);
    const u32 tb = lane; // the table of a tab_lane
    if (tid < 6) s_lm[tid] = 0;
    // initial tables, encoder.rs:379-426.  The scan produces n_part = group_num..1; its
    // k-th result is table (group_num-1-k) in selector order... which the reference then
    // reverses, so scan result k IS the table with selector value (group_num-1-k).
    if (tid == 0) {
        u32 rem = mtf_count;
        int gs = 0;
        for (u32 k = 0; k < group_num; ++k) {
            const u32 n_part = group_num - k;
            const u32 t_freq = rem / n_part;
            int ge = gs - 1;
            u32 a_freq = 0;
            while (a_freq < t_freq && ge < (int)alpha - 1) {
                ge += 1;
                a_freq += mtf_freq[ge];
            }
            if (ge > gs && n_part != group_num && n_part != 1 && (((group_num - n_part) & 1u) == 1u)) {
                a_freq -= mtf_freq[ge];
                ge -= 1;
            }
            u8 *l = s_len[group_num - 1 - k];
            for (int i = 0; i < (int)alpha; ++i) l[i] = (i >= gs && i <= ge) ? 0 : 15; // :297-298
            rem -= a_freq;
            gs = ge + 1;
        }
    }
    __syncthreads();

    for (u32 iter = 0; iter < 4; ++iter) { // BZ_N_ITERS, encoder.rs:294,433
        for (u32 i = tid; i < alpha; i += kHuffThreads) {
            unsigned long long p = 0;
            for (u32 t = 0; t < group_num; ++t) p |= (unsigned long long)s_len[t][i] << (10 * t);
            s_pack[i] = p;
        }
        for (u32 i = tid; i < 6 * kMaxAlpha; i += kHuffThreads) (&s_rfreq[0][0])[i] = 0;
        __syncthreads();
        for (u32 g0 = 0; g0 < n_selectors; g0 += kHuffThreads) {
            stage_symbols(s_sym, mtf, g0, mtf_count);
            __syncthreads();
            BZ_HT(4)
            const u32 g = g0 + tid;
            if (g < n_selectors) {
                const u32 gs = g * kGSize;
                const u32 cnt = (gs + kGSize < mtf_count) ? kGSize : mtf_count - gs;
                const u32 *my = s_sym + tid * (kGSize / 2);
                // One pass over the group's symbols gives the cost under every table (6 x 10-bit sums) and
                // the counts of the eight smallest symbols (RUNA, RUNB, ranks 1..6 -- most of a BWT block)
                // as eight 8-bit fields of one register; the rarer symbols are counted by a second, almost
                // empty pass once the table is known (one LDS atomic per symbol otherwise serialises the
                // whole workgroup on a handful of hot counters).
                unsigned long long cost = 0, c8 = 0;
                u32 big = 0;
                if (cnt == kGSize) {
                    // a full group (all but a block's last): fixed trip count, so the LDS reads are issued
                    // in batches instead of one dependent pair per symbol (the loop was bound by LDS latency:
                    // ~300 cycles per symbol with four waves per SIMD).  Five dwords = ten symbols at a time:
                    // more in flight costs the registers that let two workgroups share a CU.
#pragma unroll 1
                    for (u32 k0 = 0; k0 < kGSize / 2; k0 += 5) {
                        u32 dw[5];
#pragma unroll
                        for (u32 k = 0; k < 5; ++k) dw[k] = my[k0 + k];
#pragma unroll
                        for (u32 k = 0; k < 5; ++k) {
                            const u32 s0 = dw[k] & 0xFFFFu, s1 = dw[k] >> 16;
                            cost += s_pack[s0];
                            cost += s_pack[s1];
                            if (s0 < 8u) c8 += 1ull << (8u * s0);
                            else ++big;
                            if (s1 < 8u) c8 += 1ull << (8u * s1);
                            else ++big;
                        }
                    }
                } else {
                    for (u32 i = 0; i < cnt; ++i) {
                        const u32 sy = (my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
                        cost += s_pack[sy];
                        if (sy < 8u) c8 += 1ull << (8u * sy);
                        else ++big;
                    }
                }
                // first minimum wins (min_by, encoder.rs:466)
                u32 bt = 0, bc = (u32)(cost & 1023u);
                for (u32 t = 1; t < group_num; ++t) {
                    const u32 ct = (u32)((cost >> (10 * t)) & 1023u);
                    if (ct < bc) {
                        bc = ct;
                        bt = t;
                    }
                }
                selector[g] = (u8)bt;
                BZ_HT(5)
                if (big) {
                    if (cnt == kGSize) {
#pragma unroll 5
                        for (u32 k = 0; k < kGSize / 2; ++k) {
                            const u32 d = my[k];
                            const u32 s0 = d & 0xFFFFu, s1 = d >> 16;
                            if (s0 >= 8u) atomicAdd(&s_rfreq[bt][s0], 1u);
                            if (s1 >= 8u) atomicAdd(&s_rfreq[bt][s1], 1u);
                        }
                    } else {
                        for (u32 i = 0; i < cnt; ++i) {
                            const u32 sy = (my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
                            if (sy >= 8u) atomicAdd(&s_rfreq[bt][sy], 1u);
                        }
                    }
                }
#pragma unroll
                for (u32 q = 0; q < 8; ++q) {
                    const u32 c = (u32)(c8 >> (8u * q)) & 0xFFu;
                    if (c) atomicAdd(&s_rfreq[bt][q], c);
                }
            }
            __syncthreads();
        }
        __syncthreads();
        BZ_HT(0)
        // one lane per table replays the serial heap procedure: lanes 0..5 of wave 0, in lockstep (the
        // tables are independent and run the same code; one wave instead of six keeps the workgroup
        // free to be any size)
        if (tab_lane) s_need[tb] = (u32)heap_code_lengths(s_rfreq[tb], alpha, s_buf[tb], s_len[tb]);
        __syncthreads();
        BZ_HT(1)
        // Tables whose longest code exceeds 17 bits are redone by the package-merge procedure
        // (one lane each).  Its scratch is carved out of the symbol staging area (idle here):
        // as many tables at a time as fit; global memory (8x the latency per step) only when the
        // alphabet is too large for even one.
        {
            const u32 row = 2u * alpha + 4u;
            const u32 need = 5u * kMaxAlpha + 64u + 2u * kLim * row;
            const u32 slots = (kHuffThreads * kGSize / 2) / need;
            u32 my_rank = 0, any = 0;
            for (u32 t = 0; t < group_num; ++t) {
                if (t < tb) my_rank += s_need[t] ? 1u : 0u;
                any += s_need[t] ? 1u : 0u;
            }
            if (any) { // uniform
                const u32 rounds_lm = slots ? (any + slots - 1) / slots : 1u;
                for (u32 rd = 0; rd < rounds_lm; ++rd) {
                    if (tab_lane && s_need[tb]) {
                        if (slots == 0) {
                            if (rd == 0) {
                                lm_code_lengths(s_rfreq[tb], alpha, a.lm_scratch + ((size_t)lb * 6 + tb) * kLmWords,
                                                kLmWords, kLmRow, s_len[tb]);
                                s_lm[tb] += 1;
                            }
                        } else if (my_rank / slots == rd) {
                            lm_code_lengths(s_rfreq[tb], alpha, s_sym + (my_rank % slots) * need, need, row,
                                            s_len[tb]);
                            s_lm[tb] += 1;
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }

    BZ_HT(2)
    // canonical codes, src/huffman/mod.rs:22-67 (stable by length, then symbol)
    if (tid < 6 * 24) (&s_lcount[0][0])[tid] = 0;
    __syncthreads();
    for (u32 i = tid; i < group_num * alpha; i += kHuffThreads) {
        const u32 t = i / alpha, s = i - t * alpha;
        atomicAdd(&s_lcount[t][s_len[t][s]], 1u);
    }
    __syncthreads();
    if (tid < group_num) {
        u32 code = 0;
        for (u32 l = 1; l < 24; ++l) {
            code = (code + s_lcount[tid][l - 1]) << 1; // s_lcount[.][0] == 0: every symbol is coded
            s_first[tid][l] = code;
        }
    }
    __syncthreads();
    u32 *code_len = a.code_len + (size_t)lb * 6 * kMaxAlpha;
    u32 my_max = 0;
    for (u32 i = tid; i < group_num * alpha; i += kHuffThreads) {
        const u32 t = i / alpha, s = i - t * alpha;
        const u32 l = s_len[t][s];
        u32 r = 0;
        for (u32 q = 0; q < s; ++q) r += (s_len[t][q] == l) ? 1u : 0u;
        code_len[t * kMaxAlpha + s] = (s_first[t][l] + r) | (l << 24);
        my_max = l > my_max ? l : my_max;
    }
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) {
        const u32 o = __shfl_xor(my_max, dd, 64);
        my_max = o > my_max ? o : my_max;
    }
    if (tid == 0) s_run = 0;
    if (tid < kHuffThreads / 64) s_scan[tid] = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_scan[0], my_max);
    __syncthreads();
    const u32 max_len = s_scan[0];
    __syncthreads();

    // payload bit offset of every group (exclusive scan over the groups)
    u32 *gbo = a.group_bitoff + (size_t)lb * kGboStride;
    for (u32 g0 = 0; g0 < n_selectors; g0 += kHuffThreads) {
        stage_symbols(s_sym, mtf, g0, mtf_count);
        __syncthreads();
        const u32 g = g0 + tid;
        u32 bits = 0;
        if (g < n_selectors) {
            const u32 gs = g * kGSize;
            const u32 cnt = (gs + kGSize < mtf_count) ? kGSize : mtf_count - gs;
            const u32 *my = s_sym + tid * (kGSize / 2);
            const u8 *l = s_len[selector[g]];
            if (cnt == kGSize) {
#pragma unroll 5
                for (u32 k = 0; k < kGSize / 2; ++k) {
                    const u32 d = my[k];
                    bits += (u32)l[d & 0xFFFFu] + (u32)l[d >> 16];
                }
            } else {
                for (u32 i = 0; i < cnt; ++i) bits += l[(my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu];
            }
        }
        const u32 inc = wave_incl_sum(bits);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        u32 carry = s_run, tot = 0;
        for (u32 k = 0; k < kHuffThreads / 64; ++k) {
            if (k < wave) carry += s_scan[k];
            tot += s_scan[k];
        }
        if (g < n_selectors) gbo[g] = carry + inc - bits;
        __syncthreads();
        if (tid == 0) s_run += tot;
        __syncthreads();
    }
    const u32 payload_bits = s_run;

    // ---- block header (a few hundred to ~25k bits) ----------------------------------------
    // Fixed fields and the coding tables are written by one lane; the selector list (up to
    // 18002 unary codes of MTF positions, encoder.rs:511-517,567-574) by all lanes: the MTF
    // position over <= 6 table ids is the number of ids seen more recently, which needs only
    // each id's last occurrence -- six max-scans over the workgroup.
    u32 *stream = a.stream + (size_t)lb * kStreamWords;
    const u32 *ubits = a.inuse_bits + lb * 8;
    u32 in_use16 = 0, used_ranges = 0;
    for (u32 i = 0; i < 16; ++i) {
        const u32 half = (ubits[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
        in_use16 = (in_use16 << 1) | (half ? 1u : 0u);
        used_ranges += half ? 1u : 0u;
    }
    const u32 sel_bit0 = 48u + 32u + 1u + 24u + 16u + 16u * used_ranges + 3u + 15u; // first selector bit

    // every lane: its contiguous share of the selectors
    const u32 seg = (n_selectors + kHuffThreads - 1) / kHuffThreads;
    const u32 g_lo = tid * seg < n_selectors ? tid * seg : n_selectors;
    const u32 g_hi = (g_lo + seg < n_selectors) ? g_lo + seg : n_selectors;
    int last[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) last[t] = -1000000;
    for (u32 g = g_lo; g < g_hi; ++g) {
        const u32 v = selector[g];
#pragma unroll
        for (int t = 0; t < 6; ++t)
            if ((u32)t == v) last[t] = (int)g;
    }
    // exclusive max-scan of each id's last occurrence over the lanes
    int start_last[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int inc = wave_incl_max32(last[t]);
        if (lane == 63) s_scan[wave] = (u32)inc;
        __syncthreads();
        int carry = -1000000;
        for (u32 k = 0; k < wave; ++k) carry = (int)s_scan[k] > carry ? (int)s_scan[k] : carry;
        const int prev = __shfl_up(inc, 1, 64);
        int ex = (lane == 0) ? -1000000 : prev;
        ex = ex > carry ? ex : carry;
        // never seen yet: the initial list is 0,1,2,... (id t is behind t smaller ids)
        start_last[t] = (ex < 0) ? -(t + 1) : ex;
        __syncthreads();
    }
    // first replay: bits of this lane's unary codes
    u32 my_bits = 0;
    {
        int cur[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) cur[t] = start_last[t];
        for (u32 g = g_lo; g < g_hi; ++g) {
            const u32 v = selector[g];
            int lv = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) lv = cur[t];
            u32 pos = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) pos += (cur[t] > lv) ? 1u : 0u;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) cur[t] = (int)g;
            my_bits += pos + 1u;
        }
    }
    u32 sel_off, sel_total;
    {
        const u32 inc = wave_incl_sum(my_bits);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        u32 carry = 0, tot = 0;
        for (u32 k = 0; k < kHuffThreads / 64; ++k) {
            if (k < wave) carry += s_scan[k];
            tot += s_scan[k];
        }
        sel_off = carry + inc - my_bits;
        sel_total = tot;
        __syncthreads();
    }
    const u32 tab_bit0 = sel_bit0 + sel_total;
    // table bits: 5 + sum over symbols of (2*|delta| + 1), one lane per table
    if (tab_lane) {
        const u8 *l = s_len[tb];
        u32 nb = 5, curr = l[0];
        for (u32 i = 0; i < alpha; ++i) {
            const u32 li = l[i];
            nb += 2u * (li > curr ? li - curr : curr - li) + 1u;
            curr = li;
        }
        s_first[tb][0] = nb; // s_first[t][0] is unused by the code assignment
    }
    __syncthreads();
    u32 tab_off[7];
    tab_off[0] = tab_bit0;
    for (u32 t = 0; t < 6; ++t) tab_off[t + 1] = tab_off[t] + (t < group_num ? s_first[t][0] : 0u);
    const u32 hb = tab_off[group_num];
    // zero the header words, then everybody ORs its bits in
    for (u32 w = tid; w <= (hb >> 5); w += kHuffThreads) stream[w] = 0;
    __syncthreads();
    auto put = [&](u32 bitpos, u32 v, u32 nbits) { // nbits <= 32, MSB-first
        const u32 w = bitpos >> 5, o = bitpos & 31u;
        const unsigned long long x = (unsigned long long)v << (64u - o - nbits);
        atomicOr(&stream[w], (u32)(x >> 32));
        if ((u32)x) atomicOr(&stream[w + 1], (u32)x);
    };
    if (tid == 0) {
        put(0, 0x314159u, 24);  // encoder.rs:254-259
        put(24, 0x265359u, 24);
        put(48, a.crc[lb], 32); // :262, the randomised bit (:273) stays 0
        put(81, a.orig_ptr[lb], 24); // :333
        put(105, in_use16, 16); // mapping table, :527-554
        u32 bp = 121;
        for (u32 i = 0; i < 16; ++i) {
            const u32 half = (ubits[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
            if (half) {
                put(bp, __brev(half) >> 16, 16); // byte value 16i+j is bit j: emit j = 0 first
                bp += 16;
            }
        }
        put(bp, group_num, 3); // :569-570
        put(bp + 3, n_selectors, 15);
    }
    {
        // second replay: write the unary codes (1 << (pos+1)) - 2, pos+1 bits (:572-574)
        int cur[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) cur[t] = start_last[t];
        u32 bp = sel_bit0 + sel_off;
        for (u32 g = g_lo; g < g_hi; ++g) {
            const u32 v = selector[g];
            int lv = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) lv = cur[t];
            u32 pos = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) pos += (cur[t] > lv) ? 1u : 0u;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) cur[t] = (int)g;
            put(bp, (1u << (pos + 1u)) - 2u, pos + 1u);
            bp += pos + 1u;
        }
    }
    // coding tables, :583-601: one lane per table
    if (tab_lane) {
        const u8 *l = s_len[tb];
        u32 bp = tab_off[tb];
        u32 curr = l[0];
        put(bp, curr, 5);
        bp += 5;
        for (u32 i = 0; i < alpha; ++i) {
            const u32 li = l[i];
            while (curr < li) { put(bp, 2, 2); bp += 2; curr += 1; }
            while (curr > li) { put(bp, 3, 2); bp += 2; curr -= 1; }
            bp += 1; // the terminating 0 bit
        }
    }
    if (tid == 0) {
        bo.header_bits = hb;
        bo.pad = used_ranges | (sel_total << 5); // (the sections of the header: bz_gpu_debug_block_sections)
        bo.total_bits = (u64)hb + payload_bits;
        bo.group_num = group_num;
        bo.n_selectors = n_selectors;
        bo.max_len = max_len;
        bo.lm_tables = s_lm[0] + s_lm[1] + s_lm[2] + s_lm[3] + s_lm[4] + s_lm[5];
        bo.crc = a.crc[lb];
        bo.orig_ptr = a.orig_ptr[lb];
        if ((u64)hb + payload_bits + 96u > (u64)kStreamWords * 32u) atomicExch(a.error_flag, 1u);
    }
    // zero the words the payload kernel will OR into (the header's last, partial word stays)
    {
        const u64 tb = (u64)hb + payload_bits;
        u32 w0 = (hb >> 5) + 1u;
        u32 w1 = (u32)(tb >> 5) + 2u;
        if (w1 > kStreamWords) w1 = kStreamWords;
        for (u32 w = w0 + tid; w < w1; w += kHuffThreads) stream[w] = 0;
    }
    BZ_HT(3)
}

// ---- payload: one lane per 50-symbol group (encoder.rs:609-629) ---------------------------
__global__ __launch_bounds__(256) void k_emit_payload(HuffArgs a)
{
    __shared__ u32 s_code[6 * kMaxAlpha];
    __shared__ u32 s_sym[256 * kGSize / 2];
    const u32 lb = blockIdx.y;
    const BlockOut &bo = a.out[lb];
    const u32 mtf_count = bo.mtf_count;
    const u32 n_selectors = bo.n_selectors;
    const u32 g0 = blockIdx.x * 256u;
    if (g0 >= n_selectors) return;
    const u32 *code_len = a.code_len + (size_t)lb * 6 * kMaxAlpha;
    for (u32 i = threadIdx.x; i < 6 * kMaxAlpha; i += 256u) s_code[i] = code_len[i];
    const u16 *mtf = a.mtf + (size_t)lb * a.mtf_stride;
    stage_symbols(s_sym, mtf, g0, mtf_count);
    __syncthreads();
    const u32 g = g0 + threadIdx.x;
    if (g >= n_selectors) return;
    const u32 gs = g * kGSize;
    const u32 cnt = (gs + kGSize < mtf_count) ? kGSize : mtf_count - gs;
    const u32 *my = s_sym + threadIdx.x * (kGSize / 2);
    const u32 *tab = s_code + (u32)a.selector[(size_t)lb * kSelStride + g] * kMaxAlpha;
    const u32 bitpos = bo.header_bits + a.group_bitoff[(size_t)lb * kGboStride + g];
    u32 *stream = a.stream + (size_t)lb * kStreamWords;
    u32 widx = bitpos >> 5;
    u32 nacc = bitpos & 31u;
    u64 acc = 0;
    bool first = true;
    for (u32 i = 0; i < cnt; ++i) {
        const u32 cl = tab[(my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu];
        const u32 len = cl >> 24, code = cl & 0xFFFFFFu;
        acc |= (u64)code << (64u - nacc - len);
        nacc += len;
        if (nacc >= 32u) {
            const u32 word = (u32)(acc >> 32);
            if (first) {
                atomicOr(&stream[widx], word); // shared with the previous group / the header
                first = false;
            } else {
                stream[widx] = word;           // all 32 bits are this group's
            }
            ++widx;
            acc <<= 32;
            nacc -= 32u;
        }
    }
    if (nacc) atomicOr(&stream[widx], (u32)(acc >> 32)); // shared with the next group
}

// =====================================================================================================
// The same stage as SEVERAL kernels (the default).  k_huffman holds a block in one workgroup for ~2.3 ms -- the
// sweeps over its symbols (60 % of that) run on 512 lanes while 1189 blocks wait for 512 workgroup slots: three
// rounds, the last one a third full.  Split, the sweeps become bandwidth-shaped launches over (groups, blocks) and
// what is serial -- one lane per table replaying the reference's heap procedure -- is a 64-thread workgroup per
// block, all blocks resident at once:
//   k_huff_tables(0)            table count, initial tables (encoder.rs:370-426)
//   4 x [k_huff_sweep           cost of every 50-symbol group under every table, selector, symbol counts per
//                               selected table (:433-503)
//        k_huff_tables(i)]      code lengths from the counts: heap procedure, package-merge fallback (:504-509)
//   k_huff_gbits                payload bits of every group under the final tables
//   k_huff_header               canonical codes (src/huffman/mod.rs:22-67), group bit offsets, selector MTF and the
//                               block header (:511-601) -- the tail of k_huffman
// The tables travel between the kernels as lengths (glen), as 6 x 10-bit packed lengths per symbol (pack) and
// as counts (rfreq).  BZ_HUFF_SPLIT=0 selects k_huffman.
constexpr u32 kSweepThreads = 256;  // groups per sweep step
constexpr u32 kSweepTilesX = 11; // workgroups per block (each loops over its share of the group tiles)
// k_huff_tables' scratch arena in LDS: the heap procedure's work arrays (6 x (2 * 258 + 4) words), then -- they are
// dead by then -- the package-merge scratch of as many tables at a time as fit (one of up to 40 symbols: text; larger
// alphabets use global memory).  With the other arrays 24.6 KB per workgroup: six workgroups share a CU and every
// block of a 1 GiB batch is resident at once (at 40 KB it took two rounds of 0.6 ms each, four times over).
constexpr u32 kTabArena = 6 * (3 * kMaxAlpha + 4); // 4668 words: six tables' heap entries (u64) and parents (u16)

__device__ __forceinline__ u32 huff_group_num(u32 mtf_count) // encoder.rs:370-376
{
    if (mtf_count < 200) return 2;
    if (mtf_count < 600) return 3;
    if (mtf_count < 1200) return 4;
    if (mtf_count < 2400) return 5;
    return 6;
}

// gen_code_lm (cano_huff_table.rs:58-151) by a whole WAVE.  Single-lane, the procedure costs 0.48 ms for a
// 38-symbol table -- 1300 merge steps of a few dependent LDS accesses each -- and with it k_huff_tables is four
// times 1.1 ms.  What is serial in the reference is a two-pointer merge per level: the packages of level j (sums of
// neighbouring pairs, :113-116) against the sorted leaf weights, "package if its weight is GREATER, else leaf"
// (:117-127).  Both sequences are non-increasing -- the leaves by the sort (:64-70), the packages because level j is
// itself such a merge and weight_add is monotone on sorted pairs (equal sums of the high parts force equal high
// parts, and then the depth bytes are ordered too) -- so the merge is a matter of ranks: package k lands behind the
// k packages in front of it and the leaves whose weight is >= its own (ties go to the leaf), leaf i behind the i
// leaves in front of it and the packages whose weight is > its own; what lands at or beyond max_elem[j-1] is cut off
// (:112), and so is every package behind the last leaf (the loop ends with the leaves, :125).  Ranks are binary
// searches, one element per lane.  The stable sort (:64-70) is a rank sort.  The bookkeeping between the levels
// (take_package, :40-55, :129-141) runs level by level, the items of a level side by side.  Same scratch layout as gen_code_lm; the packages of a level are
// kept in its own val row (dead once the level below exists).  All 64 lanes of ONE wave call it (the other waves of
// the workgroup, if any, take no part).
// DOMAIN: the weights are u32 here and usize in the reference (encoder.rs:641-651; the masks keep 24 bits of
// occurrences per operand, the SUM keeps one more): equal as long as no package reaches 2^24 occurrences.  A package
// of level j holds every leaf at most once per level below it: <= 16 x the table's total, and a block's total is
// <= 900 001 + 258 -- 14.4 M < 16.7 M.  bz_gpu_debug_code_lengths refuses tables of 2^20 occurrences and more.
// (the lanes of ONE wave hand values to each other through LDS: a fence and a scheduling barrier, not a workgroup
// barrier -- k_huff_tables calls this from one of its six waves)
__device__ __forceinline__ void lm_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ void lm_code_lengths_wave(const u32 *rfreq, u32 n, u32 *scr, u32 scr_words, u32 row, u8 *out, u32 lane)
{
    const u32 lim = kLim;
    u32 *map = scr;
    u32 *sfreq = map + kMaxAlpha;
    u32 *c = sfreq + kMaxAlpha;
    u32 *misc = c + kMaxAlpha;
    u32 *max_elem = misc, *b = misc + 20, *cur = misc + 40;
    u32 *val = misc + 64 + kMaxAlpha;
    u32 *ty = val + lim * row;
    u32 *w = scr + scr_words - kMaxAlpha; // the weights, encoder.rs:642-645
    for (u32 i = lane; i < n; i += 64u) {
        const u32 f = rfreq[i];
        w[i] = (f > 1u ? f : 1u) << 8;
    }
    lm_wave_sync();
    // stable sort by weight, descending (:64-70), as ranks
    for (u32 i = lane; i < n; i += 64u) {
        const u32 f = w[i];
        u32 r = 0;
        for (u32 q = 0; q < n; ++q) {
            const u32 x = w[q];
            r += (x > f || (x == f && q < i)) ? 1u : 0u;
        }
        sfreq[r] = f;
        map[r] = i;
    }
    if (lane == 0) {
        for (u32 j = 0; j < lim; ++j) { max_elem[j] = 0; b[j] = 0; cur[j] = 0; }
        u32 excess = (1u << lim) - n;       // :75
        const u32 half = 1u << (lim - 1);   // :76
        max_elem[lim - 1] = n;              // :77
        for (u32 j = 0; j < lim; ++j) {     // :79-88
            if (excess >= half) {
                b[j] = 1;
                excess -= half;
            }
            excess <<= 1;
            if (lim >= 2 + j) max_elem[lim - 2 - j] = max_elem[lim - 1 - j] / 2 + n;
        }
        max_elem[0] = b[0];                 // :90-95
        for (u32 j = 1; j < lim; ++j)
            if (max_elem[j] > 2 * max_elem[j - 1] + b[j]) max_elem[j] = 2 * max_elem[j - 1] + b[j];
    }
    lm_wave_sync();
    for (u32 j = 0; j < lim; ++j)           // :97-98 (zero initialised vectors)
        for (u32 t = lane; t < max_elem[j]; t += 64u) { val[j * row + t] = 0; ty[j * row + t] = 0; }
    for (u32 i = lane; i < n; i += 64u) c[i] = lim; // :99
    lm_wave_sync();
    for (u32 t = lane; t < n && t < max_elem[lim - 1]; t += 64u) { // :101-104
        val[(lim - 1) * row + t] = sfreq[t];
        ty[(lim - 1) * row + t] = t;
    }
    if (lane == 0 && b[lim - 1] == 1) {     // :107-110
        c[0] -= 1;
        cur[lim - 1] += 1;
    }
    lm_wave_sync();
    const u32 last_leaf = sfreq[n - 1];
    for (u32 j = lim - 1; j > 0;) {         // :112-142
        const u32 next0 = cur[j], me = max_elem[j], cap = max_elem[j - 1];
        const u32 K = (me > next0 + 1u) ? (me - next0) / 2u : 0u; // packages: pairs (next, next + 1) with next + 1 < max_elem[j]
        u32 *vj = val + j * row, *dv = val + (j - 1) * row, *dt = ty + (j - 1) * row;
        for (u32 k0 = 0; k0 < K; k0 += 64u) {
            const u32 k = k0 + lane;
            u32 pk = 0;
            if (k < K) pk = weight_add(vj[next0 + 2u * k], vj[next0 + 2u * k + 1u]);
            lm_wave_sync();
            if (k < K) vj[k] = pk; // (writes [k0, k0 + 64) lie in front of every later read, at >= next0 + 2 (k0 + 64))
        }
        lm_wave_sync();
        for (u32 k = lane; k < K; k += 64u) {
            const u32 pk = vj[k];
            if (pk > last_leaf) {            // (a package behind the last leaf is never placed, :125)
                u32 lo = 0, hi = n;          // leaves in front of it: weight >= its own
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (sfreq[mid] >= pk) lo = mid + 1u;
                    else hi = mid;
                }
                const u32 pos = k + lo;
                if (pos < cap) {
                    dv[pos] = pk;
                    dt[pos] = n;
                }
            }
        }
        for (u32 i = lane; i < n; i += 64u) {
            const u32 sf = sfreq[i];
            u32 lo = 0, hi = K;              // packages in front of it: weight > its own
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (vj[mid] > sf) lo = mid + 1u;
                else hi = mid;
            }
            const u32 pos = i + lo;
            if (pos < cap) {
                dv[pos] = sf;
                dt[pos] = i;
            }
        }
        lm_wave_sync();
        j -= 1;
        if (lane == 0) cur[j] = 0;
        lm_wave_sync();
        if (b[j] == 1) { // (uniform)
            // take_package(ty, c, cur, j) (:40-55), level by level instead of depth first: it takes the next item of
            // level j; an item that is a package takes the next two of the level below, and so on.  Which items a
            // level loses is settled by how many packages the level above lost -- always "the next ones" -- and what
            // losing one does (a leaf's count goes down, the level's cursor up) commutes, so the order is free.  A level's
            // list holds every leaf at most once: the lanes touch different counts.
            u32 need = 1;
            for (u32 li = j; need > 0u && li < lim; ++li) {
                const u32 c0 = cur[li];
                u32 npk = 0;
                for (u32 t0 = 0; t0 < need; t0 += 64u) {
                    const u32 t = t0 + lane;
                    bool isp = false;
                    if (t < need) {
                        const u32 x = ty[li * row + c0 + t];
                        if (x == n) isp = true;
                        else c[x] -= 1;
                    }
                    npk += (u32)__popcll(__ballot(isp));
                }
                lm_wave_sync();
                if (lane == 0) cur[li] = c0 + need;
                need = 2u * npk;
                lm_wave_sync();
            }
        }
        lm_wave_sync();
    }
    for (u32 i = lane; i < n; i += 64u) out[map[i]] = (u8)c[i]; // :144-150
    lm_wave_sync();
}

// staging for any workgroup size (stage_symbols assumes its batch fits the 512-thread kernel's registers)
template <u32 THREADS>
__device__ __forceinline__ void stage_symbols_n(u32 *s_sym, const u16 *__restrict__ mtf, u32 g0, u32 mtf_count)
{
    const u32 first = g0 * kGSize;
    const u32 avail = mtf_count - first;
    const u32 want = THREADS * kGSize;
    const u32 nsym = avail < want ? avail : want;
    const u32 ndw = (nsym + 1u) >> 1;
    const u32 *src = reinterpret_cast<const u32 *>(mtf + first);
    u32 v[kGSize / 2];
#pragma unroll
    for (u32 k = 0; k < kGSize / 2; ++k) {
        const u32 i = threadIdx.x + k * THREADS;
        v[k] = (i < ndw) ? src[i] : 0u;
    }
#pragma unroll
    for (u32 k = 0; k < kGSize / 2; ++k) {
        const u32 i = threadIdx.x + k * THREADS;
        if (i < ndw) s_sym[i] = v[k];
    }
}

// (A wave per table with the heap in its registers instead of a lane per table with the heap in LDS was measured in
// round 4, profiles/r04_negatives.md: the Huffman stage of the 1 GiB text corpus
// takes 5.3 ms with it against 4.3 (38-symbol tables: six waves per block instead of one, 1189 blocks no longer
// resident at once), 13.8 against 13.2 per 256 MiB of 258-symbol binary data, 6.6 against 5.7 of random bytes -- the
// register array's writes are a compare and a select over up to sixteen registers, and that costs what the LDS round
// trips did.  The Deflate block kernel, where the same device pays, has 286-symbol tables and no LDS to spare.)
// Two forms (round 6), same lengths (the probe kernel runs both heap forms on every probed table):
//   kTabThreads = 384, six waves: a wave per table, the heap procedure pipelined (heap_code_lengths_pipe) -- a third
//     of the serial form's time for a block alone (258 symbols: 375 us a launch against 560), four times its
//     instructions in all;
//   64, one wave: a lane per table (heap_code_lengths_w) -- for batches that fill the SIMDs with one wave per block
//     (1189 blocks of text: 145 us a launch against 200 for the six-wave form).
// launch_huffman takes the six-wave form while a batch's waves find SIMDs of their own (kTabPipeBlocks).
constexpr u32 kTabPipeBlocks = 320; // 6 waves x 320 blocks: two per SIMD
template <u32 kTabThreads>
__global__ __launch_bounds__(kTabThreads) void k_huff_tables(HuffArgs a, u32 iter)
{
    __shared__ u8 s_len[6][kMaxAlpha + 6];
    __shared__ u32 s_rfreq[6][kMaxAlpha];
    __shared__ __attribute__((aligned(16))) u32 s_arena[kTabArena];
    __shared__ u32 s_need[6];
    __shared__ u32 s_lmcount;
    static_assert(kTabArena >= 6 * (3 * kMaxAlpha + 4) && (3 * kMaxAlpha + 4) % 2 == 0, "the heap work arrays fit the arena, 8-byte aligned");
    const u32 lb = blockIdx.x, tid = threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const BlockOut &bo = a.out[lb];
    const u32 mtf_count = bo.mtf_count;
    const u32 alpha = bo.in_use_count + 2; // encoder.rs:367
    const u32 group_num = huff_group_num(mtf_count);
    const u32 *mtf_freq = a.mtf_freq + (size_t)lb * kMaxAlpha;
    u32 *rfreq = a.rfreq + (size_t)lb * 6 * kMaxAlpha;
    u8 *glen = a.glen + (size_t)lb * 6 * (kMaxAlpha + 6);
    const bool tab_lane = tid < group_num; // (the single-lane paths: one lane per table)
    const u32 tb = tid;
    if (tid == 0) s_lmcount = 0;
    if (iter == 0) {
        // initial tables, encoder.rs:379-426 (see k_huffman)
        if (tid == 0) {
            u32 rem = mtf_count;
            int gs = 0;
            for (u32 k = 0; k < group_num; ++k) {
                const u32 n_part = group_num - k;
                const u32 t_freq = rem / n_part;
                int ge = gs - 1;
                u32 a_freq = 0;
                while (a_freq < t_freq && ge < (int)alpha - 1) {
                    ge += 1;
                    a_freq += mtf_freq[ge];
                }
                if (ge > gs && n_part != group_num && n_part != 1 && (((group_num - n_part) & 1u) == 1u)) {
                    a_freq -= mtf_freq[ge];
                    ge -= 1;
                }
                s_need[group_num - 1 - k] = (u32)gs | ((u32)(ge + 1) << 16); // the table's range [gs, ge]
                rem -= a_freq;
                gs = ge + 1;
            }
            a.hlm[lb] = 0;
        }
        for (u32 i = tid; i < 6 * kMaxAlpha; i += kTabThreads) rfreq[i] = 0;
        __syncthreads();
        for (u32 i = tid; i < group_num * alpha; i += kTabThreads) { // :297-298, all lanes
            const u32 t = i / alpha, sym = i - t * alpha, rg = s_need[t];
            s_len[t][sym] = (sym >= (rg & 0xFFFFu) && sym < (rg >> 16)) ? 0 : 15;
        }
        __syncthreads();
    } else {
        // the counts of the sweep come in, and are cleared for the next one
        for (u32 i = tid; i < 6 * kMaxAlpha; i += kTabThreads) {
            (&s_rfreq[0][0])[i] = rfreq[i];
            rfreq[i] = 0;
        }
        __syncthreads();
        // the heap procedure: a wave per table, eight sift-downs in flight, the heap in LDS (heap_code_lengths_pipe)
        if constexpr (kTabThreads >= 384) {
            const bool on = wv < group_num; // (uniform per wave)
            const u32 gq = on ? wv : 0u;
            const int lm = heap_code_lengths_pipe(s_rfreq[gq], alpha, s_arena + gq * (3 * kMaxAlpha + 4), s_len[gq], lane, on);
            if (on && lane == 0) s_need[wv] = (u32)lm;
        } else {
            if (tab_lane) s_need[tb] = (u32)heap_code_lengths_w(s_rfreq[tb], alpha, s_arena + tb * (3 * kMaxAlpha + 4), s_len[tb]);
        }
        __syncthreads();
        // tables whose longest code exceeds 17 bits: package-merge, one lane each, as many tables at a time as fit
        // the arena, side by side in global memory when not even one does
        const u32 row = 2u * alpha + 4u;
        const u32 need = 5u * kMaxAlpha + 64u + 2u * kLim * row;
        const u32 slots = kTabArena / need;
        u32 my_rank = 0, any = 0;
        for (u32 t = 0; t < group_num; ++t) {
            if (t < tb) my_rank += s_need[t] ? 1u : 0u;
            any += s_need[t] ? 1u : 0u;
        }
        (void)my_rank;
        if (any) { // uniform
            if (slots) {
                // one table after the other, each by one whole wave (lm_code_lengths_wave): wave 0
                if (wv == 0) {
                    for (u32 t = 0; t < group_num; ++t) {
                        if (s_need[t]) { // uniform
                            lm_code_lengths_wave(s_rfreq[t], alpha, s_arena, need, row, s_len[t], lane);
                            if (lane == 0) s_lmcount += 1;
                        }
                    }
                }
                __syncthreads();
            } else {
                // (large alphabets: the scratch of a table does not fit the arena -- it lies in global memory; the WAVE
                // builds a table there too.  Rounds 1-3 gave each table a single lane here:
                // 12 of the 13 ms the Huffman stage took per 256 MiB of 258-symbol binary data.)
                // (round 6: every table has scratch of its own there and a wave of its own here)
                for (u32 t = wv; t < group_num; t += kTabThreads / 64u) {
                    if (s_need[t]) { // uniform per wave
                        lm_code_lengths_wave(s_rfreq[t], alpha, a.lm_scratch + ((size_t)lb * 6 + t) * kLmWords, kLmWords, kLmRow,
                                             s_len[t], lane);
                        if (lane == 0) atomicAdd(&s_lmcount, 1u);
                    }
                }
                __syncthreads();
            }
        }
    }
    // lengths and the packed form the sweeps add up
    for (u32 i = tid; i < 6 * (kMaxAlpha + 6); i += kTabThreads) glen[i] = (&s_len[0][0])[i];
    unsigned long long *pack = a.pack + (size_t)lb * kMaxAlpha;
    for (u32 i = tid; i < alpha; i += kTabThreads) {
        unsigned long long p = 0;
        for (u32 t = 0; t < group_num; ++t) p |= (unsigned long long)s_len[t][i] << (10 * t);
        pack[i] = p;
    }
    if (tid == 0 && s_lmcount) a.hlm[lb] += s_lmcount;
}

// iter: the refinement pass (0 .. 3); its totc and fave (encoder.rs:435-470, the figures of the reference's
// log::debug! line :483-498) are summed into a.pass_stats for bz_gpu_debug_block_sections
__global__ __launch_bounds__(kSweepThreads) void k_huff_sweep(HuffArgs a, u32 iter)
{
    __shared__ unsigned long long s_pack[kMaxAlpha];
    __shared__ u32 s_pass[8];
    __shared__ u32 s_rfreq[6][kMaxAlpha];
    __shared__ u32 s_sym[kSweepThreads * kGSize / 2];
    const u32 lb = blockIdx.y, tid = threadIdx.x;
    const u32 mtf_count = a.out[lb].mtf_count;
    const u32 n_selectors = (mtf_count + kGSize - 1) / kGSize;
    if (blockIdx.x * kSweepThreads >= n_selectors) return;
    const u32 alpha = a.out[lb].in_use_count + 2;
    const u32 group_num = huff_group_num(mtf_count);
    const u16 *mtf = a.mtf + (size_t)lb * a.mtf_stride;
    u8 *selector = a.selector + (size_t)lb * kSelStride;
    const unsigned long long *pack = a.pack + (size_t)lb * kMaxAlpha;
    for (u32 i = tid; i < alpha; i += kSweepThreads) s_pack[i] = pack[i];
    for (u32 i = tid; i < 6 * kMaxAlpha; i += kSweepThreads) (&s_rfreq[0][0])[i] = 0;
    if (tid < 8) s_pass[tid] = 0;
    for (u32 g0 = blockIdx.x * kSweepThreads; g0 < n_selectors; g0 += kSweepTilesX * kSweepThreads) {
        __syncthreads(); // (the staging area is free again; first trip: tables and counters are set)
        stage_symbols_n<kSweepThreads>(s_sym, mtf, g0, mtf_count);
        __syncthreads();
        const u32 g = g0 + tid;
        u32 bt_mine = 0xFFFFFFFFu, bc_mine = 0; // (for the pass's figures: summed per wave below)
        if (g < n_selectors) {
            const u32 gs = g * kGSize;
            const u32 cnt = (gs + kGSize < mtf_count) ? kGSize : mtf_count - gs;
            const u32 *my = s_sym + tid * (kGSize / 2);
            // (as in k_huffman: the cost under every table as 6 x 10-bit sums, the eight smallest symbols counted
            // in eight 8-bit fields, the rarer ones by a second, almost empty pass once the table is known)
            unsigned long long cost = 0, c8 = 0;
            u32 big = 0;
            if (cnt == kGSize) {
#pragma unroll 5
                for (u32 k = 0; k < kGSize / 2; ++k) {
                    const u32 dw = my[k];
                    const u32 s0 = dw & 0xFFFFu, s1 = dw >> 16;
                    cost += s_pack[s0];
                    cost += s_pack[s1];
                    if (s0 < 8u) c8 += 1ull << (8u * s0);
                    else ++big;
                    if (s1 < 8u) c8 += 1ull << (8u * s1);
                    else ++big;
                }
            } else {
                for (u32 i = 0; i < cnt; ++i) {
                    const u32 sy = (my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
                    cost += s_pack[sy];
                    if (sy < 8u) c8 += 1ull << (8u * sy);
                    else ++big;
                }
            }
            u32 bt = 0, bc = (u32)(cost & 1023u); // first minimum wins (min_by, encoder.rs:466)
            for (u32 t = 1; t < group_num; ++t) {
                const u32 ct = (u32)((cost >> (10 * t)) & 1023u);
                if (ct < bc) {
                    bc = ct;
                    bt = t;
                }
            }
            selector[g] = (u8)bt;
            bt_mine = bt;
            bc_mine = bc;
            if (big) {
                for (u32 i = 0; i < cnt; ++i) {
                    const u32 sy = (my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
                    if (sy >= 8u) atomicAdd(&s_rfreq[bt][sy], 1u);
                }
            }
#pragma unroll
            for (u32 q = 0; q < 8; ++q) {
                const u32 c = (u32)(c8 >> (8u * q)) & 0xFFu;
                if (c) atomicAdd(&s_rfreq[bt][q], c);
            }
        }
        if (a.pass_stats) { // totc += bc, fave[bt] += 1 (:469-470), a wave at a time: one LDS add per figure instead of one per group
            const u32 wtot = wave_sum(bc_mine);
            u32 mine = 0;
#pragma unroll
            for (u32 t = 0; t < 6; ++t) {
                const u32 c = (u32)__popcll(__ballot(bt_mine == t));
                mine = (tid & 63u) == t + 1u ? c : mine;
            }
            if ((tid & 63u) == 0u) mine = wtot;
            if ((tid & 63u) < 7u && mine) atomicAdd(&s_pass[tid & 63u], mine);
        }
    }
    __syncthreads();
    u32 *rfreq = a.rfreq + (size_t)lb * 6 * kMaxAlpha;
    for (u32 i = tid; i < group_num * kMaxAlpha; i += kSweepThreads) {
        const u32 c = (&s_rfreq[0][0])[i];
        if (c) atomicAdd(&rfreq[i], c);
    }
    if (a.pass_stats && tid < 7 && s_pass[tid]) atomicAdd(&a.pass_stats[((size_t)lb * 4 + iter) * 8 + tid], s_pass[tid]);
}

// payload bits of every group under the final tables (raw, not yet summed) -> group_bitoff
__global__ __launch_bounds__(kSweepThreads) void k_huff_gbits(HuffArgs a)
{
    __shared__ u8 s_len[6][kMaxAlpha + 6];
    __shared__ u32 s_sym[kSweepThreads * kGSize / 2];
    const u32 lb = blockIdx.y, tid = threadIdx.x;
    const u32 mtf_count = a.out[lb].mtf_count;
    const u32 n_selectors = (mtf_count + kGSize - 1) / kGSize;
    if (blockIdx.x * kSweepThreads >= n_selectors) return;
    const u16 *mtf = a.mtf + (size_t)lb * a.mtf_stride;
    const u8 *selector = a.selector + (size_t)lb * kSelStride;
    const u8 *glen = a.glen + (size_t)lb * 6 * (kMaxAlpha + 6);
    u32 *gbo = a.group_bitoff + (size_t)lb * kGboStride;
    for (u32 i = tid; i < 6 * (kMaxAlpha + 6); i += kSweepThreads) (&s_len[0][0])[i] = glen[i];
    for (u32 g0 = blockIdx.x * kSweepThreads; g0 < n_selectors; g0 += kSweepTilesX * kSweepThreads) {
        __syncthreads();
        stage_symbols_n<kSweepThreads>(s_sym, mtf, g0, mtf_count);
        __syncthreads();
        const u32 g = g0 + tid;
        if (g < n_selectors) {
            const u32 gs = g * kGSize;
            const u32 cnt = (gs + kGSize < mtf_count) ? kGSize : mtf_count - gs;
            const u32 *my = s_sym + tid * (kGSize / 2);
            const u8 *l = s_len[selector[g]];
            u32 bits = 0;
            if (cnt == kGSize) {
#pragma unroll 5
                for (u32 k = 0; k < kGSize / 2; ++k) {
                    const u32 d = my[k];
                    bits += (u32)l[d & 0xFFFFu] + (u32)l[d >> 16];
                }
            } else {
                for (u32 i = 0; i < cnt; ++i) bits += l[(my[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu];
            }
            gbo[g] = bits;
        }
    }
}

// canonical codes, the exclusive scan of the group bits, selector MTF, block header: the tail of k_huffman
__global__ __launch_bounds__(kHuffThreads) void k_huff_header(HuffArgs a)
{
    __shared__ u8 s_len[6][kMaxAlpha + 6];
    __shared__ u32 s_scan[kHuffThreads / 64];
    __shared__ u32 s_run;
    __shared__ u32 s_first[6][24], s_lcount[6][24];
    const u32 lb = blockIdx.x;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    BlockOut &bo = a.out[lb];
    const u32 mtf_count = bo.mtf_count;
    const u32 alpha = bo.in_use_count + 2;
    const u32 group_num = huff_group_num(mtf_count);
    const u32 n_selectors = (mtf_count + kGSize - 1) / kGSize;
    const u8 *selector = a.selector + (size_t)lb * kSelStride;
    const u8 *glen = a.glen + (size_t)lb * 6 * (kMaxAlpha + 6);
    const bool tab_lane = (wave == 0 && lane < group_num);
    const u32 tb = lane;
    for (u32 i = tid; i < 6 * (kMaxAlpha + 6); i += kHuffThreads) (&s_len[0][0])[i] = glen[i];
    // canonical codes, src/huffman/mod.rs:22-67 (stable by length, then symbol)
    if (tid < 6 * 24) (&s_lcount[0][0])[tid] = 0;
    __syncthreads();
    for (u32 i = tid; i < group_num * alpha; i += kHuffThreads) {
        const u32 t = i / alpha, sy = i - t * alpha;
        atomicAdd(&s_lcount[t][s_len[t][sy]], 1u);
    }
    __syncthreads();
    if (tid < group_num) {
        u32 code = 0;
        for (u32 l = 1; l < 24; ++l) {
            code = (code + s_lcount[tid][l - 1]) << 1;
            s_first[tid][l] = code;
        }
    }
    __syncthreads();
    u32 *code_len = a.code_len + (size_t)lb * 6 * kMaxAlpha;
    u32 my_max = 0;
    for (u32 i = tid; i < group_num * alpha; i += kHuffThreads) {
        const u32 t = i / alpha, sy = i - t * alpha;
        const u32 l = s_len[t][sy];
        u32 r = 0;
        for (u32 q = 0; q < sy; ++q) r += (s_len[t][q] == l) ? 1u : 0u;
        code_len[t * kMaxAlpha + sy] = (s_first[t][l] + r) | (l << 24);
        my_max = l > my_max ? l : my_max;
    }
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) {
        const u32 o = __shfl_xor(my_max, dd, 64);
        my_max = o > my_max ? o : my_max;
    }
    if (tid == 0) s_run = 0;
    if (tid < kHuffThreads / 64) s_scan[tid] = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_scan[0], my_max);
    __syncthreads();
    const u32 max_len = s_scan[0];
    __syncthreads();

    // payload bit offset of every group: exclusive scan of the bits k_huff_gbits left
    u32 *gbo = a.group_bitoff + (size_t)lb * kGboStride;
    for (u32 g0 = 0; g0 < n_selectors; g0 += kHuffThreads) {
        const u32 g = g0 + tid;
        const u32 bits = (g < n_selectors) ? gbo[g] : 0u;
        const u32 inc = wave_incl_sum(bits);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        u32 carry = s_run, tot = 0;
        for (u32 k = 0; k < kHuffThreads / 64; ++k) {
            if (k < wave) carry += s_scan[k];
            tot += s_scan[k];
        }
        if (g < n_selectors) gbo[g] = carry + inc - bits;
        __syncthreads();
        if (tid == 0) s_run += tot;
        __syncthreads();
    }
    const u32 payload_bits = s_run;

    // ---- block header: as in k_huffman
    u32 *stream = a.stream + (size_t)lb * kStreamWords;
    const u32 *ubits = a.inuse_bits + lb * 8;
    u32 in_use16 = 0, used_ranges = 0;
    for (u32 i = 0; i < 16; ++i) {
        const u32 half = (ubits[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
        in_use16 = (in_use16 << 1) | (half ? 1u : 0u);
        used_ranges += half ? 1u : 0u;
    }
    const u32 sel_bit0 = 48u + 32u + 1u + 24u + 16u + 16u * used_ranges + 3u + 15u; // first selector bit
    const u32 seg = (n_selectors + kHuffThreads - 1) / kHuffThreads;
    const u32 g_lo = tid * seg < n_selectors ? tid * seg : n_selectors;
    const u32 g_hi = (g_lo + seg < n_selectors) ? g_lo + seg : n_selectors;
    int last[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) last[t] = -1000000;
    for (u32 g = g_lo; g < g_hi; ++g) {
        const u32 v = selector[g];
#pragma unroll
        for (int t = 0; t < 6; ++t)
            if ((u32)t == v) last[t] = (int)g;
    }
    int start_last[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int inc = wave_incl_max32(last[t]);
        if (lane == 63) s_scan[wave] = (u32)inc;
        __syncthreads();
        int carry = -1000000;
        for (u32 k = 0; k < wave; ++k) carry = (int)s_scan[k] > carry ? (int)s_scan[k] : carry;
        const int prev = __shfl_up(inc, 1, 64);
        int ex = (lane == 0) ? -1000000 : prev;
        ex = ex > carry ? ex : carry;
        start_last[t] = (ex < 0) ? -(t + 1) : ex;
        __syncthreads();
    }
    u32 my_bits = 0;
    {
        int cur[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) cur[t] = start_last[t];
        for (u32 g = g_lo; g < g_hi; ++g) {
            const u32 v = selector[g];
            int lv = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) lv = cur[t];
            u32 pos = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) pos += (cur[t] > lv) ? 1u : 0u;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) cur[t] = (int)g;
            my_bits += pos + 1u;
        }
    }
    u32 sel_off, sel_total;
    {
        const u32 inc = wave_incl_sum(my_bits);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        u32 carry = 0, tot = 0;
        for (u32 k = 0; k < kHuffThreads / 64; ++k) {
            if (k < wave) carry += s_scan[k];
            tot += s_scan[k];
        }
        sel_off = carry + inc - my_bits;
        sel_total = tot;
        __syncthreads();
    }
    const u32 tab_bit0 = sel_bit0 + sel_total;
    if (tab_lane) {
        const u8 *l = s_len[tb];
        u32 nb = 5, curr = l[0];
        for (u32 i = 0; i < alpha; ++i) {
            const u32 li = l[i];
            nb += 2u * (li > curr ? li - curr : curr - li) + 1u;
            curr = li;
        }
        s_first[tb][0] = nb;
    }
    __syncthreads();
    u32 tab_off[7];
    tab_off[0] = tab_bit0;
    for (u32 t = 0; t < 6; ++t) tab_off[t + 1] = tab_off[t] + (t < group_num ? s_first[t][0] : 0u);
    const u32 hb = tab_off[group_num];
    for (u32 w = tid; w <= (hb >> 5); w += kHuffThreads) stream[w] = 0;
    __syncthreads();
    auto put = [&](u32 bitpos, u32 v, u32 nbits) { // nbits <= 32, MSB-first
        const u32 w = bitpos >> 5, o = bitpos & 31u;
        const unsigned long long x = (unsigned long long)v << (64u - o - nbits);
        atomicOr(&stream[w], (u32)(x >> 32));
        if ((u32)x) atomicOr(&stream[w + 1], (u32)x);
    };
    if (tid == 0) {
        put(0, 0x314159u, 24);  // encoder.rs:254-259
        put(24, 0x265359u, 24);
        put(48, a.crc[lb], 32); // :262, the randomised bit (:273) stays 0
        put(81, a.orig_ptr[lb], 24); // :333
        put(105, in_use16, 16); // mapping table, :527-554
        u32 bp = 121;
        for (u32 i = 0; i < 16; ++i) {
            const u32 half = (ubits[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
            if (half) {
                put(bp, __brev(half) >> 16, 16);
                bp += 16;
            }
        }
        put(bp, group_num, 3); // :569-570
        put(bp + 3, n_selectors, 15);
    }
    {
        int cur[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) cur[t] = start_last[t];
        u32 bp = sel_bit0 + sel_off;
        for (u32 g = g_lo; g < g_hi; ++g) {
            const u32 v = selector[g];
            int lv = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) lv = cur[t];
            u32 pos = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) pos += (cur[t] > lv) ? 1u : 0u;
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((u32)t == v) cur[t] = (int)g;
            put(bp, (1u << (pos + 1u)) - 2u, pos + 1u); // :572-574
            bp += pos + 1u;
        }
    }
    if (tab_lane) { // coding tables, :583-601
        const u8 *l = s_len[tb];
        u32 bp = tab_off[tb];
        u32 curr = l[0];
        put(bp, curr, 5);
        bp += 5;
        for (u32 i = 0; i < alpha; ++i) {
            const u32 li = l[i];
            while (curr < li) { put(bp, 2, 2); bp += 2; curr += 1; }
            while (curr > li) { put(bp, 3, 2); bp += 2; curr -= 1; }
            bp += 1;
        }
    }
    if (tid == 0) {
        bo.header_bits = hb;
        bo.pad = used_ranges | (sel_total << 5); // (the sections of the header: bz_gpu_debug_block_sections)
        bo.total_bits = (u64)hb + payload_bits;
        bo.group_num = group_num;
        bo.n_selectors = n_selectors;
        bo.max_len = max_len;
        bo.lm_tables = a.hlm[lb];
        bo.crc = a.crc[lb];
        bo.orig_ptr = a.orig_ptr[lb];
        if ((u64)hb + payload_bits + 96u > (u64)kStreamWords * 32u) atomicExch(a.error_flag, 1u);
    }
    {
        const u64 tbits = (u64)hb + payload_bits;
        u32 w0 = (hb >> 5) + 1u;
        u32 w1 = (u32)(tbits >> 5) + 2u;
        if (w1 > kStreamWords) w1 = kStreamWords;
        for (u32 w = w0 + tid; w < w1; w += kHuffThreads) stream[w] = 0;
    }
}

void launch_huffman(hipStream_t st, const HuffArgs &a)
{
    static const bool split = !(getenv("BZ_HUFF_SPLIT") && atoi(getenv("BZ_HUFF_SPLIT")) == 0);
    if (a.pass_stats) (void)hipMemsetAsync(a.pass_stats, 0, (size_t)a.nb * 32 * sizeof(u32), st); // (BZ_HUFF_SPLIT=0: the figures stay 0)
    if (split) {
        const dim3 sweep_grid(kSweepTilesX, a.nb);
        const bool pipe = a.nb <= kTabPipeBlocks;
        auto tables = [&](u32 iter) {
            if (pipe) hipLaunchKernelGGL(k_huff_tables<384>, dim3(a.nb), dim3(384), 0, st, a, iter);
            else hipLaunchKernelGGL(k_huff_tables<64>, dim3(a.nb), dim3(64), 0, st, a, iter);
        };
        tables(0u);
        for (u32 iter = 1; iter <= 4; ++iter) { // BZ_N_ITERS, encoder.rs:294,433
            hipLaunchKernelGGL(k_huff_sweep, sweep_grid, dim3(kSweepThreads), 0, st, a, iter - 1u);
            tables(iter);
        }
        hipLaunchKernelGGL(k_huff_gbits, sweep_grid, dim3(kSweepThreads), 0, st, a);
        hipLaunchKernelGGL(k_huff_header, dim3(a.nb), dim3(kHuffThreads), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_huffman, dim3(a.nb), dim3(kHuffThreads), 0, st, a);
    }
    hipLaunchKernelGGL(k_emit_payload, dim3((kMaxSelectors + 255) / 256, a.nb), dim3(256), 0, st, a);
}

// ---- probe: code lengths of one frequency table through the device code --------------------
// (one wave; both forms of the length-limited procedure run and must agree: the single-lane replay and the
// wave-parallel one of k_huff_tables -- a disagreement is reported as lengths of 0xFF)
__global__ void k_probe_code_lengths(const u32 *freq, u32 alpha, u8 *out, u32 *lm_scr, int *lm_flag)
{
    __shared__ u32 s_buf[2 * kMaxAlpha + 4];
    __shared__ __attribute__((aligned(16))) u32 s_arena_w[3 * kMaxAlpha + 4];
    __shared__ u32 s_f[kMaxAlpha];
    __shared__ u8 s_o[kMaxAlpha + 6], s_o2[kMaxAlpha + 6], s_o3[kMaxAlpha + 6];
    __shared__ int s_lm, s_bad;
    if (threadIdx.x == 0) {
        for (u32 i = 0; i < alpha; ++i) s_f[i] = freq[i];
        const int lm = heap_code_lengths(s_f, alpha, s_buf, s_o);
        {   // (the form k_huff_tables uses must give the same lengths and the same verdict)
            const int lm2 = heap_code_lengths_w(s_f, alpha, s_arena_w, s_o2);
            s_bad = lm2 != lm ? 1 : 0;
            for (u32 i = 0; i < alpha; ++i)
                if (s_o2[i] != s_o[i]) s_bad = 1;
        }
        if (lm) lm_code_lengths(s_f, alpha, lm_scr, kLmWords, kLmRow, s_o);
        *lm_flag = lm;
        s_lm = lm;
    }
    __syncthreads();
    {   // the pipelined form (the one k_huff_tables uses), on the heap's own lengths: s_o3
        const int lm3 = heap_code_lengths_pipe(s_f, alpha, s_arena_w, s_o3, threadIdx.x, true);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (lm3 != s_lm) s_bad = 1;
            for (u32 i = 0; i < alpha; ++i)
                if (s_o3[i] != s_o2[i]) s_bad = 1;
        }
        __syncthreads();
    }
    if (s_lm) { // uniform
        lm_code_lengths_wave(s_f, alpha, lm_scr, kLmWords, kLmRow, s_o2, threadIdx.x);
        __syncthreads();
        if (threadIdx.x == 0)
            for (u32 i = 0; i < alpha; ++i)
                if (s_o2[i] != s_o[i]) s_o[i] = 0xFF;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (u32 i = 0; i < alpha; ++i) out[i] = s_bad ? (u8)0xFE : s_o[i]; // (0xFE: the two heap forms disagree)
}

void launch_probe_code_lengths(hipStream_t st, const u32 *d_freq, u32 alpha, u8 *d_out, u32 *lm_scr,
                               int *d_flag)
{
    hipLaunchKernelGGL(k_probe_code_lengths, dim3(1), dim3(64), 0, st, d_freq, alpha, d_out, lm_scr, d_flag);
}

} // namespace bzgpu
