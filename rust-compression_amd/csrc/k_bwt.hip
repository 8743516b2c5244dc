// k_bwt.hip -- batched Burrows-Wheeler transform: order of all CYCLIC ROTATIONS of
// every block of a batch.
//
// Reference being replaced: suffix_array::sais::bwt (src/suffix_array/sais.rs:266-272;
// a cyclic SA-IS, :12-264).  Only its RESULT is reproduced: the exact rotation
// order, including the tie rule for periodic blocks (equal rotations come out by
// DESCENDING (i - shift) mod n, shift = smallest start of a least rotation;
// SURVEY.md F4, pinned by tests/test_oracle_vectors.py::test_bwt_periodic_tie_rule).
// A serial induced sort is the wrong shape for a GPU; this is a prefix-doubling
// rotation sort built from LDS-tiled stable radix passes:
//
//   init   : the block's bytes are re-coded to b = ceil(log2(#symbols in use)) bits (k_pack_text keeps the
//            block as a string of b-bit symbols, so that a key is one 8-byte load and a shift) and c of them
//            (c = 30/b, at most 8; 4 when b = 8) make one key.  Phase A sorts the rotations
//            by key(j) (3 LSD radix passes); phase B walks that order, i -> j = V[i]-c, and sorts
//            the sequence stably by key(j) again (3 more passes): the result is ordered by the
//            first 2c symbols without ever holding more than 8 bytes per element.  Text of 36
//            symbols gets c = 5, i.e. the first 10 bytes, from six 10-bit passes.
//            Groups of equal 2c-prefixes get rank R[j] = group head position,
//            bit31 = "group is a singleton, rotation j is final".
//   round h: (Manber-Myers step) walk SA in order, i -> j = SA[i]-h; the non-final j
//            arrive ordered by the rank of rotation j+h.  A STABLE sort of that
//            sequence by R[j] (20 bits = two 10-bit passes; pass A also compacts
//            away final rotations) yields the 2h-order inside every old group.
//            Then flags (old group start / new group start), a max-scan, and the
//            scatter of SA and the refined ranks (binned by j >> 10, put in place by k_rank_place).
//            h doubles: 2c, 4c, ...  When fewer than a quarter of the rotations are left, a round
//            compacts them and sorts only them (four passes over the survivors).
//   stop   : no non-final rotation left (or h >= n: the block is periodic, finish
//            with the closed-form tie rule).
//
// Everything is per-block independent; a launch covers (110 tiles of 8192) x (blocks of the
// batch).  Tiles of one block are mapped to one XCD (bzgpu::xcd_remap) so the
// block's 3.6 MB rank array stays in that XCD's L2 for the gathers.
//
// Integer path: no MFMA.  Bound: HBM (streamed u32 arrays) + L2 gathers.
#include "bzgpu.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace bzgpu {
constexpr u32 kGhSpan = 8;  // tiles per workgroup (and per entry of the digit counts) of k_ghist_text
constexpr u32 kSymSpan = 8; // tiles per workgroup of k_block_symbols

enum { SRC_TEXT = 0, SRC_PAIRS = 1, SRC_MM = 2, SRC_WALK = 3, SRC_SURV = 4, SRC_LISTG = 5,
       SRC_TEXTK = 6, SRC_WALKK = 7, SRC_MMK = 8, // ..K: keys stored by the histogram kernel of the pass
       SRC_MMC = 9, // MM that also CARRIES the rank of rotation j+h along (k_radix_scatter_lb, first walk round)
       SRC_PERJ = 10,   // the survivors keyed by their own start (the period round, see k_period_find)
       SRC_PACKED = 11,   // one word per element: the digits still to be sorted by above bit 20, the rotation below (round 4:
                          // the pass in front of the last one of phase A writes these, the last one reads them)
       SRC_PERJK = 12 };  // PERJ's keys as its histogram kernel kept them (bit 31: the entry takes no part)
// sources whose sequence is the compacted list (length count[lb]) rather than all n positions
template <int SRC> struct src_is_list {
    static constexpr bool value = (SRC == SRC_PAIRS || SRC == SRC_SURV || SRC == SRC_LISTG || SRC == SRC_PERJ || SRC == SRC_PERJK || SRC == SRC_PACKED);
};

// per-block key geometry, produced by k_key_params
struct KeyInfo {
    u8 bits;   // bits per re-coded symbol
    u8 chars;  // symbols per key
    u8 pad[2];
};

// key(j): `chars` re-coded symbols of the rotation starting at j, first symbol most significant.
// The block's symbols are kept a second time as a PACKED string (k_pack_text): `bits` bits per symbol, first
// symbol in the most significant bits of the first byte, and the first 16 symbols once more behind the last one
// (rotations are cyclic).  A key is then `chars * bits` consecutive bits of that string: one unaligned 8-byte
// load, a byte swap and a shift -- instead of three dword loads, a look-up per symbol in a byte -> code table in
// LDS and the shifts that put the codes together (and 0.68 MB per block in the L2 instead of 0.9).
__device__ __forceinline__ u32 pkey(const u8 *__restrict__ pt, u32 j, u32 bits, u32 chars)
{
    const u32 bitpos = j * bits, kb = bits * chars; // (kb <= 32, bitpos < 2^23)
    u64 v;
    __builtin_memcpy(&v, pt + (bitpos >> 3), 8); // (one global_load_dwordx2: unaligned access is on for HSA code objects)
    v = __builtin_bswap64(v);
    return (u32)(v >> (64u - (bitpos & 7u) - kb)) & (u32)((1ull << kb) - 1ull);
}

// The new-group starts of a refined list as a BITMAP: one 64-bit word per row of 64 list elements (the ballot a
// refinement takes anyway), written by k_group_flags / k_group_refine, read by k_survivor_compact and by the walk
// pass that carries ranks along (k_radix_scatter_lb<SRC_MMC>).  Rows are read 16 at a time: lane r < 17 loads the
// word of row r (one 136-byte access per wave), the others take it with a lane read.  Elements at or beyond `cnt`
// count as starts (`fill`), words of rows that lie wholly beyond it were never written.
__device__ __forceinline__ u64 newbits_lane_word(const u64 *__restrict__ bits, size_t base, u32 wbase, u32 cnt, u32 l, bool fill)
{
    const u32 rowbase = wbase + l * 64u;
    u64 m = 0;
    if (l < 17u && rowbase < cnt) m = bits[(base + rowbase) >> 6];
    if (fill && l < 17u) {
        if (rowbase >= cnt) m = ~0ull;
        else if (cnt - rowbase < 64u) m |= ~0ull << (cnt - rowbase);
    }
    return m;
}

// ---- the period round's tables (k_period_find / k_period_bits / k_period_next), per block and listed distance ----
// For a block that agrees with itself at distance p (lin_p[lb][i], i < kPerK): mis = bitmap of the positions x with
// T[x] != T[(x + p) mod n], lt = bitmap of T[x] < T[(x + p) mod n], nxt[w] = the first such position at or behind 64 w
// (+ n when the search wraps).  One set is 281 608 bytes; the sets of a block live in its digit-count slot
// (gh_tiles: 2.7 MB per block, which only the fused walk rounds use -- room for eight) or, without the fused passes or
// with the pair comparison on (its verdict bytes live there), in the flag bytes' slot (room for three).
constexpr u32 kPerWords = kSlot / 64u; // 14080 words of 64 positions
constexpr u32 kPerSetBytes = 2u * kPerWords * 8u + ((kPerWords + 1u) * 4u + 7u) / 8u * 8u;
static_assert((size_t)kPerK * kPerSetBytes <= (size_t)kTilesPerBlock * 3 * kMaxBins * 4, "the period tables do not fit the digit-count slot");
__host__ __device__ __forceinline__ bool per_in_counts(const BwtArgs &a) { return a.gh_tiles != nullptr; }
__host__ __device__ __forceinline__ u32 per_kmax(const BwtArgs &a) { return per_in_counts(a) ? kPerK : (kSlot / kPerSetBytes < kPerK ? kSlot / kPerSetBytes : kPerK); }
__device__ __forceinline__ const u8 *per_set(const BwtArgs &a, u32 lb, u32 i)
{
    const u8 *base = per_in_counts(a) ? reinterpret_cast<const u8 *>(a.gh_tiles + (size_t)lb * kTilesPerBlock * 3 * kMaxBins)
                                      : a.flags + (size_t)lb * kSlot;
    return base + (size_t)i * kPerSetBytes;
}
__device__ __forceinline__ const u64 *per_mis(const BwtArgs &a, u32 lb, u32 i) { return reinterpret_cast<const u64 *>(per_set(a, lb, i)); }
__device__ __forceinline__ const u64 *per_lt(const BwtArgs &a, u32 lb, u32 i) { return per_mis(a, lb, i) + kPerWords; }
__device__ __forceinline__ const u32 *per_nxt(const BwtArgs &a, u32 lb, u32 i) { return reinterpret_cast<const u32 *>(per_mis(a, lb, i) + 2u * kPerWords); }
// Rotations x and x + p (both < n) agree until the first position m >= x (cyclic) where the block and the block
// shifted by p differ, and there rotation x reads T[m], rotation x + p reads T[m + p].
// per_first_mis: that m, in [x, x + n);  per_lt_at: T[m] < T[m + p], i.e. rot(x) < rot(x + p)
__device__ __forceinline__ u32 per_first_mis(const BwtArgs &a, u32 lb, u32 i, u32 x)
{
    const u32 w = x >> 6;
    const u64 here = per_mis(a, lb, i)[w] & (~0ull << (x & 63u));
    return here ? (w << 6) + (u32)__builtin_ctzll(here) : per_nxt(a, lb, i)[w + 1u];
}
__device__ __forceinline__ bool per_lt_at(const BwtArgs &a, u32 lb, u32 i, u32 n, u32 m)
{
    if (m >= n) m -= n;
    return (per_lt(a, lb, i)[m >> 6] >> (m & 63u)) & 1ull;
}
// The direction a survivor is keyed by in the period round (ascending: by its start, descending: by its mirrored
// start): that of a pair it belongs to at the depth reached -- for the first listed distance p under which it has a
// mate in its group, the pair (j, j + p) if those two still agree on `depth` symbols, else the pair (j - p, j) if THOSE
// do.  A heuristic only: k_period_mark checks every pair of neighbours of the sorted list exactly.
__device__ __forceinline__ bool per_key_ascending(const BwtArgs &a, u32 lb, u32 n, u32 depth, u32 j)
{
    const u32 kmax = per_kmax(a);
    for (u32 i = 0; i < kmax; ++i) {
        const u32 p = a.lin_p[(size_t)lb * kPerK + i];
        if (p == 0u) break;
        if (j + p < n) {
            const u32 m = per_first_mis(a, lb, i, j);
            if (m - j >= depth) return per_lt_at(a, lb, i, n, m);
        }
        if (j >= p) {
            const u32 m = per_first_mis(a, lb, i, j - p);
            if (m - (j - p) >= depth) return per_lt_at(a, lb, i, n, m);
        }
    }
    return true;
}

// (round 6) STRETCHES with a short period.  A block that agrees with itself at a listed distance p NOT LARGER THAN THE DEPTH
// REACHED holds stretches that repeat p bytes -- "ugh\n" thirty thousand times in libbzip2's sample3, cut into sixty
// stretches by the changed bytes of the corpus "binary": four groups of 60 000 rotations each, a quarter of the block, whose
// members agree for up to 7 KB.  The chain rule (neighbours a listed distance apart) orders such a group only when all its
// members lie in ONE stretch.  But the order of ALL of them follows from where their stretches end: rotation x follows the
// periodic pattern for e(x) = m(x) + p - x symbols (m(x): the first position at or behind x where the block differs from
// itself shifted by p) and then reads T[m + p] where the pattern goes on with T[m].  Members of one group share `depth` >= p
// symbols, so they follow the SAME pattern, and of two members with e(x) < e(y) the shorter one is the smaller iff its
// deviation is downward (T[m + p] < T[m]) -- y still reads the pattern's symbol there.  So inside a group
//     (down, e ascending, deviating byte) < (up, e descending, deviating byte)
// is the true order of classes of members; members with the same class agree on e + 1 symbols and stay one group.  Exact,
// no check needed (a member with e < depth shares its deviation with every other member of its group: one class).  The
// key: up << 28 | (up ? 0xFFFFF - e : e) << 8 | byte -- thirty bits, three passes (a.per_wide; the host asks for them when
// some block of the batch lists such a distance).  Whether a distance applies is a property of the GROUP (the first `depth`
// symbols decide it), so all members of a group are keyed under the same distance: the first listed one that applies.
__device__ __forceinline__ bool per_ekey(const BwtArgs &a, u32 lb, const u8 *__restrict__ text, u32 n, u32 depth, u32 x, u32 &key)
{
    const u32 kmax = per_kmax(a);
    for (u32 i = 0; i < kmax; ++i) {
        const u32 p = a.lin_p[(size_t)lb * kPerK + i];
        if (p == 0u) break;
        if (p > depth) continue;
        const u32 m = per_first_mis(a, lb, i, x); // in [x, x + n)
        const u32 e = m - x + p;
        if (e < depth) continue;
        u32 mp = (m >= n ? m - n : m) + p;
        mp = mp >= n ? mp - n : mp;
        const bool up = per_lt_at(a, lb, i, n, m);
        const u32 ec = e < 0xFFFFFu ? e : 0xFFFFFu;
        key = up ? ((1u << 28) | ((0xFFFFFu - ec) << 8) | (u32)text[mp]) : ((ec << 8) | (u32)text[mp]);
        return true;
    }
    return false;
}

// (round 6) SMALL groups of the survivor list -- 2 .. kLinkMax members -- are ranked member by member from direct comparisons
// of their rotations (k_link_scan), whatever the distances between the copies.  A byte per position of the compacted list,
// in the MTF stage's symbol buffer (free during the sort):
//   group_bytes  place inside the group | members << 4 (k_survivor_compact; 0: not a member of a small group); k_link_finalize
//                turns the byte of a member it has made final into 1 and every other one into 0
//   (the verdicts of the pairs: a word per list entry in the round's free key array, k_link_scan -> k_link_finalize)
constexpr u32 kLinkMax = 8; // the largest group ranked member by member (a result word holds 7 + 7 bits; members are read 4 + 4)
__device__ __forceinline__ u8 *group_bytes(const BwtArgs &a, u32 lb)
{
    return a.per_aux + (size_t)lb * kMtfStride * 2u;
}

// Batched form: the 16 rows of one lane.  All primary loads are issued back to back (clamped
// index instead of a branch), then all dependent gathers, so a wave keeps 16-32 memory
// operations in flight instead of one -- these kernels are latency-bound otherwise.
// first = index of the lane's row-0 element; rows are 64 apart.  Returns the participation mask.
template <int SRC, int ROWS = 16>
__device__ __forceinline__ u32 fetch_rows(const BwtArgs &a, u32 lb, const u8 *__restrict__ pt, u32 n, u32 hm,
                                          const u32 *__restrict__ Kin, const u32 *__restrict__ Vin, u32 first,
                                          u32 cnt, KeyInfo ki, u32 (&key)[ROWS], u32 (&val)[ROWS])
{
    const size_t base = (size_t)lb * kSlot;
    u32 ok = 0;
    if (SRC == SRC_TEXT) {
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            key[r] = pkey(pt, c, ki.bits, ki.chars);
            val[r] = c;
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
    } else if (SRC == SRC_WALK) {
        // phase B of the init: walk the key order of phase A, step back `chars` symbols
        const u32 cm = ki.chars % n;
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            const u32 s = ld_stream(Vin + base + c);
            val[r] = (s >= cm) ? s - cm : s + n - cm;
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) key[r] = pkey(pt, val[r], ki.bits, ki.chars);
    } else if (SRC == SRC_TEXTK) {
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            key[r] = ld_stream(Kin + base + c);
            val[r] = c;
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
    } else if (SRC == SRC_MMK) {
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            key[r] = ld_stream(Kin + base + c);
            const u32 s = ld_stream(a.SA + base + c);
            val[r] = (s >= hm) ? s - hm : s + n - hm;
            ok |= ((idx < cnt && !(key[r] & kFinalBit)) ? 1u : 0u) << r;
        }
    } else if (SRC == SRC_WALKK) {
        const u32 cm = ki.chars % n;
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            key[r] = ld_stream(Kin + base + c);
            const u32 s = ld_stream(Vin + base + c);
            val[r] = (s >= cm) ? s - cm : s + n - cm;
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
    } else if (SRC == SRC_PAIRS) {
        // Streamed lists: no index is clamped (a row beyond the list still lies inside the block's slot -- 110 tiles of
        // 8192 are exactly kSlot -- and takes no part), so the sixteen rows of a lane are ONE address register and sixteen
        // immediate offsets, and the participation mask is "the first k rows" (round 5; before: a compare, a select, a
        // 64-bit shift and two 64-bit adds per row and array).
        const u32 *kp = Kin + base, *vp = Vin + base;
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            key[r] = ld_stream_at(kp, first + r * 64u);
            val[r] = ld_stream_at(vp, first + r * 64u);
        }
        const u32 rows = first < cnt ? (cnt - first + 63u) >> 6 : 0u;
        ok = rows >= (u32)ROWS ? ((1u << ROWS) - 1u) : (1u << rows) - 1u;
    } else if (SRC == SRC_PACKED) {
        const u32 *vp = Vin + base;
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            key[r] = ld_stream_at(vp, first + r * 64u); // (the digit is taken with shift 20; the rotation rides in the low bits)
            val[r] = key[r] & 0xFFFFFu;
        }
        const u32 rows = first < cnt ? (cnt - first + 63u) >> 6 : 0u;
        ok = rows >= (u32)ROWS ? ((1u << ROWS) - 1u) : (1u << rows) - 1u;
    } else if (SRC == SRC_SURV) {
        // survivor round, first half: order the survivors by the rank of rotation j+h
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            val[r] = ld_stream(Vin + base + c);
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            u32 t = val[r] + hm;
            t = t >= n ? t - n : t;
            key[r] = a.R[base + t] & ~kFinalBit;
        }
    } else if (SRC == SRC_PERJ) {
        // the period round: order the survivors by where they start -- ascending or descending, as the first
        // difference between the block and the block shifted by its period decides for each of them (per_ascending)
        const u32 p = a.lin_p[(size_t)lb * kPerK]; // (hm: the depth reached, in symbols; 0: the block has no listed distance)
        // (round 6) a member of a SMALL group that k_link_scan has ranked is final already (k_link_finalize left 1 in its
        // group byte): it takes no part, the pass compacts the list (the keys it keeps carry bit 31 for such entries)
        const u8 *gb8 = (a.per_aux && a.per_links) ? group_bytes(a, lb) : nullptr;
        u32 gbv[ROWS];
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            val[r] = ld_stream(Vin + base + c);
            gbv[r] = gb8 ? gb8[c] : 0u;
            ok |= ((idx < cnt && gbv[r] != 1u) ? 1u : 0u) << r;
        }
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            u32 ek;
            if (gbv[r] == 1u) key[r] = 0x80000000u;
            else if (a.per_wide && p != 0u && per_ekey(a, lb, a.rle + a.blocks[lb].rle_off, n, hm, val[r], ek)) key[r] = ek;
            else key[r] = ((p != 0u && !per_key_ascending(a, lb, n, hm, val[r])) ? (n - 1u - val[r]) : val[r]) >> a.per_keyshift;
        }
    } else if (SRC == SRC_PERJK) {
        const u32 *kp = Kin + base, *vp = Vin + base;
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            key[r] = ld_stream(kp + c);
            val[r] = ld_stream(vp + c);
            ok |= ((idx < cnt && !(key[r] >> 31)) ? 1u : 0u) << r;
        }
    } else if (SRC == SRC_LISTG) {
        // survivor round, second half: walk that order, key = the survivor's own group
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            val[r] = ld_stream(Vin + base + c);
            ok |= (idx < cnt ? 1u : 0u) << r;
        }
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) key[r] = a.R[base + val[r]];
    } else {
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            const u32 s = ld_stream(a.SA + base + c);
            val[r] = (s >= hm) ? s - hm : s + n - hm;
        }
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) key[r] = a.R[base + val[r]];
#pragma unroll
        for (u32 r = 0; r < (u32)ROWS; ++r) {
            const u32 idx = first + r * 64u;
            ok |= ((idx < cnt && !(key[r] & kFinalBit)) ? 1u : 0u) << r;
        }
    }
    return ok;
}

// Stable rank of this lane's element among the elements of its wave that carry digit `dg` (rows are
// ranked one after the other; `cnt` = this wave's u16 counters in LDS).  0xFFFFFFFF: takes no part.
//  * match-any form: the lanes with the same digit are found with ballots, the lowest peers come first,
//    the highest peer publishes the new count.
//    (one returning LDS atomic add per row instead was measured in round 3 and is slower: profiles/r01_r03_measured_and_dropped.md)
template <int BITS>
__device__ __forceinline__ u32 rank_in_wave(u16 *cnt, u32 dg, bool ok, u32 l, u64 lt_mask)
{
    u32 rnk = 0xFFFFFFFFu;
    const u64 peers = wave_match_digit<BITS>(dg, ok);
    if (ok) {
        const u32 before = __popcll(peers & lt_mask);
        const u32 c0 = cnt[dg];
        rnk = c0 + before;
        if ((peers >> l) == 1ull) cnt[dg] = (u16)(c0 + before + 1u);
    }
    return rnk;
}

// ---- radix pass, part 1: per-tile digit histogram ---------------------------------
template <int SRC, int BITS>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(BwtArgs a, u32 shift, u32 h,
                                                              const u32 *__restrict__ Kin,
                                                              const u32 *__restrict__ Vin,
                                                              u32 *__restrict__ Kstore)
{
    constexpr u32 NB = 1u << BITS;
    // kHistCopies interleaved copies of the counters (the low lane bits pick one): text keys are skewed (the
    // top digit is a key's first two symbols) and adds to one LDS word serialise
    constexpr u32 kHistCopies = 4;
    __shared__ u32 s_hist[kHistCopies][NB];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    // (the period round's first pass compacts: its input is the whole survivor list, count2; its scan leaves the length of what
    // takes part in count)
    const u32 cnt = (SRC == SRC_PERJ || SRC == SRC_PERJK) ? a.count2[lb] : (src_is_list<SRC>::value ? a.count[lb] : n);
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = (SRC == SRC_MM || SRC == SRC_MMK || SRC == SRC_SURV) ? (((u32)ki.chars * 2u) << h) % n
                   : (SRC == SRC_PERJ ? (u32)((((u64)ki.chars * 2u) << h) < n ? (((u64)ki.chars * 2u) << h) : n) : 0u);

    for (u32 i = threadIdx.x; i < kHistCopies * NB; i += kSortThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    {
        u32 key[16], val[16];
        const u32 ok =
            fetch_rows<SRC>(a, lb, pt, n, hm, Kin, Vin, start + w * 1024u + l, cnt, ki, key, val);
#pragma unroll
        for (u32 r = 0; r < 16; ++r)
            if ((ok >> r) & 1u) atomicAdd(&s_hist[l & (kHistCopies - 1)][(key[r] >> shift) & (NB - 1)], 1u);
        // keys that cost a gather to build are kept for the scatter kernel of the same pass
        if ((SRC == SRC_TEXT || SRC == SRC_WALK || SRC == SRC_MM || SRC == SRC_SURV || SRC == SRC_LISTG || SRC == SRC_PERJ) && Kstore) {
            const size_t base = (size_t)lb * kSlot;
#pragma unroll
            for (u32 r = 0; r < 16; ++r) {
                const u32 idx = start + w * 1024u + r * 64u + l;
                if (idx < cnt) Kstore[base + idx] = key[r]; // (MM: final rotations keep their final bit)
            }
        }
    }
    __syncthreads();
    u32 *out = a.tile_hist + ((size_t)lb * kTilesPerBlock + tile) * kMaxBins;
    for (u32 i = threadIdx.x; i < NB; i += kSortThreads) {
        u32 sum = 0;
#pragma unroll
        for (u32 k = 0; k < kHistCopies; ++k) sum += s_hist[k][i];
        out[i] = sum;
    }
}

// ---- radix pass, part 2: per-block scan of the tile histograms ----------------------
// After it, bin_base[lb][bin] + tile_hist[lb][tile][bin] = first output slot of that (bin, tile).
template <int SRC, int BITS>
__global__ __launch_bounds__(kSortThreads) void k_radix_scan(BwtArgs a)
{
    constexpr u32 NB = 1u << BITS;
    constexpr u32 PER = (NB + kSortThreads - 1) / kSortThreads; // bins per thread, consecutive
    constexpr u32 CH = 11;                                      // tiles per batch of loads (110 = 10 x 11)
    __shared__ u32 s_wsum[kSortThreads / 64];
    const u32 lb = blockIdx.x;
    const u32 n = a.blocks[lb].n;
    const u32 cnt = (SRC == SRC_PERJ || SRC == SRC_PERJK) ? a.count2[lb] : (src_is_list<SRC>::value ? a.count[lb] : n);
    const u32 ntiles = (cnt + kSortTile - 1) / kSortTile;
    u32 *hist = a.tile_hist + (size_t)lb * kTilesPerBlock * kMaxBins;
    u32 *bin_base = a.bin_base + (size_t)lb * kMaxBins;

    // tile_hist[t][bin] becomes the count of that bin in tiles < t; bin_base[bin] the count of
    // all smaller bins.  Loads of a batch are issued together (the running sum is the only
    // dependency), the stores follow.
    const u32 d0 = threadIdx.x * PER;
    u32 tot[PER];
#pragma unroll
    for (u32 q = 0; q < PER; ++q) {
        const u32 dgt = d0 + q;
        u32 run = 0;
        if (dgt < NB) {
            for (u32 t0 = 0; t0 < ntiles; t0 += CH) {
                u32 v[CH];
#pragma unroll
                for (u32 k = 0; k < CH; ++k) v[k] = (t0 + k < ntiles) ? hist[(t0 + k) * kMaxBins + dgt] : 0u;
#pragma unroll
                for (u32 k = 0; k < CH; ++k) {
                    if (t0 + k < ntiles) hist[(t0 + k) * kMaxBins + dgt] = run;
                    run += v[k];
                }
            }
        }
        tot[q] = run;
    }
    u32 mine = 0;
#pragma unroll
    for (u32 q = 0; q < PER; ++q) mine += tot[q];
    const u32 inc = wave_incl_sum(mine);
    if ((threadIdx.x & 63u) == 63u) s_wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 carry = 0, total = 0;
    for (u32 k = 0; k < kSortThreads / 64; ++k) {
        if (k < (threadIdx.x >> 6)) carry += s_wsum[k];
        total += s_wsum[k];
    }
    u32 base = carry + inc - mine;
#pragma unroll
    for (u32 q = 0; q < PER; ++q) {
        const u32 dgt = d0 + q;
        if (dgt < NB) bin_base[dgt] = base;
        base += tot[q];
    }
    if ((!src_is_list<SRC>::value || SRC == SRC_PERJ) && threadIdx.x == 0) a.count[lb] = total; // list length for the next passes
}

// ---- radix pass, part 3: stable scatter -----------------------------------------------
// Wave w owns the contiguous elements [w*1024, (w+1)*1024) of the tile and walks them in
// 16 rows of 64, so (wave, row, lane) order == sequence order.  Ranking inside a row uses
// ballot matching; counts per (wave, digit) live in LDS as u16.
template <int SRC, int BITS>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(BwtArgs a, u32 shift, u32 h,
                                                                 const u32 *__restrict__ Kin,
                                                                 const u32 *__restrict__ Vin,
                                                                 u32 *__restrict__ Kout,
                                                                 u32 *__restrict__ Vout)
{
    constexpr u32 NB = 1u << BITS;
    constexpr u32 NW = kSortThreads / 64;
    // One 64 KiB LDS buffer, used twice: first as the per-(wave, digit) counters, then as the
    // staging area that puts the tile's elements in digit order so the global stores of
    // consecutive lanes hit consecutive addresses (a lane-per-bin scatter costs one cache line
    // per lane).
    __shared__ u32 s_buf[kSortTile];
    __shared__ u32 s_base[NB];
    __shared__ u16 s_tpre[NB];
    __shared__ u32 s_wsum[NW];
    __shared__ u32 s_total;
    u16 *s_cnt = reinterpret_cast<u16 *>(s_buf); // [NW][NB] u16 <= 64 KiB
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 cnt = (SRC == SRC_PERJ || SRC == SRC_PERJK) ? a.count2[lb] : (src_is_list<SRC>::value ? a.count[lb] : n);
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = (SRC == SRC_MM || SRC == SRC_MMK || SRC == SRC_SURV) ? (((u32)ki.chars * 2u) << h) % n : 0u;
    const size_t base = (size_t)lb * kSlot;

    {
        for (u32 i = threadIdx.x; i < NW * NB / 2; i += kSortThreads) s_buf[i] = 0;
        const u32 *hist = a.tile_hist + ((size_t)lb * kTilesPerBlock + tile) * kMaxBins;
        const u32 *bin_base = a.bin_base + (size_t)lb * kMaxBins;
        for (u32 i = threadIdx.x; i < NB; i += kSortThreads) s_base[i] = hist[i] + bin_base[i];
    }
    __syncthreads();

    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    u16 *my_cnt = s_cnt + w * NB;

    u32 key[16], val[16];
    u32 rnk[16]; // 0xFFFFFFFF = takes no part
    const u32 okmask =
        fetch_rows<SRC>(a, lb, pt, n, hm, Kin, Vin, start + w * 1024u + l, cnt, ki, key, val);
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const bool ok = (okmask >> r) & 1u;
        const u32 dg = (key[r] >> shift) & (NB - 1);
        rnk[r] = rank_in_wave<BITS>(my_cnt, dg, ok, l, lt_mask);
        // LDS ops of one wave retire in order: the next row's reads see this row's writes
    }
    __syncthreads();
    // exclusive prefix over waves per digit, and the tile's total per digit
    constexpr u32 PER = NB / kSortThreads; // digits per thread (1 or 2), consecutive
    u32 tot[PER];
    u32 mine = 0;
#pragma unroll
    for (u32 q = 0; q < PER; ++q) {
        const u32 dg = threadIdx.x * PER + q;
        u32 run = 0;
#pragma unroll
        for (u32 k = 0; k < NW; ++k) {
            const u32 c = s_cnt[k * NB + dg];
            s_cnt[k * NB + dg] = (u16)run;
            run += c;
        }
        tot[q] = run;
        mine += run;
    }
    // exclusive scan of the totals over digits (digit order == thread order)
    {
        const u32 inc = wave_incl_sum(mine);
        if (l == 63) s_wsum[w] = inc;
        __syncthreads();
        u32 carry = 0, total = 0;
        for (u32 k = 0; k < NW; ++k) {
            if (k < w) carry += s_wsum[k];
            total += s_wsum[k];
        }
        u32 ex = carry + inc - mine;
#pragma unroll
        for (u32 q = 0; q < PER; ++q) {
            s_tpre[threadIdx.x * PER + q] = (u16)ex;
            ex += tot[q];
        }
        if (threadIdx.x == 0) s_total = total;
    }
    __syncthreads();
    // position of every element in the tile's digit order
    u32 lpos[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 dg = (key[r] >> shift) & (NB - 1);
        lpos[r] = (rnk[r] != 0xFFFFFFFFu) ? (u32)s_tpre[dg] + (u32)my_cnt[dg] + rnk[r] : 0xFFFFFFFFu;
    }
    const u32 total = s_total;
    __syncthreads(); // counters are dead from here on: the buffer becomes the staging area
#pragma unroll
    for (u32 r = 0; r < 16; ++r)
        if (lpos[r] != 0xFFFFFFFFu) s_buf[lpos[r]] = key[r];
    __syncthreads();
    u32 dst[16];
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        dst[k] = 0xFFFFFFFFu;
        if (i < total) {
            const u32 kk = s_buf[i];
            const u32 dg = (kk >> shift) & (NB - 1);
            dst[k] = s_base[dg] + (i - (u32)s_tpre[dg]);
            Kout[base + dst[k]] = kk;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 r = 0; r < 16; ++r)
        if (lpos[r] != 0xFFFFFFFFu) s_buf[lpos[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        if (i < total) Vout[base + dst[k]] = s_buf[i];
    }
}

// =====================================================================================================
// Fused radix passes ("one sweep"): no per-pass histogram kernel.  The digit counts of a whole phase
// are taken once (the multiset of keys does not change while a phase permutes them: k_ghist_text for
// the two init phases, k_group_apply for the walk round), k_ghist_scan turns them into per-block
// digit bases, and every scatter pass finds its tile's offset inside a digit by decoupled look-back
// over the tiles in front of it: a tile publishes its digit counts (AGG), adds up its predecessors'
// until it meets an inclusive prefix (INCL), and publishes its own.  Tiles take tickets when they
// start, so every predecessor of a running tile is running or done -- no tile waits for one that
// has not been scheduled.  The words carry the pass number (epoch), so nothing is cleared between
// passes.  What it saves: the histogram kernels (and, for the passes whose keys are gathered, the
// round trip of the keys through memory).
// (kLb*, lb_give_up, st_sc1_x4 / ld_sc1_x4: bzgpu.h -- the ZLE stage uses the same look-back)
template <int B0, int B1, int B2>
__global__ __launch_bounds__(kSortThreads) void k_ghist_text(BwtArgs a, u32 *__restrict__ Kstore)
{
    constexpr u32 NB0 = 1u << B0, NB1 = 1u << B1, NB2 = 1u << B2;
    __shared__ u32 s_h0[2][NB0], s_h1[2][NB1], s_h2[2][NB2];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    // (kGhSpan tiles per workgroup and ONE set of counts for them: the counts of a tile are 12 KB for 8 KB of text, and
    // k_ghist_scan read them all back -- 1.6 GB each way per GiB)
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 start0 = tile * kGhSpan * kSortTile;
    if (start0 >= n) return;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    for (u32 i = threadIdx.x; i < 2 * NB0; i += kSortThreads) (&s_h0[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < 2 * NB1; i += kSortThreads) (&s_h1[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < 2 * NB2; i += kSortThreads) (&s_h2[0][0])[i] = 0;
    __syncthreads();
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const size_t base = (size_t)lb * kSlot;
#pragma unroll 1
    for (u32 sp = 0; sp < kGhSpan; ++sp) {
        const u32 start = start0 + sp * kSortTile;
        if (start >= n) break;
        u32 key[16], val[16];
        const u32 ok = fetch_rows<SRC_TEXT>(a, lb, pt, n, 0, nullptr, nullptr, start + w * 1024u + l, n, ki, key, val);
#pragma unroll
        for (u32 r = 0; r < 16; ++r) {
            if ((ok >> r) & 1u) {
                atomicAdd(&s_h0[l & 1u][key[r] & (NB0 - 1)], 1u);
                atomicAdd(&s_h1[l & 1u][(key[r] >> B0) & (NB1 - 1)], 1u);
                atomicAdd(&s_h2[l & 1u][(key[r] >> (B0 + B1)) & (NB2 - 1)], 1u);
                if (Kstore) Kstore[base + start + w * 1024u + r * 64u + l] = key[r];
            }
        }
    }
    __syncthreads();
    u32 *out = a.gh_tiles + ((size_t)lb * kTilesPerBlock + tile) * 3 * kMaxBins; // (entry `tile` holds the counts of kGhSpan tiles)
    for (u32 i = threadIdx.x; i < NB0; i += kSortThreads) out[i] = s_h0[0][i] + s_h0[1][i];
    for (u32 i = threadIdx.x; i < NB1; i += kSortThreads) out[kMaxBins + i] = s_h1[0][i] + s_h1[1][i];
    for (u32 i = threadIdx.x; i < NB2; i += kSortThreads) out[2 * kMaxBins + i] = s_h2[0][i] + s_h2[1][i];
}

// per block and digit position: digits smaller (exclusive scan of the summed tile counts).
// ntiles_from_count: the tiles that hold counts are those of the list of length count[lb] (the
// refinement that took them), else those of the block.  Sets count[lb] to the number of keys.
__global__ __launch_bounds__(kSortThreads) void k_ghist_scan(BwtArgs a, u32 ndig, u32 nbins0, u32 nbins1, u32 nbins2,
                                                             u32 ntiles_from_count, u32 span = 1u)
{
    __shared__ u32 s_wsum[kSortThreads / 64];
    const u32 lb = blockIdx.x;
    const u32 n = a.blocks[lb].n;
    const u32 len = ntiles_from_count ? a.count[lb] : n;
    const u32 ntiles = ((len + kSortTile - 1) / kSortTile + span - 1u) / span; // (span: tiles per entry of the counts, k_ghist_text)
    const u32 *gt = a.gh_tiles + (size_t)lb * kTilesPerBlock * 3 * kMaxBins;
    u32 total0 = 0;
    for (u32 dpos = 0; dpos < ndig; ++dpos) {
        const u32 nbins = dpos == 0 ? nbins0 : (dpos == 1 ? nbins1 : nbins2);
        const u32 per = (nbins + kSortThreads - 1) / kSortThreads; // bins per thread, consecutive (<= 4)
        u32 tot[4] = {0, 0, 0, 0};
        u32 mine = 0;
        for (u32 q = 0; q < per; ++q) {
            const u32 bin = threadIdx.x * per + q;
            u32 sum = 0;
            if (bin < nbins)
                for (u32 t = 0; t < ntiles; ++t) sum += gt[(size_t)t * 3 * kMaxBins + dpos * kMaxBins + bin];
            tot[q] = sum;
            mine += sum;
        }
        const u32 inc = wave_incl_sum(mine);
        __syncthreads();
        if ((threadIdx.x & 63u) == 63u) s_wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 carry = 0, total = 0;
        for (u32 k = 0; k < kSortThreads / 64; ++k) {
            if (k < (threadIdx.x >> 6)) carry += s_wsum[k];
            total += s_wsum[k];
        }
        u32 run = carry + inc - mine;
        for (u32 q = 0; q < per; ++q) {
            const u32 bin = threadIdx.x * per + q;
            if (bin < nbins) a.gbase[((size_t)lb * 3 + dpos) * kMaxBins + bin] = run;
            run += tot[q];
        }
        if (dpos == 0) total0 = total;
    }
    if (threadIdx.x == 0) a.count[lb] = total0;
}

// the scatter of a fused pass; dpos selects the digit position inside gbase
// WRITE_K = false: the keys are not stored (the pass that ends phase A of the init: phase B re-builds its
// keys from the block and only walks the order)
// PACK_OUT: the element leaves as ONE word, (key >> (shift + BITS)) << 20 | rotation: the digits this pass and the
// passes before it have used up are not needed again (4 bytes written instead of 8; the next pass is <SRC_PACKED>)
// (Tile shapes and occupancies that were measured and are not kept -- 1024 threads x 8 rows, 256 x 16, 512 x 12, forced
// occupancies, look-back windows, tiles without tickets: profiles/r05_sort_negatives.md, profiles/r06_sort_negatives.md.)
// -DBZ_SCATTER_TIMING: cycles (>> 6) per phase of a tile, summed over the tiles of all launches into loc_stats[32 + 8 * c + k]
// (c = 0: the streamed sources PAIRS / PACKED, c = 1: the sources that gather -- TEXT, WALK, MM, MMC; a barrier at every
// mark; the host prints them at the end of a sort): 0 ticket + set-up + counters cleared, 1 rows fetched and ranked, 2 counts over the
// waves + scan over the digits, 3 look-back, 4 places + keys staged, 5 keys read back and stored, 6 values staged,
// 7 values read back and stored
#ifdef BZ_SCATTER_TIMING
#define SC_T(k) do { __syncthreads(); const u64 t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&a.loc_stats[32 + ((SRC == SRC_PAIRS || SRC == SRC_PACKED) ? 0 : 8) + (k)], (u32)((t_ - t_prev) >> 6)); t_prev = t_; } while (0)
#else
#define SC_T(k) do { } while (0)
#endif
template <int SRC, int BITS, bool WRITE_K = true, bool PACK_OUT = false, int ROWS = 16, int TILE = (int)kSortTile>
__global__ __launch_bounds__(TILE / ROWS) void k_radix_scatter_lb(BwtArgs a, u32 shift, u32 h,
                                                                    const u32 *__restrict__ Kin,
                                                                    const u32 *__restrict__ Vin,
                                                                    u32 *__restrict__ Kout, u32 *__restrict__ Vout,
                                                                    u32 dpos, u32 epoch,
                                                                    const u32 *__restrict__ gate)
{
    constexpr u32 NB = 1u << BITS;
    constexpr u32 kT = (u32)TILE / ROWS; // threads: 512 with 16 rows per lane, 1024 with 8 (256 for a half tile)
    static_assert((u32)TILE == kSortTile, "the look-back words are laid out for tiles of kSortTile elements");
    constexpr u32 kTileStride = kTilesPerBlock; // tiles per block in the look-back words' array
    u32 *const tstate = a.tile_state;
    constexpr u32 kRows = ROWS;
    constexpr u32 NW = kT / 64;
    __shared__ u32 s_buf[TILE];
    __shared__ u32 s_base[NB];
    __shared__ u16 s_tpre[NB]; // (only between the scan over the digits and the look-back: see s_base below)
    __shared__ u32 s_wsum[NW];
    __shared__ u32 s_total;
    __shared__ u32 s_ticket;
    u16 *s_cnt = reinterpret_cast<u16 *>(s_buf); // [NW][NB] u16
    // Tiles are handed out in order, block by block, from one ticket counter per XCD -- the XCD this
    // workgroup really runs on (HW_REG_XCC_ID), so that all tiles of a block are handled on one XCD
    // and the look-back words can travel through that XCD's L2.  Workgroups are dealt to the XCDs
    // round-robin (observed: XCC_ID == linear id % 8), so every XCD gets as many workgroups as it has
    // tiles; the host checks that every counter reached its total and falls back to the three-kernel
    // passes if a dispatch ever does it differently.
#ifdef BZ_SCATTER_TIMING
    u64 t_prev = __builtin_readcyclecounter();
#endif
    const u32 xcd = (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; // XCC_ID, bits 3:0
    const u32 my_tiles = a.tiles * ((a.nb + 7u - xcd) / 8u); // (a.tiles: tiles per block the launch covers)
    if (threadIdx.x == 0) s_ticket = atomicAdd(&a.tickets[(size_t)epoch * 8u + xcd], 1u);
    __syncthreads();
    const u32 slot = s_ticket;
    if (slot >= my_tiles) return;
    const u32 b8 = slot / a.tiles;
    const u32 tile = slot - b8 * a.tiles;
    const u32 lb = b8 * 8u + xcd;
    if (gate && gate[lb] == 0u) return; // (phase B of the init: the block was ordered inside LDS)
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 cnt = src_is_list<SRC>::value ? a.count[lb] : n;
    const u32 start = tile * (u32)TILE;
    if (start >= cnt) return;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = (SRC == SRC_MM || SRC == SRC_MMC || SRC == SRC_MMK || SRC == SRC_SURV) ? (((u32)ki.chars * 2u) << h) % n : 0u;
    const size_t base = (size_t)lb * kSlot;
    for (u32 i = threadIdx.x; i < NW * NB / 2; i += kT) s_buf[i] = 0;
    __syncthreads();
    SC_T(0);

    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    u16 *my_cnt = s_cnt + w * NB;
    u32 key[kRows], val[kRows];
    u32 rnk[kRows]; // 0xFFFFFFFF = takes no part
    const u32 okmask =
        fetch_rows<SRC, ROWS>(a, lb, pt, n, hm, Kin, Vin, start + w * (64u * kRows) + l, cnt, ki, key, val);
    // SRC_MMC (the walk round that follows the first refinement): the list this pass builds is refined by
    // (group of j, rank of rotation j+h), and the rotation j+h of the element walked at position i is SA[i]: its rank
    // is the head of the group position i lies in -- the last new-group start at or before i, which the first
    // refinement left as flag bytes in SA order.  That head travels with the element in the 12 spare bits of key and
    // value (both are 20-bit numbers), so the refinement needs no gather of the rank array (k_group_refine<false>).
    // (The head goes into the spare bits at once -- the digits of this pass and the next are the low 20 bits of the
    // key; an element in front of its wave's first group start gets 0xFFFFF, "the start lies in front of this wave",
    // and is patched once the waves in front have said where: no mask is kept in registers.)
    __shared__ int s_wlast[NW];
    if (SRC == SRC_MMC) {
        int wl = -1;
        const u64 le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
        const u64 myword = newbits_lane_word(a.newbits, base, start + w * (64u * kRows), cnt, l, false);
#pragma unroll
        for (u32 r = 0; r < kRows; ++r) {
            const u32 rowbase = start + w * (64u * kRows) + r * 64u;
            const u64 mnew = wave_lane64(myword, r);
            const u64 q = mnew & le_mask;
            const u32 head = q ? rowbase + 63u - (u32)__clzll(q) : (wl >= 0 ? (u32)wl : 0xFFFFFu);
            key[r] = (key[r] & (0xFFFFFu | kFinalBit)) | ((head & 0xFFFu) << 20);
            val[r] = val[r] | ((head >> 12) << 20);
            if (mnew) wl = (int)(rowbase + 63u - (u32)__clzll(mnew));
        }
        if (l == 0) s_wlast[w] = wl;
    }
#pragma unroll
    for (u32 r = 0; r < kRows; ++r) {
        const bool ok = (okmask >> r) & 1u;
        const u32 dg = (key[r] >> shift) & (NB - 1);
        rnk[r] = rank_in_wave<BITS>(my_cnt, dg, ok, l, lt_mask);
    }
    __syncthreads();
    SC_T(1);
    constexpr u32 PER = NB / kT; // digits per thread (1, 2 or 4), consecutive
    u32 tot[PER];
    u32 mine = 0;
    // Round 5: the per-wave counts of a thread's digits stay in its registers (two u16 per register) until the scan over
    // the digits has said where each digit begins in the tile; the counters then take (tile offset of the digit +
    // elements of the waves in front) in ONE write, and an element's place in the staging buffer is one look-up
    // (my_cnt[dg] + rank) instead of two (s_tpre[dg] + my_cnt[dg] + rank).  Likewise s_base ends up as (global base -
    // tile offset): the destination of staged element i is s_base[dg] + i.  Two LDS reads and two adds less per element
    // and trip (r04_scatter_counters.md: the pass is held by its vector and LDS instructions, not by bytes).  The scan
    // over the digits comes BEFORE the look-back now, so that the counts are out of the registers while a tile waits
    // for its predecessors.
    {
        u32 cw[NW][(PER + 1) / 2];
#pragma unroll
        for (u32 q = 0; q < PER; ++q) {
            const u32 dg = threadIdx.x * PER + q;
            u32 run = 0;
#pragma unroll
            for (u32 k = 0; k < NW; ++k) {
                const u32 c = s_cnt[k * NB + dg];
                if (q & 1u) cw[k][q >> 1] |= c << 16;
                else cw[k][q >> 1] = c;
                run += c;
            }
            tot[q] = run;
            mine += run;
        }
        // exclusive scan of the totals over digits (digit order == thread order)
        const u32 inc = wave_incl_sum(mine);
        if (l == 63) s_wsum[w] = inc;
        __syncthreads();
        u32 carry = 0, total = 0;
        for (u32 k = 0; k < NW; ++k) {
            if (k < w) carry += s_wsum[k];
            total += s_wsum[k];
        }
        u32 ex = carry + inc - mine;
#pragma unroll
        for (u32 q = 0; q < PER; ++q) {
            const u32 dg = threadIdx.x * PER + q;
            u32 run = ex; // (<= 8192: fits the u16 counters)
#pragma unroll
            for (u32 k = 0; k < NW; ++k) {
                s_cnt[k * NB + dg] = (u16)run;
                run += (q & 1u) ? (cw[k][q >> 1] >> 16) : (cw[k][q >> 1] & 0xFFFFu);
            }
            s_tpre[dg] = (u16)ex;
            s_base[dg] = tot[q]; // (this tile's digit counts: what the look-back publishes)
            ex += tot[q];
        }
        if (threadIdx.x == 0) s_total = total;
    }
    __syncthreads();
    SC_T(2);
    // this tile's digit counts go out, the predecessors' come in: four digits per 16-byte word group
    // (one sc1 store / load moves 16 bytes for the price of 4, MI355X_MICROARCH.md)
    // Round 5: the counts go out HERE, but the walk over the predecessors' words waits
    // until the tile's keys have been staged and read back -- the in-kernel phase timers (-DBZ_SCATTER_TIMING,
    // profiles/r05_sort_negatives.md) put 30-35 % of a tile's time into a look-back that began the moment the counts
    // were out: consecutive tickets start together, so a tile's predecessor publishes when the tile does, and four of
    // the eight waves waited at the barrier behind the walk.  The staging in between is work the tile has to do anyway.
    const u32 lb_d0 = threadIdx.x * 4u;
    u32 *lb_mystate = tstate + ((size_t)lb * kTileStride + tile) * kMaxBins + lb_d0;
    const u32 etag = epoch << 22;
    u32 t4[4] = {0, 0, 0, 0};
    if (threadIdx.x < NB / 4u) {
#pragma unroll
        for (u32 k = 0; k < 4; ++k) t4[k] = s_base[lb_d0 + k];
        const u32 f0 = etag | (tile ? kLbAgg : kLbIncl);
        st_sc1_x4(lb_mystate, make_uint4(f0 | t4[0], f0 | t4[1], f0 | t4[2], f0 | t4[3]));
    }
    auto lookback_walk = [&]() {
        if (threadIdx.x >= NB / 4u) return;
        const u32 d0 = lb_d0;
        u32 excl[4] = {0, 0, 0, 0};
        if (tile) {
            u32 open = 0xFu; // digits whose sum has not met an inclusive prefix yet
            u32 spins = 0;
#ifdef BZ_SCATTER_TIMING
            u32 hops = 0, spins0 = 0;
#endif
            // (Asking for the words of two to four predecessors at once -- a tile walks back over five on average before it
            // meets an inclusive prefix -- was measured in round 5 and bought nothing: profiles/r05_sort_negatives.md.)
            for (u32 p = tile; p > 0 && open; --p) {
                const u32 *src = tstate + ((size_t)lb * kTileStride + (p - 1u)) * kMaxBins + d0;
                uint4 v = ld_sc1_x4(src);
#ifdef BZ_SCATTER_TIMING
                ++hops;
#endif
                while (true) {
                    const u32 x[4] = {v.x, v.y, v.z, v.w};
                    bool ready = true;
#pragma unroll
                    for (u32 k = 0; k < 4; ++k)
                        if ((x[k] >> 22) != epoch || (x[k] & kLbFlagMask) == 0u) ready = false;
                    if (ready) break; // (the four words of a group are written by one store)
                    if (lb_give_up(spins, a.sort_err, kLbSpinMax)) {
                        atomicExch(a.sort_err, 1u);
                        v = make_uint4(etag | kLbIncl, etag | kLbIncl, etag | kLbIncl, etag | kLbIncl);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    v = ld_sc1_x4(src);
#ifdef BZ_SCATTER_TIMING
                    ++spins0;
#endif
                }
                const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    if ((open >> k) & 1u) {
                        excl[k] += x[k] & kLbValMask;
                        if ((x[k] & kLbFlagMask) == kLbIncl) open &= ~(1u << k);
                    }
                }
            }
#ifdef BZ_SCATTER_TIMING
            if (threadIdx.x == 0) { // (thread 0's digit group: hops walked, loads repeated while a word was not there yet, tiles)
                atomicAdd(&a.loc_stats[48], hops);
                atomicAdd(&a.loc_stats[49], spins0);
                atomicAdd(&a.loc_stats[50], 1u);
                atomicMax(&a.loc_stats[51], hops);
            }
#endif
            const u32 fi = etag | kLbIncl;
            st_sc1_x4(lb_mystate, make_uint4(fi | (excl[0] + t4[0]), fi | (excl[1] + t4[1]), fi | (excl[2] + t4[2]),
                                             fi | (excl[3] + t4[3])));
        }
        const u32 *gb = a.gbase + ((size_t)lb * 3 + dpos) * kMaxBins + d0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) s_base[d0 + k] = gb[k] + excl[k] - (u32)s_tpre[d0 + k];
    };
    u32 lpos[kRows];
#pragma unroll
    for (u32 r = 0; r < kRows; ++r) {
        const u32 dg = (key[r] >> shift) & (NB - 1);
        lpos[r] = (rnk[r] != 0xFFFFFFFFu) ? (u32)my_cnt[dg] + rnk[r] : 0xFFFFFFFFu;
    }
    const u32 total = s_total;
    __syncthreads(); // counters are dead from here on: the buffer becomes the staging area
    if (SRC == SRC_MMC) {
        int carry = tile ? a.tile_last_new[lb * kTilesPerBlock + tile - 1u] : 0; // (settled by k_group_refine<true>)
        for (u32 k = 0; k < w; ++k) carry = s_wlast[k] > carry ? s_wlast[k] : carry;
        const u32 ck = ((u32)carry & 0xFFFu) << 20, cv = ((u32)carry >> 12) << 20;
#pragma unroll
        for (u32 r = 0; r < kRows; ++r) {
            if ((key[r] >> 20) == 0xFFFu && (val[r] >> 20) == 0xFFu) {
                key[r] = (key[r] & 0xFFFFFu) | ck;
                val[r] = (val[r] & 0xFFFFFu) | cv;
            }
        }
    }
#pragma unroll
    for (u32 r = 0; r < kRows; ++r)
        if (lpos[r] != 0xFFFFFFFFu) s_buf[lpos[r]] = key[r];
    __syncthreads();
    SC_T(4);
    u32 dst[kRows];
    u32 hi[kRows]; // (PACK_OUT) the digits above this pass's, already in place for the packed word
    u32 kks[kRows];
#pragma unroll
    for (u32 k = 0; k < kRows; ++k) {
        const u32 i = k * kT + threadIdx.x;
        kks[k] = i < total ? s_buf[i] : 0u;
    }
    lookback_walk();
    __syncthreads();
    SC_T(3);
#pragma unroll
    for (u32 k = 0; k < kRows; ++k) {
        const u32 i = k * kT + threadIdx.x;
        dst[k] = 0xFFFFFFFFu;
        hi[k] = 0;
        if (i < total) {
            const u32 kk = kks[k];
            const u32 dg = (kk >> shift) & (NB - 1);
            dst[k] = s_base[dg] + i;
            if (SRC == SRC_PACKED) st_plain_at(Vout + base, dst[k], kk & 0xFFFFFu); // (the rotation came with the digit: one staging round)
            else if (PACK_OUT) hi[k] = (kk >> (shift + BITS)) << 20;
            else if (WRITE_K) st_plain_at(Kout + base, dst[k], kk);
        }
    }
    SC_T(5);
    if (SRC == SRC_PACKED) return;
    __syncthreads();
#pragma unroll
    for (u32 r = 0; r < kRows; ++r)
        if (lpos[r] != 0xFFFFFFFFu) s_buf[lpos[r]] = val[r];
    __syncthreads();
    SC_T(6);
#pragma unroll
    for (u32 k = 0; k < kRows; ++k) {
        const u32 i = k * kT + threadIdx.x;
        if (i < total) st_plain_at(Vout + base, dst[k], PACK_OUT ? (hi[k] | s_buf[i]) : s_buf[i]);
    }
    SC_T(7);
}

// ---- phase B of the init inside LDS ----------------------------------------------------------------
// After phase A the rotations of a block are ordered by key(j) (symbols [0,c)); phase B has to order every
// group of equal key(j) by key(j+c) (symbols [c,2c)).  Three more passes over the whole list do that through
// HBM (16 bytes per element and pass); here a workgroup takes a SEGMENT of the list -- the groups that start
// inside one tile of kSortTile positions, at most kLocCap elements -- into LDS as 64-bit words
//     group number inside the segment (<= 14 bits) | key(j+c) (<= 30 bits) | j (20 bits),
// sorts the words by the bits above j with stable LSD passes of kLocBits bits that never leave LDS, and
// writes the order out once: 8 bytes read + the gathers of key(j+c) from the block text (L2), 4 written.
//   * ranks inside a wave come from one returning LDS add per element on the u16 counter of the digit; lanes
//     of one instruction that hit the same word are served in ascending lane order on gfx950
//     (tools/ubench/ldsorder.hip).  The result is CHECKED: the words are compared with their neighbours when
//     they are written out, and a block with a pair out of order is marked in a.pb_gate and sorted by the
//     three global passes like a block that does not fit.
//   * a block whose keys are wider than 30 bits, or with a segment of more than kLocCap elements (a group of
//     more than kLocCap - kSortTile equal c-symbol prefixes can do that), is marked the same way: the global
//     passes skip every block that is not marked.
constexpr u32 kLocThreads = 512, kLocWaves = kLocThreads / 64;
constexpr u32 kLocCap = 2 * kSortTile; // two halves of 8 waves x 16 rows; the second one is mostly empty
constexpr u32 kLocSets = 2 * kLocWaves; // counter sets: (half, wave), in list order
constexpr u32 kLocBits = 9, kLocBins = 1u << kLocBits;
enum { LOC_STAT_SEGS = 0, LOC_STAT_OVERFLOW, LOC_STAT_UNSORTED, LOC_STAT_PASSES };

// first position p >= p0 of the list where key(p) != key(p-1) (n if there is none); ~0 when `limit`
// positions were looked at without finding one.  One wave.
__device__ __forceinline__ u32 first_group_start(const u32 *__restrict__ Kb, u32 p0, u32 n, u32 limit)
{
    if (p0 == 0) return 0;
    if (p0 >= n) return n;
    const u32 l = threadIdx.x & 63u;
    for (u32 q0 = p0; q0 < p0 + limit; q0 += 64u) {
        if (q0 >= n) return n;
        const u32 q = q0 + l;
        bool b = false;
        if (q < n) b = Kb[q] != Kb[q - 1];
        const u64 m = __ballot(b);
        if (m) return q0 + (u32)__ffsll((long long)m) - 1u;
    }
    return 0xFFFFFFFFu;
}

// The 16 rows of one wave in one half of the segment: positions pos0 + r * 64 + lane.  (No branch around a row: the
// words live in registers, and a conditional block per row makes the compiler copy the whole array at every join.)
// loc_fill: the words come out of LDS (group-start flag in bit 63) and get their group number.
__device__ __forceinline__ void loc_fill(u64 (&e)[16], const u64 *s_e, u32 pos0, u32 len, u32 l, u64 le_mask, u32 gshift,
                                         u32 &gcarry)
{
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = pos0 + r * 64u + l;
        const u64 x = s_e[i < len ? i : len - 1u];
        const u64 m = __ballot(i < len && (x >> 63) != 0);
        const u32 gi = gcarry + (u32)__popcll(m & le_mask) - 1u; // (position 0 of the segment starts a group)
        e[r] = (x & ~(1ull << 63)) | ((u64)gi << gshift);
        gcarry += (u32)__popcll(m);
    }
}
// loc_rank: stable rank of every word among the words of its wave-half with the same digit
__device__ __forceinline__ void loc_rank(const u64 (&e)[16], u32 (&rnk)[16], u32 *cnt, u32 pos0, u32 len, u32 l, u32 shift)
{
#pragma unroll
    for (u32 q = 0; q < kLocBins / 2 / 64; ++q) cnt[q * 64u + l] = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = pos0 + r * 64u + l;
        const u32 dg = (u32)(e[r] >> shift) & (kLocBins - 1u);
        const u32 sh = (dg & 1u) * 16u;
        u32 old = 0;
        if (i < len)
            old = __hip_atomic_fetch_add(cnt + (dg >> 1), 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        rnk[r] = (old >> sh) & 0xFFFFu;
    }
}
// loc_move: every word to its place in the order by the digit
__device__ __forceinline__ void loc_move(const u64 (&e)[16], const u32 (&rnk)[16], const u32 *cnt, const u16 *s_tpre,
                                         u64 *s_e, u32 pos0, u32 len, u32 l, u32 shift)
{
    const u16 *c16 = reinterpret_cast<const u16 *>(cnt);
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = pos0 + r * 64u + l;
        const u32 dg = (u32)(e[r] >> shift) & (kLocBins - 1u);
        const u32 to = (u32)s_tpre[dg] + (u32)c16[dg] + rnk[r];
        if (i < len) s_e[to] = e[r];
    }
}
__device__ __forceinline__ void loc_reload(u64 (&e)[16], const u64 *s_e, u32 pos0, u32 len, u32 l)
{
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 i = pos0 + r * 64u + l;
        e[r] = s_e[i < len ? i : len - 1u];
    }
}

#define LOC_T(k) do { } while (0)
// ---- a survivor round inside LDS ---------------------------------------------------------------------------------------
// A survivor round orders the compacted list -- the rotations the last refinement left unordered, in the order of their
// groups -- by (group, rank of rotation j + h): two passes by the rank, two by the group, each over the whole list through
// HBM (k_radix_hist / k_radix_scan / k_radix_scatter x 4: 3.7 ms per GiB of text for its two survivor rounds).  But the
// list IS in group order and its groups are small (text: 7 % of the rotations are left behind the first walk round, in
// groups of two and three): a workgroup takes the groups that START inside one tile of the list -- at most kLocCap
// elements -- into LDS as 64-bit words  group number inside the segment | rank of j + h (20 bits) | j (20 bits),  orders
// them with k_phase_b_local's LSD passes that never leave LDS, and writes (group, j) out once.  Same rules as there: the
// result is CHECKED (the ranking inside a wave relies on the order in which LDS serves the lanes of one atomic), and a
// segment that does not fit (a group of more than kLocCap - kSortTile members) or fails the check raises a flag
// (loc_stats[LOC_STAT_SURV_FAIL]): the host then sorts the round's list with the four global passes, as before.
constexpr u32 LOC_STAT_SURV_FAIL = 80;
__device__ __forceinline__ u32 first_group_start_list(const u32 *__restrict__ R, const u32 *__restrict__ Vb, u32 p0, u32 cnt, u32 limit)
{
    if (p0 == 0) return 0;
    if (p0 >= cnt) return cnt;
    const u32 l = threadIdx.x & 63u;
    for (u32 q0 = p0; q0 < p0 + limit; q0 += 64u) {
        if (q0 >= cnt) return cnt;
        const u32 q = q0 + l;
        bool b = false;
        if (q < cnt) b = (R[Vb[q] & 0xFFFFFu] & ~kFinalBit) != (R[Vb[q - 1] & 0xFFFFFu] & ~kFinalBit);
        const u64 m = __ballot(b);
        if (m) return q0 + (u32)__ffsll((long long)m) - 1u;
    }
    return 0xFFFFFFFFu;
}
__global__ __launch_bounds__(kLocThreads) void k_surv_local(BwtArgs a, u32 step, const u32 *__restrict__ V, u32 *__restrict__ Kout,
                                                             u32 *__restrict__ Vout)
{
    __shared__ u64 s_e[kLocCap];
    __shared__ u32 s_cnt[kLocSets][kLocBins / 2]; // u16 counters, two to a word
    __shared__ u16 s_tpre[kLocBins];
    __shared__ u32 s_wsum[kLocSets];
    __shared__ u32 s_seg[2];
    __shared__ u32 s_bad;
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 n = a.blocks[lb].n;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = (((u32)ki.chars * 2u) << step) % n;
    const size_t base = (size_t)lb * kSlot;
    const u32 *Rb = a.R + base, *Vl = V + base;
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    if (w == 0) {
        const u32 s = first_group_start_list(Rb, Vl, start, cnt, kLocCap + 64u);
        if (l == 0) s_seg[0] = s;
    } else if (w == 1) {
        const u32 s = first_group_start_list(Rb, Vl, start + kSortTile, cnt, kLocCap + 64u);
        if (l == 0) s_seg[1] = s;
    }
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const u32 s0 = (u32)__builtin_amdgcn_readfirstlane((int)s_seg[0]), s1 = (u32)__builtin_amdgcn_readfirstlane((int)s_seg[1]);
    if (s0 == 0xFFFFFFFFu || s1 == 0xFFFFFFFFu || (s1 > s0 && s1 - s0 > kLocCap)) {
        if (threadIdx.x == 0) atomicAdd(&a.loc_stats[LOC_STAT_SURV_FAIL], 1u);
        return;
    }
    if (s1 <= s0) return; // no group starts inside this tile
    const u32 len = s1 - s0;
    const u32 *Vb = Vl + s0;
    const u32 wu = (u32)__builtin_amdgcn_readfirstlane((int)w);
    const u32 posA = wu * 1024u, posB = kSortTile + wu * 1024u; // this wave's first position in either half
    // ---- load: j, its group (a new one? bit 63 for now) and the rank of j + h
    u32 gcount[2] = {0, 0};
#pragma unroll 1
    for (u32 half = 0; half < 2; ++half) {
        const u32 pos0 = half ? posB : posA;
        if (pos0 >= len) break;
        u32 prev_last = 0;
        if (l == 0 && pos0 > 0) prev_last = Rb[Vb[pos0 - 1u] & 0xFFFFFu] & ~kFinalBit;
        const u32 rows = (len - pos0 + 63u) / 64u < 16u ? (len - pos0 + 63u) / 64u : 16u;
        u32 tot = 0;
#pragma unroll 4
        for (u32 r = 0; r < rows; ++r) {
            const u32 i = pos0 + r * 64u + l;
            const u32 c = i < len ? i : len - 1u;
            const u32 v = ld_stream(Vb + c) & 0xFFFFFu;
            const u32 g = Rb[v] & ~kFinalBit;
            u32 t = v + hm;
            t = t >= n ? t - n : t;
            const u32 k2 = Rb[t] & ~kFinalBit;
            u32 pg = __shfl_up(g, 1, 64);
            if (l == 0) pg = prev_last;
            prev_last = __shfl(g, 63, 64);
            const bool f = (i < len) && (i == 0 || g != pg);
            if (i < len) s_e[i] = ((u64)(f ? 1u : 0u) << 63) | ((u64)k2 << 20) | (u64)v;
            tot += (u32)__popcll(__ballot(f));
        }
        gcount[half] = tot;
    }
    if (l == 0) {
        s_wsum[w] = gcount[0];
        s_wsum[kLocWaves + w] = gcount[1];
    }
    __syncthreads();
    u32 gcarryA = 0, gcarryB = 0, groups = 0;
    for (u32 k = 0; k < kLocSets; ++k) {
        const u32 c = s_wsum[k];
        if (k < w) gcarryA += c;
        if (k < kLocWaves + w) gcarryB += c;
        groups += c;
    }
    const u64 le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
    constexpr u32 kbits = 20u;
    const u32 gshift = 20u + kbits; // the group number sits right above the rank
    const bool haveB = posB < len;  // (wave-uniform)
    u64 eA[16], eB[16];
    loc_fill(eA, s_e, posA, len, l, le_mask, gshift, gcarryA);
    if (haveB) loc_fill(eB, s_e, posB, len, l, le_mask, gshift, gcarryB);
    const u32 gbits = groups > 1u ? 32u - (u32)__builtin_clz(groups - 1u) : 0u;
    const u32 npass = (kbits + gbits + kLocBits - 1u) / kLocBits;
    u32 *cntA = s_cnt[w], *cntB = s_cnt[kLocWaves + w];
    u16 *cnt16 = reinterpret_cast<u16 *>(&s_cnt[0][0]);
#pragma unroll 1
    for (u32 p = 0; p < npass; ++p) {
        const u32 shift = 20u + p * kLocBits;
        u32 rnkA[16], rnkB[16];
        loc_rank(eA, rnkA, cntA, posA, len, l, shift);
        if (haveB) loc_rank(eB, rnkB, cntB, posB, len, l, shift);
        else {
#pragma unroll
            for (u32 q = 0; q < kLocBins / 2 / 64; ++q) cntB[q * 64u + l] = 0;
        }
        __syncthreads();
        u32 tot;
        {
            const u32 dg = threadIdx.x; // (kLocThreads == kLocBins)
            u32 run = 0;
#pragma unroll
            for (u32 k = 0; k < kLocSets; ++k) {
                const u32 c = cnt16[k * kLocBins + dg];
                cnt16[k * kLocBins + dg] = (u16)run;
                run += c;
            }
            tot = run;
        }
        const u32 inc = wave_incl_sum(tot);
        if (l == 63) s_wsum[w] = inc;
        __syncthreads();
        {
            u32 carry = 0;
            for (u32 k = 0; k < w; ++k) carry += s_wsum[k];
            s_tpre[threadIdx.x] = (u16)(carry + inc - tot);
        }
        __syncthreads();
        loc_move(eA, rnkA, cntA, s_tpre, s_e, posA, len, l, shift);
        if (haveB) loc_move(eB, rnkB, cntB, s_tpre, s_e, posB, len, l, shift);
        __syncthreads();
        if (p + 1 < npass) {
            loc_reload(eA, s_e, posA, len, l);
            if (haveB) loc_reload(eB, s_e, posB, len, l);
        }
    }
    // ---- out: (group, j) in the order found, checked
    {
        bool bad = false;
        for (u32 i = threadIdx.x; i < len; i += kLocThreads) {
            const u64 x = s_e[i];
            if (i && (s_e[i - 1] >> 20) > (x >> 20)) bad = true;
            const u32 v = (u32)x & 0xFFFFFu;
            Vout[base + s0 + i] = v;
            Kout[base + s0 + i] = Rb[v] & ~kFinalBit;
        }
        if (__ballot(bad) && l == 0) s_bad = 1u;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_bad) atomicAdd(&a.loc_stats[LOC_STAT_SURV_FAIL], 1u);
}

// ---- group refinement, part 1: boundary flags over the sorted pair list ----------------
// INIT: the list is all n rotations sorted by their first 2c symbols (one old group); K holds
//       key(j) = symbols [0,c), the second half key(j+c) is re-read from the block (L2).
// else: list sorted by old group head g = K; secondary key = rank of rotation j+h.
// `step` is the doubling step: h = 2c << step.
// `impure` != nullptr (the period round, not INIT): the list is ordered by (group, start of the rotation) and the
// secondary key is the start itself -- every member of a group becomes a group of its own -- unless the group holds
// rotations of different residues modulo the block's period (impure[group head] != 0, k_period_mark) or the block has
// no period: those groups stay as they are.
template <bool INIT>
__global__ __launch_bounds__(kSortThreads) void k_group_flags(BwtArgs a, u32 step, const u32 *__restrict__ K,
                                                               const u32 *__restrict__ V,
                                                               const u8 *__restrict__ impure = nullptr)
{
    __shared__ int s_old, s_new;
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = INIT ? (u32)ki.chars % n : (((u32)ki.chars * 2u) << step) % n;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const size_t base = (size_t)lb * kSlot;
    if (INIT && a.pb_gate[lb] == 0u) K = a.KA; // phase B was done in LDS: phase A's keys are the list's keys
    if (threadIdx.x == 0) {
        s_old = -1;
        s_new = -1;
    }
    __syncthreads();
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    int last_old = -1, last_new = -1;
    const u32 first = start + w * 1024u + l;
    // per element: old-group key g, and the secondary key as one or two words (s1, s2)
    u32 g[16], s1[16], s2[16];
    // row 0 of lane 0 also needs the element just before the wave's range
    u32 pg0 = 0, ps10 = 0, ps20 = 0;
    {
        u32 jj[16];
#pragma unroll
        for (u32 r = 0; r < 16; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            const u32 kk = ld_stream(K + base + c);
            const u32 t = ld_stream(V + base + c) + hm;
            jj[r] = t >= n ? t - n : t;
            g[r] = INIT ? 0u : kk;
            s1[r] = kk;
            s2[r] = 0;
        }
        u32 pj = 0;
        const bool need_prev = (l == 0) && (first > 0) && (first < cnt);
        if (need_prev) {
            const u32 kk = K[base + first - 1];
            pg0 = INIT ? 0u : kk;
            ps10 = kk;
            const u32 t = V[base + first - 1] + hm;
            pj = t >= n ? t - n : t;
        }
        if (INIT) {
            // (a table of the packed keys, one gather instead of three plus the re-coding, was measured: the flags
            // gain 0.6 ms, the walk pass loses 2.5 -- the table is 3.6 MB per block against 0.9 MB of text in the L2)
#pragma unroll
            for (u32 r = 0; r < 16; ++r) s2[r] = pkey(pt, jj[r], ki.bits, ki.chars);
            if (need_prev) ps20 = pkey(pt, pj, ki.bits, ki.chars);
        } else if (impure) {
            const bool per = a.lin_p[(size_t)lb * kPerK] != 0u;
            const u64 d64 = ((u64)ki.chars * 2u) << step;
            const u32 depth = d64 < n ? (u32)d64 : n;
            // every member of a group the round has put in order becomes a group of its own: a chain of the block's
            // period (impure 0 in a block that has one) or a small group ranked by comparison (impure 2); a group of
            // stretches with a short period (impure 1, per_ekey applies) splits into its classes, which the list holds
            // in their true order
#pragma unroll
            for (u32 r = 0; r < 16; ++r) {
                const u32 jr = jj[r] >= hm ? jj[r] - hm : jj[r] + n - hm; // (the rotation itself)
                const u32 im = impure[base + g[r]];
                u32 ek;
                s1[r] = (im == 2u || (per && im == 0u)) ? jr + 1u : ((per && a.per_wide && per_ekey(a, lb, text, n, depth, jr, ek)) ? ek + 1u : 0u);
            }
            if (need_prev) {
                const u32 jr = pj >= hm ? pj - hm : pj + n - hm;
                const u32 im = impure[base + pg0];
                u32 ek;
                ps10 = (im == 2u || (per && im == 0u)) ? jr + 1u : ((per && a.per_wide && per_ekey(a, lb, text, n, depth, jr, ek)) ? ek + 1u : 0u);
            }
        } else {
#pragma unroll
            for (u32 r = 0; r < 16; ++r) s1[r] = a.R[base + jj[r]];
            if (need_prev) ps10 = a.R[base + pj];
        }
    }
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = first + r * 64u;
        const bool ok = idx < cnt;
        u32 pg = wave_shr1(g[r]), p1 = wave_shr1(s1[r]), p2 = wave_shr1(s2[r]);
        // lane 0: the previous element is lane 63 of the previous row (or the pre-loaded one)
        const u32 qg = (r == 0) ? pg0 : wave_lane(g[(r + 15) & 15], 63);
        const u32 q1 = (r == 0) ? ps10 : wave_lane(s1[(r + 15) & 15], 63);
        const u32 q2 = (r == 0) ? ps20 : wave_lane(s2[(r + 15) & 15], 63);
        if (l == 0) {
            pg = qg;
            p1 = q1;
            p2 = q2;
        }
        bool ns = false;
        if (ok) {
            const bool os = (idx == 0) || (g[r] != pg);
            ns = os || (s1[r] != p1) || (s2[r] != p2);
            a.flags[base + idx] = (u8)((os ? 1u : 0u) | (ns ? 2u : 0u));
            if (os) last_old = (int)idx;
            if (ns) last_new = (int)idx;
        }
        const u64 mrow = __ballot(ns);
        if (l == 0 && idx < cnt) a.newbits[(base + idx) >> 6] = mrow;
    }
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) {
        const int o1 = __shfl_xor(last_old, dd, 64), o2 = __shfl_xor(last_new, dd, 64);
        last_old = o1 > last_old ? o1 : last_old;
        last_new = o2 > last_new ? o2 : last_new;
    }
    if (l == 0) {
        if (last_old >= 0) atomicMax(&s_old, last_old);
        if (last_new >= 0) atomicMax(&s_new, last_new);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_last_old[lb * kTilesPerBlock + tile] = s_old;
        a.tile_last_new[lb * kTilesPerBlock + tile] = s_new;
    }
}

// ---- group refinement, part 2: positions, new ranks, final bits ---------------------------
// The new rank of rotation j is NOT written to R[j] from here: 4-byte stores at random into the 3.6 MB
// rank array of a level-9 block run at 80-100 G/s (every store dirties a sector of its own; the lines
// leave the XCD's L2 before they are full).  Instead the ranks are BINNED by j >> 10: a tile counts
// its elements per bin in LDS, draws room for each bin from the block's bin cursors (one global
// atomic per tile and non-empty bin -- the order inside a bin does not matter, so no look-back is
// needed), re-orders the tile through LDS and writes runs of consecutive words; k_rank_place then
// takes one bin (1024 ranks) per wave, puts the words in place in LDS and writes the rank array in
// whole lines.  A word is (j & 1023) | head << 10 | final << 31; bin b owns the 1024 words behind
// b * 1024 of the free key array (a rotation occurs once, so a bin cannot overflow).
constexpr u32 kRankBinShift = 10, kRankBins = kSlot >> kRankBinShift; // 880 bins of 1024 rotations
constexpr u32 kRankBinSize = 1u << kRankBinShift, kRankBinMask = kRankBinSize - 1u;
static_assert(kRankBinShift + 20u + 1u <= 32u, "j inside the bin, a 20-bit head and the final bit share a word");
static_assert(kRankBins <= 1024 && (kRankBins << kRankBinShift) == kSlot, "bins tile the slot");

template <bool INIT>
__global__ __launch_bounds__(kSortThreads) void k_group_apply(BwtArgs a, u32 next_step, u32 round,
                                                               const u32 *__restrict__ K,
                                                               const u32 *__restrict__ V,
                                                               u32 *__restrict__ W)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ int s_carry_old, s_carry_new;
    __shared__ int s_wold[NW], s_wnew[NW];
    __shared__ u32 s_nonfinal;
    // digit counts of the keys of the next walk round (the group heads of the members that are not
    // final yet): what its fused radix passes need instead of histogram kernels
    __shared__ u32 s_gh[2][1024];
    // the bin scatter
    __shared__ u32 s_stage[kSortTile];
    __shared__ u16 s_binof[kSortTile];
    __shared__ u32 s_bcnt[1024];  // elements of this tile per bin, then the bin's first output word
    __shared__ u16 s_bpre[1024];  // exclusive prefix of the counts inside the tile
    __shared__ u32 s_wsum[NW];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const size_t base = (size_t)lb * kSlot;
    if (a.fused)
        for (u32 i = threadIdx.x; i < 2048u; i += kSortThreads) (&s_gh[0][0])[i] = 0;
    for (u32 i = threadIdx.x; i < 1024u; i += kSortThreads) s_bcnt[i] = 0;

    if (threadIdx.x == 0) {
        int co = -1, cn = -1;
        for (int t = (int)tile - 1; t >= 0 && co < 0; --t) co = a.tile_last_old[lb * kTilesPerBlock + t];
        for (int t = (int)tile - 1; t >= 0 && cn < 0; --t) cn = a.tile_last_new[lb * kTilesPerBlock + t];
        s_carry_old = co;
        s_carry_new = cn;
        s_nonfinal = 0;
    }
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 wbase = start + w * 1024u;
    // the pair list is loaded up front (16 rows in flight), the flag bytes likewise
    u32 gk[16], jv[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = wbase + r * 64u + l;
        const u32 c = idx < cnt ? idx : cnt - 1u;
        gk[r] = INIT ? 0u : ld_stream(K + base + c);
        jv[r] = ld_stream(V + base + c);
    }
    u64 mo[16], mn[16];
    int wl_old = -1, wl_new = -1;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = wbase + r * 64u + l;
        const u32 f = (idx < cnt) ? ld_stream(a.flags + base + idx) : 0u;
        mo[r] = __ballot(f & 1u);
        mn[r] = __ballot(f & 2u);
        if (mo[r]) wl_old = (int)(wbase + r * 64u + 63u - __clzll(mo[r]));
        if (mn[r]) wl_new = (int)(wbase + r * 64u + 63u - __clzll(mn[r]));
    }
    if (l == 0) {
        s_wold[w] = wl_old;
        s_wnew[w] = wl_new;
    }
    __syncthreads();
    int carry_old = s_carry_old, carry_new = s_carry_new;
    for (u32 k = 0; k < w; ++k) {
        carry_old = s_wold[k] > carry_old ? s_wold[k] : carry_old;
        carry_new = s_wnew[k] > carry_new ? s_wnew[k] : carry_new;
    }
    const u64 le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
    u32 my_nonfinal = 0;
    u32 word[16], lrank[16]; // the rank word of each element and its number inside (tile, bin); ~0: none
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 rowbase = wbase + r * 64u;
        const u32 idx = rowbase + l;
        lrank[r] = 0xFFFFFFFFu;
        word[r] = 0;
        bool nf = false;  // this lane's element stays unordered: it is counted for the next walk round
        u32 nf_head = 0;
        if (idx < cnt) {
            const u64 o = mo[r] & le_mask, q = mn[r] & le_mask;
            const int rs = o ? (int)(rowbase + 63u - __clzll(o)) : carry_old;
            const int ss = q ? (int)(rowbase + 63u - __clzll(q)) : carry_new;
            // is the next list element the start of a new group?
            bool next_new;
            if (idx + 1 >= cnt) next_new = true;
            else if (l < 63) next_new = (mn[r] >> (l + 1)) & 1ull;
            else if (r < 15) next_new = mn[(r + 1) & 15] & 1ull;
            else next_new = (a.flags[base + idx + 1] & 2u) != 0;
            const bool is_new = (mn[r] >> l) & 1ull;
            const bool fin = is_new && next_new;
            const u32 g = gk[r];
            const u32 j = jv[r];
            const u32 p = g + (idx - (u32)rs);
            const u32 head = g + ((u32)ss - (u32)rs);
            if (!INIT) st_stream(a.SA + base + p, j); // (INIT: the list IS SA, p == idx)
            if (fin) {
                // a rotation becomes final exactly once, here (or, in a periodic block, in k_periodic_place):
                // its byte of the last column goes out now -- L[i] = block[(SA[i]-1) mod n], origPtr = i where
                // SA[i] == 0 (src/bzip2/encoder.rs:331-338) -- instead of a pass over SA at the end
                a.L[base + p] = text[j ? j - 1u : n - 1u];
                if (j == 0) a.orig_ptr[lb] = p;
            }
            word[r] = (j & kRankBinMask) | (head << kRankBinShift) | (fin ? kFinalBit : 0u);
            lrank[r] = atomicAdd(&s_bcnt[j >> kRankBinShift], 1u);
            my_nonfinal += fin ? 0u : 1u;
            nf = !fin;
            nf_head = head;
        }
        if (a.fused) {
            // Group heads rise along the list, so the 64 heads of a row share their high digit (almost always):
            // one lane adds the count for all of them -- 64 single adds to one LDS word serialise.
            const u64 mnf = __ballot(nf);
            if (mnf) {
                if (nf) atomicAdd(&s_gh[0][nf_head & 1023u], 1u);
                const u32 lead = (u32)__ffsll((long long)mnf) - 1u;
                const u32 hi = (nf_head >> 10) & 1023u;
                const u32 hi0 = wave_lane(hi, lead); // (lead: first set bit of a ballot, the same in every lane)
                const u64 same = __ballot(nf && hi == hi0);
                if (same == mnf) {
                    if (l == lead) atomicAdd(&s_gh[1][hi0], (u32)__popcll(mnf));
                } else if (nf) {
                    atomicAdd(&s_gh[1][hi], 1u);
                }
            }
        }
        if (mo[r]) carry_old = (int)(rowbase + 63u - __clzll(mo[r]));
        if (mn[r]) carry_new = (int)(rowbase + 63u - __clzll(mn[r]));
    }
    my_nonfinal = wave_sum(my_nonfinal);
    if (l == 0 && my_nonfinal) atomicAdd(&s_nonfinal, my_nonfinal);
    __syncthreads();
    if (threadIdx.x == 0) a.tile_nf[lb * kTilesPerBlock + tile] = s_nonfinal;
    if (a.fused) {
        u32 *out = a.gh_tiles + ((size_t)lb * kTilesPerBlock + tile) * 3 * kMaxBins;
        for (u32 i = threadIdx.x; i < 1024u; i += kSortThreads) {
            out[i] = s_gh[0][i];
            out[kMaxBins + i] = s_gh[1][i];
        }
    }
    if (threadIdx.x == 0 && s_nonfinal) {
        // (the largest of these sums over a block's tiles is the block's total: the host sizes the next round's
        // launches by it)
        atomicMax(&a.maxnf[round], atomicAdd(&a.nonfinal[lb], s_nonfinal) + s_nonfinal);
        // the block still needs rounds only while the next comparison depth is below its length
        const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
        const u32 h_next = ((u32)ki.chars * 2u) << next_step;
        if (h_next < n) atomicAdd(&a.active[round], (unsigned long long)s_nonfinal);
    }
    // ---- bin scatter of the rank words
    // exclusive prefix of the bin counts inside the tile (two consecutive bins per thread), and room
    // for every non-empty bin from the block's cursors
    {
        const u32 b0 = threadIdx.x * 2u;
        const u32 c0 = s_bcnt[b0], c1 = s_bcnt[b0 + 1];
        const u32 mine = c0 + c1;
        const u32 inc = wave_incl_sum(mine);
        if (l == 63) s_wsum[w] = inc;
        u32 g0 = 0, g1 = 0;
        u32 *cur = a.bin_cursor + (size_t)lb * 1024u;
        if (c0) g0 = atomicAdd(&cur[b0], c0);
        if (c1) g1 = atomicAdd(&cur[b0 + 1], c1);
        __syncthreads();
        u32 carry = 0;
        for (u32 k = 0; k < w; ++k) carry += s_wsum[k];
        const u32 ex = carry + inc - mine;
        s_bpre[b0] = (u16)ex;
        s_bpre[b0 + 1] = (u16)(ex + c0);
        s_bcnt[b0] = (b0 << kRankBinShift) + g0;
        s_bcnt[b0 + 1] = ((b0 + 1u) << kRankBinShift) + g1;
    }
    __syncthreads();
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (lrank[r] != 0xFFFFFFFFu) {
            const u32 bin = jv[r] >> kRankBinShift;
            const u32 pos = (u32)s_bpre[bin] + lrank[r];
            s_stage[pos] = word[r];
            s_binof[pos] = (u16)bin;
        }
    }
    __syncthreads();
    const u32 total = (cnt - start) < kSortTile ? (cnt - start) : kSortTile;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        if (i < total) {
            const u32 bin = s_binof[i];
            W[base + s_bcnt[bin] + (i - (u32)s_bpre[bin])] = s_stage[i];
        }
    }
}

// ---- group refinement in ONE pass: flags + positions + rank words ----------------------------------
// k_group_flags and k_group_apply read the same pair list twice (and hand the flag bytes and two tile carries
// through memory in between).  Here a tile computes its boundary flags (the gathers of k_group_flags), keeps them as
// wave ballots, and goes straight on with the work of k_group_apply; the two things a tile needs from the tiles in
// front of it -- the last old-group start and the last new-group start before its first element -- come by the same
// decoupled look-back the fused radix passes use (tickets per XCD, epoch-tagged words in tile_state, XCD-local L2):
// a tile publishes the last starts it holds itself (AGG; "none" = 0xFFFFF) as soon as its flags are known, looks back
// until both are settled (almost always one hop: a tile without a group start is rare), and publishes the settled
// pair (INCL).  The one flag a tile needs from BEHIND itself -- does the element after a wave's span start a new
// group -- is computed from that element directly (one extra gather per wave).  The new-group starts leave as a
// bitmap (one ballot per row): k_survivor_compact and the walk pass of the next round read it.
//   INIT : the first refinement (the list is SA in the order of the first 2c symbols; the secondary key is gathered
//          from the packed text).  Its settled last new-group start per tile is kept in tile_last_new.
//   else : the refinement of the walk round that follows it.  The secondary key -- the rank of rotation j+h -- came
//          along in the spare bits of key and value (k_radix_scatter_lb<SRC_MMC>): no gather at all.
// Later rounds keep k_group_flags + k_group_apply: with few elements the 131 k tickets of a launch cost more than
// the second read of the list (1.5 ms against 0.15), and a walk round further on has no flags in SA order to carry.
// Used when the fused radix passes are (a.fused); BZ_FUSED_REFINE=0 keeps the two kernels everywhere.
constexpr u32 kLbNone = 0xFFFFFu;
// -DBZ_REFINE_TIMING: cycles (>> 4) per phase of a tile, summed over tiles into loc_stats[16 + 8 * INIT + k]
// (a barrier at every mark; the host prints them at the end of a sort): 0 ticket + set-up, 1 list + secondary keys, 2 flags,
// 3 look-back, 4 positions / rank words / last column, 5 counts out + bin offsets, 6 staging + rank words out
#ifdef BZ_REFINE_TIMING
#define RF_T(k) do { __syncthreads(); const u64 t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&a.loc_stats[16 + (INIT ? 8 : 0) + (k)], (u32)((t_ - t_prev) >> 4)); t_prev = t_; } while (0)
#else
#define RF_T(k) do { } while (0)
#endif
template <bool INIT>
__global__ __launch_bounds__(kSortThreads, 4) void k_group_refine(BwtArgs a, u32 step, u32 next_step, u32 round,
                                                                const u32 *__restrict__ K,
                                                                const u32 *__restrict__ V, u32 *__restrict__ W,
                                                                u32 epoch)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ int s_carry_old, s_carry_new;
    __shared__ int s_wold[NW], s_wnew[NW];
    __shared__ u32 s_nonfinal, s_ticket;
    // LDS: 53.4 KB, so that THREE workgroups share a CU (rounds 1-3: 62.5 KB, two).  The staging buffer of the last phase
    // holds, until that phase begins, the counters of the phases before it -- the digit counts of the next walk round
    // (s_gh) and the per-bin counts of the rank words (s_bcnt) -- and what the last phase needs of them is moved to two
    // small arrays (s_bpre, s_bdst) in front of the barrier that ends their life.
    __shared__ u32 s_stage[kSortTile];
    __shared__ u16 s_binof[kSortTile];
    __shared__ u16 s_bpre[1024];
    __shared__ u16 s_bdst[1024]; // a bin's cursor inside its 1024-word region (the region's start follows from the bin)
    __shared__ u32 s_wsum[NW];
    u32 (*s_gh)[1024] = reinterpret_cast<u32 (*)[1024]>(s_stage);          // [2][1024], words 0 .. 2047
    u32 *s_bcnt = s_stage + 2048;                                          // [1024],    words 2048 .. 3071
#ifdef BZ_REFINE_TIMING
    u64 t_prev = __builtin_readcyclecounter();
#endif
    const u32 xcd = (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u; // XCC_ID (see k_radix_scatter_lb)
    const u32 my_tiles = a.tiles * ((a.nb + 7u - xcd) / 8u); // (a.tiles: tiles per block the launch covers)
    if (threadIdx.x == 0) s_ticket = atomicAdd(&a.tickets[(size_t)epoch * 8u + xcd], 1u);
    __syncthreads();
    const u32 slot = s_ticket;
    if (slot >= my_tiles) return;
    const u32 b8 = slot / a.tiles;
    const u32 tile = slot - b8 * a.tiles;
    const u32 lb = b8 * 8u + xcd;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    if (start >= cnt) return;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u32 hm = INIT ? (u32)ki.chars % n : (((u32)ki.chars * 2u) << step) % n;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    const size_t base = (size_t)lb * kSlot;
    if (INIT && a.pb_gate[lb] == 0u) K = a.KA; // phase B was done in LDS: phase A's keys are the list's keys
    for (u32 i = threadIdx.x; i < 3072u; i += kSortThreads) s_stage[i] = 0; // (s_gh and s_bcnt)
    if (threadIdx.x == 0) s_nonfinal = 0;
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 wbase = start + w * 1024u;
    const u32 first = wbase + l;

    RF_T(0);
    // ---- the pair list and the secondary keys (k_group_flags)
    u32 gk[16], jv[16], s1[16], s2[16];
    u32 pg0 = 0, ps10 = 0, ps20 = 0; // the element in front of the wave's span (lane 0)
    u32 ng0 = 0, ns10 = 0, ns20 = 0; // the element behind it (lane 63)
    const bool need_prev = (l == 0) && (first > 0) && (first < cnt);
    const bool need_next = (l == 63) && (wbase + 1024u < cnt);
    {
#pragma unroll
        for (u32 r = 0; r < 16; ++r) {
            const u32 idx = first + r * 64u;
            const u32 c = idx < cnt ? idx : cnt - 1u;
            const u32 kk = ld_stream(K + base + c);
            const u32 vv = ld_stream(V + base + c);
            if (INIT) {
                jv[r] = vv;
                const u32 t = vv + hm;
                s2[r] = t >= n ? t - n : t; // (the position of the secondary key, for now)
                gk[r] = 0u;
                s1[r] = kk;
            } else { // key = group | rank bits 0..11, value = rotation | rank bits 12..19 (k_radix_scatter_lb<SRC_MMC>)
                jv[r] = vv & 0xFFFFFu;
                gk[r] = kk & 0xFFFFFu;
                s1[r] = (kk >> 20) | ((vv >> 20) << 12);
                s2[r] = 0u;
            }
        }
        u32 pj = 0, nj = 0;
        if (need_prev) {
            const u32 kk = K[base + wbase - 1], vv = V[base + wbase - 1];
            pg0 = INIT ? 0u : (kk & 0xFFFFFu);
            ps10 = INIT ? kk : ((kk >> 20) | ((vv >> 20) << 12));
            const u32 t = vv + hm;
            pj = t >= n ? t - n : t;
        }
        if (need_next) {
            const u32 kk = K[base + wbase + 1024u], vv = V[base + wbase + 1024u];
            ng0 = INIT ? 0u : (kk & 0xFFFFFu);
            ns10 = INIT ? kk : ((kk >> 20) | ((vv >> 20) << 12));
            const u32 t = vv + hm;
            nj = t >= n ? t - n : t;
        }
        if (INIT) {
#pragma unroll
            for (u32 r = 0; r < 16; ++r) s2[r] = pkey(pt, s2[r], ki.bits, ki.chars);
            if (need_prev) ps20 = pkey(pt, pj, ki.bits, ki.chars);
            if (need_next) ns20 = pkey(pt, nj, ki.bits, ki.chars);
        }
    }
    RF_T(1);
    // ---- boundary flags as ballots; the new-group starts go out as a bitmap
    u64 mo[16] = {}, mn[16];
    int wl_old = -1, wl_new = -1;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = first + r * 64u;
        const bool ok = idx < cnt;
        u32 pg = wave_shr1(gk[r]), p1 = wave_shr1(s1[r]), p2 = wave_shr1(s2[r]);
        const u32 qg = (r == 0) ? pg0 : wave_lane(gk[(r + 15) & 15], 63);
        const u32 q1 = (r == 0) ? ps10 : wave_lane(s1[(r + 15) & 15], 63);
        const u32 q2 = (r == 0) ? ps20 : wave_lane(s2[(r + 15) & 15], 63);
        if (l == 0) {
            pg = qg;
            p1 = q1;
            p2 = q2;
        }
        const bool os = ok && ((idx == 0) || (!INIT && gk[r] != pg));
        const bool ns = ok && (os || (s1[r] != p1) || (s2[r] != p2));
        mn[r] = __ballot(ns);
        if (l == 0 && ok) a.newbits[(base + idx) >> 6] = mn[r];
        if (mn[r]) wl_new = (int)(wbase + r * 64u + 63u - __clzll(mn[r]));
        if (!INIT) { // (INIT: the whole list is one old group that starts at 0)
            mo[r] = __ballot(os);
            if (mo[r]) wl_old = (int)(wbase + r * 64u + 63u - __clzll(mo[r]));
        }
    }
    // does the element behind the wave's span start a new group?  (lane 63 holds both sides)
    bool next_after = true;
    if (need_next) next_after = (gk[15] != ng0) || (s1[15] != ns10) || (s2[15] != ns20);
    next_after = wave_lane((u32)next_after, 63) != 0u;
    if (l == 0) {
        s_wold[w] = INIT ? 0 : wl_old;
        s_wnew[w] = wl_new;
    }
    // Which rotations become final is known from the flags alone (a new group whose successor starts one too): their
    // bytes of the last column are gathered now, all rows in flight, while the look-back below is under way.
    u32 finbits = 0;
    u32 lcol[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = first + r * 64u;
        bool next_new;
        if (idx + 1 >= cnt) next_new = true;
        else if (l < 63) next_new = (mn[r] >> (l + 1)) & 1ull;
        else if (r < 15) next_new = mn[(r + 1) & 15] & 1ull;
        else next_new = next_after;
        const bool fin = (idx < cnt) && ((mn[r] >> l) & 1ull) && next_new;
        finbits |= (fin ? 1u : 0u) << r;
        lcol[r] = 0;
        // (the walk round's refinement has no registers to spare for sixteen bytes in flight: it gathers in its loop)
        if (INIT && fin) lcol[r] = text[jv[r] ? jv[r] - 1u : n - 1u]; // src/bzip2/encoder.rs:331-338
    }
    __syncthreads();
    RF_T(2);
    // ---- the carries from the tiles in front: decoupled look-back
    if (threadIdx.x == 0) {
        int lo = INIT ? 0 : -1, ln = -1; // (INIT: one old group, it starts at 0)
        for (u32 k = 0; k < NW; ++k) {
            lo = s_wold[k] > lo ? s_wold[k] : lo;
            ln = s_wnew[k] > ln ? s_wnew[k] : ln;
        }
        u32 *mystate = a.tile_state + ((size_t)lb * kTilesPerBlock + tile) * kMaxBins;
        const u32 etag = epoch << 22;
        const u32 f0 = etag | (tile ? kLbAgg : kLbIncl);
        const u32 vo = lo >= 0 ? (u32)lo : kLbNone, vn = ln >= 0 ? (u32)ln : kLbNone;
        st_sc1_x4(mystate, make_uint4(f0 | vo, f0 | vn, f0, f0));
        int co = -1, cn = -1;
        if (tile) {
            u32 spins = 0;
            for (u32 p = tile; p > 0 && (co < 0 || cn < 0);) {
                --p;
                const u32 *src = a.tile_state + ((size_t)lb * kTilesPerBlock + p) * kMaxBins;
                uint4 v = ld_sc1_x4(src);
                while ((v.x >> 22) != epoch || (v.x & kLbFlagMask) == 0u || (v.y >> 22) != epoch || (v.y & kLbFlagMask) == 0u) {
                    if (lb_give_up(spins, a.sort_err, kLbSpinMax >> 3)) { // (a predecessor publishes some tens of microseconds after its ticket)
                        atomicExch(a.sort_err, 1u);
                        v = make_uint4(etag | kLbIncl, etag | kLbIncl, 0, 0); // (give up: the host redoes the sort)
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    v = ld_sc1_x4(src);
                }
                const u32 xo = v.x & kLbValMask, xn = v.y & kLbValMask;
                if (co < 0 && xo != kLbNone) co = (int)xo;
                if (cn < 0 && xn != kLbNone) cn = (int)xn;
            }
            const u32 fi = etag | kLbIncl;
            const u32 io = lo >= 0 ? (u32)lo : (u32)co, in = ln >= 0 ? (u32)ln : (u32)cn;
            st_sc1_x4(mystate, make_uint4(fi | (io & kLbValMask), fi | (in & kLbValMask), fi, fi));
        }
        // (the walk pass of the next round starts its tiles from these)
        a.tile_last_new[lb * kTilesPerBlock + tile] = ln >= 0 ? ln : cn;
        a.tile_last_old[lb * kTilesPerBlock + tile] = lo >= 0 ? lo : co;
        s_carry_old = co;
        s_carry_new = cn;
    }
    __syncthreads();
    RF_T(3);
    int carry_old = INIT ? 0 : s_carry_old, carry_new = s_carry_new;
    for (u32 k = 0; k < w; ++k) {
        if (!INIT) carry_old = s_wold[k] > carry_old ? s_wold[k] : carry_old;
        carry_new = s_wnew[k] > carry_new ? s_wnew[k] : carry_new;
    }
    // ---- positions, rank words, last column (k_group_apply)
    const u64 le_mask = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
    u32 my_nonfinal = 0; // (wave-uniform)
    u32 word[16], lrank[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 rowbase = wbase + r * 64u;
        const u32 idx = rowbase + l;
        lrank[r] = 0xFFFFFFFFu;
        word[r] = 0;
        bool nf = false;
        u32 nf_head = 0;
        if (idx < cnt) {
            int rs = 0;
            if (!INIT) {
                const u64 o = mo[r] & le_mask;
                rs = o ? (int)(rowbase + 63u - __clzll(o)) : carry_old;
            }
            const u64 q = mn[r] & le_mask;
            const int ss = q ? (int)(rowbase + 63u - __clzll(q)) : carry_new;
            const bool fin = (finbits >> r) & 1u;
            const u32 g = gk[r];
            const u32 j = jv[r];
            const u32 p = g + (idx - (u32)rs);
            const u32 head = g + ((u32)ss - (u32)rs);
            if (!INIT) st_stream(a.SA + base + p, j); // (INIT: the list IS SA, p == idx)
            if (fin) {
                a.L[base + p] = INIT ? (u8)lcol[r] : text[j ? j - 1u : n - 1u];
                if (j == 0) a.orig_ptr[lb] = p;
            }
            word[r] = (j & kRankBinMask) | (head << kRankBinShift) | (fin ? kFinalBit : 0u);
            lrank[r] = atomicAdd(&s_bcnt[j >> kRankBinShift], 1u);
            nf = !fin;
            nf_head = head;
        }
        {
            const u64 mnf = __ballot(nf);
            if (mnf) {
                my_nonfinal += (u32)__popcll(mnf);
                // low digit of the heads: the members of a group are neighbours, so the first member inside the row
                // adds the length of the row's stretch of the group (one LDS add per group and row, not per member)
                // (a stretch begins at a new-group start or at lane 0; the starts above a lane are mn[r]'s bits above it)
                if (nf && (l == 0 || ((mn[r] >> l) & 1ull))) {
                    const u64 rest = (l == 63) ? 0ull : (mn[r] >> (l + 1u));
                    u32 len = rest ? (u32)__ffsll((long long)rest) : 64u - l;
                    len = len < cnt - idx ? len : cnt - idx;
                    atomicAdd(&s_gh[0][nf_head & 1023u], len);
                }
                const u32 lead = (u32)__ffsll((long long)mnf) - 1u;
                const u32 hi = (nf_head >> 10) & 1023u;
                const u32 hi0 = wave_lane(hi, lead); // (lead: first set bit of a ballot, the same in every lane)
                const u64 same = __ballot(nf && hi == hi0);
                if (same == mnf) {
                    if (l == lead) atomicAdd(&s_gh[1][hi0], (u32)__popcll(mnf));
                } else if (nf) {
                    atomicAdd(&s_gh[1][hi], 1u);
                }
            }
        }
        if (!INIT && mo[r]) carry_old = (int)(rowbase + 63u - __clzll(mo[r]));
        if (mn[r]) carry_new = (int)(rowbase + 63u - __clzll(mn[r]));
    }
    if (l == 0 && my_nonfinal) atomicAdd(&s_nonfinal, my_nonfinal);
    __syncthreads();
    RF_T(4);
    if (threadIdx.x == 0) a.tile_nf[lb * kTilesPerBlock + tile] = s_nonfinal;
    {
        u32 *out = a.gh_tiles + ((size_t)lb * kTilesPerBlock + tile) * 3 * kMaxBins;
        for (u32 i = threadIdx.x; i < 1024u; i += kSortThreads) {
            out[i] = s_gh[0][i];
            out[kMaxBins + i] = s_gh[1][i];
        }
    }
    if (threadIdx.x == 0 && s_nonfinal) {
        // (the largest of these sums over a block's tiles is the block's total: the host sizes the next round's
        // launches by it)
        atomicMax(&a.maxnf[round], atomicAdd(&a.nonfinal[lb], s_nonfinal) + s_nonfinal);
        const u32 h_next = ((u32)ki.chars * 2u) << next_step;
        if (h_next < n) atomicAdd(&a.active[round], (unsigned long long)s_nonfinal);
    }
    // ---- bin scatter of the rank words (as in k_group_apply)
    {
        const u32 b0 = threadIdx.x * 2u;
        const u32 c0 = s_bcnt[b0], c1 = s_bcnt[b0 + 1];
        const u32 mine = c0 + c1;
        const u32 inc = wave_incl_sum(mine);
        if (l == 63) s_wsum[w] = inc;
        u32 g0 = 0, g1 = 0;
        u32 *cur = a.bin_cursor + (size_t)lb * 1024u;
        if (c0) g0 = atomicAdd(&cur[b0], c0);
        if (c1) g1 = atomicAdd(&cur[b0 + 1], c1);
        __syncthreads();
        u32 carry = 0;
        for (u32 k = 0; k < w; ++k) carry += s_wsum[k];
        const u32 ex = carry + inc - mine;
        s_bpre[b0] = (u16)ex;
        s_bpre[b0 + 1] = (u16)(ex + c0);
        s_bdst[b0] = (u16)g0; // (< 1024: a rotation occurs once, a bin never holds more than its 1024 words)
        s_bdst[b0 + 1] = (u16)g1;
    }
    __syncthreads(); // (s_gh and s_bcnt are dead from here on: their words become the staging buffer)
    RF_T(5);
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (lrank[r] != 0xFFFFFFFFu) {
            const u32 bin = jv[r] >> kRankBinShift;
            const u32 pos = (u32)s_bpre[bin] + lrank[r];
            s_stage[pos] = word[r];
            s_binof[pos] = (u16)bin;
        }
    }
    __syncthreads();
    const u32 total = (cnt - start) < kSortTile ? (cnt - start) : kSortTile;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        if (i < total) {
            const u32 bin = s_binof[i];
            W[base + (bin << kRankBinShift) + (u32)s_bdst[bin] + (i - (u32)s_bpre[bin])] = s_stage[i];
        }
    }
    RF_T(6);
}

// One wave per bin: the rank words of bin b (a.bin_cursor[b] of them, behind W[b * 1024]) go to
// R[b * 1024 + (word & 1023)].  A full bin (every round-0 bin but the last) is put in order in LDS and
// written as whole lines; a partly filled one updates the lines it is loaded into; a sparse one
// (later rounds) is scattered directly.
__global__ __launch_bounds__(kSortThreads) void k_rank_place(BwtArgs a, const u32 *__restrict__ W)
{
    __shared__ u32 s_r[kSortThreads / 64][kRankBinSize];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 n = a.blocks[lb].n;
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 bin = tile * (kSortThreads / 64) + w;
    const u32 j0 = bin << kRankBinShift;
    if (j0 >= n) return;
    const u32 c = a.bin_cursor[(size_t)lb * 1024u + bin];
    if (c == 0) return;
    const size_t base = (size_t)lb * kSlot + j0;
    const u32 span = (n - j0) < kRankBinSize ? (n - j0) : kRankBinSize;
    u32 *mine = s_r[w];
    if (c < 128u) {
        for (u32 i = l; i < c; i += 64u) {
            const u32 x = ld_stream(W + base + i);
            a.R[base + (x & kRankBinMask)] = ((x >> kRankBinShift) & 0xFFFFFu) | (x & kFinalBit);
        }
        return;
    }
    // (LDS accesses of one wave retire in order: no barrier between the phases of a wave's own region)
    if (c == kRankBinSize) {
        // a full bin (every bin of the first refinement but a block's last): 16 bytes per load and store
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *src = reinterpret_cast<const u32x4 *>(W + base);
        uint4 *dst = reinterpret_cast<uint4 *>(a.R + base);
        u32x4 q[kRankBinSize / 256];
#pragma unroll
        for (u32 k = 0; k < kRankBinSize / 256; ++k) q[k] = __builtin_nontemporal_load(src + k * 64u + l);
#pragma unroll
        for (u32 k = 0; k < kRankBinSize / 256; ++k) {
            const u32 x[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (u32 e = 0; e < 4; ++e) mine[x[e] & kRankBinMask] = ((x[e] >> kRankBinShift) & 0xFFFFFu) | (x[e] & kFinalBit);
        }
#pragma unroll
        for (u32 k = 0; k < kRankBinSize / 256; ++k) dst[k * 64u + l] = reinterpret_cast<const uint4 *>(mine)[k * 64u + l];
        return;
    }
    if (c < span)
        for (u32 i = l; i < span; i += 64u) mine[i] = a.R[base + i];
    for (u32 i = l; i < c; i += 64u) {
        const u32 x = ld_stream(W + base + i);
        mine[x & kRankBinMask] = ((x >> kRankBinShift) & 0xFFFFFu) | (x & kFinalBit);
    }
    for (u32 i = l; i < span; i += 64u) a.R[base + i] = mine[i];
}

// ---- survivors of a round, compacted in list order ------------------------------------------------
// The list (V, flags) a round has just refined -> the rotations that are still not final, in the
// same order (i.e. sorted by their new group head).  Used when few survive: the next round then
// sorts the survivors explicitly instead of walking all of SA.
// midx != nullptr (the period round and the link rounds): every member of a SMALL group of survivors (2 .. kLinkMax members:
// the place inside the group and its size are read off the start bits the compaction reads anyway) leaves
// midx[its start] = its place in the compacted list | place inside the group << 20 | members << 24 (the host cleared the
// array: 0 = not such a member) and its group byte.
// dense_only (the period round): only in blocks whose list holds a quarter of the block and more -- k_link_scan walks a block
// in text order, which costs a text block with 2 % of its rotations left more than the doubling round it saves.
__global__ __launch_bounds__(kSortThreads) void k_survivor_compact(BwtArgs a, const u32 *__restrict__ V,
                                                                    u32 *__restrict__ VS, u32 *__restrict__ midx = nullptr, u32 dense_only = 0)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ u32 s_off, s_wsum[NW], s_total;
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    const u32 ntiles = (cnt + kSortTile - 1) / kSortTile;
    if (tile == 0 && threadIdx.x == 0) {
        u32 tot = 0;
        for (u32 t = 0; t < ntiles; ++t) tot += a.tile_nf[lb * kTilesPerBlock + t];
        a.count2[lb] = tot;
    }
    if (start >= cnt) return;
    if (a.tile_nf[lb * kTilesPerBlock + tile] == 0) return;
    const size_t base = (size_t)lb * kSlot;
    if (threadIdx.x == 0) {
        u32 off = 0, tot = 0;
        for (u32 t = 0; t < tile; ++t) off += a.tile_nf[lb * kTilesPerBlock + t];
        if (midx && dense_only)
            for (u32 t = 0; t < ntiles; ++t) tot += a.tile_nf[lb * kTilesPerBlock + t];
        s_off = off;
        s_total = tot;
    }
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 wbase = start + w * 1024u;
    u64 surv[16];
    u32 jv[16];
    u32 wcount = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = wbase + r * 64u + l;
        const u32 c = idx < cnt ? idx : cnt - 1u;
        jv[r] = ld_stream(V + base + c) & 0xFFFFFu; // (the list of a walk round may carry ranks in the spare bits)
    }
    u64 mn[17]; // (row 16: the element right after the wave's span)
    {
        const u64 myword = newbits_lane_word(a.newbits, base, wbase, cnt, l, true);
#pragma unroll
        for (u32 r = 0; r < 17; ++r) mn[r] = wave_lane64(myword, r);
    }
    // (links) the starts of the 64 elements in front of the wave's span: a group may begin there
    const u64 mn_prev = (midx && wbase >= 64u) ? a.newbits[(base + wbase - 64u) >> 6] : 0ull;
    u8 *gb8 = midx ? group_bytes(a, lb) : nullptr;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = wbase + r * 64u + l;
        bool next_new;
        if (idx + 1 >= cnt) next_new = true;
        else if (l < 63) next_new = (mn[r] >> (l + 1)) & 1ull;
        else next_new = mn[r + 1] & 1ull;
        const bool is_new = (mn[r] >> l) & 1ull;
        const bool sv = (idx < cnt) && !(is_new && next_new);
        surv[r] = __ballot(sv);
        wcount += (u32)__popcll(surv[r]);
    }
    if (l == 0) s_wsum[w] = wcount;
    __syncthreads();
    u32 off = s_off;
    for (u32 k = 0; k < w; ++k) off += s_wsum[k];
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    bool any_small = false;
    // (a sparse list: its groups go on doubling -- their group bytes are cleared all the same, the period round's keys look at them)
    // (a block of less than 64 bytes -- the two bytes behind a cut -- has no small groups either: k_link_scan does not look at it,
    // and a group byte without a link byte written in this round would be read with a stale one: fuzz cases of round 6,
    // tests/golden/fuzz_r6_*.bin)
    const bool sparse = midx && ((dense_only && (u64)s_total * 4u < (u64)a.blocks[lb].n) || a.blocks[lb].n < 64u);
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 at = off + (u32)__popcll(surv[r] & lt_mask);
        if ((surv[r] >> l) & 1ull) VS[base + at] = jv[r];
        if (midx && ((surv[r] >> l) & 1ull)) {
            // place inside its group and the group's size, from the start bits around the element
            const u64 cur = mn[r], prv = r ? mn[r - 1] : mn_prev, nxw = mn[r + 1];
            const u64 lo = (l == 63u) ? cur : ((cur << (63u - l)) | (prv >> (l + 1u)));
            const u64 hi = (l == 63u) ? nxw : ((cur >> (l + 1u)) | (nxw << (63u - l)));
            const u32 back = lo ? (u32)__clzll(lo) : 64u, fwd = hi ? (u32)__builtin_ctzll(hi) + 1u : 65u;
            const u32 size = back + fwd;
            const bool small = size >= 2u && size <= kLinkMax && !sparse;
            gb8[at] = small ? (u8)(back | (size << 4)) : (u8)0;
            if (small) {
                midx[base + jv[r]] = at | (back << 20) | (size << 24); // (never 0: the host cleared the array, 0 = no member)
                any_small = true;
            }
        }
        off += (u32)__popcll(surv[r]);
    }
    if (any_small) a.ptext[base + kSlot - 2u] = 1; // the block has small groups: k_link_scan and k_link_finalize look at it
}

__global__ void k_copy_counts(u32 *__restrict__ dst, const u32 *__restrict__ src, u32 nb)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) dst[i] = src[i];
}

// ---- periodic blocks (block = u^k): closed-form tie rule -------------------------------------
__global__ __launch_bounds__(kSortThreads) void k_periodic_stats(BwtArgs a)
{
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    if (a.nonfinal[lb] == 0) return;
    const u32 n = a.blocks[lb].n;
    const size_t base = (size_t)lb * kSlot;
    u32 c = 0, mj = 0xFFFFFFFFu;
    for (u32 j = tile * kSortTile + threadIdx.x; j < n && j < (tile + 1) * kSortTile; j += kSortThreads) {
        if (a.R[base + j] == 0u) { // non-final member of the group at position 0 (least rotation)
            ++c;
            mj = j < mj ? j : mj;
        }
    }
    c = wave_sum(c);
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) {
        const u32 o = __shfl_xor(mj, dd, 64);
        mj = o < mj ? o : mj;
    }
    if ((threadIdx.x & 63u) == 0 && c) {
        atomicAdd(&a.per_k[lb], c);
        atomicMin(&a.per_shift[lb], mj);
    }
}

__global__ __launch_bounds__(kSortThreads) void k_periodic_place(BwtArgs a)
{
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    if (a.nonfinal[lb] == 0) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    const size_t base = (size_t)lb * kSlot;
    const u32 k = a.per_k[lb], shift = a.per_shift[lb];
    if (k == 0) return;
    const u32 p = n / k; // the period
    for (u32 j = tile * kSortTile + threadIdx.x; j < n && j < (tile + 1) * kSortTile; j += kSortThreads) {
        const u32 r = a.R[base + j];
        if (r & kFinalBit) continue;
        const u32 i0 = j % p, t = j / p;
        // members j = i0 + t*p of one group, DESCENDING (j - shift) mod n
        const u32 pos = (i0 >= shift) ? (k - 1u - t) : (t == 0 ? 0u : k - t);
        a.SA[base + r + pos] = j;
        a.L[base + r + pos] = text[j ? j - 1u : n - 1u]; // (the last column of what was not final before)
        if (j == 0) a.orig_ptr[lb] = r + pos;
    }
}

// ---- blocks with a PERIOD: deep repeats without the doubling rounds ---------------------------------------------------
// A block that holds a stretch repeated at distance p -- a paragraph repeated (stress corpus T2: 4 KiB x 220), a file
// that occurs twice, a record format -- has common prefixes as long as the stretch, and prefix doubling pays
// log2(length / 2c) rounds for it (17 for T2).  The reference's SA-IS (sais.rs:127-264) costs the same whatever the data.
// What makes such data cheap: rotations i and i + p agree exactly until the first position m >= i (cyclic) where the
// block differs from itself shifted by p, and THERE rotation i reads T[m] and rotation i + p reads T[m + p].  So
//     rot(i) < rot(i + p)  <=>  T[m(i)] < T[m(i) + p],   m(i) = min { m >= i : T[m] != T[m + p] }   (indices mod n),
// whatever the depth the doubling has reached -- one bitmap of the mismatch positions and one of their directions
// answer it for every i.  A group of still-equal rotations whose members, ordered by start, step by exactly p and agree on
// the direction is therefore a chain whose order is its order by start (ascending or descending): the period round
// sorts the survivors by (group, start or mirrored start) and makes every member of such a group a group of its own.
// Groups that mix residues, skip a member, or mix directions are marked impure (k_period_mark) and go on doubling.
// Rounds 2-3 knew one case of this -- a block that is periodic from its first byte to its last (one direction for the
// whole block); a block with a few foreign bytes in front of the repeated paragraph (a block cut that does not fall on a
// paragraph boundary: every real file) got no help.  Now the period is looked for at four anchors inside the block, it
// need not hold everywhere (half the block is enough), and the direction is per mismatch.  Only the ORDER is the
// reference's business (sais.rs:266-272), and the comparison above is exact.
//   k_period_find  one workgroup per block with most of its rotations unordered: the distance at which the 16 bytes at
//                  an anchor recur (four anchors; the candidate with the widest agreement wins); kept when the block agrees with itself shifted by it at
//                  half of its positions or more, and not everywhere (a block periodic as a cycle has equal rotations:
//                  k_periodic_place's case)
//   k_period_bits  the two bitmaps;  k_period_next  first mismatch at or behind every 64-position word
__global__ __launch_bounds__(kSortThreads) void k_period_find(BwtArgs a, u32 quarters)
{
    constexpr u32 kLongAnchors = 32, kShortAnchors = 64, kShortMax = 64, kAnchors = kLongAnchors + kShortAnchors;
    __shared__ u32 s_best, s_cand[kAnchors], s_agree[kAnchors];
    const u32 lb = blockIdx.x, tid = threadIdx.x;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    if (tid < kPerK) {
        a.lin_p[(size_t)lb * kPerK + tid] = 0;
        a.lin_sig[(size_t)lb * kPerK + tid] = 0;
    }
    // (count2: the survivors of the last refinement, k_survivor_compact) only blocks that are deep in repeats
    // (round 6: the SHORT distances are looked for in every block that has a sixteenth of its rotations left -- a block may
    // hold one stretch of "ugh\n" x 13 000 and nothing else that repeats; the search costs one trip)
    if (n < 256u || (u64)a.count2[lb] * 16u < (u64)n) return;
    const bool deep_block = (u64)a.count2[lb] * 4u >= (u64)n * quarters;
    // Round 5: up to kPerK distances per block instead of one.  Data whose copies DRIFT (a file and an edited copy of it,
    // a tar of similar files; the corpus "binary": RLE1 turns a changed byte into a shift) agrees with itself at five to
    // seven distances per block, none of them over half of it: thirty-two anchors over the first half of the block each
    // name the distance at which their 16 bytes recur first, and the candidates with the widest agreement are listed.
    for (u32 k = 0; k < kLongAnchors; ++k) { // (uniform)
        if (!deep_block) {
            if (tid == 0) s_cand[k] = 0xFFFFFFFFu;
            continue;
        }
        const u32 at = (u32)(((u64)n * (k + 1u)) / (2u * (kLongAnchors + 1u))); // 1.5 % .. 48.5 % of the block: room for distances beyond n / 2
        u64 h0, h1;
        __builtin_memcpy(&h0, text + at, 8);
        __builtin_memcpy(&h1, text + at + 8, 8);
        if (tid == 0) s_best = 0xFFFFFFFFu;
        __syncthreads();
        // the farthest shift that keeps the 16 bytes inside the block -- and (round 6) not beyond an eighth of it: stretches that far
        // apart make groups of eight members at most, which k_link_scan ranks whatever their distances; the chains of a listed
        // distance are for the groups of MANY copies (a paragraph repeated).  An anchor without a hit costs a pass over its range.
        const u32 last = (n - 16u - at) < (n >> 3) ? (n - 16u - at) : (n >> 3);
        // (four shifts per thread and trip, their loads in flight together; the second half of the 16 bytes is only
        // looked at where the first agrees: a block without a repeat reads 8 bytes per shift and anchor, not 16)
        // (s_best is read through a volatile pointer: without it the compiler keeps the first value in a register, no thread
        // ever sees a hit of another one, and every anchor costs a pass over the rest of the block -- rounds 3-4 and the first
        // build of round 5: 33 ms per GiB of the stress corpus T2 with thirty-two anchors)
        for (u32 q0 = 1u + tid; q0 <= last && q0 < *(volatile u32 *)&s_best; q0 += 4u * kSortThreads) {
            u64 x[4];
#pragma unroll
            for (u32 u = 0; u < 4u; ++u) {
                const u32 q = q0 + u * kSortThreads;
                x[u] = ~h0;
                if (q <= last) __builtin_memcpy(&x[u], text + at + q, 8);
            }
            bool hit = false;
#pragma unroll
            for (u32 u = 0; u < 4u; ++u) {
                if (!hit && x[u] == h0) {
                    const u32 q = q0 + u * kSortThreads;
                    u64 x1;
                    __builtin_memcpy(&x1, text + at + q + 8, 8);
                    if (x1 == h1) {
                        atomicMin(&s_best, q);
                        hit = true;
                    }
                }
            }
            if (hit) break;
        }
        __syncthreads();
        if (tid == 0) s_cand[k] = s_best;
        __syncthreads();
    }
    // (round 6) SHORT distances -- stretches that repeat a few bytes ("ugh\n" x 30 000), whose groups per_ekey splits by where
    // the stretches end -- are looked for all over the block, not only in its first half: sixty-four more anchors, shifts of
    // 1 .. kShortMax only (one trip: a wave per anchor, a lane per shift).  Such a stretch may be a twentieth of the block.
    for (u32 k = tid >> 6; k < kShortAnchors; k += kSortThreads / 64u) {
        const u32 at = (u32)(((u64)n * (2u * k + 1u)) / (2u * kShortAnchors));
        const u32 q = (tid & 63u) + 1u;
        bool hit = false;
        if (at + q + 16u <= n && q <= kShortMax) {
            u64 h0, h1, x0, x1;
            __builtin_memcpy(&h0, text + at, 8);
            __builtin_memcpy(&h1, text + at + 8, 8);
            __builtin_memcpy(&x0, text + at + q, 8);
            __builtin_memcpy(&x1, text + at + q + 8, 8);
            hit = x0 == h0 && x1 == h1;
        }
        const u64 hits = __ballot(hit);
        if ((tid & 63u) == 0u) s_cand[kLongAnchors + k] = hits ? (u32)__builtin_ctzll(hits) + 1u : 0xFFFFFFFFu;
    }
    __syncthreads();
    // The candidates worth a look: the distinct ones, by the number of anchors that named them, a dozen at most (an anchor
    // inside a run of equal bytes answers "1", one inside a short inner repeat answers that repeat's distance; text has such
    // repeats everywhere, and every candidate looked at costs a pass over the block -- the first build of this kernel looked
    // at all of them: 33 ms per GiB of the stress corpus T2, a fifth of its step).
    constexpr u32 kLook = 12;
    __shared__ u32 s_lookc[kLook], s_nlook;
    __shared__ u32 votes[kAnchors];
    if (tid == 0) {
        for (u32 k = 0; k < kAnchors; ++k) {
            votes[k] = 0;
            const u32 c = s_cand[k];
            if (c == 0xFFFFFFFFu) continue;
            bool first = true;
            for (u32 j = 0; j < k; ++j) first = first && s_cand[j] != c;
            if (!first) continue;
            for (u32 j = k; j < kAnchors; ++j) votes[k] += s_cand[j] == c ? 1u : 0u;
        }
        u32 nl = 0;
        for (; nl < kLook; ++nl) {
            u32 bi = 0xFFFFFFFFu, bv = 0;
            for (u32 k = 0; k < kAnchors; ++k)
                if (votes[k] > bv || (votes[k] == bv && bv && s_cand[k] < s_cand[bi])) {
                    bv = votes[k];
                    bi = k;
                }
            if (bi == 0xFFFFFFFFu) break;
            s_lookc[nl] = s_cand[bi];
            votes[bi] = 0;
        }
        s_nlook = nl;
    }
    __syncthreads();
    // how widely the block agrees with itself under each of them: on every fourth stripe of 4 KiB first (a ranking needs no
    // more), on all of it when the sample agrees everywhere (periodic as a cycle or nearly so: that decides whether the
    // distance is listed at all)
    const u32 nlook = s_nlook;
    for (u32 k = 0; k < kAnchors; ++k) { // (the agreement of candidate k goes to s_agree[k], k < nlook)
        if (tid == 0) s_agree[k] = 0;
    }
    __syncthreads();
    for (u32 k = 0; k < nlook; ++k) { // (uniform)
        const u32 c = s_lookc[k];
        for (u32 pass = 0; pass < 2u; ++pass) {
            u32 mine = 0, seen = 0;
            for (u32 i = tid * 8u; i < n; i += kSortThreads * 8u) {
                if (pass == 0 && ((i / (kSortThreads * 8u)) & 3u) != 0u) continue; // every fourth stripe
                if (i + 8u + c <= n) {
                    u64 x, y;
                    __builtin_memcpy(&x, text + i, 8);
                    __builtin_memcpy(&y, text + i + c, 8);
                    const u64 z = x ^ y; // bytes that agree are zero bytes
                    mine += 8u - (u32)__popcll(((z | (z >> 1) | (z >> 2) | (z >> 3) | (z >> 4) | (z >> 5) | (z >> 6) | (z >> 7)) & 0x0101010101010101ull));
                    seen += 8u;
                } else {
                    for (u32 q = i; q < i + 8u && q < n; ++q) {
                        const u32 t = q + c;
                        mine += text[q] == text[t >= n ? t - n : t] ? 1u : 0u;
                        seen += 1u;
                    }
                }
            }
            mine = wave_sum(mine);
            seen = wave_sum(seen);
            if (tid == 0) {
                s_agree[k] = 0;
                s_best = 0; // (borrowed: the positions looked at)
            }
            __syncthreads();
            if ((tid & 63u) == 0) {
                atomicAdd(&s_agree[k], mine);
                atomicAdd(&s_best, seen);
            }
            __syncthreads();
            const u32 ag = s_agree[k], sn = s_best;
            __syncthreads();
            if (pass == 0) {
                if (ag != sn) { // the sample, scaled to the block
                    if (tid == 0) s_agree[k] = (u32)(((u64)ag * n) / (sn ? sn : 1u));
                    break;
                }
            }
            // (pass 1: the exact count stands)
        }
        __syncthreads();
    }
    if (tid == 0)
        for (u32 k = 0; k < kAnchors; ++k) s_cand[k] = k < nlook ? s_lookc[k] : 0xFFFFFFFFu; // (the list below reads s_cand / s_agree)
    __syncthreads();
    // the list: widest agreement first (ties: the shorter distance); a distance is worth its tables when the block agrees
    // with itself under it at a sixteenth of its positions or more -- and not everywhere (periodic as a cycle: equal
    // rotations, k_periodic_place's case)
    if (tid == 0) {
        const u32 kmax = per_kmax(a);
        u32 listed = 0;
        for (u32 round = 0; round < kmax; ++round) {
            u32 bi = 0xFFFFFFFFu, best = 0, bp = 0;
            for (u32 k = 0; k < kAnchors; ++k) {
                const u32 c = s_cand[k], ag = s_agree[k];
                if (c == 0xFFFFFFFFu || ag == 0u || ag >= n) continue;
                if ((u64)ag * (c <= kShortMax ? 128u : 16u) < n) continue; // (a short period is worth its tables at a 128th of the block)
                if (ag > best || (ag == best && c < bp)) {
                    best = ag;
                    bp = c;
                    bi = k;
                }
            }
            if (bi == 0xFFFFFFFFu) break;
            a.lin_p[(size_t)lb * kPerK + listed] = bp;
            a.lin_sig[(size_t)lb * kPerK + listed] = (u32)(((u64)best * 1000u) / n); // (for the trace: agreement in permille)
            ++listed;
            s_agree[bi] = 0; // taken
        }
    }
}

__global__ __launch_bounds__(kSortThreads) void k_period_bits(BwtArgs a)
{
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    if (a.lin_p[(size_t)lb * kPerK] == 0u) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 start = tile * kSortTile;
    if (start >= n) return;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 kmax = per_kmax(a);
    for (u32 pi = 0; pi < kmax; ++pi) { // (uniform)
        const u32 p = a.lin_p[(size_t)lb * kPerK + pi];
        if (p == 0u) break;
        u64 *mis = const_cast<u64 *>(per_mis(a, lb, pi)), *lt = const_cast<u64 *>(per_lt(a, lb, pi));
#pragma unroll
        for (u32 r = 0; r < 16; ++r) {
            const u32 i = start + w * 1024u + r * 64u + l;
            bool m = false, less = false;
            if (i < n) {
                const u32 t = i + p;
                const u8 x = text[i], y = text[t >= n ? t - n : t];
                m = x != y;
                less = x < y;
            }
            const u64 bm = __ballot(m), bl = __ballot(less);
            if (l == 0 && start + w * 1024u + r * 64u < n) {
                mis[(start + w * 1024u + r * 64u) >> 6] = bm;
                lt[(start + w * 1024u + r * 64u) >> 6] = bl;
            }
        }
    }
}

// one workgroup per (block, listed distance)
__global__ __launch_bounds__(kSortThreads) void k_period_next(BwtArgs a)
{
    __shared__ u32 s_first[kSortThreads];
    const u32 lb = blockIdx.x, pi = blockIdx.y, tid = threadIdx.x;
    if (pi >= per_kmax(a) || a.lin_p[(size_t)lb * kPerK + pi] == 0u) return;
    const u32 n = a.blocks[lb].n;
    const u32 nw = (n + 63u) >> 6;
    const u64 *mis = per_mis(a, lb, pi);
    u32 *nxt = const_cast<u32 *>(per_nxt(a, lb, pi));
    const u32 per = (nw + kSortThreads - 1u) / kSortThreads; // words per thread, consecutive
    const u32 w0 = tid * per, w1 = w0 + per < nw ? w0 + per : nw;
    u32 first = 0xFFFFFFFFu; // the first mismatch inside this thread's words
    for (u32 w = w0; w < w1; ++w) {
        const u64 m = mis[w];
        if (m) {
            first = (w << 6) + (u32)__builtin_ctzll(m);
            break;
        }
    }
    s_first[tid] = first;
    __syncthreads();
    // the first mismatch behind this thread's words (threads above it, then -- wrapping -- the block's first one + n)
    u32 behind = 0xFFFFFFFFu;
    for (u32 t = tid + 1u; t < kSortThreads; ++t)
        if (s_first[t] != 0xFFFFFFFFu) {
            behind = s_first[t];
            break;
        }
    if (behind == 0xFFFFFFFFu)
        for (u32 t = 0; t < kSortThreads; ++t)
            if (s_first[t] != 0xFFFFFFFFu) {
                behind = s_first[t] + n;
                break;
            }
    u32 run = behind;
    for (u32 w = w1; w > w0;) {
        --w;
        const u64 m = mis[w];
        if (m) run = (w << 6) + (u32)__builtin_ctzll(m);
        nxt[w] = run;
    }
    if (tid == 0) { // (the word behind the last one: the wrap)
        u32 f = 0xFFFFFFFFu;
        for (u32 t = 0; t < kSortThreads; ++t)
            if (s_first[t] != 0xFFFFFFFFu) {
                f = s_first[t];
                break;
            }
        nxt[nw] = f + n;
    }
}

// ---- SMALL groups: ranked member by member from direct comparisons, one scan per STRETCH of copies ----------------------------
// Data that holds a stretch several times (a file and edited copies of it, a tar of similar files; the corpus "binary": the
// reference's fixtures tiled with a byte changed every 4 KiB, and the fixtures repeat inside themselves) leaves, after the
// first rounds, almost every rotation in a group of two to four whose members agree for kilobytes: eight and more doubling
// rounds over the whole block to find out what one look at the texts tells -- they are 0.9 MB, they sit in the L2.  And the
// comparisons of a copied stretch are SHARED: rotations x and y = x + d agree up to the same position m as x + 1 and
// y + 1 do -- the first m at or behind the later start + depth with T[m] != T[m + d] -- whatever d is and however many
// distances a block has (copies that DRIFT: every changed byte that RLE1 turns into a shift starts a new d).  So:
//   k_survivor_compact  every member of a group of 2 .. kLinkMax survivors leaves its place in the list under its START;
//   k_link_scan  walks the block in TEXT order, a wave per 64 consecutive starts.  A lane whose start x is such a member
//                reads its group and takes the members that start behind x one after the other, nearest first; the
//                lanes whose partner lies the same distance away are resolved together -- the wave compares 512 bytes of
//                T and of T shifted by d per step, eight per lane, until a difference turns up, and every lane whose
//                known-equal prefix ends in front of it takes it (the lanes behind it go on from there).  EVERY pair of
//                members is compared (by the member with the smaller start): a chain of neighbours in start order would
//                do for most groups, but a copy with a changed byte between two that agree beyond it leaves the two
//                undecided -- and that is exactly what drifting copies look like.  A decided comparison adds 1 to the
//                "decided" nibble of both members' link bytes and 1 to the "members below" nibble of the greater one;
//   k_link_finalize  a group all of whose members have all their comparisons decided is ordered, and its members are final
//                at once (SA, last column, rank word); the period round's passes and the link round's k_link_compact run
//                on what is left.
// The comparison is the definition of the order (sais.rs:266-272 fixes nothing else); a group that is not decided (equal
// rotations, more stretches in one row of starts than kLinkTries) goes on doubling.
constexpr u32 kLinkTries = 8;      // stretches per row of 64 starts and partner that the whole wave scans
__device__ __forceinline__ u64 link_load8(const u8 *__restrict__ text, u32 n, u32 p) // eight bytes from p on, cyclic (p < n)
{
    u64 v;
    if (p + 8u <= n) {
        __builtin_memcpy(&v, text + p, 8);
    } else {
        v = 0;
        for (u32 k = 0; k < 8u; ++k) {
            const u32 q = p + k;
            v |= (u64)text[q >= n ? q - n : q] << (8u * k);
        }
    }
    return v;
}

__global__ __launch_bounds__(kSortThreads) void k_link_scan(BwtArgs a, u32 step, const u32 *__restrict__ VS, const u32 *__restrict__ midx,
                                                             const u8 *__restrict__ impure, u32 *__restrict__ RES)
{
    // what the last scan of this wave found, per partner slot: {distance, leader's start, first difference (absolute, not
    // wrapped), smaller start is the smaller rotation}.  The next row of 64 starts almost always lies inside the same
    // stretch: its lanes take the verdict without a scan.
    __shared__ u32 s_carry[kSortThreads / 64][kLinkMax][4];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 start = tile * kSortTile;
    if (start >= n || n < 64u) return;
    const size_t base = (size_t)lb * kSlot;
    if (impure[base + kSlot - 2u] == 0) return; // (k_survivor_compact found no small group in this block)
    const u8 *__restrict__ text = a.rle + d.rle_off;
    const KeyInfo ki = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb];
    const u64 d64 = ((u64)ki.chars * 2u) << step;
    const u32 depth = d64 < n ? (u32)d64 : n;
    const u32 cnt = a.count[lb];
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    if (l < kLinkMax * 4u) (&s_carry[w][0][0])[l] = 0u; // (distance 0: no carry)
    u32 mi[16];
#pragma unroll
    for (u32 r = 0; r < 16u; ++r) {
        const u32 x = start + w * 1024u + r * 64u + l;
        mi[r] = x < n ? ld_stream(midx + base + x) : 0u;
    }
#pragma unroll 1
    for (u32 r = 0; r < 16u; ++r) {
        const u32 x = start + w * 1024u + r * 64u + l;
        u32 e = 0;
#pragma unroll
        for (u32 q = 0; q < 16u; ++q) e = q == r ? mi[q] : e; // (a register array under a loop that is not unrolled)
        // the group of x: its place in the list, the members, and for each member behind x how many lie between
        u32 idx = e & 0xFFFFFu, size = e >> 24;
        const u32 off = (e >> 20) & 15u;
        if (size < 2u || size > kLinkMax || off >= size || idx < off || idx - off + size > cnt) size = 0; // (0: no member of a small group)
        if (!__ballot(size != 0u)) continue;
        const u32 g0 = idx - off;
        u32 mem[kLinkMax], slot[kLinkMax];
        {
            // (the members lie side by side in the list: two 16-byte loads instead of eight gathers; entries behind the
            // group's last one are read and not used -- the list's array is kSlot words, so is the slack behind cnt)
            uint4 m0 = make_uint4(0, 0, 0, 0), m1 = make_uint4(0, 0, 0, 0);
            if (size) __builtin_memcpy(&m0, VS + base + g0, 16);
            if (size > 4u) __builtin_memcpy(&m1, VS + base + g0 + 4u, 16);
            mem[0] = m0.x; mem[1] = m0.y; mem[2] = m0.z; mem[3] = m0.w;
            mem[4] = m1.x; mem[5] = m1.y; mem[6] = m1.z; mem[7] = m1.w;
        }
        // the members that start BEHIND x, nearest first (a pair is compared by its member with the smaller start; the other one
        // reads the verdict in k_link_finalize): consecutive starts of a copied stretch meet the same distances in the same slots
        u32 nothers = 0;
#pragma unroll
        for (u32 t = 0; t < kLinkMax; ++t) {
            u32 c = 0;
#pragma unroll
            for (u32 u = 0; u < kLinkMax; ++u) c += (u < size && mem[u] > x && mem[u] < mem[t]) ? 1u : 0u;
            const bool behind = t < size && mem[t] > x && mem[t] < n;
            slot[t] = behind ? c : 0xFFFFFFFFu;
            nothers += behind ? 1u : 0u;
        }
        u32 own = 0; // bit s: the comparison with the s-th member behind x is decided; bit 8 + s: rot(x) is the smaller one
#pragma unroll 1
        for (u32 s = 0; s + 1u < kLinkMax; ++s) {
            const bool has = s < nothers;
            u64 todo = __ballot(has);
            if (!todo) break; // (no lane has a partner left)
            u32 y = x;
#pragma unroll
            for (u32 t = 0; t < kLinkMax; ++t)
                if (slot[t] == s) y = mem[t];
            const u32 dist = y - x;
            bool solved = false, lo_less = false;
            {
                // inside the stretch the wave's last scan for this slot resolved?
                const u32 cd = s_carry[w][s][0], cx = s_carry[w][s][1], cD = s_carry[w][s][2], cl = s_carry[w][s][3];
                if (has && cd != 0u && dist == cd && x >= cx && x + depth <= cD) {
                    solved = true;
                    lo_less = cl != 0u;
                }
                todo &= ~__ballot(solved);
            }
            if (todo) {
                // a look of its own for every lane first: sixteen bytes behind the depth reached.  Rows whose lanes all have
                // partners somewhere else (text: short repeats all over the block) are no stretch of one copy, a scan by the
                // whole wave resolves one lane of them; copies that agree for kilobytes pay four loads for nothing.  (Leaving
                // the look out in rows whose neighbouring lanes lie the same distance from their partners was measured:
                // 7.2 -> 8.3 ms per 256 MiB of the corpus "binary" -- at sixteen symbols many of ITS pairs end within the look.)
                if (has && !solved && depth + 16u < n) {
                    u32 pa = x + depth, pb = y + depth;
                    pa = pa >= n ? pa - n : pa;
                    pb = pb >= n ? pb - n : pb;
                    u32 pa8 = pa + 8u, pb8 = pb + 8u;
                    pa8 = pa8 >= n ? pa8 - n : pa8;
                    pb8 = pb8 >= n ? pb8 - n : pb8;
                    const u64 xa = link_load8(text, n, pa), xb = link_load8(text, n, pb);
                    const u64 xa8 = link_load8(text, n, pa8), xb8 = link_load8(text, n, pb8);
                    const u64 df = xa != xb ? xa ^ xb : xa8 ^ xb8;
                    if (df) {
                        const u32 byte = (u32)__builtin_ctzll(df) >> 3;
                        const u64 ua = xa != xb ? xa : xa8, ub = xa != xb ? xb : xb8;
                        solved = true;
                        lo_less = (u32)((ua >> (8u * byte)) & 0xFFull) < (u32)((ub >> (8u * byte)) & 0xFFull);
                    }
                }
                todo &= ~__ballot(solved);
            }
            for (u32 tries = 0; todo && tries < kLinkTries; ++tries) {
                const u32 L = (u32)__builtin_ctzll(todo);
                const u32 dL = wave_lane(dist, L), xL = wave_lane(x, L);
                const u64 grp = __ballot(has && !solved && dist == dL) & todo; // (lane L is in it; its start is the group's smallest)
                u32 sp = xL + depth;                                            // where the leader's known-equal prefix ends
                sp = sp >= n ? sp - n : sp;
                u32 scanned = 0;
                bool found = false;
                while (scanned + depth < n) { // (all the way: two members that agree to the end are equal rotations, a periodic block)
                    // (a lane whose eight bytes begin a whole turn behind the leader's start and more has nothing new to look at --
                    // and in a block shorter than the 512 bytes of a step its position would wrap twice: a periodic block
                    // of 72 bytes came out in the wrong order, tests/golden/fuzz_r6_links_2414.bin)
                    const bool in_turn = scanned + 8u * l + depth < n;
                    u32 pa = sp + (in_turn ? 8u * l : 0u);
                    pa = pa >= n ? pa - n : pa;
                    u32 pb = pa + dL;
                    pb = pb >= n ? pb - n : pb;
                    const u64 xa = link_load8(text, n, pa), xb = in_turn ? link_load8(text, n, pb) : xa;
                    const u64 df = xa ^ xb;
                    const u64 bm = __ballot(df != 0ull);
                    if (bm) {
                        const u32 f = (u32)__builtin_ctzll(bm);
                        const u64 dff = wave_lane64(df, f), xaf = wave_lane64(xa, f), xbf = wave_lane64(xb, f);
                        const u32 byte = (u32)__builtin_ctzll(dff) >> 3;
                        const u32 moff = scanned + 8u * f + byte; // the first difference, counted from the leader's start + depth
                        const bool less = (u32)((xaf >> (8u * byte)) & 0xFFull) < (u32)((xbf >> (8u * byte)) & 0xFFull);
                        // every lane of the group whose own known-equal prefix ends at or in front of the difference takes it
                        const bool mine = ((grp >> l) & 1ull) && (x - xL) <= moff;
                        if (mine) {
                            solved = true;
                            lo_less = less;
                        }
                        todo &= ~__ballot(mine);
                        found = true;
                        if (l == 0) {
                            s_carry[w][s][0] = dL;
                            s_carry[w][s][1] = xL;
                            s_carry[w][s][2] = xL + depth + moff;
                            s_carry[w][s][3] = less ? 1u : 0u;
                        }
                        break;
                    }
                    sp += 512u;
                    sp = sp >= n ? sp - n : sp;
                    scanned += 512u;
                }
                if (!found) todo &= ~grp; // (equal rotations: a periodic block -- k_periodic_place's case)
            }
            if (solved && y != x) own |= (1u << s) | (lo_less ? (0x100u << s) : 0u);
        }
        // (every member writes its word, decided or not: k_link_finalize reads the words of a whole group.  The first build had
        // the smaller start add to its partner's byte -- 375 M byte atomics per 256 MiB of the corpus "binary", 7.6 of the
        // kernel's 11.5 ms; the second had both members of a pair scan it: twice the partners per lane, 8.2 ms.)
        if (size) RES[base + idx] = own;
    }
}

// A group all of whose pairs are decided is ORDERED, and its members are final at once: member x stands at the group's first
// place in SA + the number of members below it.  What k_group_apply does for a rotation that becomes final is done right
// here -- SA, the last column, origPtr, the rank word (a 4-byte store at random: they are a few per group, and the
// rotations leave the list for good) -- and the entry's group byte becomes 1: the period round's passes and the link
// round's compaction leave it out, so everything behind this kernel runs on what is LEFT (a third of the list on the corpus
// "binary", where the passes, the flags and the apply kernel over the whole list were 14 of the round's 30 ms).
// A member reads the group's members and their result words (k_link_scan: the verdict of a pair lies with its smaller
// start, under the number of members between the two), counts the members below it and says whether ITS pairs are all
// decided; the group is decided when every member says so -- a ballot when the group's entries lie in one row of 64 list
// entries (all but one group in twenty), else every pair is looked at.
// tile_nf[tile] = entries of the tile that stay (for k_link_compact).
__device__ __forceinline__ bool link_pair(const u32 (&m)[kLinkMax], const u32 (&res)[kLinkMax], u32 size, u32 ai, u32 bi, bool &a_less)
{
    // the pair (m[ai], m[bi]): decided?  a_less: rot(m[ai]) < rot(m[bi])
    const bool a_lo = m[ai] < m[bi];
    const u32 lo = a_lo ? m[ai] : m[bi], hi = a_lo ? m[bi] : m[ai];
    const u32 r = a_lo ? res[ai] : res[bi];
    u32 sl = 0;
#pragma unroll
    for (u32 u = 0; u < kLinkMax; ++u) sl += (u < size && m[u] > lo && m[u] < hi) ? 1u : 0u;
    const bool lo_less = (r >> (8u + sl)) & 1u;
    a_less = a_lo ? lo_less : !lo_less;
    return (r >> sl) & 1u;
}

__global__ __launch_bounds__(kSortThreads) void k_link_finalize(BwtArgs a, const u32 *__restrict__ VS, const u8 *__restrict__ impure,
                                                                 const u32 *__restrict__ RES)
{
    __shared__ u32 s_stay;
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    const size_t base = (size_t)lb * kSlot;
    if (start >= cnt) return;
    const u32 here = cnt - start < kSortTile ? cnt - start : kSortTile;
    if (impure[base + kSlot - 2u] == 0) { // (no small group: every group byte of the block's list is 0 already, everything stays)
        if (threadIdx.x == 0) a.tile_nf[lb * kTilesPerBlock + tile] = here;
        return;
    }
    if (threadIdx.x == 0) s_stay = 0;
    __syncthreads();
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u8 *__restrict__ text = a.rle + d.rle_off;
    u8 *gb8 = group_bytes(a, lb);
    const u32 l = threadIdx.x & 63u;
    u32 stay = 0;
    // (the trips of the loop are uniform for the workgroup -- the ballots below need every lane of a wave)
    for (u32 i0 = start; i0 < start + kSortTile && i0 < cnt; i0 += kSortThreads) {
        const u32 idx = i0 + threadIdx.x;
        const bool live = idx < cnt;
        const u32 gb = live ? gb8[idx] : 0u;
        u32 size = gb >> 4;
        const u32 off = gb & 15u;
        if (size < 2u || size > kLinkMax || off >= size || idx < off || idx - off + size > cnt) size = 0;
        const u32 g0 = idx - off;
        u32 m[kLinkMax], res[kLinkMax];
        {
            uint4 m0 = make_uint4(0, 0, 0, 0), m1 = make_uint4(0, 0, 0, 0), r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
            if (size) {
                __builtin_memcpy(&m0, VS + base + g0, 16);
                __builtin_memcpy(&r0, RES + base + g0, 16);
            }
            if (size > 4u) {
                __builtin_memcpy(&m1, VS + base + g0 + 4u, 16);
                __builtin_memcpy(&r1, RES + base + g0 + 4u, 16);
            }
            m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
            res[0] = r0.x; res[1] = r0.y; res[2] = r0.z; res[3] = r0.w; res[4] = r1.x; res[5] = r1.y; res[6] = r1.z; res[7] = r1.w;
        }
        // this member's pairs
        u32 x = 0;
#pragma unroll
        for (u32 t = 0; t < kLinkMax; ++t) x = t == off ? m[t] : x;
        u32 rmine = 0;
#pragma unroll
        for (u32 t = 0; t < kLinkMax; ++t) rmine = t == off ? res[t] : rmine;
        bool own_ok = size != 0u;
        u32 rank = 0;
#pragma unroll
        for (u32 t = 0; t < kLinkMax; ++t) {
            if (t >= size || m[t] == x) continue;
            // (the pair (x, m[t]) through link_pair's arithmetic, with x's own word at hand)
            const bool x_lo = x < m[t];
            const u32 lo = x_lo ? x : m[t], hi = x_lo ? m[t] : x;
            const u32 r = x_lo ? rmine : res[t];
            u32 sl = 0;
#pragma unroll
            for (u32 u = 0; u < kLinkMax; ++u) sl += (u < size && m[u] > lo && m[u] < hi) ? 1u : 0u;
            const bool lo_less = (r >> (8u + sl)) & 1u;
            own_ok = own_ok && ((r >> sl) & 1u);
            rank += (x_lo ? !lo_less : lo_less) ? 1u : 0u; // the other member is the smaller rotation
        }
        // the group: every member's pairs decided
        bool ranked;
        {
            const u64 okm = __ballot(own_ok);
            const bool in_row = size != 0u && off <= l && l - off + size <= 64u;
            const u64 want = (size >= 64u ? ~0ull : ((1ull << size) - 1ull)) << (in_row ? l - off : 0u);
            ranked = in_row && (okm & want) == want;
            if (size != 0u && !in_row && own_ok) { // (the group lies across two rows: every pair)
                bool all = true;
#pragma unroll
                for (u32 p = 0; p < kLinkMax; ++p)
#pragma unroll
                    for (u32 q = p + 1u; q < kLinkMax; ++q) {
                        if (q >= size) continue;
                        bool dummy;
                        all = all && link_pair(m, res, size, p, q, dummy);
                    }
                ranked = all;
            }
        }
        // (the group's first place in SA is the rank of any of its members: the entry that stands first in the list looks it up,
        // the others take it from that lane when it belongs to the same wave)
        const bool first = off == 0u;
        const bool same_wave = ranked && off <= l;
        u32 head = (ranked && (!same_wave || first)) ? (a.R[base + x] & ~kFinalBit) : 0u;
        {
            const u32 from = __shfl(head, (int)(l - (same_wave ? off : 0u)), 64);
            if (same_wave && !first) head = from;
        }
        if (ranked) {
            const u32 pos = head + rank;
            a.SA[base + pos] = x;
            a.L[base + pos] = text[x ? x - 1u : n - 1u];
            if (x == 0) a.orig_ptr[lb] = pos;
            a.R[base + x] = pos | kFinalBit;
        } else if (live) {
            stay += 1;
        }
        if (live) gb8[idx] = ranked ? (u8)1 : (u8)0; // (each entry reads its own group byte only: nobody else's read is disturbed)
    }
    stay = wave_sum(stay);
    if (l == 0 && stay) atomicAdd(&s_stay, stay);
    __syncthreads();
    if (threadIdx.x == 0) a.tile_nf[lb * kTilesPerBlock + tile] = s_stay;
}

// a link round's list: what k_link_finalize left, in list (= group) order; keys = the group heads
__global__ __launch_bounds__(kSortThreads) void k_link_compact(BwtArgs a, const u32 *__restrict__ VS, u32 *__restrict__ Kout, u32 *__restrict__ Vout)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ u32 s_off, s_wsum[NW];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 cnt = a.count2[lb]; // (the whole survivor list; count becomes what is left)
    const u32 start = tile * kSortTile;
    const u32 ntiles = (cnt + kSortTile - 1) / kSortTile;
    if (tile == 0 && threadIdx.x == 0) {
        u32 tot = 0;
        for (u32 t = 0; t < ntiles; ++t) tot += a.tile_nf[lb * kTilesPerBlock + t];
        a.count[lb] = tot;
    }
    if (start >= cnt) return;
    const size_t base = (size_t)lb * kSlot;
    if (threadIdx.x == 0) {
        u32 off = 0;
        for (u32 t = 0; t < tile; ++t) off += a.tile_nf[lb * kTilesPerBlock + t];
        s_off = off;
    }
    const u8 *gb8 = group_bytes(a, lb);
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 wbase = start + w * 1024u;
    u64 keep[16];
    u32 xv[16];
    u32 wcount = 0;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = wbase + r * 64u + l;
        const u32 c = idx < cnt ? idx : cnt - 1u;
        xv[r] = ld_stream(VS + base + c);
        keep[r] = __ballot(idx < cnt && gb8[c] != 1u);
        wcount += (u32)__popcll(keep[r]);
    }
    if (l == 0) s_wsum[w] = wcount;
    __syncthreads();
    u32 off = s_off;
    for (u32 k = 0; k < w; ++k) off += s_wsum[k];
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if ((keep[r] >> l) & 1ull) {
            const u32 at = off + (u32)__popcll(keep[r] & lt_mask);
            Kout[base + at] = a.R[base + xv[r]] & ~kFinalBit;
            Vout[base + at] = xv[r];
        }
        off += (u32)__popcll(keep[r]);
    }
}

// the period round's list, ordered by (group, start or mirrored start): every pair of neighbours inside a group must lie
// a LISTED distance apart and stand in the order the first difference behind them gives (rot(lo) < rot(lo + p) <=>
// T[m] < T[m + p] at the first m >= lo where the block and the block shifted by p differ: exact, whatever depth the
// doubling has reached); a group with a pair that does not is marked impure (one byte per group head) and goes on
// doubling.  The pairs of a group are checked one by one, so a group that passes is in its true order.
__global__ __launch_bounds__(kSortThreads) void k_period_mark(BwtArgs a, u32 step, const u32 *__restrict__ K,
                                                               const u32 *__restrict__ V, u8 *__restrict__ impure)
{
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 cnt = a.count[lb];
    const u32 start = tile * kSortTile;
    if (start >= cnt || a.lin_p[(size_t)lb * kPerK] == 0u) return;
    const u32 n = a.blocks[lb].n;
    const u32 kmax = per_kmax(a);
    (void)step;
    const size_t base = (size_t)lb * kSlot;
    u32 lp[kPerK];
#pragma unroll
    for (u32 i = 0; i < kPerK; ++i) lp[i] = i < kmax ? a.lin_p[(size_t)lb * kPerK + i] : 0u;
    // (round 6) the sixteen rows of a lane together, every phase's loads in flight at once (the kernel took an entry at a
    // time: a chain of five dependent loads each, 11 ms per GiB of the stress corpus T2 -- a tenth of its step)
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 first = start + w * 1024u + l;
    u32 g[16], v[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = first + r * 64u;
        const u32 c = idx < cnt ? idx : cnt - 1u;
        g[r] = ld_stream(K + base + c);
        v[r] = ld_stream(V + base + c);
    }
    u32 pg0 = 0, pv0 = 0;
    if (l == 0 && first > 0 && first < cnt) {
        pg0 = K[base + first - 1];
        pv0 = V[base + first - 1];
    }
    u32 lo[16], wh[16];
    u32 act = 0, vp_lo = 0; // per row: the pair is checked against the tables; the element in front is the pair's smaller start
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = first + r * 64u;
        u32 pg = wave_shr1(g[r]), pv = wave_shr1(v[r]);
        const u32 qg = (r == 0) ? pg0 : wave_lane(g[(r + 15) & 15], 63);
        const u32 qv = (r == 0) ? pv0 : wave_lane(v[(r + 15) & 15], 63);
        if (l == 0) {
            pg = qg;
            pv = qv;
        }
        lo[r] = v[r] < pv ? v[r] : pv;
        wh[r] = kPerK;
        if (idx < cnt && idx > 0 && g[r] == pg) {
            const u32 dist = v[r] < pv ? pv - v[r] : v[r] - pv;
#pragma unroll
            for (u32 i = 0; i < kPerK; ++i)
                if (lp[i] != 0u && lp[i] == dist) wh[r] = i;
            if (wh[r] == kPerK) impure[base + g[r]] = 1; // (no listed distance apart)
            else {
                act |= 1u << r;
                vp_lo |= (pv == lo[r] ? 1u : 0u) << r;
            }
        }
    }
    u64 here[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) here[r] = ((act >> r) & 1u) ? (per_mis(a, lb, wh[r])[lo[r] >> 6] & (~0ull << (lo[r] & 63u))) : 1ull;
    u32 m[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r)
        m[r] = here[r] ? ((lo[r] >> 6) << 6) + (u32)__builtin_ctzll(here[r]) : per_nxt(a, lb, wh[r])[(lo[r] >> 6) + 1u];
    u64 ltw[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (m[r] >= n) m[r] -= n;
        ltw[r] = ((act >> r) & 1u) ? per_lt(a, lb, wh[r])[m[r] >> 6] : 0ull;
    }
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (!((act >> r) & 1u)) continue;
        const bool lo_first = (ltw[r] >> (m[r] & 63u)) & 1ull;   // rot(lo) < rot(lo + p)
        if (lo_first != (((vp_lo >> r) & 1u) != 0u)) impure[base + g[r]] = 1; // (the list has the element in front first)
    }
}

// ---- symbols in use and key geometry (before the sort) ----------------------------------------------
__global__ __launch_bounds__(kSortThreads) void k_block_symbols(BwtArgs a, u32 *__restrict__ inuse_bits /*[nb][8]*/)
{
    __shared__ u32 s_bits[8];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    // (eight tiles per workgroup: one tile is ONE 16-byte load per thread, and the reductions and atomics behind it cost more than
    // the load -- 0.98 ms per GiB for a kernel that reads the image once)
    constexpr u32 kSpan = kSymSpan * kSortTile;
    const u32 start = tile * kSpan;
    if (start >= n) return;
    const u8 *text = a.rle + d.rle_off;
    const u8 *pt = a.ptext + (size_t)lb * kSlot;
    if (threadIdx.x < 8) s_bits[threadIdx.x] = 0;
    __syncthreads();
    // 16 bytes per load, from the 16-byte boundary in front of the tile (the image is padded at both ends of a block's
    // bytes by its neighbours or by the buffer's slack); four 64-bit sets, picked with selects
    const u32 cnt = (n - start) < kSpan ? (n - start) : kSpan;
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(text + start);
    const u32 lead = (u32)(p0 & 15u);
    const uint4 *src = reinterpret_cast<const uint4 *>(p0 - lead);
    u64 set[4] = {0, 0, 0, 0};
    for (u32 c = threadIdx.x; c * 16u < lead + cnt; c += kSortThreads) {
        const uint4 v = src[c];
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            const u32 pos = c * 16u + k;
            const u32 b = (w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu;
            const u64 bit = (pos >= lead && pos < lead + cnt) ? (1ull << (b & 63u)) : 0ull;
            const u32 hi = b >> 6;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) set[q] |= (hi == q) ? bit : 0ull;
        }
    }
    u32 seen[8];
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        seen[2 * q] = (u32)set[q];
        seen[2 * q + 1] = (u32)(set[q] >> 32);
    }
#pragma unroll
    for (u32 q = 0; q < 8; ++q) {
        u32 v = seen[q];
#pragma unroll
        for (u32 dd = 32; dd >= 1; dd >>= 1) v |= __shfl_xor(v, dd, 64);
        if ((threadIdx.x & 63u) == 0 && v) atomicOr(&s_bits[q], v);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_bits[threadIdx.x]) atomicOr(&inuse_bits[lb * 8 + threadIdx.x], s_bits[threadIdx.x]);
}

// one 256-thread workgroup per block: byte -> code table (rank among the bytes in use, the
// reference's unseq2seq, src/bzip2/encoder.rs:304-314) and the key geometry
__global__ __launch_bounds__(256) void k_key_params(const u32 *__restrict__ inuse_bits, u8 *__restrict__ sym_code,
                                                     u8 *__restrict__ keyinfo)
{
    const u32 lb = blockIdx.x;
    const u32 *bits = inuse_bits + lb * 8;
    const u32 v = threadIdx.x;
    u32 before = 0;
    for (u32 q = 0; q < (v >> 5); ++q) before += __popc(bits[q]);
    before += __popc(bits[v >> 5] & ((1u << (v & 31u)) - 1u));
    sym_code[(size_t)lb * 256 + v] = (u8)before;
    if (v == 0) {
        u32 alpha = 0;
        for (u32 q = 0; q < 8; ++q) alpha += __popc(bits[q]);
        u32 nbits = 1;
        while ((1u << nbits) < alpha) ++nbits;
        u32 chars = (nbits >= 8) ? 4u : 30u / nbits;
        if (chars > 8u) chars = 8u;
        KeyInfo ki;
        ki.bits = (u8)nbits;
        ki.chars = (u8)chars;
        ki.pad[0] = ki.pad[1] = 0;
        reinterpret_cast<KeyInfo *>(keyinfo)[lb] = ki;
    }
}

// The block's symbols as a packed string for pkey(): `bits` bits per re-coded symbol, most significant bit first,
// symbols n .. n+15 = symbols 0 .. 15 of the cyclic text.  A thread packs groups of eight symbols (= `bits` whole
// bytes); a tile's bytes leave through LDS as dwords.
__global__ __launch_bounds__(kSortThreads) void k_pack_text(BwtArgs a, u8 *__restrict__ ptext)
{
    __shared__ u8 s_code[256];
    __shared__ u32 s_out[kSortTile / 4]; // 1024 groups x at most 8 bytes
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu) return;
    const BlockDesc d = a.blocks[lb];
    const u32 n = d.n;
    const u32 start = tile * kSortTile;
    if (n == 0 || start >= n + 16u) return;
    const u8 *text = a.rle + d.rle_off;
    const u32 bits = reinterpret_cast<const KeyInfo *>(a.keyinfo)[lb].bits;
    for (u32 i = threadIdx.x; i < 256; i += kSortThreads) s_code[i] = a.sym_code[(size_t)lb * 256 + i];
    __syncthreads();
    u8 *s_bytes = reinterpret_cast<u8 *>(s_out);
#pragma unroll
    for (u32 h = 0; h < 2; ++h) {
        const u32 g = h * kSortThreads + threadIdx.x; // group inside the tile
        const u32 i0 = start + g * 8u;
        u32 lo = 0, hi = 0;
        if (i0 + 8u <= n) {
            const uintptr_t p = reinterpret_cast<uintptr_t>(text + i0);
            const u32 *ap = reinterpret_cast<const u32 *>(p & ~(uintptr_t)3);
            const u32 w0 = ap[0], w1 = ap[1], w2 = ap[2];
            lo = __builtin_amdgcn_alignbyte(w1, w0, (u32)(p & 3u));
            hi = __builtin_amdgcn_alignbyte(w2, w1, (u32)(p & 3u));
        } else if (i0 < n + 16u) {
            for (u32 q = 0; q < 8; ++q) {
                const u32 i = i0 + q;
                const u32 c = (i < n + 16u) ? (u32)text[i < n ? i : (i - n) % n] : 0u;
                if (q < 4) lo |= c << (8u * q);
                else hi |= c << (8u * (q - 4u));
            }
        }
        u64 v = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) {
            const u32 byte = ((q < 4 ? lo >> (8u * q) : hi >> (8u * (q - 4u)))) & 0xFFu;
            v = (v << bits) | (u64)s_code[byte];
        }
        // the 8 * bits bits of v, most significant byte first
        for (u32 k = 0; k < bits; ++k) s_bytes[g * bits + k] = (u8)(v >> (8u * (bits - 1u - k)));
    }
    __syncthreads();
    u32 *dst = reinterpret_cast<u32 *>(ptext + (size_t)lb * kSlot + (size_t)tile * 1024u * bits);
    for (u32 i = threadIdx.x; i < 256u * bits; i += kSortThreads) dst[i] = s_out[i];
}

// ---- host side ------------------------------------------------------------------------------------
int KernelProf::begin(hipStream_t st, int id, u64 nbytes)
{
    if (!on) return -1;
    if (nrecs == cap) {
        const u32 ncap = cap ? cap * 2 : 256;
        Rec *nr = (Rec *)realloc(recs, ncap * sizeof(Rec));
        if (!nr) return -1;
        recs = nr;
        cap = ncap;
    }
    Rec &r = recs[nrecs];
    r.id = id;
    r.bytes = nbytes;
    if (hipEventCreate(&r.a) != hipSuccess) return -1;
    if (hipEventCreate(&r.b) != hipSuccess) {
        (void)hipEventDestroy(r.a);
        return -1;
    }
    (void)hipEventRecord(r.a, st);
    return (int)nrecs++;
}
void KernelProf::end(hipStream_t st, int idx)
{
    if (idx >= 0) (void)hipEventRecord(recs[idx].b, st);
}
void KernelProf::collect()
{
    for (u32 i = 0; i < nrecs; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, recs[i].a, recs[i].b) == hipSuccess) {
            launches[recs[i].id] += 1;
            bytes[recs[i].id] += recs[i].bytes;
            seconds[recs[i].id] += ms * 1e-3;
        }
        (void)hipEventDestroy(recs[i].a);
        (void)hipEventDestroy(recs[i].b);
    }
    nrecs = 0;
}
void KernelProf::reset()
{
    collect();
    for (int i = 0; i < KID_COUNT; ++i) {
        launches[i] = 0;
        bytes[i] = 0;
        seconds[i] = 0;
    }
}

// Algorithmic bytes per element of each radix kernel, by source:
//   hist   : TEXT 1 (block byte), PAIRS 4 (key), MM 8 (SA + rank), WALK 5 (order + block byte)
//   scatter: the same reads (+4 for the value of PAIRS) + 8 written (key, value)
// `tiles`: tiles per block the launches cover (a list of a round is often a few tiles long: launching 110 tiles for
// every block costs ~60 us per kernel in workgroups that have nothing to do)
template <int SRC, int BITS>
static void radix_pass(hipStream_t st, const BwtArgs &a, u32 shift, u32 h, const u32 *Kin, const u32 *Vin,
                       u32 *Kout, u32 *Vout, u64 elems, KernelProf *prof, u32 *Ktmp = nullptr, u32 tiles = kTilesPerBlock)
{
    const dim3 grid(tiles, xcd_grid_y(a.nb));
    const u64 rd_hist = (SRC == SRC_TEXT) ? 1 : (SRC == SRC_PAIRS ? 4 : (SRC == SRC_WALK ? 5 : 8));
    const u64 rd_scat = (SRC == SRC_TEXT) ? 1 : (SRC == SRC_WALK ? 5 : 8); // SURV / LISTG: list + rank = 8
    int p = prof ? prof->begin(st, KID_RADIX_HIST, elems * rd_hist) : -1;
    hipLaunchKernelGGL((k_radix_hist<SRC, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Kin, Vin, Ktmp);
    if (prof) prof->end(st, p);
    p = prof ? prof->begin(st, KID_RADIX_SCAN, (u64)a.nb * kTilesPerBlock * (1u << BITS) * 8u) : -1;
    hipLaunchKernelGGL((k_radix_scan<SRC, BITS>), dim3(a.nb), dim3(kSortThreads), 0, st, a);
    if (prof) prof->end(st, p);
    p = prof ? prof->begin(st, KID_RADIX_SCATTER, elems * (rd_scat + 8)) : -1;
    if (SRC == SRC_TEXT && Ktmp)
        hipLaunchKernelGGL((k_radix_scatter<SRC_TEXTK, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Ktmp,
                           Vin, Kout, Vout);
    else if (SRC == SRC_WALK && Ktmp)
        hipLaunchKernelGGL((k_radix_scatter<SRC_WALKK, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Ktmp,
                           Vin, Kout, Vout);
    else if (SRC == SRC_MM && Ktmp)
        hipLaunchKernelGGL((k_radix_scatter<SRC_MMK, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Ktmp,
                           Vin, Kout, Vout);
    else if (SRC == SRC_PERJ && Ktmp) // (bit 31 of a kept key: a member of a ranked group, final already)
        hipLaunchKernelGGL((k_radix_scatter<SRC_PERJK, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Ktmp,
                           Vin, Kout, Vout);
    else if ((SRC == SRC_SURV || SRC == SRC_LISTG) && Ktmp) // the gathered keys were kept: a plain pair list now
        hipLaunchKernelGGL((k_radix_scatter<SRC_PAIRS, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Ktmp,
                           Vin, Kout, Vout);
    else
        hipLaunchKernelGGL((k_radix_scatter<SRC, BITS>), grid, dim3(kSortThreads), 0, st, a, shift, h, Kin, Vin,
                           Kout, Vout);
    if (prof) prof->end(st, p);
}

// the tag of the next fused pass; the look-back words and tickets are cleared when the counter wraps
static u32 next_epoch(hipStream_t st, const BwtArgs &a)
{
    u32 e = *a.epoch + 1u;
    if (e >= kSortEpochs) {
        (void)hipMemsetAsync(a.tile_state_all, 0, a.tile_state_bytes, st);
        (void)hipMemsetAsync(a.tickets, 0, (size_t)kSortEpochs * 8 * 4, st);
        e = 1;
    }
    *a.epoch = e;
    return e;
}

template <int SRC, int BITS, bool WRITE_K = true, bool PACK_OUT = false>
static void fused_pass(hipStream_t st, const BwtArgs &a, u32 shift, u32 h, const u32 *Kin, const u32 *Vin, u32 *Kout,
                       u32 *Vout, u32 dpos, u64 elems, KernelProf *prof, u64 out_elems = ~0ull,
                       const u32 *gate = nullptr)
{
    const dim3 grid(a.tiles, xcd_grid_y(a.nb));
    const u64 rd = (SRC == SRC_TEXTK || SRC == SRC_PACKED) ? 4 : ((SRC == SRC_TEXT) ? 1 : ((SRC == SRC_WALK) ? 5 : (SRC == SRC_MMC ? 9 : 8)));
    const u32 e = next_epoch(st, a);
    if (out_elems == ~0ull) out_elems = elems;
    const int p = prof ? prof->begin(st, KID_RADIX_SCATTER_LB, elems * rd + out_elems * ((WRITE_K && !PACK_OUT) ? 8 : 4)) : -1;
    hipLaunchKernelGGL((k_radix_scatter_lb<SRC, BITS, WRITE_K, PACK_OUT>), grid, dim3(kSortTile / 16), 0, st, a, shift, h, Kin, Vin, Kout, Vout, dpos, e, gate);
    if (prof) prof->end(st, p);
}

// did the fused pass tagged `epoch` hand out every tile on every XCD, and did no look-back give up?
// (one small copy and a stream synchronisation)
static bool fused_pass_ok(hipStream_t st, const BwtArgs &a, u32 epoch)
{
    static const bool fail_test = getenv("BZ_ONESWEEP_FAILTEST") != nullptr; // (tests: exercise the fallback)
    if (fail_test) return false;
    u32 tk[8], gave_up = 0;
    {
        const MailSeg sg[2] = {{tk, a.tickets + (size_t)epoch * 8u, sizeof(tk)}, {&gave_up, a.sort_err, 4}};
        if (mail_fetch(st, sg, 2) != 0) return false;
    }
    if (gave_up) return false;
    // Every workgroup of the launch draws one ticket from the counter of the XCD it runs on, and the launch has
    // a.tiles * xcd_grid_y(nb) workgroups dealt evenly to the eight XCDs: each counter must stand at EXACTLY
    // that share.  Fewer: an XCD ran fewer workgroups than it has tiles (tiles left out).  More: tickets were drawn
    // twice -- the counters were cleared while the pass ran, or two launches shared an epoch -- and tiles ran twice.
    const u32 want = a.tiles * (xcd_grid_y(a.nb) / 8u);
    for (u32 x = 0; x < 8; ++x)
        if (tk[x] != want) return false;
    return true;
}

// false: the first pass did not behave (workgroups not dealt evenly to the XCDs, or a look-back gave
// up); nothing of it is used then and the caller sorts with the three-kernel passes
template <int B0, int B1, int B2>
static bool init_sort_fused(hipStream_t st, const BwtArgs &a, u64 total_n, KernelProf *prof)
{
    const dim3 grid(a.tiles, xcd_grid_y(a.nb));
    // the digit counts of key(j), once: both init phases sort the same multiset of keys
    // (the keys are not kept: with the packed text a key costs one load, less than a stored key's write and read)
    int p = prof ? prof->begin(st, KID_GHIST_TEXT, total_n * 1) : -1;
    hipLaunchKernelGGL((k_ghist_text<B0, B1, B2>), dim3((a.tiles + kGhSpan - 1u) / kGhSpan, xcd_grid_y(a.nb)), dim3(kSortThreads), 0, st, a, (u32 *)nullptr);
    if (prof) prof->end(st, p);
    p = prof ? prof->begin(st, KID_GHIST_SCAN, (u64)a.nb * kTilesPerBlock * 3 * 1024 * 4 / kGhSpan) : -1;
    hipLaunchKernelGGL(k_ghist_scan, dim3(a.nb), dim3(kSortThreads), 0, st, a, 3u, 1u << B0, 1u << B1, 1u << B2, 0u, kGhSpan);
    if (prof) prof->end(st, p);
    // phase A: order by key(j); phase B: walk it, order by key(j - c) -> 2c symbols
    // (tests) BZ_TEST_STALE_TICKETS=1: the ticket counters of the first pass already stand at a full launch's count (a
    // counter that was not cleared): every workgroup draws a ticket beyond its XCD's tiles and leaves, no tile is
    // sorted, and each counter ends at twice its share -- a ">= tiles" check is satisfied, "== share" is not.  (The
    // first pass, because its check comes before anything reads what the pass wrote.)
    static const bool stale_tickets_test = getenv("BZ_TEST_STALE_TICKETS") && atoi(getenv("BZ_TEST_STALE_TICKETS")) != 0;
    if (stale_tickets_test && *a.epoch + 1u < kSortEpochs)
        (void)hipMemsetD32Async((hipDeviceptr_t)(a.tickets + (size_t)(*a.epoch + 1u) * 8u),
                                (int)(a.tiles * (xcd_grid_y(a.nb) / 8u)), 8, st);
    fused_pass<SRC_TEXT, B0>(st, a, 0, 0, nullptr, nullptr, a.KA, a.VA, 0, total_n, prof);
    if (!fused_pass_ok(st, a, *a.epoch)) return false;
    // (round 4) the middle pass of phase A writes ONE word per element -- the last digit above bit 20, the rotation below:
    // the first two digits are used up, and 10 + 20 bits fit a word -- and the last pass reads that word: 8 of the 32
    // bytes the two passes moved per element, and one of the last pass's two trips through LDS.
    fused_pass<SRC_PAIRS, B1, true, true>(st, a, B0, 0, a.KA, a.VA, nullptr, a.VB, 1, total_n, prof);
    (void)hipMemsetAsync(a.pb_gate, 1, a.nb * sizeof(u32), st); // every block: keys in KB, order by the passes
    fused_pass<SRC_PACKED, B2, false>(st, a, 20, 0, nullptr, a.VB, nullptr, a.VA, 2, total_n, prof); // (order only)
    const u32 *gate = nullptr;
    fused_pass<SRC_WALK, B0>(st, a, 0, 0, a.KA, a.VA, a.KB, a.VB, 0, total_n, prof, ~0ull, gate);
    fused_pass<SRC_PAIRS, B1>(st, a, B0, 0, a.KB, a.VB, a.KA, a.VA, 1, total_n, prof, ~0ull, gate);
    // the last pass puts the order straight into SA: the first refinement leaves every rotation where it is
    fused_pass<SRC_PAIRS, B2>(st, a, B0 + B1, 0, a.KA, a.VA, a.KB, a.SA, 2, total_n, prof, ~0ull, gate);
    return true;
}

void launch_block_symbols(hipStream_t st, const BwtArgs &a, u32 *inuse_bits, u8 *sym_code, u8 *keyinfo)
{
    const dim3 grid(kTilesPerBlock, xcd_grid_y(a.nb));
    (void)hipMemsetAsync(inuse_bits, 0, (size_t)a.nb * 8 * sizeof(u32), st);
    hipLaunchKernelGGL(k_block_symbols, dim3((a.tiles + kSymSpan - 1u) / kSymSpan, xcd_grid_y(a.nb)), dim3(kSortThreads), 0, st, a, inuse_bits);
    hipLaunchKernelGGL(k_key_params, dim3(a.nb), dim3(256), 0, st, inuse_bits, sym_code, keyinfo);
    hipLaunchKernelGGL(k_pack_text, grid, dim3(kSortThreads), 0, st, a, a.ptext);
}

template <int B0, int B1, int B2>
static void init_sort(hipStream_t st, const BwtArgs &a, u64 total_n, KernelProf *prof)
{
    // phase A: order by key(j); phase B: walk it, order by key(j - c) -> 2c symbols
    radix_pass<SRC_TEXT, B0>(st, a, 0, 0, nullptr, nullptr, a.KA, a.VA, total_n, prof, a.KB);
    radix_pass<SRC_PAIRS, B1>(st, a, B0, 0, a.KA, a.VA, a.KB, a.VB, total_n, prof);
    radix_pass<SRC_PAIRS, B2>(st, a, B0 + B1, 0, a.KB, a.VB, a.KA, a.VA, total_n, prof);
    radix_pass<SRC_WALK, B0>(st, a, 0, 0, a.KA, a.VA, a.KB, a.VB, total_n, prof, a.KA);
    radix_pass<SRC_PAIRS, B1>(st, a, B0, 0, a.KB, a.VB, a.KA, a.VA, total_n, prof);
    radix_pass<SRC_PAIRS, B2>(st, a, B0 + B1, 0, a.KA, a.VA, a.KB, a.SA, total_n, prof); // (the order goes straight into SA)
}

// Sorts the rotations of every block of the batch; leaves the order in a.SA.
// h_active is a pinned host word used to poll the per-round counters.  total_n = sum of block
// lengths (for the profiler's byte accounting).  wide_keys: some block uses more than 128
// symbols (32-bit keys, 11+11+10 bit digits; otherwise 30-bit keys, 10+10+10).  min_chars: the
// smallest symbols-per-key of the batch.
// Returns the number of doubling rounds executed, <0 on HIP error.
// One attempt.  Returns the rounds executed, -1 on a HIP error, -2 when a fused pass did not behave
// (a look-back gave up or an XCD handed out fewer tickets than it has tiles -- another kernel shared
// the device, or workgroups were not dealt to the XCDs as assumed): nothing of the attempt can be
// trusted then, but the block text is untouched and the caller sorts again with the three-kernel passes.
// (a.fused_state is the engine's: an engine whose fused passes misbehaved once keeps off them; other engines --
// the other lane of a host context, other devices -- are not affected)

static int run_bwt_once(hipStream_t st, const BwtArgs &a_in, u32 max_n, u64 total_n, unsigned long long *h_active,
                        u64 *sorted_elems, KernelProf *prof, u64 *round_active, bool wide_keys, u32 min_chars,
                        bool allow_fused)
{
    BwtArgs a = a_in;
    if (!allow_fused || a.fused_state[0]) a.fused = 0;
    // A launch covers the tiles the batch's LARGEST block needs, not a level-9 block's 110: at level 1 (100 KB blocks,
    // 13 tiles) seven of eight workgroups of every launch had nothing to do.  Array strides stay kTilesPerBlock.
    a.tiles = std::min<u32>(kTilesPerBlock, std::max<u32>(1u, (max_n + kSortTile - 1u) / kSortTile));
    const dim3 grid(a.tiles, xcd_grid_y(a.nb));
    (void)hipMemsetAsync(a.active, 0, 64 * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(a.maxnf, 0, 64 * sizeof(u32), st);
    (void)hipMemsetAsync(a.nonfinal, 0, a.nb * sizeof(u32), st);
    if (a.sort_err) (void)hipMemsetAsync(a.sort_err, 0, 4, st); // a failure of an earlier call must not stick

    // Fused passes (k_radix_scatter_lb) unless switched off (BZ_ONESWEEP=0) or found not to behave: their
    // first pass is checked, and the three-kernel passes redo the sort from the block if it fails.
    bool fused = a.fused != 0;
    const u32 epoch_first = *a.epoch + 1u; // (fused) the passes of this call, for the coverage check below
    if (fused) {
        // (phase B of the init inside LDS was built in round 2 and took 37 ms per GiB of text against 15 ms for the three
        // global passes it replaces: profiles/r02_phase_b_in_lds.md; its machinery lives on in k_surv_local)
        const bool ok = wide_keys ? init_sort_fused<11, 11, 10>(st, a, total_n, prof) : init_sort_fused<10, 10, 10>(st, a, total_n, prof);
        if (!ok) {
            fprintf(stderr, "bz2_mi355x: fused radix passes disabled (tile tickets / look-back check failed)\n");
            a.fused_state[0] = 1;
            a.fused_state[1] += 1;
            fused = false;
            a.fused = 0;
            (void)hipMemsetAsync(a.sort_err, 0, 4, st);
        }
    }
    if (!fused) {
        (void)hipMemsetAsync(a.pb_gate, 1, a.nb * sizeof(u32), st);
        if (wide_keys) init_sort<11, 11, 10>(st, a, total_n, prof);
        else init_sort<10, 10, 10>(st, a, total_n, prof);
    }
    // flags + apply in one pass (k_group_refine) whenever the fused radix passes are in use; BZ_FUSED_REFINE=0 keeps
    // the two kernels (A/B measurements, tests)
    static const bool want_refine = !(getenv("BZ_FUSED_REFINE") && atoi(getenv("BZ_FUSED_REFINE")) == 0);
    const bool refine = fused && want_refine;
    int p;
    (void)hipMemsetAsync(a.bin_cursor, 0, (size_t)a.nb * 1024 * sizeof(u32), st);
    if (refine) {
        const u32 e = next_epoch(st, a);
        p = prof ? prof->begin(st, KID_GROUP_REFINE, total_n * 14) : -1; // K 4, V 4, flag byte 1, rank word 4, last column 1
        hipLaunchKernelGGL((k_group_refine<true>), grid, dim3(kSortThreads), 0, st, a, 0u, 0u, 0u, a.KB, a.SA, a.KA, e);
        if (prof) prof->end(st, p);
    } else {
        p = prof ? prof->begin(st, KID_GROUP_FLAGS, total_n * 9) : -1;
        hipLaunchKernelGGL((k_group_flags<true>), grid, dim3(kSortThreads), 0, st, a, 0u, a.KB, a.SA);
        if (prof) prof->end(st, p);
        p = prof ? prof->begin(st, KID_GROUP_APPLY, total_n * 13) : -1;
        hipLaunchKernelGGL((k_group_apply<true>), grid, dim3(kSortThreads), 0, st, a, 0u, 0u, a.KB, a.SA, a.KA);
        if (prof) prof->end(st, p);
    }
    p = prof ? prof->begin(st, KID_RANK_PLACE, total_n * 8) : -1;
    hipLaunchKernelGGL(k_rank_place, grid, dim3(kSortThreads), 0, st, a, a.KA);
    if (prof) prof->end(st, p);

    u32 *cK = a.KB, *cV = a.VB; // the list the last refinement ran on (after the init: keys in KB, order in SA)
    u32 *fK = a.KA, *fV = a.VA; // the free pair
    const u32 *lastV = a.SA;    // the order that refinement ran on
    u32 step = 0; // this round compares at depth h = 2c << step
    int rounds = 0;
    u32 slot = 0;
    bool period_done = false;
    // (round 6) link rounds: behind the period round, whenever a doubling round has just run and enough is left unordered --
    // until one of them orders less than a quarter of what it was given (the groups that are left are large)
    bool last_doubled = false, links_pay = true;
    u64 link_m = 0; // what the last link round was given (0: the last round was none)
    if (!(a.gh_tiles && a.per_aux)) a.per_links = 0; // (only with the fused passes' workspace: the period tables then live in the digit counts' slot)
    const u32 links_on = a.per_links;
    bool surv_local_ok = true; // (no segment of a survivor round of this sort has overflowed LDS so far)
    u32 list_tiles = a.tiles; // (the first refinement ran on all of SA)
    while (true) {
        {
            const MailSeg sg[2] = {{h_active, a.active + slot, sizeof(unsigned long long)}, {h_active + 1, a.maxnf + slot, sizeof(u32)}};
            if (mail_fetch(st, sg, 2) != 0) return -1;
        }
        const u64 m = *h_active; // rotations still to be ordered (in unfinished blocks)
        // tiles of the longest list of the coming round (its members are the rotations the last refinement left
        // unordered), and of the list that refinement ran on
        const u32 list_tiles_prev = list_tiles;
        const u32 mx = *reinterpret_cast<const u32 *>(h_active + 1); // the most rotations any block is left with
        {
            list_tiles = (mx + kSortTile - 1) / kSortTile + 1u;
            if (list_tiles > kTilesPerBlock) list_tiles = kTilesPerBlock;
        }
        const dim3 grid_list(list_tiles, xcd_grid_y(a.nb)), grid_prev(list_tiles_prev, xcd_grid_y(a.nb));
        if (sorted_elems) *sorted_elems += m;
        if (round_active && slot < 64) round_active[slot] += m;
        if (m == 0 || step > 24 || ((u64)(2u * min_chars) << step) >= max_n) break;
        ++slot;
        ++rounds;
        bool carried = false;
        u32 trace_worst = 0;
        static const bool bwt_trace = getenv("BZ_BWT_TRACE") != nullptr;
        if (bwt_trace) { // (which block holds the most: a.nonfinal still has the last refinement's counts)
            std::vector<u32> nf(a.nb);
            (void)hipMemcpyAsync(nf.data(), a.nonfinal, a.nb * 4, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            trace_worst = (u32)(std::max_element(nf.begin(), nf.end()) - nf.begin());
        }
        (void)hipMemsetAsync(a.nonfinal, 0, a.nb * sizeof(u32), st);
        // SOME block with most of its rotations still unordered after a doubling round (4c symbols and more compared;
        // text is down to 7 % by then): deep repeats.  One period round (k_period_find ...) finishes the groups of the
        // blocks that have a linear period; what it cannot take goes on doubling, and blocks without a period pay a
        // survivor round that leaves their groups as they are (O(their survivors)).  The test is per BLOCK (round 4;
        // rounds 1-3 asked for three quarters of the whole BATCH, so a batch in which every third block was a deep
        // repeat took seventeen full-width walk rounds for everybody -- the reference's SA-IS costs the same whatever
        // the data, sais.rs:127-264): the largest per-block count against the largest block.  BZ_PERIOD_ROUND=batch
        // restores the old trigger, =0 turns the round off.
        static const int want_period = [] {
            const char *e = getenv("BZ_PERIOD_ROUND");
            return !e ? 1 : (e[0] == 'b' ? 2 : (atoi(e) != 0 ? 1 : 0));
        }();
        // (round 5: a batch that is unordered to the last rotation behind the init -- 63 of 64: a paragraph repeated; text
        // stands at 56 % there, the corpus "binary" at 96 % -- takes the period round at once instead of behind a full-width
        // walk round that orders nothing.  Only when the WHOLE batch looks like that: a period round in front of the walk
        // round costs the blocks that do not need it four passes over their survivors and the carried ranks of that round,
        // and at 8-12 symbols most groups of data that merely holds a stretch twice are still a mixture of both copies'
        // neighbourhoods: binary loses 2 % with the round at h = 8 instead of 16.)
        const bool deep = want_period == 2 ? (rounds >= 3 && m * 4 >= total_n * 3)
                                           : ((rounds >= 2 && ((u64)mx * 4 >= (u64)max_n * 3 || m * 4 >= total_n * 3)) ||
                                              (rounds == 1 && m * 64 >= total_n * 63));
        const bool per_round = want_period != 0 && !period_done && deep;
        // A LINK round (round 6; round 5 had a pair round at 32 symbols here): the small groups of the survivor list -- two to
        // kLinkMax members -- ranked by direct comparison (k_link_scan), no tables, no radix passes.  Data that holds a stretch
        // several times leaves most of its rotations in such groups once a doubling round or two have told the copies'
        // neighbourhoods apart; their members agree for kilobytes, which is eight and more doubling rounds.  Taken whenever
        // a doubling round has just run behind the period round (so: only batches that are deep in repeats) and a 256th of
        // the rotations is still unordered, as long as the rounds pay.
        if (link_m) {
            if ((link_m - m) * 4 < link_m) links_pay = false;
            link_m = 0;
        }
        const bool link_round = !per_round && links_on && period_done && last_doubled && links_pay && m * 256 >= total_n;
        // survivor form below this share of the rotations (1/8 ... 1/2 measured in round 5: flat)
        constexpr u64 surv_num = 1, surv_den = 4;
        if (bwt_trace)
            fprintf(stderr, "bz2_mi355x: sort round %d (h = %llu): %llu of %llu rotations unordered, at most %u in one block (block %u; of %u)%s\n",
                    rounds, (unsigned long long)(2u * min_chars) << step, (unsigned long long)m, (unsigned long long)total_n, mx,
                    trace_worst, max_n, per_round ? ": period round" : (link_round ? ": link round" : (m * surv_den < total_n * surv_num ? ": survivor form" : ": walk form")));
        u8 *impure = a.ptext; // (the packed text is not read any more once the init is over)
        last_doubled = !(per_round || link_round);
        if (per_round || link_round) {
            if (per_round) period_done = true;
            else link_m = m;
            (void)hipMemsetAsync(impure, 0, (size_t)a.nb * kSlot, st);
            // The members of small groups leave their place in the compacted list under their start (midx: the keys of the
            // list the last refinement ran on, which nothing reads any more); k_link_scan ranks them, k_link_finalize makes
            // the members of decided groups final, and the rest of the round runs on what is left.  BZ_LINK_ROUND=0: none
            // of it (the period round keeps its tables and chains).
            // (a period round taken at once -- the batch unordered to the last rotation behind the init: a paragraph repeated --
            // meets groups of hundreds, none of eight: clearing midx and the compaction's extra work cost 2 % of such a batch)
            const bool links = links_on != 0 && !(per_round && rounds == 1);
            a.per_links = links ? 1u : 0u; // (the round's kernels look at the group bytes only when the round has written them)
            u32 *midx = cK;
            if (links) (void)hipMemsetAsync(midx, 0, (size_t)a.nb * kSlot * sizeof(u32), st); // (0: the start is no member of a small group)
            hipLaunchKernelGGL(k_survivor_compact, grid_prev, dim3(kSortThreads), 0, st, a, lastV, fV, links ? midx : nullptr, per_round ? 1u : 0u);
            hipLaunchKernelGGL(k_copy_counts, dim3((a.nb + 255) / 256), dim3(256), 0, st, a.count, a.count2, a.nb);
            // (the pairs' verdicts: a word per list entry in the free key array, which nothing else touches before the passes)
            if (links) hipLaunchKernelGGL(k_link_scan, grid, dim3(kSortThreads), 0, st, a, step, fV, midx, impure, fK);
            if (links) hipLaunchKernelGGL(k_link_finalize, grid_list, dim3(kSortThreads), 0, st, a, fV, impure, fK);
            if (link_round) {
                // no tables, no passes: what the ranked groups leave behind, in list order (count2: the whole list, count: the rest)
                (void)hipMemsetAsync(a.lin_p, 0, (size_t)a.nb * kPerK * 4, st);
                hipLaunchKernelGGL(k_link_compact, grid_list, dim3(kSortThreads), 0, st, a, fV, cK, cV);
            } else {
            // the blocks' distances and the bitmaps that order rotation i against rotation i + p (in the digit counts' slot)
            hipLaunchKernelGGL(k_period_find, dim3(a.nb), dim3(kSortThreads), 0, st, a, 3u);
            hipLaunchKernelGGL(k_period_bits, grid, dim3(kSortThreads), 0, st, a);
            hipLaunchKernelGGL(k_period_next, dim3(a.nb, per_kmax(a)), dim3(kSortThreads), 0, st, a);
            // The start-based keys are heuristic (k_period_mark decides): when every listed distance of the batch is 1024 and more
            // -- a paragraph repeated every 4 KiB --, the members of a group differ above bit 10 and the keys go down by ten bits:
            // ranks, pair verdicts and shifted starts all fit ONE digit, and the second pass over the key is left out (a pass over
            // the whole list: 4.4 of T2's 119 ms per GiB).
            bool one_pass = false, wide = false;
            {
                std::vector<u32> lp((size_t)a.nb * kPerK), ls((size_t)a.nb * kPerK);
                {
                    const MailSeg sg[2] = {{lp.data(), a.lin_p, lp.size() * 4}, {ls.data(), a.lin_sig, ls.size() * 4}};
                    if (mail_fetch(st, sg, 2) != 0) return -1;
                }
                one_pass = true;
                u32 small = 0, smallest = 0xFFFFFFFFu;
                // (a distance below 1024 that a fifth of its block agrees at; the ones that barely made the list -- a paragraph
                // repeated lists a chance distance at 6 % in one block of forty -- only leave their few groups to the doubling)
                // (round 6) a distance that is not larger than the depth reached -- sixteen symbols per key at most -- means
                // stretches with a short period, whose groups are split by where the stretches end (per_ekey): thirty-bit keys
                const u64 depth_max = 16ull << step;
                for (size_t i = 0; i < lp.size(); ++i) {
                    if (lp[i] != 0u && lp[i] < 1024u && ls[i] >= 200u) {
                        one_pass = false;
                        small += 1;
                        smallest = lp[i] < smallest ? lp[i] : smallest;
                    }
                    if (lp[i] != 0u && lp[i] <= depth_max) wide = true;
                }
                if (wide) one_pass = false;
                if (bwt_trace) fprintf(stderr, "  period round: %s (%u distances below 1024 that a fifth of their block agrees at, the smallest %u)\n", wide ? "three passes over thirty-bit keys" : (one_pass ? "one pass over the keys" : "two passes over the keys"), small, small ? smallest : 0u);
            }
            a.per_keyshift = one_pass ? 10u : 0u;
            a.per_wide = wide ? 1u : 0u;
            if (one_pass) {
                radix_pass<SRC_PERJ, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
                radix_pass<SRC_LISTG, 10>(st, a, 0, step, nullptr, cV, fK, cK, m, prof, fV, list_tiles); // (values cV -> cK, keys kept in fV)
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, fK, cK, cV, fV, m, prof, nullptr, list_tiles); // (result: keys in cV, values in fV)
                u32 *rk = cV, *rv = fV, *ok = fK, *ov = cK;
                cK = rk; cV = rv; fK = ok; fV = ov;
            } else if (wide) {
                radix_pass<SRC_PERJ, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, cK, cV, fK, fV, m, prof, nullptr, list_tiles);
                radix_pass<SRC_PAIRS, 10>(st, a, 20, step, fK, fV, cK, cV, m, prof, nullptr, list_tiles);
                radix_pass<SRC_LISTG, 10>(st, a, 0, step, nullptr, cV, fK, fV, m, prof, cK, list_tiles);
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, fK, fV, cK, cV, m, prof, nullptr, list_tiles); // (the result is in cK, cV)
            } else {
                radix_pass<SRC_PERJ, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, cK, cV, fK, fV, m, prof, nullptr, list_tiles);
                radix_pass<SRC_LISTG, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, cK, cV, fK, fV, m, prof, nullptr, list_tiles);
                u32 *t = cK; cK = fK; fK = t;
                t = cV; cV = fV; fV = t;
            }
            a.per_keyshift = 0;
            hipLaunchKernelGGL(k_period_mark, grid_list, dim3(kSortThreads), 0, st, a, step, cK, cV, impure);
            }
        } else if (m * surv_den < total_n * surv_num) {
            // few survivors: compact them (list order = sorted by group), order them by the rank
            // of rotation j+h (2 passes), then stably by their own group (2 passes): O(m) work
            hipLaunchKernelGGL(k_survivor_compact, grid_prev, dim3(kSortThreads), 0, st, a, lastV, fV);
            hipLaunchKernelGGL(k_copy_counts, dim3((a.nb + 255) / 256), dim3(256), 0, st, a.count, a.count2, a.nb);
            // The list inside LDS, segment by segment (k_surv_local) -- unless a segment of an earlier round
            // of this sort did not fit: then, and when one of this round does not, the four global passes.
            bool local_done = false;
            // (only lists of less than an eighth of the rotations: text's are 7 % and 0.001 %; a list of 18 % -- the corpus "binary"
            // behind its walk rounds -- still holds groups that do not fit, and the failed attempt costs 3 ms per 256 MiB)
            if (surv_local_ok && m * 8 < total_n) { // (cK, cV are free: the compaction above was the last reader of the old list)
                u32 fails = 0;
                (void)hipMemsetAsync(a.loc_stats + LOC_STAT_SURV_FAIL, 0, 4, st);
                hipLaunchKernelGGL(k_surv_local, grid_list, dim3(kLocThreads), 0, st, a, step, fV, cK, cV);
                {
                    const MailSeg sg = {&fails, a.loc_stats + LOC_STAT_SURV_FAIL, 4};
                    if (mail_fetch(st, &sg, 1) != 0) return -1;
                }
                local_done = fails == 0;
                if (!local_done) surv_local_ok = false;
                if (bwt_trace) fprintf(stderr, "  survivor round inside LDS: %s\n", local_done ? "done" : "a segment did not fit or failed its check: the global passes");
            }
            if (!local_done) {
            // (the ranks a histogram kernel gathers are kept in the free key array for its scatter kernel: with few
            // survivors per block the rank arrays of the blocks in flight do not fit the L2, a gather costs a sector)
            radix_pass<SRC_SURV, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
            radix_pass<SRC_PAIRS, 10>(st, a, 10, step, cK, cV, fK, fV, m, prof, nullptr, list_tiles);
            radix_pass<SRC_LISTG, 10>(st, a, 0, step, nullptr, fV, cK, cV, m, prof, fK, list_tiles);
            radix_pass<SRC_PAIRS, 10>(st, a, 10, step, cK, cV, fK, fV, m, prof, nullptr, list_tiles);
            u32 *t = cK; cK = fK; fK = t;
            t = cV; cV = fV; fV = t;
            }
        } else {
            // pass A walks all of SA (total_n), pass B and the refinement only the m survivors
            if (fused) {
                // the keys are the group heads of the rotations that are not final: k_group_apply counted them
                hipLaunchKernelGGL(k_ghist_scan, dim3(a.nb), dim3(kSortThreads), 0, st, a, 2u, 1024u, 1024u, 0u, 1u);
                // the round behind the first refinement: its flags lie in SA order, the rank of rotation j+h travels
                // with the element and the refinement gathers nothing (k_group_refine<false>)
                carried = refine && rounds == 1;
                if (carried) fused_pass<SRC_MMC, 10>(st, a, 0, step, nullptr, nullptr, fK, fV, 0, total_n, prof, m);
                else fused_pass<SRC_MM, 10>(st, a, 0, step, nullptr, nullptr, fK, fV, 0, total_n, prof, m);
                fused_pass<SRC_PAIRS, 10>(st, a, 10, step, fK, fV, cK, cV, 1, m, prof);
            } else {
                radix_pass<SRC_MM, 10>(st, a, 0, step, nullptr, nullptr, fK, fV, total_n, prof, cK);
                radix_pass<SRC_PAIRS, 10>(st, a, 10, step, fK, fV, cK, cV, m, prof);
            }
        }
        (void)hipMemsetAsync(a.bin_cursor, 0, (size_t)a.nb * 1024 * sizeof(u32), st);
        if (per_round || link_round) {
            // (the comparison depth does not move: the next round doubles from where the last one stood)
            p = prof ? prof->begin(st, KID_GROUP_FLAGS, m * 13) : -1;
            hipLaunchKernelGGL((k_group_flags<false>), grid_list, dim3(kSortThreads), 0, st, a, step, cK, cV, impure);
            a.per_wide = 0;
            a.per_links = links_on;
            if (prof) prof->end(st, p);
            p = prof ? prof->begin(st, KID_GROUP_APPLY, m * 17) : -1;
            hipLaunchKernelGGL((k_group_apply<false>), grid_list, dim3(kSortThreads), 0, st, a, step, slot, cK, cV, fK);
            if (prof) prof->end(st, p);
        } else if (carried) {
            const u32 e = next_epoch(st, a);
            p = prof ? prof->begin(st, KID_GROUP_REFINE, m * 18) : -1; // K 4, V 4, flag 1, SA 4, rank word 4, column 1
            hipLaunchKernelGGL((k_group_refine<false>), grid, dim3(kSortThreads), 0, st, a, step, step + 1, slot, cK, cV, fK, e);
            if (prof) prof->end(st, p);
        } else {
            p = prof ? prof->begin(st, KID_GROUP_FLAGS, m * 13) : -1;
            hipLaunchKernelGGL((k_group_flags<false>), grid_list, dim3(kSortThreads), 0, st, a, step, cK, cV);
            if (prof) prof->end(st, p);
            p = prof ? prof->begin(st, KID_GROUP_APPLY, m * 17) : -1;
            hipLaunchKernelGGL((k_group_apply<false>), grid_list, dim3(kSortThreads), 0, st, a, step + 1, slot, cK, cV, fK);
            if (prof) prof->end(st, p);
        }
        p = prof ? prof->begin(st, KID_RANK_PLACE, m * 8) : -1;
        hipLaunchKernelGGL(k_rank_place, grid, dim3(kSortThreads), 0, st, a, fK);
        if (prof) prof->end(st, p);
        if (per_round && bwt_trace) {
            std::vector<u32> lp((size_t)a.nb * kPerK), ls((size_t)a.nb * kPerK), nf(a.nb), cn(a.nb);
            (void)hipMemcpyAsync(lp.data(), a.lin_p, (size_t)a.nb * kPerK * 4, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(ls.data(), a.lin_sig, (size_t)a.nb * kPerK * 4, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(nf.data(), a.nonfinal, a.nb * 4, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(cn.data(), a.count, a.nb * 4, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            for (u32 b = 0; b < a.nb; ++b)
                if (lp[(size_t)b * kPerK] || nf[b] * 2u > max_n) {
                    fprintf(stderr, "  block %u: distances (agreement in permille of the block)", b);
                    for (u32 i = 0; i < kPerK && lp[(size_t)b * kPerK + i]; ++i) fprintf(stderr, " %u (%u)", lp[(size_t)b * kPerK + i], ls[(size_t)b * kPerK + i]);
                    fprintf(stderr, "; list %u, %u rotations left unordered\n", cn[b], nf[b]);
                }
        }
        lastV = cV;
        if (!per_round && !link_round) ++step;
    }
    if (fused) {
        // a look-back that gave up (it never should) must not pass for a sorted block, and every pass
        // must have handed out all its tiles on every XCD (it does unless an XCD ran no workgroup)
        u32 gave_up = 0;
        std::vector<u32> tk;
        const u32 epoch_last = *a.epoch;
        if (epoch_last >= epoch_first) // (no wrap of the pass counter in between)
            tk.resize((size_t)(epoch_last - epoch_first + 1u) * 8u);
        {
            const MailSeg sg[2] = {{&gave_up, a.sort_err, 4}, {tk.data(), a.tickets + (size_t)epoch_first * 8u, tk.size() * 4}};
            if (mail_fetch(st, sg, tk.empty() ? 1 : 2) != 0) return -1;
        }
        static const bool late_fail_test = getenv("BZ_ONESWEEP_LATEFAILTEST") != nullptr; // (tests: exercise the redo)
        bool bad = gave_up != 0 || late_fail_test;
        for (size_t i = 0; i < tk.size(); ++i)
            if (tk[i] != a.tiles * (xcd_grid_y(a.nb) / 8u)) bad = true; // (exactly: see fused_pass_ok)
        if (bad) {
            (void)hipMemsetAsync(a.sort_err, 0, 4, st);
            return -2;
        }
    }
#if defined(BZ_SCATTER_TIMING) || defined(BZ_REFINE_TIMING)
    { // (the phase timers of the instrumented builds: tools/build_variant.sh <name> -DBZ_SCATTER_TIMING / -DBZ_REFINE_TIMING)
        (void)hipStreamSynchronize(st);
#ifdef BZ_SCATTER_TIMING
        {
            u32 rt[16] = {};
            (void)hipMemcpy(rt, a.loc_stats + 32, sizeof(rt), hipMemcpyDeviceToHost);
            fprintf(stderr, "  k_radix_scatter_lb cycles/64 per phase (set-up, fetch+rank, scans, look-back, stage K, store K, stage V, store V): "
                            "streamed sources %u %u %u %u %u %u %u %u; gathering sources %u %u %u %u %u %u %u %u\n", rt[0], rt[1], rt[2], rt[3],
                    rt[4], rt[5], rt[6], rt[7], rt[8], rt[9], rt[10], rt[11], rt[12], rt[13], rt[14], rt[15]);
            (void)hipMemset(a.loc_stats + 32, 0, sizeof(rt));
            u32 lk[4] = {};
            (void)hipMemcpy(lk, a.loc_stats + 48, sizeof(lk), hipMemcpyDeviceToHost);
            fprintf(stderr, "  look-back of thread 0's digits: %u tiles, %u hops (%.2f per tile, most %u), %u repeated loads (%.2f per tile)\n", lk[2], lk[0],
                    lk[2] ? (double)lk[0] / lk[2] : 0.0, lk[3], lk[1], lk[2] ? (double)lk[1] / lk[2] : 0.0);
            (void)hipMemset(a.loc_stats + 48, 0, sizeof(lk));
        }
#endif
#ifdef BZ_REFINE_TIMING
        {
            u32 rt[16] = {};
            (void)hipMemcpy(rt, a.loc_stats + 16, sizeof(rt), hipMemcpyDeviceToHost);
            fprintf(stderr, "  k_group_refine cycles/16 per phase (set-up, list+keys, flags, look-back, apply, counts, out): first refinement "
                            "%u %u %u %u %u %u %u; walk round %u %u %u %u %u %u %u\n", rt[8], rt[9], rt[10], rt[11], rt[12], rt[13], rt[14],
                    rt[0], rt[1], rt[2], rt[3], rt[4], rt[5], rt[6]);
            (void)hipMemset(a.loc_stats + 16, 0, sizeof(rt));
        }
#endif
    }
#endif
    // periodic blocks: whatever is still non-final is a set of equal rotations
    (void)hipMemsetAsync(a.per_k, 0, a.nb * sizeof(u32), st);
    (void)hipMemsetAsync(a.per_shift, 0xFF, a.nb * sizeof(u32), st);
    hipLaunchKernelGGL(k_periodic_stats, grid, dim3(kSortThreads), 0, st, a);
    hipLaunchKernelGGL(k_periodic_place, grid, dim3(kSortThreads), 0, st, a);
    return rounds;
}

int run_bwt(hipStream_t st, const BwtArgs &a, u32 max_n, u64 total_n, unsigned long long *h_active,
            u64 *sorted_elems, KernelProf *prof, u64 *round_active, bool wide_keys, u32 min_chars)
{
    const u64 sorted0 = sorted_elems ? *sorted_elems : 0;
    u64 ra0[64];
    if (round_active)
        for (int i = 0; i < 64; ++i) ra0[i] = round_active[i];
    int r = run_bwt_once(st, a, max_n, total_n, h_active, sorted_elems, prof, round_active, wide_keys, min_chars, true);
    if (r == -2) {
        // a pass after the first one misbehaved: redo this batch's sort from the block text without them
        fprintf(stderr, "bz2_mi355x: a fused radix pass misbehaved; batch sorted again with the three-kernel passes, "
                        "fused passes disabled\n");
        a.fused_state[0] = 1;
        a.fused_state[1] += 1;
        if (sorted_elems) *sorted_elems = sorted0;
        if (round_active)
            for (int i = 0; i < 64; ++i) round_active[i] = ra0[i];
        r = run_bwt_once(st, a, max_n, total_n, h_active, sorted_elems, prof, round_active, wide_keys, min_chars, false);
        if (r == -2) r = -1;
    }
    return r;
}

} // namespace bzgpu
