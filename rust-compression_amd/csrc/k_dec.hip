// k_dec.hip -- BZip2 block DECODE kernels (SURVEY.md row a18 / BASELINE configs[3]).
//
// Reference being replaced: BZip2DecoderBase::{init_block, get_next_lfm, next}
// (src/bzip2/decoder.rs:163-581), HuffmanDecoder (src/huffman/decoder.rs:98-233),
// MtfPositionDecoder::pop (src/bzip2/mtf.rs:51-64), BitReader<Left> (src/bitio/reader.rs:70-186).
//
// A .bz2 stream is serial in two ways: blocks are not byte aligned and end where their Huffman
// data ends, and inside a block Huffman decode, inverse MTF and the inverse BWT pointer chase are
// recurrences.  What is done about each:
//   D0  every bit position is tested for the 48-bit block magic (one thread per input byte, 8
//       shifts): candidates.  D1 decodes ALL candidates in parallel (one 256-thread workgroup each:
//       256 candidate code starts per round, pointer doubling in LDS, the r-th true start in at most
//       six dependent reads); the host then links true blocks (start == previous end) and drops
//       the false positives.
//   D2  RUNA/RUNB runs and inverse MTF: the effect of a 512-symbol chunk on the MTF list is a
//       permutation of list positions, independent of the list's content; per-chunk permutations
//       are composed left to right, then every chunk is replayed by one lane.
//   D3  T vector = stable counting sort of positions by byte (the radix-pass shape of k_bwt.hip with
//       an 8-bit digit); the n-step pointer chase is cut at sample nodes (every 128th index): all
//       segments are walked once in parallel by persistent work-stealing lanes that keep the bytes
//       they pass, the ~7000 samples are ranked serially in LDS, and the kept bytes are copied to
//       their final offsets.
//   D4  RLE1 undo as function composition over 64-byte sub-tiles; block CRCs from coalesced 16-byte
//       pieces folded in GF(2).
#include "bzgpu.h"
#include "bz2_rnums.h"

namespace bzgpu {

constexpr u64 kBlockMagic = 0x314159265359ull;
constexpr u64 kEosMagic = 0x177245385090ull;

// ---- D0: magic scan ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_scan(const u8 *__restrict__ in, u64 nbytes, DecCand *__restrict__ cands,
                                                   u32 cap, u32 *__restrict__ count)
{
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    if (p >= nbytes) return;
    // 56 bits starting at byte p
    u64 wdw = 0;
#pragma unroll
    for (u32 k = 0; k < 7; ++k) wdw = (wdw << 8) | (u64)((p + k < nbytes) ? in[p + k] : 0);
#pragma unroll
    for (u32 s = 0; s < 8; ++s) {
        const u64 v = (wdw >> (8u - s)) & 0xFFFFFFFFFFFFull;
        const u32 type = (v == kBlockMagic) ? 1u : (v == kEosMagic ? 2u : 0u);
        if (type && p * 8 + s + 48 <= nbytes * 8) {
            const u32 i = atomicAdd(count, 1u);
            if (i < cap) {
                cands[i].bitpos = p * 8 + s;
                cands[i].type = type;
                cands[i].pad = 0;
            }
        }
    }
}

// ---- MSB-first bit cursor over big-endian words (zero past the end, like the reference's short reads)
struct BitCur {
    const u32 *w;
    u64 nwords;    // words that hold input bytes (the rest reads as zero)
    u32 tail_mask; // keeps the input bytes of the last word (big-endian view)
    u64 widx;      // next word to load
    u64 buf;       // left aligned
    int cnt;
    __device__ __forceinline__ void open(const u8 *in, u64 nbytes)
    {
        w = reinterpret_cast<const u32 *>(in);
        nwords = (nbytes + 3) / 4;
        const u32 r = (u32)(nbytes & 3u);
        tail_mask = r ? ~(0xFFFFFFFFu >> (8u * r)) : 0xFFFFFFFFu;
    }
    __device__ __forceinline__ u32 load(u64 i) const
    {
        if (i >= nwords) return 0u;
        const u32 x = __builtin_bswap32(w[i]);
        return (i + 1 == nwords) ? (x & tail_mask) : x;
    }
    __device__ __forceinline__ void seek(u64 bitpos)
    {
        widx = bitpos >> 5;
        const u32 sh = (u32)bitpos & 31u;
        const u32 x = load(widx);
        ++widx;
        buf = (u64)x << (32u + sh);
        cnt = 32 - (int)sh;
    }
    __device__ __forceinline__ void fill() // makes at least 32 bits available
    {
        if (cnt <= 32) {
            const u32 x = load(widx);
            ++widx;
            buf |= (u64)x << (32 - cnt);
            cnt += 32;
        }
    }
    __device__ __forceinline__ u32 peek(u32 n) { return n ? (u32)(buf >> (64u - n)) : 0u; } // n <= 32, after fill()
    __device__ __forceinline__ void skip(u32 n)
    {
        buf <<= n;
        cnt -= (int)n;
    }
    __device__ __forceinline__ u32 read(u32 n)
    {
        fill();
        const u32 v = peek(n);
        skip(n);
        return v;
    }
    __device__ __forceinline__ u64 pos() const { return widx * 32ull - (u64)cnt; }
};

// Lookup-table width.  The tables are the bulk of the kernel's LDS, and LDS decides how many blocks
// are resident at once (10 bits: 21 KB per block, 7 per CU, 1792 on the chip; 11 bits: 4 per CU --
// the 1189 blocks of a 1 GiB file then run as two rounds, twice the time).  Longer codes take the
// canonical walk.
constexpr u32 kLutBits = 10;
constexpr u16 kLutLong = 0xFFFE, kLutBad = 0xFFFF;
constexpr u32 kRingWords = 256;  // staged input window (words), power of two
constexpr u32 kOutBuf = 1024;    // staged output symbols

constexpr u32 kD1Threads = 256; // threads = candidate code starts per round
// ---- the selectors of a block's header by the whole workgroup ------------------------------------------------------------
// n_selectors unary codes (j one bits and a zero, j < n_groups: the position in a move-to-front list of the tables,
// decoder.rs:294-316).  Thread 0 took them one code per step: 18 002 dependent steps, 1.5 M of the 1.9 M cycles a header
// costs, 0.8 ms of a block's 12.5.  Here:
//   A  every thread takes 32 bits of the code string per trip: the zeros in them are code ends; a block scan of their
//      number says which selectors they end, a block max-scan of the last zero's position says where the first of them
//      began; j goes to sel[s].  A code with n_groups one bits (the reference gives up there) is noted with the smallest
//      selector number -- also when its zero lies behind the n_selectors * n_groups bits looked at (then fewer zeros
//      than selectors turn up, and the code behind the last zero is the one).
//   B  the move-to-front of the j: "take position j to the front" is a permutation of POSITIONS whatever the list holds,
//      so a thread composes the permutations of its stretch of selectors, an exclusive scan over the threads gives the
//      list in front of each stretch, and the stretch is replayed from it (the values replace the j in sel[]).
// out64[0] = (selector << 32 | first bit of its code, relative to p0) of the first bad code or ~0; out64[1] = the bit
// behind the last selector, relative to p0.  scratch: 144 words of LDS.  All threads call it; it ends with a barrier.
__device__ __forceinline__ u32 d1_perm_compose(u32 x, u32 y) // (x o y)[k] = x[y[k]]: y's move applied to the list x
{
    u32 r = 0;
#pragma unroll
    for (u32 k = 0; k < 6; ++k) {
        const u32 yk = (y >> (4u * k)) & 15u;
        r |= ((x >> (4u * yk)) & 15u) << (4u * k);
    }
    return r;
}
__device__ __forceinline__ u32 d1_move_front(u32 lst, u32 j, u32 &v)
{
    v = (lst >> (4u * j)) & 15u;
    if (j) {
        const u32 lowmask = (1u << (4u * j)) - 1u;
        lst = (lst & ~((lowmask << 4) | 15u)) | ((lst & lowmask) << 4) | v;
    }
    return lst;
}
__device__ void d1_selectors(const BitCur &bc, u8 *__restrict__ sel, u64 p0, u32 nsel, u32 ng, u32 l, u32 *scratch, u64 *out64)
{
    constexpr u32 NW = kD1Threads / 64;
    u32 *s_zsum = scratch;                 // [NW] zeros per wave
    int *s_zmax = reinterpret_cast<int *>(scratch + 8);  // [NW] last zero per wave
    u32 *s_pw = scratch + 16;              // [NW] permutation per wave
    const u32 wv = l >> 6, ln = l & 63u;
    if (l == 0) {
        out64[0] = ~0ull;
        out64[1] = 0ull;
    }
    __syncthreads();
    // ---- A
    const u32 maxbits = nsel * ng;
    u32 zdone = 0;  // selectors ended so far
    int lastz = -1; // the last zero so far (relative bit position)
    for (u32 base = 0; base < maxbits && zdone < nsel; base += kD1Threads * 32u) {
        const u64 bp = p0 + base + 32u * l;
        const u64 wi = bp >> 5;
        const u32 sh = (u32)bp & 31u;
        const u32 hw = bc.load(wi), lw = bc.load(wi + 1);
        const u32 word = sh ? ((hw << sh) | (lw >> (32u - sh))) : hw;
        const u32 zm = ~word; // bit (31 - b) set: a zero at bit b of this thread's 32 (b = 0 first)
        const u32 zc = (u32)__popc(zm);
        const int myz = zm ? (int)(base + 32u * l + 31u - (u32)__builtin_ctz(zm)) : -1;
        // scans over the wave, then over the waves
        u32 zin = zc;
        int zmx = myz;
#pragma unroll
        for (u32 d = 1; d < 64; d <<= 1) {
            const u32 a = (u32)__shfl_up((int)zin, d, 64);
            const int b = __shfl_up(zmx, d, 64);
            if (ln >= d) {
                zin += a;
                zmx = b > zmx ? b : zmx;
            }
        }
        if (ln == 63u) {
            s_zsum[wv] = zin;
            s_zmax[wv] = zmx;
        }
        __syncthreads();
        u32 zbase = zdone, ztot = 0;
        int pmax = lastz, tmax = lastz;
        for (u32 k = 0; k < NW; ++k) {
            if (k < wv) {
                zbase += s_zsum[k];
                pmax = s_zmax[k] > pmax ? s_zmax[k] : pmax;
            }
            ztot += s_zsum[k];
            tmax = s_zmax[k] > tmax ? s_zmax[k] : tmax;
        }
        const int prevlane = __shfl_up(zmx, 1, 64); // (inclusive max of the lanes in front)
        int prev = (ln ? (prevlane > pmax ? prevlane : pmax) : pmax);
        u32 sidx = zbase + zin - zc;
        u32 m = zm;
        while (m && sidx < nsel) {
            const u32 b = (u32)__builtin_clz(m);
            const int pos = (int)(base + 32u * l + b);
            const u32 j = (u32)(pos - prev - 1);
            if (j >= ng) atomicMin(reinterpret_cast<unsigned long long *>(&out64[0]), ((unsigned long long)sidx << 32) | (u32)(prev + 1));
            else sel[sidx] = (u8)j;
            if (sidx + 1u == nsel) out64[1] = (u64)(pos + 1);
            prev = pos;
            ++sidx;
            m &= ~(0x80000000u >> b);
        }
        zdone += ztot;
        lastz = tmax;
        __syncthreads(); // (the wave sums are written again in the next trip)
    }
    if (zdone < nsel && l == 0) // fewer zeros than selectors inside n_selectors * n_groups bits: the code behind the last zero is too long
        atomicMin(reinterpret_cast<unsigned long long *>(&out64[0]), ((unsigned long long)zdone << 32) | (u32)(lastz + 1));
    __syncthreads();
    if (out64[0] != ~0ull) return; // (uniform: every thread reads the same word behind the barrier)
    // ---- B
    const u32 per = (((nsel + kD1Threads - 1u) / kD1Threads) + 15u) & ~15u; // selectors per thread, whole 16-byte groups
    const u32 s0 = l * per, s1 = s0 + per < nsel ? s0 + per : nsel;
    u32 P = 0x543210u;
    for (u32 q = s0; q < s1; q += 16u) {
        const uint4 v4 = *reinterpret_cast<const uint4 *>(sel + q);
        const u32 wd[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (q + k < s1) {
                u32 v;
                P = d1_move_front(P, (wd[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu, v);
            }
        }
    }
    u32 inc = P;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)inc, d, 64);
        if (ln >= d) inc = d1_perm_compose(o, inc);
    }
    if (ln == 63u) s_pw[wv] = inc;
    __syncthreads();
    u32 E = 0x543210u; // the list in front of this thread's stretch
    for (u32 k = 0; k < wv; ++k) E = d1_perm_compose(E, s_pw[k]);
    {
        const u32 before = (u32)__shfl_up((int)inc, 1, 64);
        if (ln) E = d1_perm_compose(E, before);
    }
    for (u32 q = s0; q < s1; q += 16u) {
        const uint4 v4 = *reinterpret_cast<const uint4 *>(sel + q);
        const u32 wd[4] = {v4.x, v4.y, v4.z, v4.w};
        u32 od[4] = {0, 0, 0, 0};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            u32 v = 0;
            if (q + k < s1) E = d1_move_front(E, (wd[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu, v);
            od[k >> 2] |= v << ((k & 3u) * 8u);
        }
        *reinterpret_cast<uint4 *>(sel + q) = make_uint4(od[0], od[1], od[2], od[3]);
    }
    __syncthreads();
}

// ---- D1: header + Huffman decode of one candidate -------------------------------------------------------
// One workgroup of four waves per candidate.  Thread 0 parses the header; all threads build the
// decode tables; the symbols are then found 256 candidate code starts at a time (below).
__global__ __launch_bounds__(kD1Threads) void k_dec_block(const u8 *__restrict__ in, u64 nbytes,
                                                   const DecCand *__restrict__ cands, u32 ncand,
                                                   DecBlockInfo *__restrict__ info, u16 *__restrict__ sym_out,
                                                   u8 *__restrict__ sel_scratch)
{
    __shared__ u16 s_lut[6][1u << kLutBits];
    __shared__ u8 s_len[6][260];
    __shared__ u32 s_cnt[6][24], s_first[6][24], s_idx[6][24];
    __shared__ u16 s_perm[6][260];
    __shared__ u32 s_hdr[8]; // n_groups, alpha, n_selectors, status
    __shared__ u32 s_ring[kRingWords];
    __shared__ u16 s_out[kOutBuf];
    __shared__ u64 s_pos;
    __shared__ u16 s_j[6][kD1Threads + 4]; // s_j[k][i]: the code start 2^k symbols behind start i (256 = outside the window)
    __shared__ u16 s_e[kD1Threads];        // table entry of the code that would start at i
    __shared__ u32 s_ctl[4];               // symbols taken, bits consumed, stop reason
    __shared__ u64 s_sel64[2];
    __shared__ u8 s_sel[256];              // the selectors of the groups g .. g | 255
    const u32 c = blockIdx.x;
    if (c >= ncand) return;
    const u32 l = threadIdx.x;
#define BZ_T(i)
    DecBlockInfo &bi = info[c];
    BitCur bc;
    bc.open(in, nbytes);
    const DecCand cd = cands[c];
    u8 *sel = sel_scratch + (size_t)c * 32768u;
    if (l == 0) {
        bc.seek(cd.bitpos);
        int status = 0;
        u32 n_groups = 0, alpha = 0, n_selectors = 0;
        (void)bc.read(24);
        (void)bc.read(24); // the magic (only its first byte is compared by the reference: the host does that)
        {
            // a read at the end of the input returns the bits that are left as a SHORTER number
            // (bitio/reader.rs:70-186), not a zero-padded one
            const u64 at = bc.pos(), total = nbytes * 8ull;
            const u32 v = bc.read(32);
            const u64 avail = total > at ? total - at : 0ull;
            bi.stored_crc = avail >= 32 ? v : (avail ? v >> (32u - (u32)avail) : 0u);
        }
        if (cd.type == 2u) {
            bi.end_bit = bc.pos();
            bi.nsym = 0;
            bi.status = 0;
            s_hdr[3] = 1; // nothing more to do
        } else {
            bi.randomised = bc.read(1);
            bi.orig_ptr = bc.read(24);
            u32 in_use16 = bc.read(16);
            u32 n_in_use = 0;
            for (u32 i = 0; i < 16; ++i) {
                if ((in_use16 >> (15u - i)) & 1u) {
                    const u32 m = bc.read(16);
                    for (u32 j = 0; j < 16; ++j)
                        if ((m >> (15u - j)) & 1u) bi.seq2unseq[n_in_use++] = (u8)(i * 16u + j);
                }
            }
            bi.n_in_use = n_in_use;
            if (n_in_use == 0) status = BZ_DEC_E_DATA;                // decoder.rs:273-275
            alpha = n_in_use + 2;
            if (!status) {
                n_groups = bc.read(3);
                if (n_groups < 2 || n_groups > 6) status = BZ_DEC_E_DATA; // :283-285
            }
            if (!status) {
                n_selectors = bc.read(15);
                if (n_selectors < 1) status = BZ_DEC_E_DATA;            // :290-292
            }
            // (the selectors are taken by all threads behind the barrier below; thread 0 goes on with the coding tables then)
            s_hdr[0] = n_groups;
            s_hdr[1] = alpha;
            s_hdr[2] = n_selectors;
            s_hdr[3] = status ? 1u : 0u;
            s_pos = bc.pos();
            if (status) {
                bi.status = (u32)status;
                bi.end_bit = bc.pos();
                bi.nsym = 0;
            }
        }
    }
    __syncthreads();
    if (s_hdr[3]) return;
    d1_selectors(bc, sel, s_pos, s_hdr[2], s_hdr[0], l, &s_cnt[0][0], s_sel64);
    // (s_sel64[0]: (selector << 32 | its first bit, relative) of the first selector with n_groups one bits, or ~0;
    //  s_sel64[1]: the bit behind the last selector, relative)
    if (l == 0) {
        int status = 0;
        const u32 n_groups = s_hdr[0], alpha = s_hdr[1];
        if (s_sel64[0] != ~0ull) {
            status = BZ_DEC_E_DATA; // the reference gives up at the n_groups-th one bit of a selector (:294-316)
            bc.seek(s_pos + (s_sel64[0] & 0xFFFFFFFFull));
        } else {
            bc.seek(s_pos + s_sel64[1]);
        }
        {
            if (!status) { // coding tables (:318-348); "10" = +1, "11" = -1, "0" = next symbol
                for (u32 t = 0; t < n_groups && !status; ++t) {
                    u32 curr = bc.read(5);
                    for (u32 i = 0; i < alpha && !status; ++i) {
                        while (true) {
                            bc.fill();
                            const u32 two = bc.peek(2);
                            if ((two & 2u) == 0) {
                                bc.skip(1);
                                break;
                            }
                            if (curr < 1 || curr > 20) {
                                status = BZ_DEC_E_DATA;
                                break;
                            }
                            curr += (two & 1u) ? 0xFFFFFFFFu : 1u;
                            bc.skip(2);
                        }
                        s_len[t][i] = (u8)curr;
                    }
                }
            }
            s_hdr[3] = status ? 1u : 0u;
            s_pos = bc.pos();
            if (status) {
                bi.status = (u32)status;
                bi.end_bit = bc.pos();
                bi.nsym = 0;
            }
        }
    }
    __syncthreads();
    BZ_T(0)
    if (s_hdr[3]) return;
    const u32 n_groups = s_hdr[0], alpha = s_hdr[1], n_selectors = s_hdr[2];
    // ---- decode tables (all lanes): canonical codes, lookup table, per-length arrays for longer codes
    for (u32 i = l; i < 6 * 24; i += kD1Threads) (&s_cnt[0][0])[i] = 0;
    for (u32 i = l; i < 6u << kLutBits; i += kD1Threads) (&s_lut[0][0])[i] = kLutBad;
    __syncthreads();
    for (u32 i = l; i < n_groups * alpha; i += kD1Threads) {
        const u32 t = i / alpha, s = i - t * alpha;
        const u32 ln = s_len[t][s];
        if (ln >= 1 && ln <= 23) atomicAdd(&s_cnt[t][ln], 1u);
    }
    __syncthreads();
    if (l < n_groups) {
        u32 code = 0, idx = 0, bad = 0;
        for (u32 ln = 1; ln < 24; ++ln) {
            code = (code + s_cnt[l][ln - 1]) << 1;
            s_first[l][ln] = code;
            s_idx[l][ln] = idx;
            idx += s_cnt[l][ln];
            if ((u64)code + s_cnt[l][ln] > (1ull << ln)) bad = 1; // over-subscribed: the reference indexes out of bounds
        }
        for (u32 s = 0; s < alpha; ++s)
            if (s_len[l][s] > 23) bad = 1;
        if (bad) atomicExch(&s_hdr[3], 1u);
    }
    __syncthreads();
    if (s_hdr[3]) {
        if (l == 0) {
            bi.status = (u32)BZ_DEC_E_DATA;
            bi.end_bit = cd.bitpos;
            bi.nsym = 0;
        }
        return;
    }
    for (u32 i = l; i < n_groups * alpha; i += kD1Threads) {
        const u32 t = i / alpha, s = i - t * alpha;
        const u32 ln = s_len[t][s];
        if (ln == 0) continue; // no code for this symbol (huffman/mod.rs:31-35)
        u32 r = 0;
        for (u32 q = 0; q < s; ++q) r += (s_len[t][q] == ln) ? 1u : 0u;
        const u32 code = s_first[t][ln] + r;
        s_perm[t][s_idx[t][ln] + r] = (u16)s;
        if (ln <= kLutBits) {
            const u32 base = code << (kLutBits - ln);
            for (u32 j = 0; j < (1u << (kLutBits - ln)); ++j) s_lut[t][base + j] = (u16)(s | (ln << 9));
        } else {
            s_lut[t][code >> (ln - kLutBits)] = kLutLong;
        }
    }
    __syncthreads();
    BZ_T(1)
    // ---- the symbols (:367-437)
    // Huffman decode is serial bit by bit, but not thread by thread.  In every round thread i looks up
    // the code that WOULD start at bit pos + i (256 candidate starts, one LDS gather); s_j[0][i] is
    // where the next code would start.  Five pointer-doubling steps give the start 2, 4, .. 32 codes
    // ahead of every candidate, and lane r of wave 0 then reaches the r-th TRUE code start from bit
    // pos by following the set bits of r -- at most six dependent LDS reads instead of r hops.  The
    // table is fixed inside a 50-symbol group, so a round ends at the group's end, after 64 symbols,
    // at the end of the 256-bit window, at EOB, or at a code the lookup table does not resolve
    // (longer than kLutBits bits / owned by no symbol), which is decoded on its own.
    u16 *out = sym_out + (size_t)c * kMtfStride;
    const u32 eob = alpha - 1;
    const u64 total_bits = nbytes * 8ull;
    const u32 wv = l >> 6, ln_ = l & 63u; // wave, lane
    u64 pos = s_pos;
    u32 flushed = 0;
    u32 g = 0, krem = 0, t = 0; // krem == 0: open the next group before the next symbol
    // Round 5: the selectors come through a 256-entry window in LDS (refilled by all threads every 256 groups) instead of one
    // global load per group issued a round ahead of its use (k_dec_block 12.74 -> 12.49 ms)
    for (u32 q = l; q < 256u; q += kD1Threads) s_sel[q] = sel[q < n_selectors ? q : 0u];
    __syncthreads();
    u32 t_next = s_sel[0];
    u32 state = 0;              // 0 running, 1 end of block, 2 error
    if (l < 6) s_j[l][kD1Threads] = (u16)kD1Threads; // outside stays outside
    while (true) {
        const u64 wb = pos >> 5;
        for (u32 q = l; q < kRingWords; q += kD1Threads) s_ring[(wb + q) & (kRingWords - 1)] = bc.load(wb + q);
        __syncthreads();
        BZ_T(2)
        u32 nout = 0;
        const u64 wend = (wb + kRingWords) * 32ull; // first bit not staged
        while (nout + 65u <= kOutBuf && pos + (u64)(kD1Threads + 96u) <= wend) { // (a candidate reads 64 bits from its start)
            if (krem == 0) {
                if (g >= n_selectors) { // group_no > n_selectors (:381-383)
                    state = 2;
                    break;
                }
                t = t_next;
                ++g;
                if ((g & 255u) == 0u) { // (g, krem are the same in every thread: the barriers are met by all)
                    __syncthreads();
                    for (u32 q = l; q < 256u; q += kD1Threads) s_sel[q] = sel[g + q < n_selectors ? g + q : 0u];
                    __syncthreads();
                }
                t_next = s_sel[g & 255u];
                krem = kGSize;
            }
            if (pos >= total_bits) { // peek returns no bits: Ok(None) -> DataError (:387-390)
                state = 2;
                break;
            }
            // the candidate code at bit pos + l
            {
                const u64 bp = pos + l;
                const u32 wi = (u32)(bp >> 5), sh = (u32)bp & 31u;
                const u64 two = ((u64)s_ring[wi & (kRingWords - 1)] << 32) | s_ring[(wi + 1u) & (kRingWords - 1)];
                const u32 e = s_lut[t][(u32)((two << sh) >> (64u - kLutBits))];
                const u32 len = (e < kLutLong) ? (e >> 9) : 0u;
                u32 nx = len ? l + len : kD1Threads; // an unresolved code ends the chain
                nx = nx < kD1Threads ? nx : kD1Threads;
                s_e[l] = (u16)e;
                s_j[0][l] = (u16)nx;
                __syncthreads();
                BZ_T(3)
                u32 cur = nx;
#pragma unroll
                for (u32 k = 1; k < 6; ++k) {
                    cur = s_j[k - 1][cur];
                    s_j[k][l] = (u16)cur;
                    __syncthreads();
                }
            }
            BZ_T(4)
            if (wv == 0) {
                const u32 r = ln_;
                u32 p = 0;
#pragma unroll
                for (u32 k = 0; k < 6; ++k)
                    if ((r >> k) & 1u) p = s_j[k][p];
                const bool inw = (p < kD1Threads) && (pos + p < total_bits);
                const u32 e = inw ? (u32)s_e[p] : (u32)kLutBad;
                const u32 len = (e < kLutLong) ? (e >> 9) : 0u;
                const u32 sy = e & 511u;
                const u64 vm = __ballot(inw); // a prefix of ones: code starts only move forward
                const u32 nvalid = (vm == ~0ull) ? 64u : (u32)__builtin_ctzll(~vm);
                const u32 lim = nvalid < krem ? nvalid : krem;
                const u64 limmask = (lim >= 64u) ? ~0ull : ((1ull << lim) - 1ull);
                const u64 su = __ballot(inw && len == 0u) & limmask;               // unresolved codes
                const u64 se = __ballot(inw && len != 0u && sy == eob) & limmask;  // EOB
                u32 cnt = lim, stop = 0; // stop: 1 EOB, 2 unresolved code
                if (su | se) {
                    const u32 f = (u32)__builtin_ctzll(su | se);
                    if ((su >> f) & 1ull) {
                        cnt = f;
                        stop = 2;
                    } else {
                        cnt = f + 1u;
                        stop = 1;
                    }
                }
                // bits consumed = start of the first symbol not taken
                const u32 lastr = cnt ? cnt - 1u : 0u;
                const u32 pl = (u32)__builtin_amdgcn_readlane((int)p, (int)lastr);
                const u32 ll = (u32)__builtin_amdgcn_readlane((int)len, (int)lastr);
                if (r < cnt) s_out[nout + r] = (u16)sy;
                if (r == 0) {
                    s_ctl[0] = cnt;
                    s_ctl[1] = cnt ? pl + ll : 0u;
                    s_ctl[2] = stop;
                }
            }
            __syncthreads();
            const u32 cnt = s_ctl[0], off = s_ctl[1], stop = s_ctl[2];
            nout += cnt;
            krem -= cnt;
            pos += off;
            BZ_T(5)
            if (stop == 1) {
                state = 1;
                break;
            }
            if (stop == 2) {
                // one symbol the table does not resolve: canonical walk from kLutBits + 1 bits up
                const u32 wi0 = (u32)(pos >> 5), sh0 = (u32)pos & 31u;
                const u64 two0 = ((u64)s_ring[wi0 & (kRingWords - 1)] << 32) | s_ring[(wi0 + 1u) & (kRingWords - 1)];
                const u64 bits = two0 << sh0; // the next 32 bits, left aligned
                const u32 e0 = s_lut[t][(u32)(bits >> (64u - kLutBits))];
                u32 ln = kLutBits, sy = 0;
                bool hit = false;
                if (e0 == kLutLong) {
                    while (ln < 23u) {
                        ++ln;
                        const u32 code = (u32)(bits >> (64u - ln));
                        const u32 rel = code - s_first[t][ln];
                        if (code >= s_first[t][ln] && rel < s_cnt[t][ln]) {
                            sy = s_perm[t][s_idx[t][ln] + rel];
                            hit = true;
                            break;
                        }
                    }
                }
                // the reference's table is min(max_len, 12) bits wide; bits beyond it are taken one by one
                // and fail at the end of the input (huffman/decoder.rs:150-233); a pattern no code owns
                // is unreachable!() there
                if (!hit || (ln > 12u && pos + ln > total_bits)) {
                    state = 2;
                    break;
                }
                if (l == 0) s_out[nout] = (u16)sy;
                ++nout;
                --krem;
                pos += ln;
                if (sy == eob) {
                    state = 1;
                    break;
                }
            }
            if (flushed + nout > kMaxBlockLen + 1u) { // more symbols than any block the reference accepts
                state = 2;
                break;
            }
        }
        __syncthreads();
        for (u32 i = l; i < nout; i += kD1Threads) out[flushed + i] = s_out[i];
        flushed += nout;
        BZ_T(6)
        if (state) {
            if (l == 0) {
                bi.status = state == 2 ? (u32)BZ_DEC_E_DATA : 0u;
                bi.end_bit = pos;
                bi.nsym = flushed;
                // the host's record chain wants the head byte of whatever follows
                const u64 avail = total_bits > pos ? total_bits - pos : 0ull;
                const u32 k = avail < 8ull ? (u32)avail : 8u;
                bc.seek(pos);
                const u32 v = bc.read(8);
                bi.next_bits = k;
                bi.next_head = k ? (v >> (8u - k)) : 0u;
            }
            break;
        }
        __syncthreads();
    }
}

// ---- D2: zero runs + inverse MTF --------------------------------------------------------------------

// inverse MTF step on a byte list held as dwords: returns the byte at rank `r` and moves it to the
// front.  The first eight entries (w0, w1) live in registers -- most ranks of a BWT block are tiny, and
// a lane's steps are serial, so an LDS round trip per symbol would be the whole cost; words 2.. of the
// list are in LDS (list[0], list[1] are stale while a chunk is being replayed).
__device__ __forceinline__ u32 imtf_pop(u32 &w0, u32 &w1, u32 *list, u32 r)
{
    if (r < 4u) {
        const u32 sh = 8u * r;
        const u32 v = (w0 >> sh) & 0xFFu;
        if (r) {
            const u32 upto = (r == 3u) ? 0xFFFFFFFFu : ((1u << (sh + 8u)) - 1u);
            w0 = (w0 & ~upto) | ((((w0 & (upto >> 8)) << 8) | v) & upto);
        }
        return v;
    }
    if (r < 8u) {
        const u32 sh = 8u * (r - 4u);
        const u32 v = (w1 >> sh) & 0xFFu;
        const u32 carry = w0 >> 24;
        w0 = (w0 << 8) | v;
        const u32 upto = (r == 7u) ? 0xFFFFFFFFu : ((1u << (sh + 8u)) - 1u);
        w1 = (w1 & ~upto) | ((((w1 & (upto >> 8)) << 8) | carry) & upto);
        return v;
    }
    const u32 dq = r >> 2, bq = r & 3u;
    const u32 wq = list[dq];
    const u32 v = (wq >> (8u * bq)) & 0xFFu;
    u32 carry = w1 >> 24;
    w1 = (w1 << 8) | (w0 >> 24);
    w0 = (w0 << 8) | v;
    for (u32 d = 2; d < dq; ++d) {
        const u32 w = list[d];
        list[d] = (w << 8) | carry;
        carry = w >> 24;
    }
    const u32 upto = (bq == 3u) ? 0xFFFFFFFFu : ((1u << (8u * (bq + 1u))) - 1u);
    list[dq] = (wq & ~upto) | ((((wq & (upto >> 8)) << 8) | carry) & upto);
    return v;
}

__global__ __launch_bounds__(256) void k_dec_chunk_perm(DecArgs a)
{
    __shared__ u32 s_list[256 * 65];
    // (all workgroups of a block on one XCD, blocks dealt round-robin: with the block as grid.y the XCD of a block's
    // FILLED workgroups was (workgroups per block x block) mod 8 -- a launch with an even number of workgroups per block,
    // most of them idle, put the work on half, a quarter or one of the eight XCDs)
    u32 wgx, lb;
    xcd_remap(gridDim.x, a.nb, wgx, lb);
    if (lb == 0xFFFFFFFFu) return;
    const u32 sl = a.slot[lb];
    const u32 nsym = a.info[sl].nsym;
    const u32 chunk = wgx * 256u + threadIdx.x;
    const u32 beg = chunk * kMtfChunk;
    if (beg >= nsym) return;
    const u32 end = (beg + kMtfChunk < nsym) ? beg + kMtfChunk : nsym;
    const u16 *sym = a.sym + (size_t)sl * kMtfStride;
    const u32 eob = a.info[sl].n_in_use + 1u;
    u32 *list = s_list + threadIdx.x * 65u;
    for (u32 d = 0; d < 64; ++d) list[d] = (4u * d) | ((4u * d + 1u) << 8) | ((4u * d + 2u) << 16) | ((4u * d + 3u) << 24);
    u32 w0 = 0x03020100u, w1 = 0x07060504u;
    u32 emit = 0, err = 0;
    // zero-run in progress: value so far and the weight of the next digit (decoder.rs:419-424); a run
    // belongs to the chunk that holds the symbol behind it, so digits hanging over from the previous
    // chunk are picked up first
    u32 es = 0, nw = 1;
    if (beg > 0 && sym[beg - 1] <= 1u) {
        u32 s0 = beg - 1;
        while (s0 > 0 && sym[s0 - 1] <= 1u && beg - s0 < 32u) --s0;
        for (u32 i = s0; i < beg; ++i) {
            if (nw >= 2u * 1024u * 1024u) err = 1; // :413-416
            else {
                es += nw << sym[i];
                nw <<= 1;
            }
        }
    }
    // sixteen symbols per trip to memory (a lane's chunk is contiguous, but the 64 lanes of a wave read
    // 64 different lines: one load per symbol would cost a full memory round trip each)
    for (u32 i0 = beg; i0 < end; i0 += 16u) {
        const uint4 qa = *reinterpret_cast<const uint4 *>(sym + i0), qb = *reinterpret_cast<const uint4 *>(sym + i0 + 8u);
        const u32 wv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (u32 k = 0; k < 16u; ++k) {
            if (i0 + k < end) {
                const u32 sy = (wv[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu;
                if (sy <= 1u) {
                    if (nw >= 2u * 1024u * 1024u) err = 1;
                    else {
                        es += nw << sy; // RUNA adds the weight, RUNB twice the weight
                        nw <<= 1;
                    }
                } else {
                    emit += es;
                    es = 0;
                    nw = 1;
                    if (sy != eob) {
                        (void)imtf_pop(w0, w1, list, sy - 1u);
                        emit += 1;
                    }
                }
            }
        }
    }
    u8 *out = a.perm + ((size_t)lb * kMaxMtfChunks + chunk) * 256u;
    list[0] = w0;
    list[1] = w1;
    for (u32 d = 0; d < 64; ++d) reinterpret_cast<u32 *>(out)[d] = list[d];
    a.chunk_emit[(size_t)lb * kMaxMtfChunks + chunk] = emit;
    if (err) atomicMax(&a.err[lb], 1u);
}

// M2': compose the permutations into start lists (of byte values) and scan the emit counts
__global__ __launch_bounds__(64) void k_dec_compose(DecArgs a)
{
    __shared__ u8 s_state[256], s_new[256];
    const u32 lb = blockIdx.x, l = threadIdx.x;
    const DecBlockInfo &bi = a.info[a.slot[lb]];
    const u32 nchunks = (bi.nsym + kMtfChunk - 1) / kMtfChunk;
    for (u32 k = 0; k < 4; ++k) s_state[l * 4 + k] = (l * 4 + k < bi.n_in_use) ? bi.seq2unseq[l * 4 + k] : 0;
    __syncthreads();
    u32 run = 0;
    for (u32 c = 0; c < nchunks; ++c) {
        u8 *pp = a.perm + ((size_t)lb * kMaxMtfChunks + c) * 256u;
        const u32 pw = reinterpret_cast<const u32 *>(pp)[l];
        reinterpret_cast<u32 *>(pp)[l] = reinterpret_cast<const u32 *>(s_state)[l]; // start list of chunk c
        for (u32 k = 0; k < 4; ++k) s_new[l * 4 + k] = s_state[(pw >> (8u * k)) & 0xFFu];
        __syncthreads();
        reinterpret_cast<u32 *>(s_state)[l] = reinterpret_cast<const u32 *>(s_new)[l];
        if (l == 0) {
            const u32 e = a.chunk_emit[(size_t)lb * kMaxMtfChunks + c];
            a.chunk_emit[(size_t)lb * kMaxMtfChunks + c] = run;
            run = (run + e < run) ? 0xFFFFFFFFu : run + e;
        }
        __syncthreads();
    }
    if (l == 0) {
        a.tt_len[lb] = run;
        if (run == 0xFFFFFFFFu || run > a.nblock_max[lb]) atomicMax(&a.err[lb], 1u); // tt overflows (:399,427)
        // decoder.rs:238 (orig_pos > 10 + 100000 * level) and :441-443 (orig_pos >= tt.len())
        if (bi.orig_ptr > 10u + a.nblock_max[lb] || bi.orig_ptr >= run) atomicMax(&a.err[lb], 1u);
    }
}

// M3': replay every chunk from its start list and write the BWT column
__global__ __launch_bounds__(256) void k_dec_chunk_emit(DecArgs a)
{
    __shared__ u32 s_list[256 * 65];
    u32 wgx, lb;
    xcd_remap(gridDim.x, a.nb, wgx, lb); // (see k_dec_chunk_perm)
    if (lb == 0xFFFFFFFFu) return;
    if (a.err[lb]) return;
    const u32 sl = a.slot[lb];
    const u32 nsym = a.info[sl].nsym;
    const u32 chunk0 = wgx * 256u;
    if (chunk0 * kMtfChunk >= nsym) return;
    const u32 nchunks = (nsym + kMtfChunk - 1) / kMtfChunk;
    {
        const u32 *src = reinterpret_cast<const u32 *>(a.perm + ((size_t)lb * kMaxMtfChunks + chunk0) * 256u);
        const u32 avail = (nchunks - chunk0 < 256u ? nchunks - chunk0 : 256u) * 64u;
        for (u32 i = threadIdx.x; i < avail; i += 256u) s_list[(i >> 6) * 65u + (i & 63u)] = src[i];
    }
    __syncthreads();
    const u32 chunk = chunk0 + threadIdx.x;
    const u32 beg = chunk * kMtfChunk;
    if (beg >= nsym) return;
    const u32 end = (beg + kMtfChunk < nsym) ? beg + kMtfChunk : nsym;
    const u16 *sym = a.sym + (size_t)sl * kMtfStride;
    const u32 eob = a.info[sl].n_in_use + 1u;
    const u32 maxlen = a.nblock_max[lb];
    u32 *list = s_list + threadIdx.x * 65u;
    u32 w0 = list[0], w1 = list[1];
    u8 *L = a.L + (size_t)lb * kSlot;
    u32 o = a.chunk_emit[(size_t)lb * kMaxMtfChunks + chunk];
    u32 err = 0;
    u32 es = 0, nw = 1;
    if (beg > 0 && sym[beg - 1] <= 1u) {
        u32 s0 = beg - 1;
        while (s0 > 0 && sym[s0 - 1] <= 1u && beg - s0 < 32u) --s0;
        for (u32 i = s0; i < beg; ++i) {
            if (nw >= 2u * 1024u * 1024u) err = 1;
            else {
                es += nw << sym[i];
                nw <<= 1;
            }
        }
    }
    for (u32 i0 = beg; i0 < end && !err; i0 += 16u) {
        const uint4 qa = *reinterpret_cast<const uint4 *>(sym + i0), qb = *reinterpret_cast<const uint4 *>(sym + i0 + 8u);
        const u32 wv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (u32 k = 0; k < 16u; ++k) {
            if (i0 + k < end && !err) {
                const u32 sy = (wv[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu;
                if (sy <= 1u) {
                    if (nw >= 2u * 1024u * 1024u) err = 1;
                    else {
                        es += nw << sy;
                        nw <<= 1;
                    }
                } else {
                    if (es) {
                        const u32 front = w0 & 0xFFu;
                        if (o + es >= maxlen) err = 1; // tt.len() >= nblock_max after a run (:399-401)
                        else
                            for (u32 q = 0; q < es; ++q) L[o + q] = (u8)front;
                        o += es;
                        es = 0;
                        nw = 1;
                    }
                    if (sy != eob) {
                        if (o >= maxlen) err = 1;    // :427-429
                        else L[o] = (u8)imtf_pop(w0, w1, list, sy - 1u);
                        o += 1;
                    }
                }
            }
        }
    }
    if (err) atomicMax(&a.err[lb], 1u);
}

// ---- D3: the pointer chase, cut at sample nodes ----------------------------------------------------------
// T[slot] = (position << 8) | byte, where `byte` is the digit of the bucket the slot lies in -- the
// byte the chase emits when it LEAVES the slot (L[T[slot]] == first-column byte of slot).  One
// random 4-byte load per step, like the reference's tt[] (decoder.rs:468-473,533-536).
//
// Segments between sample nodes have geometric lengths; a lane-per-segment launch would idle most
// lanes of a wave while its longest segment finishes.  The walkers are persistent instead: a lane
// that finishes a segment takes the next (block, sample) item from a global counter (one atomic
// per wave and refill).  Items are block-major, so the blocks being walked at any time are a
// window of a few dozen whose T vectors stay in the 256 MB Infinity Cache.
__device__ __forceinline__ u32 sample_id(u32 node, u32 p0)
{
    if (node == p0) return kDecSamples - 1u;
    return (node % kDecSampleStep == 0) ? node / kDecSampleStep : 0xFFFFFFFFu;
}

// hands out work items to the idle lanes of a wave; returns false when nothing is left for this lane
__device__ __forceinline__ bool take_item(u32 *ctr, u32 total, bool idle, u32 &id, u32 &base)
{
    const u64 need = __ballot(idle);
    const u32 l = lane_id();
    base = 0;
    const u32 leader = (u32)__builtin_ctzll(need);
    if (l == leader) base = atomicAdd(ctr, (u32)__popcll(need));
    base = __shfl(base, (int)leader);
    const u64 lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
    id = base + (u32)__popcll(need & lt);
    return idle && id < total;
}

constexpr u32 kWalkBurst = 8;   // steps between looks at the idle lanes
constexpr u32 kWalkRefill = 16; // idle lanes that make a wave fetch new items (an atomic round trip)

// per-block constants of the walkers: {length (0 = skip the block), start node, first byte}
__global__ __launch_bounds__(64) void k_dec_walk_meta(DecArgs a)
{
    const u32 lb = blockIdx.x * 64u + threadIdx.x;
    if (lb >= a.nb) return;
    uint4 m = {0u, 0u, 0u, 0u};
    if (!a.err[lb]) {
        const u32 v = a.T[(size_t)lb * kSlot + a.info[a.slot[lb]].orig_ptr];
        m.x = a.tt_len[lb];
        m.y = v >> 8;    // tt[orig_pos] >> 8: the first node (decoder.rs:476)
        m.z = v & 0xFFu;
    }
    a.walk_meta[lb] = m;
}

// Walk 1: every segment (sample node -> next sample node) is walked once.  The bytes it passes are
// kept in the segment's scratch row (dword stores, kSegCap bytes); its length and successor go to
// the sample arrays.  The node reached after kSegCap steps is remembered for the few long segments.
// the chase's load: a plain one (nt / sc0 / sc1 variants were measured in round 3 and change nothing: profiles/r03_decode_walk_window.md)
__device__ __forceinline__ u32 walk_load(const u32 *p)
{
    return *p;
}

constexpr u32 kWalkGroup = 1;                          // consecutive sample nodes a lane takes at a time
constexpr u32 kWalkItems = (kDecSamples + kWalkGroup - 1u) / kWalkGroup; // items per block
__global__ __launch_bounds__(256) void k_dec_walk_lengths(DecArgs a)
{
    // blocks are dealt to the XCDs (workgroup i runs on XCD i % 8): a block's T vector is walked by
    // one XCD only and stays in that XCD's 4 MiB L2
    // (reading XCC_ID instead gives the same assignment and the same time: checked in round 2)
    const u32 xcd = blockIdx.x & 7u;
    const u32 total = ((a.nb + 7u - xcd) / 8u) * kWalkItems;
    u32 *ctr = a.work_ctr + xcd * 16u;
    bool have = false;
    u32 cur = 0, len = 0, n = 0, p0 = 0, slot = 0, acc = 0;
    u32 sid = 0, sid_end = 0, lb = 0; // the lane's item: sample nodes [sid, sid_end) of block lb
    const u32 *T = nullptr;
    u8 *row = nullptr;
    bool dry = false;
    // the next sample node of the lane's item that starts a segment
    auto next_segment = [&]() {
        while (sid < sid_end) {
            const u32 node = (sid == kDecSamples - 1u) ? p0 : sid * kDecSampleStep;
            // (p0's segment belongs to the start sample)
            if (sid == kDecSamples - 1u || (node < n && node != p0)) {
                slot = lb * kDecSamples + sid;
                cur = node;
                len = 0;
                acc = 0;
                row = a.seg_buf + (size_t)slot * kSegCap;
                have = true;
                ++sid;
                return;
            }
            ++sid;
        }
    };
    while (true) {
        const u64 idle = __ballot(!have);
        if ((!dry && (u32)__popcll(idle) >= kWalkRefill) || idle == ~0ull) {
            u32 id, first_id;
            const bool got = take_item(ctr, total, !have, id, first_id);
            if (got) {
                const u32 q = id / kWalkItems, grp = id - q * kWalkItems;
                lb = q * 8u + xcd;
                const uint4 m = a.walk_meta[lb];
                if (m.x) {
                    n = m.x;
                    T = a.T + (size_t)lb * kSlot;
                    p0 = m.y;
                    sid = grp * kWalkGroup;
                    sid_end = sid + kWalkGroup < kDecSamples ? sid + kWalkGroup : kDecSamples;
                    next_segment();
                }
            }
            if (first_id + (u32)__popcll(idle) >= total) dry = true; // the last items are handed out
            if (!__ballot(have)) {
                if (first_id >= total) break; // the counter ran dry and no lane holds a segment
                continue;                     // every item taken was an empty one: take more
            }
        }
#pragma unroll 1
        for (u32 s = 0; s < kWalkBurst; ++s) {
            if (have) {
                const u32 v = walk_load(T + cur); // tt[pos]: the byte in the low 8 bits, the next position above (decoder.rs:533-536)
                cur = v >> 8;
                if (len < kSegCap) {
                    acc |= (v & 0xFFu) << (8u * (len & 3u));
                    if ((len & 3u) == 3u) {
                        __builtin_nontemporal_store(acc, reinterpret_cast<u32 *>(row + (len - 3u)));
                        acc = 0;
                    }
                }
                ++len;
                if (len == kSegCap) { // a long segment: remembered for the second walk
                    a.seg_cont[slot] = cur;
                    a.long_list[atomicAdd(a.work_ctr + 256, 1u)] = slot;
                }
                const u32 nid = sample_id(cur, p0);
                if (nid != 0xFFFFFFFFu || len > n) {
                    if (len < kSegCap && (len & 3u)) __builtin_nontemporal_store(acc, reinterpret_cast<u32 *>(row + (len & ~3u)));
                    a.samp_next[slot] = nid;
                    a.samp_len[slot] = len;
                    have = false;
                    next_segment(); // (the other sample nodes of the lane's item: no trip to the counter)
                }
            }
        }
    }
}

// order of the samples along the chain: one wave per block, the chain itself in LDS
__global__ __launch_bounds__(64) void k_dec_rank_samples(DecArgs a)
{
    // one word per sample: the segment's length (16 bits; the rare longer one is fetched from memory) and its
    // successor (15 bits), so that a hop of the serial chain is ONE dependent LDS read (rounds 1-3: three -- length,
    // successor and a "seen" bit -- and a global store; 2.0 ms per GiB for 7000 hops per block); the offsets replace the
    // words as the chain passes and leave together at the end.  28 KB of LDS per block: all blocks of a 1 GiB file are
    // resident at once.
    static_assert(kDecSamples < 0x7FFFu, "a sample's successor must fit 15 bits");
    __shared__ u32 s_hop[kDecSamples];
    const u32 lb = blockIdx.x, l = threadIdx.x;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const size_t base = (size_t)lb * kDecSamples;
    for (u32 i = l; i < kDecSamples; i += 64) {
        const bool live = (i == kDecSamples - 1u) || (i * kDecSampleStep < n);
        const u32 nx = live ? a.samp_next[base + i] : 0xFFFFFFFFu;
        const u32 ln = live ? a.samp_len[base + i] : 0u;
        s_hop[i] = (nx < kDecSamples ? nx : 0x7FFFu) << 16 | (ln < 0xFFFFu ? ln : 0xFFFFu);
    }
    __syncthreads();
    if (l == 0) {
        // T is a permutation: the first sample the chain comes back to is the one it started from (a periodic block's
        // chain closes before n steps, decoder.rs:527-542 reads on around it)
        const u32 start = kDecSamples - 1u;
        u32 s = start, o = 0, cyc = n;
        while (o < n) {
            const u32 w = s_hop[s];
            s_hop[s] = 0x80000000u | o; // (a sample is visited once: its word now holds its offset, bit 31 = visited)
            const u32 ln = w & 0xFFFFu;
            o += (ln == 0xFFFFu) ? a.samp_len[base + s] : ln;
            s = w >> 16;
            if (s == 0x7FFFu) break;
            if (s == start) {
                if (o < n) cyc = o;
                break;
            }
        }
        a.cycle_len[lb] = cyc;
    }
    __syncthreads();
    for (u32 i = l; i < kDecSamples; i += 64) {
        const u32 w = s_hop[i];
        a.samp_off[base + i] = (w & 0x80000000u) ? (w & 0x7FFFFFFFu) : 0xFFFFFFFFu;
    }
}

// Copy: the scratch rows go to their places in the RLE1 image.  Sixteen lanes per segment (a row is
// 128 bytes on average); aligned dword stores built from two aligned source dwords, single bytes
// only at the ragged ends.
__global__ __launch_bounds__(256) void k_dec_seg_copy(DecArgs a)
{
    const u32 lb = blockIdx.y;
    const uint4 m = a.walk_meta[lb];
    if (!m.x) return;
    const u32 n = m.x;
    const u32 sid = blockIdx.x * 16u + (threadIdx.x >> 4), sl = threadIdx.x & 15u;
    if (sid >= kDecSamples) return;
    const u32 id = lb * kDecSamples + sid;
    const u32 o0 = a.samp_off[id];
    if (o0 == 0xFFFFFFFFu) return;
    u8 *X = a.X + (size_t)lb * kSlot;
    if (sid == kDecSamples - 1u && sl == 0) X[0] = (u8)m.z; // tt[orig_pos] & 0xFF is the first byte (decoder.rs:476)
    const u32 len = a.samp_len[id];
    u32 cnt = len < kSegCap ? len : kSegCap;
    const u32 o = o0 + 1u;
    if (o >= n) return;
    if (o + cnt > n) cnt = n - o;
    const u8 *row = a.seg_buf + (size_t)id * kSegCap;
    const u32 *rdw = reinterpret_cast<const u32 *>(row);
    u8 *dst = X + o;
    u32 head = (4u - (u32)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u;
    if (head > cnt) head = cnt;
    if (sl < head) dst[sl] = row[sl];
    const u32 nd = (cnt - head) >> 2;
    u32 *ddw = reinterpret_cast<u32 *>(dst + head);
    const u32 sh = (head & 3u) * 8u;
    for (u32 j = sl; j < nd; j += 16u) {
        const u32 si = (head >> 2) + j; // (head < 4: source dword j, shifted by head bytes)
        const u32 lo = rdw[si], hi = rdw[si + 1u];
        ddw[j] = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
    }
    const u32 done = head + nd * 4u;
    if (sl < cnt - done) dst[done + sl] = row[done + sl];
}

// Walk 2: only what the scratch rows could not hold (the few segments longer than kSegCap, listed by
// walk 1) is walked again, one lane per segment.
__global__ __launch_bounds__(256) void k_dec_walk_write(DecArgs a)
{
    const u32 total = a.work_ctr[256];
    for (u32 it = blockIdx.x * 256u + threadIdx.x; it < total; it += gridDim.x * 256u) {
        const u32 id = a.long_list[it];
        const u32 lb = id / kDecSamples;
        const uint4 m = a.walk_meta[lb];
        const u32 o0 = a.samp_off[id];
        if (!m.x || o0 == 0xFFFFFFFFu) continue;
        const u32 n = m.x;
        const u32 *T = a.T + (size_t)lb * kSlot;
        u8 *X = a.X + (size_t)lb * kSlot;
        u32 cur = a.seg_cont[id];
        u32 o = o0 + 1u + kSegCap;
        for (u32 left = a.samp_len[id] - kSegCap; left && o < n; --left) {
            const u32 v = T[cur];
            X[o++] = (u8)v;
            cur = v >> 8;
        }
    }
}

// periodic blocks (the chain closes before n steps) and randomised blocks (decoder.rs:27-92,537-539)
__global__ __launch_bounds__(256) void k_dec_fixups(DecArgs a)
{
    const u32 lb = blockIdx.x;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    u8 *X = a.X + (size_t)lb * kSlot;
    const u32 c = a.cycle_len[lb];
    if (c && c < n)
        for (u32 q = c + threadIdx.x; q < n; q += 256u) X[q] = X[q % c];
    __syncthreads();
    if (a.info[a.slot[lb]].randomised && threadIdx.x == 0) {
        u32 tpos = 0, q = 0;
        while (true) {
            const u32 step = kBz2RNums[tpos];
            tpos = (tpos + 1u) & 511u;
            // n2go is loaded with rNums[tpos] when it is 0, then decremented for every byte; the byte at
            // which it reaches 1 is flipped: the last but one of each stretch of rNums[tpos] bytes
            if (q + step - 2u >= n) break;
            X[q + step - 2u] ^= 1u;
            q += step;
        }
    }
}

// ---- D4: RLE1 undo (decoder.rs:550-578) ---------------------------------------------------------------------
// The undo is a little state machine (last byte, equal bytes seen; after four equal bytes the next
// byte is a repeat count).  A 64-byte sub-tile is a function from its entering state to its exit
// state and output size; relative to the sub-tile's first byte only five entering states differ:
//   e = 0  no equal byte in front (fresh)      e = 1..3  that many bytes equal to the first byte
//   e = 4  the first byte is a count
// k_dec_rle_sub tabulates that function for every sub-tile, k_dec_rle_chain composes them along the
// block (1024 threads, each folds its share; one thread links the 1024 partial results), and
// k_dec_rle_expand replays every sub-tile from its now known entering state.
struct RleSt {
    u32 last, cnt, out;
};
__device__ __forceinline__ void rle_step(RleSt &s, u32 b)
{
    if (s.cnt == 4u) {
        s.out += b;
        s.cnt = 0;
        s.last = 0x100u;
    } else if (b == s.last) {
        ++s.cnt;
        ++s.out;
    } else {
        s.last = b;
        s.cnt = 1;
        ++s.out;
    }
}

__global__ __launch_bounds__(256) void k_dec_rle_sub(DecArgs a)
{
    const u32 lb = blockIdx.y;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 sub = blockIdx.x * 256u + threadIdx.x;
    const u32 p = sub * 64u;
    if (p >= n) return;
    const u8 *X = a.X + (size_t)lb * kSlot;
    const u32 b0 = X[p];
    RleSt st[5];
    st[0] = {0x100u, 0u, 0u};
    st[1] = {b0, 1u, 0u};
    st[2] = {b0, 2u, 0u};
    st[3] = {b0, 3u, 0u};
    st[4] = {0x100u, 4u, 0u};
    const u32 end = (p + 64u < n) ? 64u : n - p;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(X + p + q * 16u);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (q * 16u + k < end) {
                const u32 b = (wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
#pragma unroll
                for (u32 e = 0; e < 5; ++e) rle_step(st[e], b);
            }
        }
    }
    const u32 nxt = (p + 64u < n) ? (u32)X[p + 64u] : 0x200u;
    u32 epack = 0;
#pragma unroll
    for (u32 e = 0; e < 5; ++e) {
        const u32 ex = (st[e].cnt == 4u) ? 4u : ((st[e].last == nxt) ? st[e].cnt : 0u);
        epack |= ex << (3u * e);
    }
    uint4 tr;
    tr.x = st[0].out | (st[1].out << 16);
    tr.y = st[2].out | (st[3].out << 16);
    tr.z = st[4].out | (epack << 16);
    tr.w = 0;
    a.sub_trans[(size_t)lb * kDecSubs + sub] = tr;
}

__device__ __forceinline__ void rle_apply(const uint4 tr, u32 &e, u32 &off)
{
    const u32 o01 = tr.x, o23 = tr.y;
    const u32 out = (e == 0) ? (o01 & 0xFFFFu) : (e == 1) ? (o01 >> 16) : (e == 2) ? (o23 & 0xFFFFu)
                  : (e == 3) ? (o23 >> 16) : (tr.z & 0xFFFFu);
    off += out;
    e = ((tr.z >> 16) >> (3u * e)) & 7u;
}

__global__ __launch_bounds__(1024) void k_dec_rle_chain(DecArgs a)
{
    __shared__ u32 s_out[1024][5];
    __shared__ u16 s_ep[1024];
    __shared__ u32 s_in_off[1024];
    __shared__ u8 s_in_e[1024];
    const u32 lb = blockIdx.x, j = threadIdx.x;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 nsub = (n + 63u) / 64u;
    const u32 K = (nsub + 1023u) / 1024u;
    const u32 s0 = j * K, s1 = (s0 + K < nsub) ? s0 + K : nsub;
    const uint4 *tr = a.sub_trans + (size_t)lb * kDecSubs;
    {
        u32 e[5] = {0, 1, 2, 3, 4}, off[5] = {0, 0, 0, 0, 0};
        for (u32 s = s0; s < s1; ++s) {
            const uint4 t = tr[s];
#pragma unroll
            for (u32 q = 0; q < 5; ++q) rle_apply(t, e[q], off[q]);
        }
        u32 ep = 0;
#pragma unroll
        for (u32 q = 0; q < 5; ++q) {
            s_out[j][q] = off[q];
            ep |= e[q] << (3u * q);
        }
        s_ep[j] = (u16)ep;
    }
    __syncthreads();
    if (j == 0) {
        u32 e = 0, off = 0;
        for (u32 t = 0; t < 1024; ++t) {
            s_in_e[t] = (u8)e;
            s_in_off[t] = off;
            off += s_out[t][e];
            e = ((u32)s_ep[t] >> (3u * e)) & 7u;
        }
        a.out_len[lb] = off;
        a.sub_off[(size_t)lb * (kDecSubs + 1) + nsub] = off;
        // four equal bytes with no count byte behind them: the reference reads on around the closed
        // pointer chain and never terminates (decoder.rs:566-570); reported as DataError here
        if (e == 4u) a.err[lb] = 1u;
    }
    __syncthreads();
    {
        u32 e = s_in_e[j], off = s_in_off[j];
        for (u32 s = s0; s < s1; ++s) {
            a.sub_state[(size_t)lb * kDecSubs + s] = (u8)e;
            a.sub_off[(size_t)lb * (kDecSubs + 1) + s] = off;
            rle_apply(tr[s], e, off);
        }
    }
}

// replays one sub-tile from its entering state; WP is an LDS or a global byte pointer
template <class WP>
__device__ __forceinline__ void rle_replay(const u8 *__restrict__ src, u32 end, u32 last, u32 cnt, WP wp)
{
    u32 o = 0;
#pragma unroll 1
    for (u32 q = 0; q < 4; ++q) {
        if (q * 16u >= end) break;
        const uint4 v = *reinterpret_cast<const uint4 *>(src + q * 16u);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (q * 16u + k < end) {
                const u32 b = (wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                if (cnt == 4u) {
                    for (u32 r = 0; r < b; ++r) wp[o + r] = (u8)last;
                    o += b;
                    cnt = 0;
                    last = 0x100u;
                } else {
                    if (b == last) ++cnt;
                    else {
                        last = b;
                        cnt = 1;
                    }
                    wp[o++] = (u8)b;
                }
            }
        }
    }
}

constexpr u32 kExpStage = 8192; // staged output bytes per wave (4 KiB of image; long runs bypass it)

__global__ __launch_bounds__(256) void k_dec_rle_expand(DecArgs a, const u64 *__restrict__ out_base, u8 *__restrict__ out)
{
    __shared__ u32 s_stage[4][kExpStage / 4 + 2];
    const u32 lb = blockIdx.y;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u32 sub0 = (blockIdx.x * 4u + w) * 64u; // first sub-tile of this wave
    if (sub0 * 64u >= n) return;
    const u32 nsub = (n + 63u) / 64u;
    const u32 sub = sub0 + l;
    const u32 *soff = a.sub_off + (size_t)lb * (kDecSubs + 1);
    const u32 wave_last = (sub0 + 64u < nsub) ? sub0 + 64u : nsub;
    const u32 base = soff[sub0], total = soff[wave_last] - base;
    const u8 *X = a.X + (size_t)lb * kSlot;
    u8 *dst = out + out_base[lb] + base;
    const u32 shift = (u32)(reinterpret_cast<uintptr_t>(dst) & 3u);
    const bool staged = total + shift <= kExpStage;
    u8 *stage = reinterpret_cast<u8 *>(s_stage[w]);
    if (sub < nsub) {
        const u32 p = sub * 64u;
        const u32 e = a.sub_state[(size_t)lb * kDecSubs + sub];
        u32 last = 0x100u, cnt = 0;
        if (e >= 1u && e <= 3u) {
            last = X[p];
            cnt = e;
        } else if (e == 4u) {
            last = X[p - 1u];
            cnt = 4u;
        }
        const u32 o = soff[sub] - base;
        const u32 end = (p + 64u < n) ? 64u : n - p;
        if (staged) rle_replay(X + p, end, last, cnt, stage + shift + o);
        else rle_replay(X + p, end, last, cnt, dst + o);
    }
    if (!staged) return;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // flush: aligned dwords in the middle, bytes at both ends
    u8 *gbase = dst - shift;                 // 4-byte aligned
    const u32 first = shift, lastb = shift + total; // byte range inside the staging buffer
    const u32 dw0 = (first + 3u) / 4u, dw1 = lastb / 4u;
    if (dw0 <= dw1) {
        for (u32 i = dw0 + l; i < dw1; i += 64u) reinterpret_cast<u32 *>(gbase)[i] = s_stage[w][i];
        if (l < dw0 * 4u - first) gbase[first + l] = stage[first + l];                 // head bytes (< 4)
        if (l < lastb - dw1 * 4u) gbase[dw1 * 4u + l] = stage[dw1 * 4u + l];           // tail bytes (< 4)
    } else {
        if (l < total) gbase[first + l] = stage[first + l]; // everything inside one dword
    }
}

// ---- D3a: T vector = positions of the BWT column in stable byte order (decoder.rs:446-473) ---------------
// Same three-kernel shape as the encoder's radix pass (k_bwt.hip) with one 8-bit digit and no keys
// to carry: per-tile byte counts, per-block scan, stable scatter ranked with wave ballots.
__global__ __launch_bounds__(kSortThreads) void k_dec_thist(DecArgs a)
{
    // four interleaved copies of the counters, and a thread adds a run of equal bytes at once: the BWT
    // column is mostly runs, and single adds to one LDS word serialise
    __shared__ u32 s_hist[4][256];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu || a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 start = tile * kSortTile;
    if (start >= n) return;
    for (u32 i = threadIdx.x; i < 4u * 256u; i += kSortThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    const u8 *L = a.L + (size_t)lb * kSlot;
    u32 *mine = s_hist[threadIdx.x & 3u];
    // 16 bytes per thread
    const u32 i0 = start + threadIdx.x * 16u;
    if (i0 < n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(L + i0);
        const u32 wv[4] = {q.x, q.y, q.z, q.w};
        u32 prev = wv[0] & 0xFFu, run = 0;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) {
            if (i0 + k < n) {
                const u32 b = (wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                if (b == prev) ++run;
                else {
                    atomicAdd(&mine[prev], run);
                    prev = b;
                    run = 1;
                }
            }
        }
        if (run) atomicAdd(&mine[prev], run);
    }
    __syncthreads();
    if (threadIdx.x < 256)
        a.thist[((size_t)lb * kTilesPerBlock + tile) * 256u + threadIdx.x] =
            s_hist[0][threadIdx.x] + s_hist[1][threadIdx.x] + s_hist[2][threadIdx.x] + s_hist[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_dec_tscan(DecArgs a)
{
    __shared__ u32 s_w[4];
    const u32 lb = blockIdx.x;
    if (a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 ntiles = (n + kSortTile - 1) / kSortTile;
    u32 *hist = a.thist + (size_t)lb * kTilesPerBlock * 256u;
    const u32 d = threadIdx.x;
    u32 run = 0;
    for (u32 t = 0; t < ntiles; ++t) {
        const u32 v = hist[t * 256u + d];
        hist[t * 256u + d] = run;
        run += v;
    }
    const u32 inc = wave_incl_sum(run);
    if ((d & 63u) == 63u) s_w[d >> 6] = inc;
    __syncthreads();
    u32 carry = 0;
    for (u32 k = 0; k < (d >> 6); ++k) carry += s_w[k];
    a.tbase[(size_t)lb * 256u + d] = carry + inc - run;
}

__global__ __launch_bounds__(kSortThreads) void k_dec_tscatter(DecArgs a)
{
    constexpr u32 NW = kSortThreads / 64;
    __shared__ u32 s_buf[kSortTile];
    __shared__ u32 s_base[256];
    __shared__ u16 s_tpre[256];
    __shared__ u16 s_cnt[NW][256];
    __shared__ u32 s_wsum[4];
    u32 tile, lb;
    xcd_remap(gridDim.x, a.nb, tile, lb);
    if (lb == 0xFFFFFFFFu || a.err[lb]) return;
    const u32 n = a.tt_len[lb];
    const u32 start = tile * kSortTile;
    if (start >= n) return;
    const u8 *L = a.L + (size_t)lb * kSlot;
    u32 *T = a.T + (size_t)lb * kSlot;
    for (u32 i = threadIdx.x; i < NW * 256u / 2u; i += kSortThreads) reinterpret_cast<u32 *>(&s_cnt[0][0])[i] = 0;
    if (threadIdx.x < 256)
        s_base[threadIdx.x] = a.thist[((size_t)lb * kTilesPerBlock + tile) * 256u + threadIdx.x] +
                              a.tbase[(size_t)lb * 256u + threadIdx.x];
    __syncthreads();
    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    u16 *my_cnt = s_cnt[w];
    u32 dgv[16], rnk[16];
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = start + w * 1024u + r * 64u + l;
        dgv[r] = idx < n ? (u32)L[idx] : 0u;
    }
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        const u32 idx = start + w * 1024u + r * 64u + l;
        const bool ok = idx < n;
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
            const u32 c = s_cnt[k][threadIdx.x];
            s_cnt[k][threadIdx.x] = (u16)tot;
            tot += c;
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
    const u32 cnt_tile = (n - start < kSortTile) ? n - start : kSortTile;
#pragma unroll
    for (u32 r = 0; r < 16; ++r) {
        if (rnk[r] != 0xFFFFFFFFu) {
            const u32 dg = dgv[r];
            const u32 lpos = (u32)s_tpre[dg] + (u32)my_cnt[dg] + rnk[r];
            // digit order inside the tile: value (position) in the low 24 bits, digit on top
            s_buf[lpos] = (start + w * 1024u + r * 64u + l) | (dg << 24);
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        const u32 i = k * kSortThreads + threadIdx.x;
        if (i < cnt_tile) {
            const u32 e = s_buf[i];
            const u32 dg = e >> 24;
            T[s_base[dg] + (i - (u32)s_tpre[dg])] = ((e & 0xFFFFFFu) << 8) | dg;
        }
    }
}

// ---- D4b: CRC of every block's output (crc32.rs:116-131) ------------------------------------------------
// Lanes read consecutive 16-byte pieces (aligned in memory, so the first piece may begin in front of
// the block: those bytes are fed as zeros, which a CRC register that starts at zero ignores).  A
// piece's CRC is moved to the end of its 4 KiB tile with the x^(128 k) table, tiles are chained with
// x^(8*4096), and the slice (64 KiB per workgroup) is moved to the end of the block once.
__global__ __launch_bounds__(256) void k_dec_crc(DecArgs a, const u64 *__restrict__ out_base,
                                                 const u8 *__restrict__ out, const u32 *__restrict__ crc_tab,
                                                 const u32 *__restrict__ xp2, const u32 *__restrict__ xp16)
{
    __shared__ u32 s_tab[256];
    __shared__ u32 s_x[4];
    __shared__ u32 s_tile[16];
    const u32 lb = blockIdx.y;
    if (a.err[lb]) return;
    const u64 len = a.out_len[lb];
    const uintptr_t A = reinterpret_cast<uintptr_t>(out + out_base[lb]); // first byte of the block
    const uintptr_t a0 = A & ~(uintptr_t)15;
    const u64 R = (u64)(A - a0) + len;  // bytes from the aligned start to the end of the block
    const u64 npieces = (R + 15) / 16;  // the last one may be partial
    const u64 p0 = (u64)blockIdx.x * 4096ull; // first piece of this workgroup's slice
    if (p0 >= npieces && blockIdx.x != 0) return;
    s_tab[threadIdx.x] = crc_tab[threadIdx.x];
    __syncthreads();
    const u32 t = threadIdx.x;
    for (u32 tile = 0; tile < 16; ++tile) {
        const u64 i = p0 + tile * 256ull + t;
        u32 c = 0;
        if (i < npieces) {
            const uint4 q = *reinterpret_cast<const uint4 *>(a0 + i * 16);
            const u32 wv[4] = {q.x, q.y, q.z, q.w};
            const u64 pbeg = i * 16, pend = (pbeg + 16 < R) ? pbeg + 16 : R; // relative to a0
            const u32 lo = (i == 0) ? (u32)(A - a0) : 0u, hi = (u32)(pend - pbeg);
#pragma unroll
            for (u32 k = 0; k < 16; ++k) {
                if (k < hi) {
                    const u32 b = (k >= lo) ? ((wv[k >> 2] >> ((k & 3) * 8)) & 0xFFu) : 0u;
                    c = s_tab[(c >> 24) ^ b] ^ (c << 8);
                }
            }
            // to the end of the tile (or, in the last tile of the block, to the end of the block)
            const u64 tile_last = p0 + tile * 256ull + 255ull;            // last piece of a full tile
            const u64 last = tile_last < npieces - 1 ? tile_last : npieces - 1; // last piece of this tile
            if (pend == R) {
                // (the block's last piece: nothing behind it)
            } else if (last == npieces - 1 && (R & 15u)) {
                // pieces in front of a ragged last piece: x^(128 * pieces between) * x^(8 * its bytes)
                c = gf_mulmod(c, xp16[(u32)(last - 1 - i)]);
                c = gf_mulmod(c, gf_xpow_bytes(R & 15u, xp2));
            } else {
                c = gf_mulmod(c, xp16[(u32)(last - i)]);
            }
        }
        const u32 wx = wave_xor(c);
        if (lane_id() == 0) s_x[t >> 6] = wx;
        __syncthreads();
        if (t == 0) s_tile[tile] = s_x[0] ^ s_x[1] ^ s_x[2] ^ s_x[3];
        __syncthreads();
    }
    if (t == 0) {
        // chain the tiles of the slice, then move the slice to the end of the block
        u32 acc = 0;
        u64 end_piece = 0; // one past the last piece covered so far
        for (u32 tile = 0; tile < 16; ++tile) {
            const u64 first = p0 + tile * 256ull;
            if (first >= npieces) break;
            const u64 lastp = (first + 255ull < npieces - 1) ? first + 255ull : npieces - 1;
            // bytes this tile's result covers = up to the end of piece `lastp` (the block's end in the last tile)
            const u64 cover_beg = first * 16, cover_end = (lastp == npieces - 1) ? R : (lastp + 1) * 16;
            acc = gf_mulmod(acc, gf_xpow_bytes(cover_end - cover_beg, xp2)) ^ s_tile[tile];
            end_piece = cover_end;
        }
        if (npieces) acc = gf_mulmod(acc, gf_xpow_bytes(R - end_piece, xp2));
        // the initial 0xFFFFFFFF rides through the whole message
        if (blockIdx.x == 0) acc ^= gf_mulmod(0xFFFFFFFFu, gf_xpow_bytes(len, xp2));
        atomicXor(&a.crc[lb], acc); // (host applies the final NOT)
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------
// The same scan with FOUR byte positions per thread out of three aligned big-endian words (k_dec_scan loads seven single
// bytes per position: 1.6 G byte loads for a 226 MB stream, 0.9 ms; round 5).  `in` must be 4-byte aligned.
__global__ __launch_bounds__(256) void k_dec_scan4(const u8 *__restrict__ in, u64 nbytes, DecCand *__restrict__ cands,
                                                    u32 cap, u32 *__restrict__ count)
{
    const u64 t = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 p0 = t * 4u;
    if (p0 >= nbytes) return;
    const u32 *w = reinterpret_cast<const u32 *>(in);
    const u64 nwords = (nbytes + 3u) / 4u;
    const u32 r = (u32)(nbytes & 3u);
    u32 x[3];
#pragma unroll
    for (u32 k = 0; k < 3; ++k) {
        const u64 i = t + k;
        u32 v = 0;
        if (i + 1 == nwords && r) {
            // (the last, partial word byte by byte: a caller's buffer of exactly nbytes ends here -- bz_gpu_decode_device may be
            // handed a sub-range of a larger allocation or the last bytes of one)
            for (u32 b = 0; b < r; ++b) v |= (u32)in[i * 4u + b] << (24u - 8u * b);
        } else if (i < nwords) {
            v = __builtin_bswap32(w[i]);
        }
        x[k] = v;
    }
    const u64 hi = ((u64)x[0] << 32) | x[1];
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
        const u64 p = p0 + j;
        if (p >= nbytes) break;
        const u64 top = j ? ((hi << (8u * j)) | ((u64)x[2] >> (32u - 8u * j))) : hi; // the 64 bits from byte p on
        const u64 wdw = top >> 8;                                                       // 56 of them
#pragma unroll
        for (u32 s = 0; s < 8; ++s) {
            const u64 v = (wdw >> (8u - s)) & 0xFFFFFFFFFFFFull;
            const u32 type = (v == kBlockMagic) ? 1u : (v == kEosMagic ? 2u : 0u);
            if (type && p * 8 + s + 48 <= nbytes * 8) {
                const u32 i = atomicAdd(count, 1u);
                if (i < cap) {
                    cands[i].bitpos = p * 8 + s;
                    cands[i].type = type;
                    cands[i].pad = 0;
                }
            }
        }
    }
}

void launch_dec_scan(hipStream_t st, const u8 *in, u64 nbytes, DecCand *cands, u32 cap, u32 *count)
{
    (void)hipMemsetAsync(count, 0, 4, st);
    if (nbytes == 0) return;
    if ((reinterpret_cast<uintptr_t>(in) & 3u) == 0)
        hipLaunchKernelGGL(k_dec_scan4, dim3((u32)(((nbytes + 3) / 4 + 255) / 256)), dim3(256), 0, st, in, nbytes, cands, cap, count);
    else
        hipLaunchKernelGGL(k_dec_scan, dim3((u32)((nbytes + 255) / 256)), dim3(256), 0, st, in, nbytes, cands, cap, count);
}

void launch_dec_blocks(hipStream_t st, const u8 *in, u64 nbytes, const DecCand *cands, u32 ncand, DecBlockInfo *info,
                       u16 *sym, u8 *sel_scratch)
{
    if (ncand == 0) return;
    hipLaunchKernelGGL(k_dec_block, dim3(ncand), dim3(kD1Threads), 0, st, in, nbytes, cands, ncand, info, sym, sel_scratch);
}

// 64-byte windows of the input at the given (16-byte aligned) offsets, zeros behind the end of the input: what the host's
// record chain reads at stream ends, fetched for a whole batch at once
__global__ __launch_bounds__(256) void k_dec_gather_windows(const u8 *__restrict__ in, u64 nbytes, const u64 *__restrict__ bases, u32 nw,
                                                            u8 *__restrict__ out)
{
    const u32 i = blockIdx.x * 256u + threadIdx.x; // one thread per 16-byte piece
    if (i >= nw * 4u) return;
    const u64 p = bases[i >> 2] + (u64)(i & 3u) * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p + 16 <= nbytes) { // (the input is read as 32-bit words everywhere: no more alignment is asked of it)
        const u32 *w = reinterpret_cast<const u32 *>(in + p);
        v = make_uint4(w[0], w[1], w[2], w[3]);
    } else if (p < nbytes) {
        u32 t[4] = {0, 0, 0, 0};
        for (u64 b = p; b < nbytes; ++b) t[(b - p) >> 2] |= (u32)in[b] << (8u * ((b - p) & 3u));
        v = make_uint4(t[0], t[1], t[2], t[3]);
    }
    reinterpret_cast<uint4 *>(out)[i] = v;
}
void launch_dec_gather_windows(hipStream_t st, const u8 *in, u64 nbytes, const u64 *bases, u32 nw, u8 *out)
{
    if (nw) hipLaunchKernelGGL(k_dec_gather_windows, dim3((nw * 4u + 255u) / 256u), dim3(256), 0, st, in, nbytes, bases, nw, out);
}

void launch_dec_mtf(hipStream_t st, const DecArgs &a, KernelProf *prof, int *rec)
{
    *rec = prof ? prof->begin(st, KID_DEC_MTF, 0) : -1;
    // (the chunk kernels keep the launch of a full slot, 14 workgroups per block, most of which leave at once: launched
    // with just the workgroups the chunks need the stage is SLOWER -- level 9: 9.5 ms per GiB against 8.3, level 1: 30.0
    // against 17.1 --, with one workgroup per CU twice as slow; profiles/r04_off_default_configs.md)
    const u32 cw = a.cw; // (odd: see dec_engine.hip)
    const dim3 cgrid(cw, xcd_grid_y(a.nb));
    hipLaunchKernelGGL(k_dec_chunk_perm, cgrid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dec_compose, dim3(a.nb), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_dec_chunk_emit, cgrid, dim3(256), 0, st, a);
    if (prof) prof->end(st, *rec);
}

// st2 / ev: a second stream and two events, so that the second walk (little work, but as long as the
// longest segment's tail) runs beside the copy instead of behind it
void launch_dec_walks(hipStream_t st, const DecArgs &a, u32 walk_wgs, hipStream_t st2, hipEvent_t ev_a, hipEvent_t ev_b,
                      KernelProf *prof, int rec[4])
{
    const dim3 tiles(a.tiles, xcd_grid_y(a.nb));
    rec[0] = prof ? prof->begin(st, KID_DEC_TSORT, 0) : -1;
    hipLaunchKernelGGL(k_dec_thist, tiles, dim3(kSortThreads), 0, st, a);
    hipLaunchKernelGGL(k_dec_tscan, dim3(a.nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dec_tscatter, tiles, dim3(kSortThreads), 0, st, a);
    if (prof) prof->end(st, rec[0]);
    (void)hipMemsetAsync(a.work_ctr, 0, 2048, st);
    hipLaunchKernelGGL(k_dec_walk_meta, dim3((a.nb + 63) / 64), dim3(64), 0, st, a);
    const u32 wgs = (walk_wgs + 7u) & ~7u; // the same number of walkers on every XCD
    rec[1] = prof ? prof->begin(st, KID_DEC_WALK, 0) : -1;
    hipLaunchKernelGGL(k_dec_walk_lengths, dim3(wgs), dim3(256), 0, st, a);
    if (prof) prof->end(st, rec[1]);
    rec[2] = prof ? prof->begin(st, KID_DEC_PLACE, 0) : -1;
    hipLaunchKernelGGL(k_dec_rank_samples, dim3(a.nb), dim3(64), 0, st, a);
    (void)hipEventRecord(ev_a, st);
    (void)hipStreamWaitEvent(st2, ev_a, 0);
    hipLaunchKernelGGL(k_dec_walk_write, dim3(wgs), dim3(256), 0, st2, a);
    (void)hipEventRecord(ev_b, st2);
    hipLaunchKernelGGL(k_dec_seg_copy, dim3((kDecSamples + 15) / 16, a.nb), dim3(256), 0, st, a);
    (void)hipStreamWaitEvent(st, ev_b, 0);
    hipLaunchKernelGGL(k_dec_fixups, dim3(a.nb), dim3(256), 0, st, a);
    if (prof) prof->end(st, rec[2]);
    rec[3] = prof ? prof->begin(st, KID_DEC_RLE, 0) : -1;
    hipLaunchKernelGGL(k_dec_rle_sub, dim3(a.sub_wgs, a.nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dec_rle_chain, dim3(a.nb), dim3(1024), 0, st, a);
    if (prof) prof->end(st, rec[3]);
}

void launch_dec_expand(hipStream_t st, const DecArgs &a, const u64 *out_base, u8 *out)
{
    hipLaunchKernelGGL(k_dec_rle_expand, dim3(a.sub_wgs, a.nb), dim3(256), 0, st, a, out_base, out);
}

void launch_dec_crc(hipStream_t st, const DecArgs &a, const u64 *out_base, const u8 *out, u32 max_out_len,
                    const u32 *crc_tab, const u32 *xp2, const u32 *xp16)
{
    (void)hipMemsetAsync(a.crc, 0, (size_t)a.nb * 4u, st);
    const u32 slices = (max_out_len + 15u + 65535u) / 65536u + 1u; // (+15: the aligned start may lie in front of the block)
    hipLaunchKernelGGL(k_dec_crc, dim3(slices, a.nb), dim3(256), 0, st, a, out_base, out, crc_tab, xp2, xp16);
}

} // namespace bzgpu
