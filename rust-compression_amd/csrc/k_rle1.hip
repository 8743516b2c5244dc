// k_rle1.hip -- RLE1, block split and block CRC as data-parallel kernels.
//
// Reference being replaced: EncoderInner::next / write_rle
// (src/bzip2/encoder.rs:671-716), the block cut at :692 and the CRC update at
// :701-702 (src/crc32.rs:58-84, 128-149).
//
// Restatement that makes it parallel.  A "chunk" is a maximal same-byte run cut
// every 255 bytes from the run's start (encoder.rs:682).  For input position p
// with run start rs(p):  q = p - rs(p),  c = q mod 255 (position inside its
// chunk).  Position p emits its byte iff c < 4, and a count byte (c-3) iff it
// is the last byte of its chunk and c >= 3.  The RLE1 image of the whole input
// is therefore one exclusive prefix sum away, and blocks are contiguous slices
// of it: a block is cut right after the first chunk whose end brings the block
// to >= 100000*level-19 bytes (encoder.rs:186,692).  Only the cut positions are
// a serial chain (about 1200 per GiB).
//
// HBM traffic: the input is read 3x (tile scan+CRC, count, scatter) and the
// image written once -- all coalesced 16 B/lane.
#include "bzgpu.h"

namespace bzgpu {

constexpr u32 RT = 256; // threads per RLE tile

__device__ __forceinline__ i64 block_excl_max64(i64 v, i64 *sh /*[RT/64 + 1]*/, i64 &total)
{
    // exclusive max-scan over the workgroup in thread order
    const u32 l = lane_id(), w = threadIdx.x >> 6;
    i64 inc = wave_incl_max64(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    i64 carry = -1;
    for (u32 k = 0; k < w; ++k) carry = sh[k] > carry ? sh[k] : carry;
    i64 tot = -1;
    for (u32 k = 0; k < RT / 64; ++k) tot = sh[k] > tot ? sh[k] : tot;
    total = tot;
    i64 prev = __shfl_up(inc, 1, 64);
    i64 ex = (l == 0) ? (i64)-1 : prev;
    __syncthreads();
    return ex > carry ? ex : carry;
}

__device__ __forceinline__ u32 block_excl_sum(u32 v, u32 *sh /*[RT/64]*/, u32 &total)
{
    const u32 l = lane_id(), w = threadIdx.x >> 6;
    u32 inc = wave_incl_sum(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    u32 carry = 0, tot = 0;
    for (u32 k = 0; k < RT / 64; ++k) {
        if (k < w) carry += sh[k];
        tot += sh[k];
    }
    total = tot;
    __syncthreads();
    return carry + inc - v;
}

// load the 16 bytes of this thread (tile t, thread tid) + the byte before and after
struct Seg {
    u8 b[16];
    int prev;  // byte before p0 or -1
    int next;  // byte after the segment or -1 (end of input)
    u64 p0;
    u32 valid; // number of valid bytes (0..16)
    u32 skip;  // leading bytes that lie before the region's first byte (treated as not there)
};

// in_begin: first byte of the region being coded.  RLE1 restarted at a block cut is identical to
// RLE1 continued (a cut is a chunk start), so a rank may start at the cut it was handed.
__device__ __forceinline__ void load_seg(const u8 *__restrict__ in, u64 n, u64 tile, Seg &s, u64 in_begin = 0)
{
    s.p0 = tile * (u64)kRleTile + (u64)threadIdx.x * 16u;
    if (s.p0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + s.p0);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) s.b[k] = (u8)(w[k >> 2] >> ((k & 3) * 8));
        s.valid = 16;
    } else {
        s.valid = s.p0 < n ? (u32)(n - s.p0) : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) s.b[k] = (u32)k < s.valid ? in[s.p0 + k] : (u8)0;
    }
    s.prev = (s.p0 > in_begin && s.valid > 0) ? (int)in[s.p0 - 1] : -1;
    s.next = (s.p0 + s.valid < n && s.valid > 0) ? (int)in[s.p0 + s.valid] : -1;
    s.skip = (in_begin > s.p0) ? (u32)((in_begin - s.p0) < 16u ? (in_begin - s.p0) : 16u) : 0u;
}

// ---- kernel A: per-tile last run start + per-tile raw CRC ---------------------
// crc_tab: 256-entry byte table (src/crc32.rs:58-72); xp16: x^(8*16*k) mod P, k=0..255
__global__ __launch_bounds__(RT) void k_rle_tile_scan(const u8 *__restrict__ in, u64 n, u64 t0, u64 in_begin,
                                                       const u32 *__restrict__ crc_tab,
                                                       const u32 *__restrict__ xp16,
                                                       i64 *__restrict__ tile_last_start,
                                                       u32 *__restrict__ tile_crc)
{
    __shared__ u32 s_tab[256];
    __shared__ i64 s_max[RT / 64];
    __shared__ u32 s_x[RT / 64];
    s_tab[threadIdx.x] = crc_tab[threadIdx.x];
    __syncthreads();

    const u64 tile = t0 + blockIdx.x;
    Seg s;
    load_seg(in, n, tile, s, in_begin);

    i64 last = -1;
    int pb = s.prev;
    u32 crc = 0;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        if (k < s.valid) {
            if (k >= s.skip) {
                if ((int)s.b[k] != pb) last = (i64)(s.p0 + k);
                pb = s.b[k];
            }
            crc = s_tab[(crc >> 24) ^ s.b[k]] ^ (crc << 8); // the tile CRC covers every byte of the tile
        }
    }
    // shift this thread's CRC to the end of the tile: bytes after it inside the tile
    const u64 tile_beg = tile * (u64)kRleTile;
    const u32 tile_len = (u32)((n - tile_beg) < (u64)kRleTile ? (n - tile_beg) : (u64)kRleTile);
    if (s.valid > 0) {
        const u32 after = tile_len - (threadIdx.x * 16u + s.valid);
        if ((after & 15u) == 0) {
            crc = gf_mulmod(crc, xp16[after >> 4]);
        } else { // only in the last, partial tile
            u32 m = gf_mulmod(xp16[after >> 4], 1u);
            for (u32 k = 0; k < (after & 15u); ++k) m = gf_mulmod(m, 0x100u); // * x^8
            crc = gf_mulmod(crc, m);
        }
    } else {
        crc = 0;
    }
    i64 wl = last;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        i64 o = __shfl_xor(wl, d, 64);
        wl = o > wl ? o : wl;
    }
    const u32 wx = wave_xor(crc);
    if (lane_id() == 0) {
        s_max[threadIdx.x >> 6] = wl;
        s_x[threadIdx.x >> 6] = wx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        i64 m = -1;
        u32 x = 0;
        for (u32 k = 0; k < RT / 64; ++k) {
            m = s_max[k] > m ? s_max[k] : m;
            x ^= s_x[k];
        }
        tile_last_start[tile] = m;
        tile_crc[tile] = x;
    }
}

// ---- kernels B and D: exclusive scans over the tiles of a range, in three small launches -----------------
// (one workgroup walking 262 144 tiles took 0.44 ms per scan and left the chip idle)
//   _local : a workgroup scans 1024 consecutive tiles, writes the tile's value relative to the workgroup and the
//            workgroup's total;   _parts : one workgroup scans the totals;   _add : every tile takes its workgroup's.
// IS_MAX: max-scan of i64 run starts (identity -1, carried in from `init`); else sum of u32 counts into u64 offsets.
template <bool IS_MAX> struct TileScan;
template <> struct TileScan<true> {
    typedef i64 In;
    typedef i64 Out;
    static __device__ __forceinline__ i64 ident() { return -1; }
    static __device__ __forceinline__ i64 op(i64 a, i64 b) { return a > b ? a : b; }
};
template <> struct TileScan<false> {
    typedef u32 In;
    typedef u64 Out;
    static __device__ __forceinline__ u64 ident() { return 0; }
    static __device__ __forceinline__ u64 op(u64 a, u64 b) { return a + b; }
};
template <bool IS_MAX> __device__ __forceinline__ typename TileScan<IS_MAX>::Out wave_incl(typename TileScan<IS_MAX>::Out v)
{
    typedef TileScan<IS_MAX> S;
    const u32 l = lane_id();
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const typename S::Out t = __shfl_up(v, d, 64);
        if (l >= d) v = S::op(t, v);
    }
    return v;
}
// exclusive scan of one value per thread over a 1024-thread workgroup (thread order); total = the whole
template <bool IS_MAX>
__device__ __forceinline__ typename TileScan<IS_MAX>::Out wg_excl(typename TileScan<IS_MAX>::Out v,
                                                                     typename TileScan<IS_MAX>::Out *sh /*[16]*/,
                                                                     typename TileScan<IS_MAX>::Out &total)
{
    typedef TileScan<IS_MAX> S;
    const u32 l = lane_id(), w = threadIdx.x >> 6;
    const typename S::Out inc = wave_incl<IS_MAX>(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    typename S::Out carry = S::ident(), tot = S::ident();
    for (u32 k = 0; k < 16; ++k) {
        if (k < w) carry = S::op(carry, sh[k]);
        tot = S::op(tot, sh[k]);
    }
    total = tot;
    typename S::Out prev = __shfl_up(inc, 1, 64);
    if (l == 0) prev = S::ident();
    __syncthreads();
    return S::op(carry, prev);
}
template <bool IS_MAX>
__global__ __launch_bounds__(1024) void k_tiles_scan_local(const typename TileScan<IS_MAX>::In *__restrict__ in,
                                                           typename TileScan<IS_MAX>::Out *__restrict__ out, u64 t0, u64 t1,
                                                           typename TileScan<IS_MAX>::Out *__restrict__ part)
{
    typedef TileScan<IS_MAX> S;
    __shared__ typename S::Out sh[16];
    const u64 t = t0 + (u64)blockIdx.x * 1024u + threadIdx.x;
    const typename S::Out v = t < t1 ? (typename S::Out)in[t] : S::ident();
    typename S::Out total;
    const typename S::Out ex = wg_excl<IS_MAX>(v, sh, total);
    if (t < t1) out[t] = ex;
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}
// part[b] <- what is carried into workgroup b (from `init`); *whole (and *whole2, if given) <- the scan's end value
template <bool IS_MAX>
__global__ __launch_bounds__(1024) void k_tiles_scan_parts(typename TileScan<IS_MAX>::Out *__restrict__ part, u32 nparts,
                                                           typename TileScan<IS_MAX>::Out init,
                                                           typename TileScan<IS_MAX>::Out *__restrict__ whole,
                                                           typename TileScan<IS_MAX>::Out *__restrict__ whole2)
{
    typedef TileScan<IS_MAX> S;
    __shared__ typename S::Out sh[16];
    typename S::Out run = init;
    for (u32 b0 = 0; b0 < nparts; b0 += 1024u) {
        const u32 b = b0 + threadIdx.x;
        const typename S::Out v = b < nparts ? part[b] : S::ident();
        typename S::Out total;
        const typename S::Out ex = wg_excl<IS_MAX>(v, sh, total);
        if (b < nparts) part[b] = S::op(run, ex);
        run = S::op(run, total);
    }
    if (threadIdx.x == 0) {
        if (whole) *whole = run;
        if (whole2) *whole2 = run;
    }
}
template <bool IS_MAX>
__global__ __launch_bounds__(1024) void k_tiles_scan_add(typename TileScan<IS_MAX>::Out *__restrict__ out, u64 t0, u64 t1,
                                                         const typename TileScan<IS_MAX>::Out *__restrict__ part)
{
    typedef TileScan<IS_MAX> S;
    const u64 t = t0 + (u64)blockIdx.x * 1024u + threadIdx.x;
    if (t < t1) out[t] = S::op(part[blockIdx.x], out[t]);
}
template <bool IS_MAX>
static void tiles_scan(hipStream_t st, const typename TileScan<IS_MAX>::In *in, typename TileScan<IS_MAX>::Out *out, u64 t0,
                       u64 t1, typename TileScan<IS_MAX>::Out *part, typename TileScan<IS_MAX>::Out init,
                       typename TileScan<IS_MAX>::Out *whole, typename TileScan<IS_MAX>::Out *whole2)
{
    const u32 nparts = (u32)((t1 - t0 + 1023u) / 1024u);
    hipLaunchKernelGGL((k_tiles_scan_local<IS_MAX>), dim3(nparts), dim3(1024), 0, st, in, out, t0, t1, part);
    hipLaunchKernelGGL((k_tiles_scan_parts<IS_MAX>), dim3(1), dim3(1024), 0, st, part, nparts, init, whole, whole2);
    hipLaunchKernelGGL((k_tiles_scan_add<IS_MAX>), dim3(nparts), dim3(1024), 0, st, out, t0, t1, part);
}

// Per-thread RLE1 evaluation of its 16 positions.
//   rs_in : run start that is live when the segment begins (from earlier threads/tiles), -1 if p0==0
// Produces the emitted-byte count, and (optionally) the bytes.
struct SegEval {
    u32 count;      // bytes this segment emits
    u32 c_first;    // chunk phase of the first position
};

__device__ __forceinline__ u32 phase_of(u64 p, i64 rs)
{
    return (u32)((p - (u64)rs) % 255u);
}

// fills e[k] = number of bytes position k emits (0,1,2); returns sum
__device__ __forceinline__ u32 eval_seg(const Seg &s, i64 rs_in, u8 e[16], u8 cph[16])
{
    u32 sum = 0;
    i64 rs = rs_in;
    int pb = s.prev;
    u32 c = 0;
    bool have_c = false;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        u32 ek = 0;
        u32 ck = 0;
        if (k < s.valid && k >= s.skip) {
            const u64 p = s.p0 + k;
            if ((int)s.b[k] != pb) {
                rs = (i64)p;
                c = 0;
                have_c = true;
            } else if (!have_c) {
                c = phase_of(p, rs);
                have_c = true;
            } else {
                c = (c == 254u) ? 0u : c + 1u;
            }
            pb = s.b[k];
            const int nb = (k + 1 < s.valid) ? (int)s.b[k + 1 < 16 ? k + 1 : 15] : s.next;
            const bool chunk_end = (nb != (int)s.b[k]) || (c == 254u);
            ek = (c < 4u ? 1u : 0u) + ((chunk_end && c >= 3u) ? 1u : 0u);
            ck = c;
        }
        e[k] = (u8)ek;
        cph[k] = (u8)ck;
        sum += ek;
    }
    return sum;
}

// run start live at the beginning of each thread's segment
__device__ __forceinline__ i64 seg_run_start(const Seg &s, i64 tile_carry, i64 *sh)
{
    i64 last = -1;
    int pb = s.prev;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        if (k < s.valid && k >= s.skip) {
            if ((int)s.b[k] != pb) last = (i64)(s.p0 + k);
            pb = s.b[k];
        }
    }
    i64 tot;
    i64 ex = block_excl_max64(last, sh, tot);
    return ex > tile_carry ? ex : tile_carry;
}

// ---- kernel C: bytes emitted per tile -------------------------------------------
// Also records, for each 256-byte sub-tile, the bytes emitted before it inside its tile and the run
// start live at its first byte -- the cut chain uses them to look at 2 KiB instead of whole tiles.
__global__ __launch_bounds__(RT) void k_rle_count(const u8 *__restrict__ in, u64 n, u64 t0, u64 in_begin,
                                                   const i64 *__restrict__ carry_in,
                                                   u32 *__restrict__ tile_count, u16 *__restrict__ sub_off,
                                                   i64 *__restrict__ sub_rs)
{
    __shared__ i64 s_m[RT / 64 + 1];
    __shared__ u32 s_s[RT / 64];
    const u64 tile = t0 + blockIdx.x;
    Seg s;
    load_seg(in, n, tile, s, in_begin);
    const i64 rs = seg_run_start(s, carry_in[tile], s_m);
    u8 e[16], cph[16];
    const u32 cnt = eval_seg(s, rs, e, cph);
    u32 tot;
    const u32 ex = block_excl_sum(cnt, s_s, tot);
    if (threadIdx.x == 0) tile_count[tile] = tot;
    if ((threadIdx.x & 15u) == 0) {
        sub_off[tile * 16u + (threadIdx.x >> 4)] = (u16)ex;
        sub_rs[tile * 16u + (threadIdx.x >> 4)] = rs;
    }
}

// ---- kernel E: scatter the RLE1 image -------------------------------------------
__global__ __launch_bounds__(RT) void k_rle_scatter(const u8 *__restrict__ in, u64 n, u64 t0, u64 in_begin,
                                                     const i64 *__restrict__ carry_in,
                                                     const u64 *__restrict__ tile_off,
                                                     u8 *__restrict__ rle)
{
    __shared__ i64 s_m[RT / 64 + 1];
    __shared__ u32 s_s[RT / 64];
    __shared__ __attribute__((aligned(16))) u8 s_out[2 * kRleTile + 32];
    const u64 tile = t0 + blockIdx.x;
    Seg s;
    load_seg(in, n, tile, s, in_begin);
    const i64 rs = seg_run_start(s, carry_in[tile], s_m);
    u8 e[16], cph[16];
    const u32 cnt = eval_seg(s, rs, e, cph);
    u32 tot;
    // The tile's bytes are staged at the offset their destination has inside a 16-byte line, so that the copy out is
    // aligned 16-byte stores of aligned 16-byte LDS reads (rounds 1-4a: one BYTE per lane and store instruction, sixteen
    // store instructions per thread); only the ragged first and last line go out byte by byte.
    u8 *dst = rle + tile_off[tile];
    const u32 a0 = (u32)(reinterpret_cast<uintptr_t>(dst) & 15u);
    u32 o = a0 + block_excl_sum(cnt, s_s, tot);
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        if (e[k]) {
            if (cph[k] < 4u) s_out[o++] = s.b[k];
            if (e[k] == 2u || cph[k] >= 4u) s_out[o++] = (u8)(cph[k] - 3u);
        }
    }
    __syncthreads();
    u8 *line0 = dst - a0; // 16-byte aligned
    const u32 end = a0 + tot, nlines = (end + 15u) >> 4;
    for (u32 j = threadIdx.x; j < nlines; j += RT) {
        const u32 lo = j * 16u;
        if (lo >= a0 && lo + 16u <= end) {
            *reinterpret_cast<uint4 *>(line0 + lo) = *reinterpret_cast<const uint4 *>(s_out + lo);
        } else {
            const u32 b0 = lo > a0 ? lo : a0, b1 = lo + 16u < end ? lo + 16u : end;
            for (u32 b = b0; b < b1; ++b) line0[b] = s_out[b];
        }
    }
}

// ---- kernel F: the chain of block cuts (one workgroup) -------------------------------------
// A cut depends on the previous one, but only weakly: a block holds between L and L+4 bytes
// (L = 100000*level-19; the last chunk adds at most 5), so the k-th next block starts inside
// [s + kL, s + k(L+4)].  Each of the CW waves therefore works on one FUTURE cut speculatively: it
// finds the 256-byte sub-tile holding its earliest possible target, evaluates RLE1 over a window
// from there into LDS, and lane i resolves the cut for the candidate target T+i.  The real chain
// is then CW table lookups.  One round costs the latency of ~4 dependent global loads instead of
// CW x that.  Windows are 2 KiB; data with long runs (63 output bytes can span 3.2 KB of input)
// falls back to 12 KiB windows for the round that needs it.
// Emits BlockDesc records.  emit_tail: also emit the remaining partial block.
constexpr u32 CW = 16;                               // speculative cuts per round (waves)
constexpr u32 CT = CW * 64;                          // threads
constexpr u32 CSW_SMALL = 2, CSW_BIG = 12;           // sweeps of 64 segments (1 KiB) per window
constexpr u32 CSEG = CSW_BIG * 64;                   // LDS slots per wave
constexpr u64 CUT_END = ~0ull, CUT_OUT = ~0ull - 1;

// Tiles [t_begin, ntiles) are in play; tile_off is relative to t_begin; the first block starts at
// input byte in_begin (a cut handed over by the previous rank, or 0).
__global__ __launch_bounds__(CT) void k_rle_cuts(const u8 *__restrict__ in, u64 n, u64 t_begin, u64 in_begin,
                                                  const u16 *__restrict__ sub_off,
                                                  const i64 *__restrict__ sub_rs,
                                                  const u64 *__restrict__ tile_off, u64 ntiles,
                                                  u32 block_max_len, int emit_tail,
                                                  BlockDesc *__restrict__ blocks, u32 max_blocks,
                                                  u64 *__restrict__ out_nblocks_consumed /*[3]*/)
{
    __shared__ u16 s_pre[CW][CSEG];    // inclusive count of emitted bytes, per segment, from the window start
    __shared__ u32 s_eb[CW][CSEG];     // 2 bits per position: bytes it emits
    __shared__ u16 s_ce[CW][CSEG];     // chunk-end mask
    __shared__ u64 s_res_rle[CW][64], s_res_in[CW][64];
    __shared__ u64 s_tlo[CW];

    const u32 w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const u64 L = block_max_len;
    const u64 M = tile_off[ntiles];
    u64 s_rle = 0, s_in = in_begin;
    u64 cur_tile = t_begin; // tile_off[cur_tile] < every later target
    u32 nb = 0;
    bool broken = false, ended = false; // ended: the next target's chunk ends behind the range (see phase 2b)
    u32 nsw = CSW_SMALL; // sweeps per window in this round

    while (!broken && !ended) {
        if (M < s_rle + L || M == s_rle) break; // no chunk end reaches the limit any more
        // ---- phase 1: this wave's window ------------------------------------------------
        const u64 T_lo = s_rle + (u64)(w + 1) * L; // smallest possible target of step w
        const bool alive = T_lo <= M;
        u64 t_lo = cur_tile;
        u64 in0 = 0, base_off = 0;
        i64 rs_carry = -1;
        if (alive) {
            // largest t with tile_off[t] < T_lo, galloping from cur_tile (tile_off[ntiles] = M >= T_lo)
            u64 base = cur_tile;
            while (true) { // coarse: 64 probes, 64 tiles apart
                u64 t = base + (u64)(l + 1) * 64u;
                if (t > ntiles) t = ntiles;
                const u64 bal = __ballot(tile_off[t] >= T_lo);
                if (bal) {
                    base += (u64)__builtin_ctzll(bal) * 64u;
                    break;
                }
                base += 64u * 64u;
            }
            {
                u64 t = base + (u64)(l + 1);
                if (t > ntiles) t = ntiles;
                const u64 bal = __ballot(tile_off[t] >= T_lo); // lane 63 always hits
                t_lo = base + (u64)__builtin_ctzll(bal);
            }
            // sub-tile: the last of the 16 whose start offset is still below T_lo
            const u64 toff = tile_off[t_lo];
            const u32 so = sub_off[t_lo * 16u + (l & 15u)];
            const u64 bal = __ballot((l < 16u) && (toff + so < T_lo)); // bit 0 always set
            const u32 u = 63u - (u32)__builtin_clzll(bal);
            in0 = t_lo * (u64)kRleTile + (u64)u * 256u;
            base_off = toff + (u64)__shfl(so, (int)u, 64);
            rs_carry = sub_rs[t_lo * 16u + u];
        }
        // ---- phase 2: RLE1 over the window, 64 consecutive segments (1 KiB) per sweep ----------------
        if (alive) {
            u32 off_carry = 0;
            int prev_carry = (in0 > in_begin) ? (int)in[in0 - 1] : -1;
            uint4 cur;
            u32 cur_valid;
            {
                const u64 p0 = in0 + (u64)l * 16u;
                if (p0 + 16 <= n) {
                    cur = *reinterpret_cast<const uint4 *>(in + p0);
                    cur_valid = 16;
                } else {
                    u32 t[4] = {0, 0, 0, 0};
                    cur_valid = p0 < n ? (u32)(n - p0) : 0u;
                    for (u32 k = 0; k < cur_valid; ++k) t[k >> 2] |= (u32)in[p0 + k] << ((k & 3u) * 8u);
                    cur = make_uint4(t[0], t[1], t[2], t[3]);
                }
            }
            for (u32 q = 0; q < nsw; ++q) {
                // prefetch the next sweep while this one is evaluated
                uint4 nxt = make_uint4(0, 0, 0, 0);
                u32 nxt_valid = 0;
                {
                    const u64 p0 = in0 + (u64)((q + 1) * 64u + l) * 16u;
                    if (p0 + 16 <= n) {
                        nxt = *reinterpret_cast<const uint4 *>(in + p0);
                        nxt_valid = 16;
                    } else {
                        u32 t[4] = {0, 0, 0, 0};
                        nxt_valid = p0 < n ? (u32)(n - p0) : 0u;
                        for (u32 k = 0; k < nxt_valid; ++k) t[k >> 2] |= (u32)in[p0 + k] << ((k & 3u) * 8u);
                        nxt = make_uint4(t[0], t[1], t[2], t[3]);
                    }
                }
                Seg sg;
                sg.p0 = in0 + (u64)(q * 64u + l) * 16u;
                sg.valid = cur_valid;
                sg.skip = (in_begin > sg.p0) ? (u32)((in_begin - sg.p0) < 16u ? (in_begin - sg.p0) : 16u) : 0u;
                const u32 wv[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) sg.b[k] = (u8)(wv[k >> 2] >> ((k & 3) * 8));
                // neighbours' bytes by shuffle: previous segment's last byte, next segment's first
                const int lastb = sg.valid ? (int)sg.b[(sg.valid - 1) & 15] : -1;
                const int firstb = sg.valid ? (int)sg.b[0] : -1;
                int pv = __shfl_up(lastb, 1, 64);
                if (l == 0) pv = prev_carry;
                int nx = __shfl_down(firstb, 1, 64);
                const u32 nfw = __shfl(nxt.x, 0, 64), nfv = __shfl(nxt_valid, 0, 64);
                if (l == 63) nx = nfv ? (int)(nfw & 0xFFu) : -1;
                sg.prev = (sg.valid && sg.p0 > in_begin) ? pv : -1;
                sg.next = (sg.valid == 16) ? nx : -1; // a short segment ends the input
                i64 last = -1;
                int pb = sg.prev;
#pragma unroll
                for (u32 k = 0; k < 16; ++k) {
                    if (k < sg.valid && k >= sg.skip) {
                        if ((int)sg.b[k] != pb) last = (i64)(sg.p0 + k);
                        pb = sg.b[k];
                    }
                }
                const i64 inc = wave_incl_max64(last);
                const i64 prv = __shfl_up(inc, 1, 64);
                i64 rs = (l == 0) ? (i64)-1 : prv;
                rs = rs > rs_carry ? rs : rs_carry;
                u8 e[16], cph[16];
                const u32 c = eval_seg(sg, rs, e, cph);
                u32 eb = 0, ce = 0;
#pragma unroll
                for (u32 k = 0; k < 16; ++k) {
                    if (k < sg.valid && k >= sg.skip) {
                        const int nbyte = (k + 1 < sg.valid) ? (int)sg.b[k + 1 < 16 ? k + 1 : 15] : sg.next;
                        if (nbyte != (int)sg.b[k] || cph[k] == 254u) ce |= 1u << k;
                        eb |= (u32)e[k] << (2u * k);
                    }
                }
                const u32 incs = wave_incl_sum(c);
                s_pre[w][q * 64u + l] = (u16)(off_carry + incs);
                s_eb[w][q * 64u + l] = eb;
                s_ce[w][q * 64u + l] = (u16)ce;
                const i64 winc = __shfl(inc, 63, 64);
                rs_carry = winc > rs_carry ? winc : rs_carry;
                off_carry += __shfl(incs, 63, 64);
                prev_carry = __shfl(lastb, 63, 64);
                cur = nxt;
                cur_valid = nxt_valid;
            }
        }
        __syncthreads();
        // ---- phase 2b: lane i resolves the cut for target T_lo + i -----------------------------
        {
            const u32 nseg = nsw * 64u;
            u64 r_rle = CUT_OUT, r_in = 0;
            const u64 tgt = T_lo + l;
            if (!alive || tgt > M) {
                r_rle = CUT_END;
            } else if (tgt - base_off <= (u64)s_pre[w][nseg - 1]) {
                const u32 rel = (u32)(tgt - base_off);
                u32 lo = 0, hi = nseg - 1;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if ((u32)s_pre[w][mid] >= rel) hi = mid; else lo = mid + 1;
                }
                u32 seg = lo, k = 0;
                u32 acc = seg ? (u32)s_pre[w][seg - 1] : 0u;
                u32 eb = s_eb[w][seg];
                for (; k < 16; ++k) {
                    acc += (eb >> (2u * k)) & 3u;
                    if (acc >= rel) break;
                }
                // on to the end of the chunk
                bool ok = true;
                while (!((s_ce[w][seg] >> k) & 1u)) {
                    if (++k == 16u) {
                        k = 0;
                        if (++seg == nseg) {
                            ok = false;
                            break;
                        }
                        eb = s_eb[w][seg];
                    }
                    acc += (eb >> (2u * k)) & 3u;
                }
                if (ok) {
                    r_rle = base_off + acc;
                    r_in = in0 + (u64)seg * 16u + k + 1u;
                    // A chunk that BEGINS in the range and ends behind it (a run across the edge of a slab: its count
                    // byte belongs to the run's last input byte, so the range's image ends in mid-chunk) closes no
                    // block of this range: the block it would close begins here and is the next rank's first one
                    // (that rank codes the bytes in front of its slab afresh, from this block's start).
                    if (r_rle > M) r_rle = CUT_END;
                }
            }
            s_res_rle[w][l] = r_rle;
            s_res_in[w][l] = r_in;
            if (l == 0) s_tlo[w] = t_lo;
        }
        __syncthreads();
        // ---- phase 3: the real chain, CW table lookups (every thread does the same) ---------------
        const u64 round_s = s_rle;
        bool need_big = false;
        for (u32 q = 0; q < CW; ++q) {
            const u64 target = s_rle + L;
            if (M < target || M == s_rle) break;
            const u64 idx = target - (round_s + (u64)(q + 1) * L);
            if (idx > 63) { // cannot happen for q <= 15 (jitter <= 4 per step); stop the round
                if (q == 0) broken = true;
                break;
            }
            const u64 r = s_res_rle[q][idx];
            if (r == CUT_END) {
                ended = true;
                break;
            }
            if (r == CUT_OUT) {
                if (q == 0) {
                    if (nsw == CSW_BIG) broken = true; // 12 KiB always hold step 0
                    need_big = true;
                }
                break;
            }
            const u64 cut_in = s_res_in[q][idx];
            if (threadIdx.x == 0 && nb < max_blocks) {
                BlockDesc d;
                d.rle_off = s_rle;
                d.in_off = s_in;
                d.in_end = cut_in;
                d.n = (u32)(r - s_rle);
                d.pad = 0;
                blocks[nb] = d;
            }
            ++nb;
            s_rle = r;
            s_in = cut_in;
            cur_tile = s_tlo[q];
        }
        nsw = need_big ? CSW_BIG : CSW_SMALL;
        __syncthreads();
    }
    u32 tail = 0;
    if (emit_tail && M > s_rle && !broken) {
        tail = 1;
        if (threadIdx.x == 0 && nb < max_blocks) {
            BlockDesc d;
            d.rle_off = s_rle;
            d.in_off = s_in;
            d.in_end = n;
            d.n = (u32)(M - s_rle);
            d.pad = 0;
            blocks[nb] = d;
        }
        ++nb;
        s_in = n;
    }
    if (threadIdx.x == 0) {
        out_nblocks_consumed[0] = broken ? ~0ull : nb;
        out_nblocks_consumed[1] = s_in;
        out_nblocks_consumed[2] = tail; // 1: the last block is the unfinished tail, not one closed by a cut
    }
}

// ---- kernels H: the cuts from tables (no chain of launches-worth of latency, and nothing to wait for) ------
// Block j of the whole input starts at image offset S_j, and S_j lies in [j L, j (L + 4)] (a block holds L ... L + 4
// bytes), so the step S_j -> S_{j+1} -- "the end of the first chunk that ends at or behind the TARGET S_j + L" -- has
// only 4 j + 1 possible arguments.  All of them are resolved in parallel, position by position: a chunk that ends at
// image offset c_e and holds `em` bytes answers every target in (c_e - em, c_e].  A table entry is
//     (c_e - target)  [3 bits: 0 .. 4]   |   (input position behind the chunk - first byte of the step's first tile) << 3
// and the real chain is then one look-up per block -- or one per SIXTEEN blocks through the composed table
// (`k_cut_compose`).  Offsets are those of the image of the WHOLE input (g_base = offset of the range's first byte:
// a rank of a sharded job knows it from an all-gather of the slabs' totals) and so is j: a rank fills the tables for
// the targets inside its slab, (g_base, own_hi], BEFORE the cut of the rank in front of it arrives, and what is left
// of its link of the chain is `k_cut_select`: about a hundred dependent loads.  0xFFFFFFFF = a target of another rank.
__host__ __device__ __forceinline__ u64 cut_tbase(u64 j_lo, u64 j) // entries of the steps j_lo ... j - 1
{
    return 2u * (j * (j - 1u) - j_lo * (j_lo - 1u)) + (j - j_lo);
}
__host__ __device__ __forceinline__ u64 cut_cbase(u64 g_lo, u64 g) // composed entries of the groups g_lo ... g - 1
{
    return 32u * (g * (g - 1u) - g_lo * (g_lo - 1u)) + (g - g_lo);
}

// largest t in [lo, hi] with off[t] < v (off[lo] < v is given)
__device__ __forceinline__ u64 cut_last_below(const u64 *__restrict__ off, u64 lo, u64 hi, u64 v)
{
    while (lo < hi) {
        const u64 mid = (lo + hi + 1) >> 1;
        if (off[mid] < v) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// one thread per step: the tiles that hold the chunk ends of its targets
__global__ __launch_bounds__(256) void k_cut_steps(CutPlan pl, const u64 *__restrict__ tile_off, u64 *__restrict__ step_t0,
                                                    u32 *__restrict__ step_nt)
{
    const u64 gid = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 j = pl.j_lo + gid;
    if (j > pl.j_hi) return;
    const u64 lb = (j + 1u) * pl.L;
    const u64 w_lo = lb > pl.w_min ? lb : pl.w_min;
    const u64 w_hi = lb + 4u * j < pl.own_hi ? lb + 4u * j : pl.own_hi;
    u64 ta = pl.tb;
    u32 nt = 0;
    if (w_lo <= w_hi) {
        // (targets at or below g_base: the chunk that answers them began in front of the range and ends in its first tiles)
        if (w_lo > pl.g_base) ta = cut_last_below(tile_off, pl.tb, pl.t1, w_lo - pl.g_base);
        u64 te = (w_hi > pl.g_base ? cut_last_below(tile_off, ta, pl.t1, w_hi - pl.g_base) : ta) + 1u; // (a chunk may end in the tile behind)
        if (te > pl.t_eval) te = pl.t_eval;
        if (ta > pl.t_eval) ta = pl.t_eval;
        nt = (u32)(te - ta + 1u);
    }
    step_t0[gid] = ta;
    step_nt[gid] = nt;
}

// exclusive sum of step_nt -> step_w0[0 .. nsteps], one workgroup
__global__ __launch_bounds__(1024) void k_cut_steps_scan(const u32 *__restrict__ step_nt, u64 nsteps, u64 *__restrict__ step_w0)
{
    __shared__ u64 sh[16];
    u64 run = 0;
    for (u64 b0 = 0; b0 < nsteps; b0 += 1024u) {
        const u64 b = b0 + threadIdx.x;
        const u64 v = b < nsteps ? (u64)step_nt[b] : 0u;
        u64 total;
        const u64 ex = wg_excl<false>(v, sh, total);
        if (b < nsteps) step_w0[b] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) step_w0[nsteps] = run;
}

// one workgroup per (step, tile): the RLE1 evaluation of k_rle_count, and every chunk end answers its targets
__global__ __launch_bounds__(RT) void k_cut_table(CutPlan pl, const u8 *__restrict__ in, u64 n_lim,
                                                   const i64 *__restrict__ carry_in, const u64 *__restrict__ tile_off,
                                                   const u64 *__restrict__ step_t0, const u64 *__restrict__ step_w0,
                                                   u64 nsteps, u32 *__restrict__ tab)
{
    __shared__ i64 s_m[RT / 64 + 1];
    __shared__ u32 s_s[RT / 64];
    // the step this workgroup belongs to: last gid with step_w0[gid] <= blockIdx.x
    u64 lo = 0, hi = nsteps - 1u;
    while (lo < hi) {
        const u64 mid = (lo + hi + 1) >> 1;
        if (step_w0[mid] <= (u64)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const u64 gid = lo, j = pl.j_lo + gid;
    const u64 t_first = step_t0[gid];
    const u64 tile = t_first + ((u64)blockIdx.x - step_w0[gid]);
    const u64 lb = (j + 1u) * pl.L;
    const u64 w_lo = lb > pl.w_min ? lb : pl.w_min;
    const u64 w_hi = lb + 4u * j < pl.own_hi ? lb + 4u * j : pl.own_hi;
    const u64 in_ref = t_first * (u64)kRleTile;
    u32 *__restrict__ row = tab + cut_tbase(pl.j_lo, j);

    Seg s;
    load_seg(in, n_lim, tile, s, 0);
    const i64 rs = seg_run_start(s, carry_in[tile], s_m);
    u8 e[16], cph[16];
    const u32 cnt = eval_seg(s, rs, e, cph);
    u32 tot;
    const u32 ex = block_excl_sum(cnt, s_s, tot);
    u64 cum = pl.g_base + tile_off[tile] + ex;
    if (cum + cnt < w_lo || cum > w_hi + 4u) return; // none of this segment's chunk ends answers a target of the step
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        if (k < s.valid && k >= s.skip) {
            cum += e[k];
            const int nbyte = (k + 1 < s.valid) ? (int)s.b[k + 1 < 16 ? k + 1 : 15] : s.next;
            if (nbyte != (int)s.b[k] || cph[k] == 254u) {
                const u32 c = cph[k];
                const u64 em = c < 3u ? c + 1u : 5u;
                const u64 t_lo = cum - em + 1u > w_lo ? cum - em + 1u : w_lo;
                const u64 t_hi = cum < w_hi ? cum : w_hi;
                const u32 rel = (u32)(s.p0 + k + 1u - in_ref) << 3;
                for (u64 T = t_lo; T <= t_hi; ++T) row[T - lb] = (u32)(cum - T) | rel;
            }
        }
    }
}

// sixteen steps in one: comp[g][i] = how far a block start with index i at step 16 g has drifted sixteen steps on
// (0xFFFF: one of them leaves this rank's targets)
__global__ __launch_bounds__(256) void k_cut_compose(CutPlan pl, u64 g_lo, const u32 *__restrict__ tab, u16 *__restrict__ comp)
{
    const u64 g = g_lo + blockIdx.y;
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 j0 = g * 16u;
    if (i > 4u * j0) return;
    u64 idx = i;
    bool ok = true;
    for (u32 q = 0; q < 16u; ++q) {
        const u32 en = tab[cut_tbase(pl.j_lo, j0 + q) + idx];
        if (en == 0xFFFFFFFFu) {
            ok = false;
            break;
        }
        idx += en & 7u;
    }
    comp[cut_cbase(g_lo, g) + i] = ok ? (u16)(idx - i) : (u16)0xFFFFu;
}

// The chain itself: from block j0 with image offset s0 (its first input byte: start_in).  Thread 0 follows the
// composed tables and notes where every stretch begins; the threads then walk the stretches and write the records.
// res[0] blocks (~0: the tables did not answer an owned target -- never seen; the caller falls back to k_rle_cuts),
// res[1] input bytes consumed, res[2] tail flag, res[3] image offset and res[4] number of the next rank's first block.
constexpr u32 kCutSegCap = 4096;
__global__ __launch_bounds__(256) void k_cut_select(CutPlan pl, u64 j0, u64 s0, u64 start_in, u64 g_lo, u64 g_hi, u64 n,
                                                    long long rle_bias, int emit_tail, const u32 *__restrict__ tab,
                                                    const u16 *__restrict__ comp, const u64 *__restrict__ step_t0,
                                                    BlockDesc *__restrict__ blocks, u32 max_blocks, u64 *__restrict__ res)
{
    __shared__ u32 sg_j[kCutSegCap]; // relative to j0
    __shared__ u32 sg_idx[kCutSegCap];
    __shared__ u8 sg_n[kCutSegCap];
    __shared__ u64 s_stop[3]; // j, idx at the stop; 1 if the tables failed
    __shared__ u32 s_nseg;
    if (threadIdx.x == 0) {
        u64 j = j0, idx = s0 - j0 * pl.L;
        u32 ns = 0;
        u64 bad = (s0 < j0 * pl.L || idx > 4u * j0) ? 1u : 0u;
        while (!bad) {
            const u64 target = (j + 1u) * pl.L + idx;
            if (target > pl.own_hi) break; // the target lies behind this rank's image
            if (ns >= kCutSegCap) {
                bad = 1;
                break;
            }
            // A target among the last four bytes of this rank's image may belong to a chunk that ends BEHIND the image (a
            // run across the edge of the slab: its count byte is the next slab's): such a chunk is evaluated by the next
            // rank (w_min), the block it closes is that rank's first one, and the chain of this rank stops here.
            if (j > pl.j_hi || j < pl.j_lo) {
                if (target + 4u <= pl.own_hi) bad = 1;
                break;
            }
            if ((j & 15u) == 0 && g_hi >= g_lo && (j >> 4) >= g_lo && (j >> 4) <= g_hi) {
                const u16 c = comp[cut_cbase(g_lo, j >> 4) + idx];
                if (c != 0xFFFFu) {
                    sg_j[ns] = (u32)(j - j0);
                    sg_idx[ns] = (u32)idx;
                    sg_n[ns] = 16;
                    ++ns;
                    idx += c;
                    j += 16u;
                    continue;
                }
            }
            const u32 en = tab[cut_tbase(pl.j_lo, j) + idx];
            if (en == 0xFFFFFFFFu) {
                if (target + 4u <= pl.own_hi) bad = 1; // an owned target without an answer (else: the next rank's, see above)
                break;
            }
            sg_j[ns] = (u32)(j - j0);
            sg_idx[ns] = (u32)idx;
            sg_n[ns] = 1;
            ++ns;
            idx += en & 7u;
            j += 1u;
        }
        s_stop[0] = j;
        s_stop[1] = idx;
        s_stop[2] = bad;
        s_nseg = ns;
    }
    __syncthreads();
    const u64 j_stop = s_stop[0], s_end = j_stop * pl.L + s_stop[1];
    const u32 ns = s_nseg;
    const u64 nb = j_stop - j0;
    const bool tail = emit_tail && !s_stop[2] && pl.own_hi > s_end;
    if (s_stop[2]) {
        if (threadIdx.x == 0) {
            res[0] = ~0ull;
            res[1] = start_in;
            res[2] = 0;
            res[3] = s0;
            res[4] = j0;
        }
        return;
    }
    for (u32 q = threadIdx.x; q < ns; q += 256u) {
        u64 j = j0 + sg_j[q], idx = sg_idx[q];
        for (u32 w = 0; w < sg_n[q]; ++w, ++j) {
            const u32 en = tab[cut_tbase(pl.j_lo, j) + idx];
            const u64 cut_in = step_t0[j - pl.j_lo] * (u64)kRleTile + (u64)(en >> 3);
            const u64 k = j - j0;
            if (k < max_blocks) {
                blocks[k].rle_off = (u64)((long long)(j * pl.L + idx) + rle_bias);
                blocks[k].in_end = cut_in;
                blocks[k].n = (u32)(pl.L + (en & 7u));
                blocks[k].pad = 0;
                if (k == 0) blocks[k].in_off = start_in;
            }
            if (k + 1u < max_blocks) blocks[k + 1u].in_off = cut_in;
            if (j + 1u == j_stop) res[1] = tail ? n : cut_in;
            idx += en & 7u;
        }
    }
    if (threadIdx.x == 0) {
        if (nb == 0) res[1] = tail ? n : start_in;
        if (tail && nb < max_blocks) {
            blocks[nb].rle_off = (u64)((long long)s_end + rle_bias);
            if (nb == 0) blocks[nb].in_off = start_in;
            blocks[nb].in_end = n;
            blocks[nb].n = (u32)(pl.own_hi - s_end);
            blocks[nb].pad = 0;
        }
        res[0] = nb + (tail ? 1u : 0u);
        res[2] = tail ? 1u : 0u;
        res[3] = s_end;
        res[4] = j_stop;
    }
}

// ---- kernel G: block CRC from tile CRCs -------------------------------------------
__global__ __launch_bounds__(RT) void k_block_crc(const u8 *__restrict__ in,
                                                   const BlockDesc *__restrict__ blocks,
                                                   const u32 *__restrict__ crc_tab,
                                                   const u32 *__restrict__ xp2,
                                                   const u32 *__restrict__ tile_crc,
                                                   u32 *__restrict__ out_crc)
{
    __shared__ u32 s_tab[256];
    __shared__ u32 s_x[RT / 64];
    s_tab[threadIdx.x] = crc_tab[threadIdx.x];
    __syncthreads();
    const BlockDesc d = blocks[blockIdx.x];
    const u64 a = d.in_off, b = d.in_end;
    const u64 ft = (a + kRleTile - 1) / kRleTile; // first full tile
    const u64 lt = b / kRleTile;                  // one past the last full tile
    u32 acc = 0;
    if (ft >= lt) {
        // short range: 16-byte pieces straight from the bytes
        for (u64 p = a + (u64)threadIdx.x * 16u; p < b; p += (u64)RT * 16u) {
            const u64 len = (b - p) < 16u ? (b - p) : 16u;
            const u32 c = crc_bytes_raw(in + p, len, s_tab);
            acc ^= gf_mulmod(c, gf_xpow_bytes(b - (p + len), xp2));
        }
    } else {
        const u64 head_end = ft * kRleTile, tail_beg = lt * kRleTile;
        // head and tail pieces
        for (u64 p = a + (u64)threadIdx.x * 16u; p < head_end; p += (u64)RT * 16u) {
            const u64 len = (head_end - p) < 16u ? (head_end - p) : 16u;
            const u32 c = crc_bytes_raw(in + p, len, s_tab);
            acc ^= gf_mulmod(c, gf_xpow_bytes(b - (p + len), xp2));
        }
        for (u64 p = tail_beg + (u64)threadIdx.x * 16u; p < b; p += (u64)RT * 16u) {
            const u64 len = (b - p) < 16u ? (b - p) : 16u;
            const u32 c = crc_bytes_raw(in + p, len, s_tab);
            acc ^= gf_mulmod(c, gf_xpow_bytes(b - (p + len), xp2));
        }
        // full tiles: each thread folds a contiguous run of tiles
        const u64 nt = lt - ft;
        const u64 per = (nt + RT - 1) / RT;
        const u64 t0 = ft + (u64)threadIdx.x * per;
        const u64 t1 = (t0 + per < lt) ? t0 + per : lt;
        if (t0 < t1) {
            const u32 xt = xp2[12]; // x^(8*4096)
            u32 c = 0;
            for (u64 t = t0; t < t1; ++t) c = gf_mulmod(c, xt) ^ tile_crc[t];
            acc ^= gf_mulmod(c, gf_xpow_bytes(b - t1 * kRleTile, xp2));
        }
    }
    const u32 wx = wave_xor(acc);
    if (lane_id() == 0) s_x[threadIdx.x >> 6] = wx;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 x = 0;
        for (u32 k = 0; k < RT / 64; ++k) x ^= s_x[k];
        // init 0xFFFFFFFF rides through the whole message; final NOT (crc32.rs:128-131)
        x ^= gf_mulmod(0xFFFFFFFFu, gf_xpow_bytes(b - a, xp2));
        out_crc[blockIdx.x] = ~x;
    }
}

// ---- host launchers -----------------------------------------------------------------
// last run start inside tiles [t0,t1) (-1: none)
__global__ __launch_bounds__(1024) void k_slab_last(const i64 *__restrict__ tile_last, u64 t0, u64 t1,
                                                     i64 *__restrict__ out)
{
    __shared__ i64 s_m[1024];
    i64 m = -1;
    for (u64 t = t0 + threadIdx.x; t < t1; t += 1024) m = tile_last[t] > m ? tile_last[t] : m;
    s_m[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (u32 k = 1; k < 1024; ++k) m = s_m[k] > m ? s_m[k] : m;
        *out = m;
    }
}

void launch_slab_last(hipStream_t st, const RleBuffers &rb, u64 t0, u64 t1, i64 *d_out)
{
    hipLaunchKernelGGL(k_slab_last, dim3(1), dim3(1024), 0, st, rb.tile_last, t0, t1, d_out);
}

// The split is driven in three steps so that a multi-GPU job can shard it by slabs of tiles:
//   scan   : tiles [t0,t1): run starts + tile CRCs; *out_last = last run start of the range
//   count  : tiles [t0,t1): carries from `init_carry`, RLE1 byte counts, sub-tile index
//   finish : prefix over [tb,t1), the cut chain from byte in_begin, the RLE1 image
void launch_rle_scan(hipStream_t st, const u8 *d_in, u64 n, u64 t0, u64 t1, u64 in_begin, const u32 *crc_tab,
                     const u32 *xp16, const RleBuffers &rb)
{
    if (t1 <= t0) return;
    hipLaunchKernelGGL(k_rle_tile_scan, dim3((u32)(t1 - t0)), dim3(RT), 0, st, d_in, n, t0, in_begin, crc_tab, xp16,
                       rb.tile_last, rb.tile_crc);
}

void launch_rle_count(hipStream_t st, const u8 *d_in, u64 n, u64 t0, u64 t1, u64 in_begin, i64 init_carry,
                      const RleBuffers &rb, i64 *d_out_last)
{
    if (t1 <= t0) return;
    // carry_in[t] = last run start before tile t (from init_carry); *d_out_last = the range's last run start
    tiles_scan<true>(st, rb.tile_last, rb.carry_in, t0, t1, reinterpret_cast<i64 *>(rb.scan_part), init_carry, d_out_last,
                     nullptr);
    hipLaunchKernelGGL(k_rle_count, dim3((u32)(t1 - t0)), dim3(RT), 0, st, d_in, n, t0, in_begin, rb.carry_in,
                       rb.tile_count, rb.sub_off, rb.sub_rs);
}

void launch_rle_prefix(hipStream_t st, u64 tb, u64 t1, const RleBuffers &rb, bool write_end)
{
    // tile_off[tb] = 0 ... tile_off[t1] = total (offsets are relative to the range); write_end = false leaves
    // tile_off[t1] alone (it is the first offset of a range that has its own)
    if (t1 <= tb) {
        (void)hipMemsetAsync(rb.total, 0, 8, st);
        if (write_end) (void)hipMemsetAsync(rb.tile_off + tb, 0, 8, st);
        return;
    }
    tiles_scan<false>(st, rb.tile_count, rb.tile_off, tb, t1, rb.scan_part, 0ull, rb.total, write_end ? rb.tile_off + t1 : nullptr);
}

// The chain of cuts is one workgroup that reads the input and the tile tables, not the image: the two are
// separate launches (one GPU: side by side on two streams; a rank of a sharded job: the cuts first, the cut is
// handed to the next rank, then the image).
void launch_rle_cuts(hipStream_t st, const u8 *d_in, u64 n, u64 tb, u64 t1, u64 in_begin, const RleBuffers &rb,
                     u32 block_max_len, int emit_tail, BlockDesc *d_blocks, u32 max_blocks)
{
    if (t1 <= tb) {
        (void)hipMemsetAsync(rb.cut_result, 0, 3 * sizeof(u64), st);
        return;
    }
    hipLaunchKernelGGL(k_rle_cuts, dim3(1), dim3(CT), 0, st, d_in, n, tb, in_begin, rb.sub_off, rb.sub_rs,
                       rb.tile_off, t1, block_max_len, emit_tail, d_blocks, max_blocks, rb.cut_result);
}
void launch_rle_image(hipStream_t st, const u8 *d_in, u64 n, u64 tb, u64 t1, u64 in_begin, const RleBuffers &rb, u8 *d_rle)
{
    if (t1 <= tb) return;
    hipLaunchKernelGGL(k_rle_scatter, dim3((u32)(t1 - tb)), dim3(RT), 0, st, d_in, n, tb, in_begin, rb.carry_in,
                       rb.tile_off, d_rle);
}

// The cuts from tables, in three steps (see "kernels H"): the steps' tiles (the caller reads step_w0[nsteps], the number
// of workgroups of the table kernel), the tables, the chain.
u64 cut_table_entries(const CutPlan &pl) { return pl.j_hi >= pl.j_lo ? cut_tbase(pl.j_lo, pl.j_hi + 1u) : 0u; }
void cut_groups(const CutPlan &pl, u64 *g_lo, u64 *g_hi, u64 *entries)
{
    // groups of sixteen steps that lie inside [j_lo, j_hi]; *g_hi < *g_lo: none
    *g_lo = (pl.j_lo + 15u) / 16u;
    *g_hi = *g_lo;
    *entries = 0;
    if (pl.j_hi < pl.j_lo || pl.j_hi + 1u < (*g_lo + 1u) * 16u) {
        *g_lo = 1;
        *g_hi = 0;
        return;
    }
    *g_hi = (pl.j_hi + 1u) / 16u - 1u;
    *entries = cut_cbase(*g_lo, *g_hi + 1u);
}
u32 cut_seg_cap() { return kCutSegCap; }
void launch_cut_steps(hipStream_t st, const CutPlan &pl, const RleBuffers &rb, const CutBuffers &cb)
{
    const u64 nsteps = pl.j_hi - pl.j_lo + 1u;
    if (nsteps == 0) return; // (cut_tables_prepare does not get here without steps)
    hipLaunchKernelGGL(k_cut_steps, dim3((u32)((nsteps + 255u) / 256u)), dim3(256), 0, st, pl, rb.tile_off, cb.step_t0, cb.step_nt);
    // (the exclusive sum of the steps' tile counts, by the three-kernel scan of the tile offsets instead of one workgroup
    // that walks the steps 1024 at a time (k_cut_steps_scan, rounds 4-5).  For the 1 189 blocks of a GiB there are about
    // 1 200 steps and either form is microseconds of work; the kernel trace shows it at 1.5 ms only because it runs beside
    // k_rle_scatter and waits for a CU -- nothing the step time sees.  The workgroup totals live behind step_w0's
    // nsteps + 1 entries.)
    tiles_scan<false>(st, cb.step_nt, cb.step_w0, 0, nsteps, cb.step_w0 + nsteps + 2, 0ull, cb.step_w0 + nsteps, nullptr);
}
void launch_cut_tables(hipStream_t st, const CutPlan &pl, const u8 *d_in, u64 n_lim, const RleBuffers &rb, const CutBuffers &cb,
                       u64 work_total)
{
    const u64 nsteps = pl.j_hi - pl.j_lo + 1u;
    (void)hipMemsetAsync(cb.tab, 0xFF, cut_table_entries(pl) * 4u, st);
    if (work_total)
        hipLaunchKernelGGL(k_cut_table, dim3((u32)work_total), dim3(RT), 0, st, pl, d_in, n_lim, rb.carry_in, rb.tile_off, cb.step_t0,
                           cb.step_w0, nsteps, cb.tab);
    u64 g_lo, g_hi, ce;
    cut_groups(pl, &g_lo, &g_hi, &ce);
    if (g_hi >= g_lo)
        hipLaunchKernelGGL(k_cut_compose, dim3((u32)((4u * 16u * g_hi + 256u) / 256u), (u32)(g_hi - g_lo + 1u)), dim3(256), 0, st, pl,
                           g_lo, cb.tab, cb.comp);
}
void launch_cut_select(hipStream_t st, const CutPlan &pl, u64 j0, u64 s0, u64 start_in, u64 n, long long rle_bias, int emit_tail,
                       const RleBuffers &rb, const CutBuffers &cb, BlockDesc *d_blocks, u32 max_blocks)
{
    u64 g_lo, g_hi, ce;
    cut_groups(pl, &g_lo, &g_hi, &ce);
    hipLaunchKernelGGL(k_cut_select, dim3(1), dim3(256), 0, st, pl, j0, s0, start_in, g_lo, g_hi, n, rle_bias, emit_tail, cb.tab,
                       cb.comp, cb.step_t0, d_blocks, max_blocks, rb.cut_result);
}

void launch_block_crc(hipStream_t st, const u8 *d_in, const BlockDesc *d_blocks, u32 nblocks,
                      const u32 *crc_tab, const u32 *xp2, const u32 *tile_crc, u32 *d_crc)
{
    if (nblocks == 0) return;
    hipLaunchKernelGGL(k_block_crc, dim3(nblocks), dim3(RT), 0, st, d_in, d_blocks, crc_tab, xp2, tile_crc,
                       d_crc);
}

} // namespace bzgpu
