// bzgpu.h -- internal declarations shared by the HIP translation units.
// gfx950 only: wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace bzgpu {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

// ---- sizes (level 9 worst case; lower levels just use less of each slot) ----
constexpr u32 kWave = 64;
constexpr u32 kMaxBlockLen = 900000;        // 100000*level, >= n (n <= 100000*level - 19 + 4)
constexpr u32 kSortTile = 8192;             // elements per radix tile: 8 waves x 16 rows x 64 lanes
constexpr u32 kSortThreads = 512;
constexpr u32 kTilesPerBlock = 110;         // ceil(900000 / 8192)
constexpr u32 kSlot = kTilesPerBlock * kSortTile; // 901120: per-block stride of the u32 work arrays
constexpr u32 kMaxBins = 2048;              // 11-bit digits for the initial 32-bit key sort
constexpr u32 kPerK = 8;                    // distances per block the period round orders copies by (round 5: one before)
constexpr u32 kGSize = 50;                  // BZ_G_SIZE, src/bzip2/mod.rs:20
constexpr u32 kMaxSelectors = 18002;        // BZ_MAX_SELECTORS, src/bzip2/encoder.rs:295
constexpr u32 kMaxAlpha = 258;
constexpr u32 kRleTile = 4096;              // input bytes per RLE1/CRC tile (256 threads x 16 B)
constexpr u32 kMtfChunk = 512;     // symbols per serial MTF chunk (one lane each)
constexpr u32 kMaxMtfChunks = (kMaxBlockLen + kMtfChunk - 1) / kMtfChunk; // 3516
// words reserved per block bit string: header (<= ~27k bits) + 900001 symbols x 17 bits (+ slack)
constexpr u32 kStreamWords = 480000;        // 1.92 MB

constexpr u32 kFinalBit = 0x80000000u;      // rank word: rotation is alone in its group

// Lanes of the wave that hold the same `bits`-bit digit as this lane (and take part): the "match any" of a
// radix pass.  Per bit: the lanes that DIFFER from this one are ballot ^ (all ones if my bit is set), OR-ed
// up in one three-input boolean instruction per half (v_bitop3_b32 on gfx950): 4 vector instructions per
// bit instead of the 9 that the select form `peers &= bit ? m : ~m` compiles to.
template <int BITS> __device__ __forceinline__ u64 wave_match_digit(u32 dg, bool ok)
{
    u32 diff_lo = 0, diff_hi = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const u32 mine = (u32)((int)(dg << (31 - b)) >> 31); // all ones if bit b of dg is set
        const u64 m = __ballot((int)mine < 0);
        // diff | (ballot ^ mine) in one v_bitop3_b32 (truth table 0xF6: source 0 is the most significant index bit)
        diff_lo = __builtin_amdgcn_bitop3_b32(diff_lo, (u32)m, mine, 0xF6);
        diff_hi = __builtin_amdgcn_bitop3_b32(diff_hi, (u32)(m >> 32), mine, 0xF6);
    }
    const u64 diff = ((u64)diff_hi << 32) | diff_lo;
    return __ballot(ok) & ~diff;
}

// per-block record produced by the partition step (device + host mirror)
struct BlockDesc {
    u64 rle_off;   // offset of the block's bytes in the RLE1 image
    u64 in_off;    // offset of the first input byte the block covers
    u64 in_end;    // one past the last input byte it covers
    u32 n;         // post-RLE1 length
    u32 pad;
};

// per-block results of the encode step
struct BlockOut {
    u32 crc;
    u32 orig_ptr;
    u32 in_use_count;
    u32 mtf_count;
    u32 group_num;
    u32 n_selectors;
    u32 max_len;
    u32 lm_tables;
    u32 header_bits;
    u32 pad;       // (k_huff_header) 16-byte ranges of the byte map in use | bits of the selectors' unary codes << 5
    u64 total_bits;
};

// ---- streaming access hints -----------------------------------------------------
// Arrays that are swept once per kernel (sorted pair lists, SA as the walk source) are loaded
// non-temporally so they do not push the block's rank array (3.6 MB, gathered at random) out of the
// XCD's 4 MiB L2 (+2.4 % of the step, measured in round 2).
template <class T> __device__ __forceinline__ T ld_stream(const T *p)
{
    return __builtin_nontemporal_load(p);
}
template <class T> __device__ __forceinline__ void st_stream(T *p, T v)
{
    __builtin_nontemporal_store(v, p);
}

// The same for element `idx` of an array whose start is wave-uniform, with the byte offset kept in 32 bits (idx < 2^30):
// the access is then `global_load_dword v, v_off, s[base:base+1]` -- one shift for the address instead of a 64-bit shift
// and a 64-bit add, one address register instead of two, and rows a constant distance apart share the register (the
// distance goes into the instruction's immediate offset).
__device__ __forceinline__ u32 ld_stream_at(const u32 *base_uniform, u32 idx)
{
    return ld_stream(reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(base_uniform) + (size_t)(u32)(idx << 2)));
}
__device__ __forceinline__ void st_stream_at(u32 *base_uniform, u32 idx, u32 v)
{
    st_stream(reinterpret_cast<u32 *>(reinterpret_cast<char *>(base_uniform) + (size_t)(u32)(idx << 2)), v);
}

__device__ __forceinline__ void st_plain_at(u32 *base_uniform, u32 idx, u32 v)
{
    *reinterpret_cast<u32 *>(reinterpret_cast<char *>(base_uniform) + (size_t)(u32)(idx << 2)) = v;
}

// ---- wave64 primitives -------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }

// The value of the lane below (lane 0: 0) and of a lane known when the kernel is compiled, without the LDS crossbar:
// __shfl_up(v, 1) / __shfl(v, k) compile to ds_bpermute_b32 plus the arithmetic of its byte address -- an LDS
// instruction and four or five vector ones --, a DPP move shifted by one lane across the whole wave (wave_shr:1) and a
// v_readlane_b32 are one instruction each.
__device__ __forceinline__ u32 wave_shr1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ u32 wave_shl1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); } // the lane ABOVE's value (lane 63: 0)
__device__ __forceinline__ u32 wave_lane(u32 v, u32 lane_uniform) { return (u32)__builtin_amdgcn_readlane((int)v, (int)lane_uniform); }
__device__ __forceinline__ u64 wave_lane64(u64 v, u32 lane_uniform)
{
    return ((u64)wave_lane((u32)(v >> 32), lane_uniform) << 32) | wave_lane((u32)v, lane_uniform);
}

__device__ __forceinline__ u32 wave_incl_sum(u32 v)
{
    const u32 l = lane_id();
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ i64 wave_incl_max64(i64 v)
{
    const u32 l = lane_id();
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        i64 t = __shfl_up(v, d, 64);
        if (l >= d && t > v) v = t;
    }
    return v;
}

__device__ __forceinline__ int wave_incl_max32(int v)
{
    const u32 l = lane_id();
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (l >= d && t > v) v = t;
    }
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v)
{
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ u32 wave_xor(u32 v)
{
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}

// A small array that lives in the registers of a wave: element e in lane e & 63 of register e >> 6, read with a
// v_readlane at a wave-uniform index (the compiler addresses the register through M0), written with a compare and a
// select.  For procedures that are one chain of dependent accesses by construction (the reference's heap Huffman):
// ~30 cycles per access instead of an LDS round trip.  All lanes of the wave make the same calls.
template <u32 NR> struct WaveArr {
    typedef u32 vec_t __attribute__((ext_vector_type(NR <= 1 ? 1 : (NR <= 4 ? 4 : 16))));
    vec_t r;
    __device__ __forceinline__ u32 get(u32 x) const
    {
        const u32 word = NR <= 1 ? r[0] : r[x >> 6];
        return (u32)__builtin_amdgcn_readlane((int)word, (int)(x & 63u));
    }
    __device__ __forceinline__ void set(u32 x, u32 val)
    {
        const u32 lane = (u32)__builtin_amdgcn_mbcnt_hi(~0u, (u32)__builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (NR <= 1) r[0] = lane == (x & 63u) ? val : r[0];
        else {
            const u32 word = r[x >> 6];
            r[x >> 6] = lane == (x & 63u) ? val : word;
        }
    }
};

// XCD-aware remap of a 2-D (tile, block) launch.  Workgroups are dealt
// round-robin over the 8 XCDs by linear id, so lid%8 names the XCD group.  We
// hand each XCD group whole blocks (all tiles of a block land on one XCD and
// its 4 MiB L2 keeps that block's rank array hot for the gathers).
// Speed only: any placement gives the same result.
__device__ __forceinline__ void xcd_remap(u32 tiles, u32 nblocks, u32 &tile, u32 &blk)
{
    const u32 lid = blockIdx.x + gridDim.x * blockIdx.y;
    const u32 xcd = lid & 7u;
    const u32 slot = lid >> 3;
    const u32 b8 = slot / tiles;         // which group of 8 blocks
    tile = slot - b8 * tiles;
    blk = b8 * 8u + xcd;
    if (blk >= nblocks) blk = 0xFFFFFFFFu;
}
// grid.y for such a launch: blocks rounded up to a multiple of 8
inline u32 xcd_grid_y(u32 nblocks) { return (nblocks + 7u) & ~7u; }

// ---- decoupled look-back between the tiles of a block (fused radix passes, k_group_refine, k_zle_fused) ----
constexpr u32 kLbValMask = 0xFFFFFu, kLbAgg = 1u << 20, kLbIncl = 2u << 20, kLbFlagMask = 3u << 20;
// A spin gives up after about a second -- or as soon as another tile of the launch has given up (sort_err): one
// look-back that cannot complete leaves every tile behind it waiting, and each of them for the full bound otherwise.
constexpr u32 kLbSpinMax = 1u << 20;
__device__ __forceinline__ bool lb_give_up(u32 &spins, const u32 *sort_err, u32 limit)
{
    ++spins;
    if (spins > limit) return true;
    return (spins & 255u) == 0u && __hip_atomic_load(sort_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// 16-byte accesses that the other compute units OF THE SAME XCD observe: plain stores are written
// through to the XCD's L2 and stay there, `nt` loads bypass the L1 and are served by that L2
// (MI355X_MICROARCH.md).  Producer and consumer of a look-back word always share an XCD (tickets are
// taken from the counter of the XCD a workgroup runs on).  (Accesses that are coherent across XCDs -- sc1 -- have 2-3x
// the latency, and such stores drop the L2 line: the first build used them and won 2.5 % where this one wins 7.)  A 16-byte
// granule written by one store is seen whole.
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1_x4(u32 *p, uint4 v)
{
    const u32x4_t r = {v.x, v.y, v.z, v.w};
    // (s_nop: a store of more than 8 bytes reads its data registers a cycle late; the compiler's hazard
    // recogniser does not look inside inline assembly)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ uint4 ld_sc1_x4(const u32 *p)
{
    u32x4_t r;
    asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return make_uint4(r.x, r.y, r.z, r.w);
}

// the same load WITHOUT the wait, for several words asked for together (ld_x4_wait_all before the first use of any of
// them: the compiler does not know that the registers are filled late)
__device__ __forceinline__ void ld_sc1_x4_issue(const u32 *p, u32x4_t &r)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld_x4_wait_all() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }

// ---- CRC-32/BZIP2 arithmetic in GF(2)[x] / 0x104C11DB7 ------------------------
constexpr u32 kCrcPoly = 0x04C11DB7u;

__host__ __device__ inline u32 gf_mulmod(u32 a, u32 b)
{
    u32 r = 0;
#pragma unroll 4
    for (int i = 31; i >= 0; --i) {
        r = (r << 1) ^ ((r & 0x80000000u) ? kCrcPoly : 0u);
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}

// xp2[k] = x^(8 * 2^k) mod P, k = 0..47
__device__ __forceinline__ u32 gf_xpow_bytes(u64 nbytes, const u32 *__restrict__ xp2)
{
    u32 r = 1u;
    for (u32 k = 0; nbytes; ++k, nbytes >>= 1)
        if (nbytes & 1u) r = gf_mulmod(r, xp2[k]);
    return r;
}

__device__ __forceinline__ u32 crc_bytes_raw(const u8 *__restrict__ p, u64 len, const u32 *s_tab)
{
    u32 crc = 0;
    for (u64 i = 0; i < len; ++i) crc = s_tab[(crc >> 24) ^ p[i]] ^ (crc << 8);
    return crc;
}


// ---- argument blocks shared by the kernels (k_*.hip) and the host code (engine.hip) ----------
struct RleBuffers {
    i64 *tile_last;   // [ntiles] last run start inside the tile, -1 if none
    i64 *carry_in;    // [ntiles] last run start before the tile
    u32 *tile_crc;    // [ntiles] raw CRC of the tile's bytes
    u32 *tile_count;  // [ntiles] RLE1 bytes the tile emits
    u16 *sub_off;     // [ntiles][16] bytes emitted before each 256-byte sub-tile, inside its tile
    i64 *sub_rs;      // [ntiles][16] run start live at each sub-tile's first byte
    u64 *tile_off;    // [ntiles+1] exclusive sum of tile_count
    u64 *total;       // [1]
    u64 *cut_result;  // [3] number of blocks, input bytes consumed, tail-block flag
    u64 *scan_part;   // [ntiles / 1024 + 2] workgroup totals of the tile scans
};

// the cuts from tables (k_rle1.hip, "kernels H"); offsets are those of the image of the WHOLE input
struct CutPlan {
    u64 L;          // bytes a block holds before the chunk that closes it (100000 level - 19)
    u64 g_base;     // image offset of the first byte of tile tb (tile_off is relative to it)
    u64 own_hi;     // g_base + tile_off[t1]: a target is answered here if the chunk that closes its block ENDS in (g_base, own_hi]
    u64 w_min;      // smallest such target: g_base - 3 (a chunk that ends at g_base + 1 holds up to five bytes), 1 for g_base < 4
    u64 j_lo, j_hi; // the steps (= blocks of the whole input) with such targets
    u64 tb, t1;     // the tiles of the range
    u64 t_last;     // last tile whose bytes may be read (t1: there is a tile behind the range; else t1 - 1)
    u64 t_eval;     // last tile whose chunk ends answer targets (t1 - 1: a chunk that ends behind the range is the next rank's)
};
struct CutBuffers {
    u64 *step_t0; // [steps] first tile of a step's targets
    u32 *step_nt; // [steps] tiles of the step
    u64 *step_w0; // [steps + 1] exclusive sum of step_nt
    u32 *tab;     // the steps' tables
    u16 *comp;    // sixteen steps composed
};

struct BwtArgs {
    const u8 *rle;
    const BlockDesc *blocks; // descriptors of the batch's blocks
    u32 nb;
    u32 tiles;               // tiles per block a launch of the sort covers (run_bwt: what the largest block needs)
    u32 *SA, *R, *KA, *VA, *KB, *VB; // [nb * kSlot]
    u32 *tile_hist;                  // [nb][kTilesPerBlock][kMaxBins]
    u32 *bin_base;                   // [nb][kMaxBins]
    u32 *count;                      // [nb] length of the compacted pair list
    u32 *count2;                     // [nb] scratch: survivors of the last round
    u32 *tile_nf;                    // [nb][kTilesPerBlock] survivors per tile of the last refinement
    u8 *flags;                       // [nb * kSlot]
    u64 *newbits;                    // [nb * kSlot / 64] new-group starts of the last refined list, one bit per element
    int *tile_last_old;              // [nb][kTilesPerBlock]
    int *tile_last_new;              // [nb][kTilesPerBlock]
    u32 *nonfinal;                   // [nb]
    unsigned long long *active;      // [64] per-round count of non-final rotations in unfinished blocks
    u32 *maxnf;                      // [64] per round: the largest number of non-final rotations any block is left with
    u32 *per_k, *per_shift;          // [nb] periodic blocks: repetition count, least-rotation start
    u32 *lin_p, *lin_sig;            // [nb][kPerK] blocks deep in repeats (k_period_find): the distances at which they agree
                                     //   with themselves, widest agreement first (0: none / no more), and that agreement
                                     //   in permille of the block
    u32 *bin_cursor;                 // [nb][1024] rank words binned so far (k_group_apply -> k_rank_place)
    u32 *pb_gate;                    // [nb] != 0: phase B of the init by the global passes (always, since round 6: the LDS form is gone)
    u32 *loc_stats;                  // [64] k_surv_local's counters (LOC_STAT_*), the phase timers of the instrumented builds
    u8 *L;                           // [nb * kSlot] last column, written as rotations become final
    u32 *orig_ptr;                   // [nb] position of rotation 0 in the sorted order
    u8 *ptext;                       // [nb * kSlot] the blocks' symbols packed for pkey() (k_pack_text); borrowed: the
                                     //   MTF stage's rank bytes live here once the sort is done
    const u8 *sym_code;              // [nb][256] byte -> code (rank among the bytes in use)
    const u8 *keyinfo;               // [nb] KeyInfo {bits per symbol, symbols per key}
    // fused radix passes (no per-pass histogram kernel; tile offsets by decoupled look-back)
    u32 *gh_tiles;                   // [nb][kTilesPerBlock][3][kMaxBins] per-tile digit counts of a whole phase
    u32 *gbase;                      // [nb][3][kMaxBins] digits smaller, per digit position
    u32 *tile_state;                 // [nb][kTilesPerBlock][kMaxBins] look-back words: epoch | flag | value
    u32 *tickets;                    // [kSortEpochs][8] tile tickets per pass and XCD
    u32 *sort_err;                   // [1] a look-back that gave up
    u32 *epoch;                      // host: the engine's pass counter (1 .. kSortEpochs-1)
    u8 *per_aux;                     // [nb][2 kMtfStride] bytes (the MTF stage's symbol buffer, free during the sort): a group byte and a
                                     // link byte per position of the survivor list (k_survivor_compact, k_link_scan; k_bwt.hip)
    u32 per_keyshift;                // the period round's start-based keys are shifted down by this (10 when every listed distance of the
                                     // batch is at least 1024: the members of a group then differ above bit 10, ONE pass orders them)
    u32 per_wide;                    // the period round under way sorts thirty-bit keys (per_ekey, k_bwt.hip)
    u32 per_links;                   // != 0: small groups of survivors are ranked by direct comparison in the period round and in link
                                     //   rounds behind it (BZ_LINK_ROUND=0: off)
    u32 *fused_state;                // host, per engine: [0] != 0: the fused passes misbehaved once and stay off for
                                     //   this engine, [1] sorts that fell back to the three-kernel passes
    u32 *tile_state_all;             // host: whole look-back buffer (cleared when the counter wraps)
    u32 fused;                       // 1: use the fused passes (BZ_ONESWEEP=1); 0: histogram + scan + scatter
    size_t tile_state_bytes;
};
constexpr u32 kSortEpochs = 1024;

constexpr u32 kMtfStride = kSlot + 64;
struct MtfArgs {
    const BlockDesc *blocks;
    u32 nb;
    u32 tiles;               // tiles of kSortTile bytes per block the ZLE launches cover (what the largest block needs)
    const u8 *L;             // [nb * kSlot] last column (raw bytes)
    const u32 *inuse_bits;   // [nb][8]
    u8 *summ;                // [nb][kMaxMtfChunks][256] recency lists
    u16 *summ_len;           // [nb][kMaxMtfChunks]
    u8 *init_state;          // [nb][kMaxMtfChunks][256]
    u8 *rank8;               // [nb * kSlot]
    int *ztile_last;         // [nb][kTilesPerBlock] last position with a non-zero rank
    u32 *ztile_cnt;          // [nb][kTilesPerBlock]
    u32 *zstate;             // [nb][kTilesPerBlock][4] look-back words of k_zle_fused
    u32 *ztick;              // [16] its tile tickets per XCD, [8]: a look-back gave up
    u32 fused_zle;           // 1: k_zle_fused; 0: k_zle_last + k_zle_emit<false> + k_zle_emit<true>
    u16 *mtf;                // [nb][kMtfStride] output symbols
    u32 *mtf_freq;           // [nb][kMaxAlpha]
    BlockOut *out;           // [nb]
};

constexpr u32 kLim = 17;                       // maxLen, src/bzip2/encoder.rs:504-507
constexpr u32 kLmRow = 2 * kMaxAlpha + 4;      // >= max_elem[j] for every package-merge row
constexpr u32 kLmWords = 2 * kLim * kLmRow + 5 * kMaxAlpha + 64; // last kMaxAlpha words: the weights
constexpr u32 kSelStride = kMaxSelectors + 14;
constexpr u32 kGboStride = kMaxSelectors + 2;
struct HuffArgs {
    const BlockDesc *blocks;
    u32 nb;
    const u16 *mtf;        // [nb][kMtfStride]
    u32 mtf_stride;
    const u32 *mtf_freq;   // [nb][kMaxAlpha]
    const u32 *inuse_bits; // [nb][8]
    const u32 *crc;        // [nb]
    const u32 *orig_ptr;   // [nb]
    u8 *selector;          // [nb][kSelStride]
    u32 *code_len;         // [nb][6][kMaxAlpha]   code | len << 24
    u32 *group_bitoff;     // [nb][kGboStride]  payload bit offset of each 50-symbol group
    u32 *lm_scratch;       // [nb][6][kLmWords]
    // between the kernels of the split stage (k_huff_tables / k_huff_sweep / k_huff_gbits / k_huff_header)
    u8 *glen;              // [nb][6][kMaxAlpha + 6] code lengths, tables in selector order
    unsigned long long *pack; // [nb][kMaxAlpha] the same as 6 x 10 bits per symbol
    u32 *rfreq;            // [nb][6][kMaxAlpha] symbol counts per selected table of the sweep under way
    u32 *hlm;              // [nb] tables that took the length-limited path
    u32 *pass_stats;       // [nb][4][8] per refinement pass: the groups' cost under their chosen tables (totc), groups per table (fave[6])
    u32 *stream;           // [nb][kStreamWords]  block bit string, logical MSB-first words
    BlockOut *out;         // [nb]
    u32 *error_flag;       // [1]
};

struct AsmBlock {
    u64 src_word;   // first word of the block's bit string in `packed`
    u64 bit_len;    // bits
    u64 dst_bit;    // position of its first bit in the output stream
};
struct PackBlock {
    u64 src_word, dst_word, nwords;
};

// ---- decoder records -------------------------------------------------------------------------------
constexpr int BZ_DEC_E_DATA = -1; // BZip2Error::DataError (src/bzip2/error.rs:5-11)
struct DecCand {
    u64 bitpos; // first bit of a 48-bit magic
    u32 type;   // 1 block, 2 end of stream
    u32 pad;
};
struct DecBlockInfo {
    u64 end_bit;    // first bit after the block's data (after the stored CRC for an end-of-stream record)
    u32 status;     // 0 or a BZ_DEC_E_* code (as u32)
    u32 stored_crc; // block CRC / combined CRC
    u32 orig_ptr;
    u32 randomised;
    u32 n_in_use;
    u32 nsym;       // symbols decoded, EOB included
    u32 next_head;  // the 8 bits behind the block (the head byte of the next record), read like BitReader does:
    u32 next_bits;  // ... next_bits (< 8 at the end of the input) bits were there
    u8 seq2unseq[256];
};

constexpr u32 kDecSampleStep = 128;             // every n-th T slot is a sample node
constexpr u32 kDecSamples = kMaxBlockLen / kDecSampleStep + 3; // sample nodes per block (the last one = the start node)
constexpr u32 kSegCap = 4 * kDecSampleStep;                    // scratch bytes per segment (mean length = the step)
constexpr u32 kDecSubs = kSlot / 64;                           // 64-byte RLE1-undo sub-tiles per block
struct KernelProf;
struct DecArgs {
    u32 nb;
    // launch shapes that follow the batch (a level-1 block is a ninth of a level-9 one): 8192-byte tiles of the T-vector
    // sort and workgroups of 256 x 64-byte sub-tiles of the RLE1 undo per block
    u32 tiles, sub_wgs;
    u32 cw;                       // workgroups of 256 MTF chunks per block the chunk kernels are launched with
    const u32 *slot;              // [nb] candidate slot (index into info / sym) of each true block, stream order
    const DecBlockInfo *info;     // [slots]
    const u16 *sym;               // [slots][kMtfStride] Huffman symbols
    const u32 *nblock_max;        // [nb] 100000 * level of the block's stream
    u8 *perm;                     // [nb][kMaxMtfChunks][256] chunk permutations, then start lists
    u32 *chunk_emit;              // [nb][kMaxMtfChunks] bytes a chunk emits, then their exclusive sums
    u32 *tt_len;                  // [nb] length of the BWT column
    u32 *err;                     // [nb] 0 or 1 (DataError found while rebuilding the block)
    u8 *L;                        // [nb * kSlot] BWT column
    u32 *T;                       // [nb * kSlot] T vector
    u8 *X;                        // [nb * kSlot] RLE1 image
    u32 *samp_next, *samp_len, *samp_off; // [nb][kDecSamples]
    u32 *cycle_len;               // [nb]
    uint4 *sub_trans;             // [nb][kDecSubs] RLE1 undo: sub-tile transition tables
    u32 *sub_off;                 // [nb][kDecSubs + 1] output offset of every sub-tile (and the total)
    u8 *sub_state;                // [nb][kDecSubs] entering state of every sub-tile
    u32 *work_ctr;                // [2] work counters of the persistent walkers
    uint4 *walk_meta;             // [nb] {length, start node, first byte} per block
    u8 *seg_buf;                  // [nb][kDecSamples][kSegCap] bytes passed by each segment's walk
    u32 *seg_cont;                // [nb][kDecSamples] node reached after kSegCap steps
    u32 *long_list;               // [nb * kDecSamples] segments longer than kSegCap (count in work_ctr[256])
    u32 *out_len;                 // [nb]
    u32 *thist;                   // [nb][kTilesPerBlock][256] T-vector sort: per-tile byte counts
    u32 *tbase;                   // [nb][256]
    u32 *crc;                     // [nb] CRC of the block's output bytes
};
void launch_dec_scan(hipStream_t st, const u8 *in, u64 nbytes, DecCand *cands, u32 cap, u32 *count);
void launch_dec_blocks(hipStream_t st, const u8 *in, u64 nbytes, const DecCand *cands, u32 ncand, DecBlockInfo *info,
                       u16 *sym, u8 *sel_scratch);
void launch_dec_mtf(hipStream_t st, const DecArgs &a, KernelProf *prof, int *rec);
void launch_dec_gather_windows(hipStream_t st, const u8 *in, u64 nbytes, const u64 *bases, u32 nw, u8 *out);
void launch_dec_walks(hipStream_t st, const DecArgs &a, u32 walk_wgs, hipStream_t st2, hipEvent_t ev_a, hipEvent_t ev_b,
                      KernelProf *prof, int rec[4]);
void launch_dec_expand(hipStream_t st, const DecArgs &a, const u64 *out_base, u8 *out);
void launch_dec_crc(hipStream_t st, const DecArgs &a, const u64 *out_base, const u8 *out, u32 max_out_len,
                    const u32 *crc_tab, const u32 *xp2, const u32 *xp16);

// ---- per-kernel timing (HIP events on the launch stream) -------------------------------------------
// Off by default.  When on, every launch of the listed kernels is bracketed by two events; the
// records are resolved after the stream is synchronised.  `bytes` is the ALGORITHMIC traffic of the
// launch ("Kernels and rooflines").
enum KernelId {
    KID_RADIX_HIST = 0,
    KID_RADIX_SCAN,
    KID_RADIX_SCATTER,
    KID_GROUP_FLAGS,
    KID_GROUP_APPLY,
    KID_LAST_COLUMN,
    KID_RADIX_SCATTER_LB, // fused pass: tile counts + look-back + scatter in one kernel
    KID_GHIST_TEXT,       // the three digit counts of a phase, one read of the text
    KID_GHIST_SCAN,
    KID_RANK_PLACE,       // binned rank words -> the rank array, whole lines
    KID_PHASE_B_LOCAL,    // phase B of the init inside LDS
    KID_GROUP_REFINE,     // flags + apply in one pass (look-back for the two group carries)
    // decode path (k_dec.hip)
    KID_DEC_BLOCK,   // header + Huffman
    KID_DEC_MTF,     // chunk_perm + compose + chunk_emit
    KID_DEC_TSORT,   // thist + tscan + tscatter
    KID_DEC_WALK,    // k_dec_walk_lengths (the inverse BWT's random loads)
    KID_DEC_PLACE,   // rank_samples + seg_copy + walk_write + fixups
    KID_DEC_RLE,     // rle_sub + rle_chain + rle_expand
    KID_DEC_CRC,
    KID_COUNT
};
struct KernelProf {
    bool on = false;
    struct Rec {
        int id;
        u64 bytes;
        hipEvent_t a, b;
    };
    Rec *recs = nullptr;
    u32 nrecs = 0, cap = 0;
    // accumulated results
    u64 launches[KID_COUNT] = {};
    u64 bytes[KID_COUNT] = {};
    double seconds[KID_COUNT] = {};
    int begin(hipStream_t st, int id, u64 nbytes);
    void end(hipStream_t st, int idx);
    void set_bytes(int idx, u64 nbytes) { if (idx >= 0) recs[idx].bytes = nbytes; } // when only known later
    void collect(); // call after the stream has been synchronised
    void reset();
};

// ---- launchers -----------------------------------------------------------------------------------
void launch_slab_last(hipStream_t st, const RleBuffers &rb, u64 t0, u64 t1, i64 *d_out);
void launch_rle_scan(hipStream_t st, const u8 *d_in, u64 n, u64 t0, u64 t1, u64 in_begin, const u32 *crc_tab,
                     const u32 *xp16, const RleBuffers &rb);
void launch_rle_count(hipStream_t st, const u8 *d_in, u64 n, u64 t0, u64 t1, u64 in_begin, i64 init_carry,
                      const RleBuffers &rb, i64 *d_out_last);
void launch_rle_prefix(hipStream_t st, u64 tb, u64 t1, const RleBuffers &rb, bool write_end = true);
void launch_rle_cuts(hipStream_t st, const u8 *d_in, u64 n, u64 tb, u64 t1, u64 in_begin, const RleBuffers &rb,
                     u32 block_max_len, int emit_tail, BlockDesc *d_blocks, u32 max_blocks);
void launch_rle_image(hipStream_t st, const u8 *d_in, u64 n, u64 tb, u64 t1, u64 in_begin, const RleBuffers &rb, u8 *d_rle);
u64 cut_table_entries(const CutPlan &pl);
void cut_groups(const CutPlan &pl, u64 *g_lo, u64 *g_hi, u64 *entries);
u32 cut_seg_cap();
void launch_cut_steps(hipStream_t st, const CutPlan &pl, const RleBuffers &rb, const CutBuffers &cb);
void launch_cut_tables(hipStream_t st, const CutPlan &pl, const u8 *d_in, u64 n_lim, const RleBuffers &rb, const CutBuffers &cb,
                       u64 work_total);
void launch_cut_select(hipStream_t st, const CutPlan &pl, u64 j0, u64 s0, u64 start_in, u64 n, long long rle_bias, int emit_tail,
                       const RleBuffers &rb, const CutBuffers &cb, BlockDesc *d_blocks, u32 max_blocks);
void launch_block_crc(hipStream_t st, const u8 *d_in, const BlockDesc *d_blocks, u32 nblocks,
                      const u32 *crc_tab, const u32 *xp2, const u32 *tile_crc, u32 *d_crc);
void launch_block_symbols(hipStream_t st, const BwtArgs &a, u32 *inuse_bits, u8 *sym_code, u8 *keyinfo);
int run_bwt(hipStream_t st, const BwtArgs &a, u32 max_n, u64 total_n, unsigned long long *h_active,
            u64 *sorted_elems, KernelProf *prof, u64 *round_active /*[64] or null*/, bool wide_keys,
            u32 min_chars);
void launch_mtf(hipStream_t st, const MtfArgs &a);
inline bool env_verify() { return getenv("BZ_VERIFY") && atoi(getenv("BZ_VERIFY")) != 0; } // the self-check for every context and engine of the process
// a few bytes for the host between launches (k_emit.hip): up to four device ranges, 0 = they are there and the stream's
// earlier work is done
struct MailSeg {
    void *h;
    const void *d;
    size_t n;
};
int mail_fetch(hipStream_t st, const MailSeg *segs, int nseg);
// a few bytes from the host in front of the next launch (up to 1 KB as a kernel argument, more by hipMemcpyAsync); 0 = queued
int mail_poke(hipStream_t st, void *d_dst, const void *h_src, size_t n);
void launch_huffman(hipStream_t st, const HuffArgs &a);
void launch_probe_code_lengths(hipStream_t st, const u32 *d_freq, u32 alpha, u8 *d_out, u32 *lm_scr,
                               int *d_flag);
void launch_assemble(hipStream_t st, const u32 *packed, const AsmBlock *d_blocks, u32 n_blocks, u64 max_words,
                     u32 *out_words);
void launch_frame(hipStream_t st, u32 *out_words, int write_header, u32 level, u32 carry_bits, u32 carry_byte,
                  int write_trailer, u64 trailer_bit, u32 combined_crc);
void launch_pack(hipStream_t st, const u32 *src, const PackBlock *d_pb, u32 n_blocks, u32 *dst);

} // namespace bzgpu
