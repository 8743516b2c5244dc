// k_deflate.h -- declarations shared by the Deflate kernels (k_deflate.hip) and their orchestration
// (deflate_engine.hip).  SURVEY.md rows f-2 (Inflater) and f-3 (zlib / gzip containers).
#pragma once
#include "bzgpu.h"

namespace dfgpu {
using bzgpu::u8;
using bzgpu::u16;
using bzgpu::u32;
using bzgpu::u64;
using bzgpu::i64;

constexpr u32 kWin = 0x8000;       // Inflater::new: window (deflate/encoder.rs:117)
constexpr u32 kMaxMatch = 258;     // LZSS_MAX_MATCH :110
constexpr u32 kMinMatch = 3;       // LZSS_MIN_MATCH :109
constexpr u32 kChain = 255;        // MATCH_SEARCH_COUNT - 1 (lzss/slidedict.rs:129, 232)
constexpr u32 kBlockMax = 0xFFFF;  // MAX_BLOCK_SIZE (deflate/encoder.rs:270)
#ifndef DF_CHUNK_LOG2
#define DF_CHUNK_LOG2 19
#endif
constexpr u32 kChunk = 1u << DF_CHUNK_LOG2; // positions per sort chunk (hash chains are built chunk by chunk)
constexpr u32 kChunkStride = kChunk + kWin;          // entries of one sort chunk: its positions and the 32 KiB in front
constexpr u32 kChunkTiles = kChunkStride / bzgpu::kSortTile; // 132 tiles of 8192 entries
constexpr u32 kPrevSpan = kChunkStride / 256;         // 256-entry slices of one sort chunk
constexpr u32 kMTile = 8192;       // positions per match workgroup
constexpr u32 kMThreads = 1024;
#ifndef DF_M2_THREADS
#define DF_M2_THREADS 256          // entries per workgroup of the match kernel (k_df_match2)
#endif
constexpr u32 kPTile = 4096;       // positions per parse tile
constexpr u32 kEntries = 260;      // a step advances by at most 258 + 2: entry offsets 0..259
constexpr u32 kFan = 64;           // tiles composed per level
#ifndef DF_BLOCK_THREADS
#define DF_BLOCK_THREADS 128
#endif
constexpr u32 kBThreads = DF_BLOCK_THREADS;
#ifndef DF_EMIT_THREADS
#define DF_EMIT_THREADS 64
#endif
constexpr u32 kEThreads = DF_EMIT_THREADS;
constexpr u32 kHdrWords = 160;     // bits of BFINAL + dynamic header: < 74 + 316 * 14
constexpr u32 kDfLmTable = 3 * 288 + 64 + 2 * 15 * (2 * 288 + 4); // scratch words of one length-limited table
constexpr u32 kDfLmWords = 2 * kDfLmTable;
constexpr u32 kSumPiece = 65536;   // bytes per checksum piece
constexpr u32 F_CODE = 0x80000000u; // code[q]: an LZSS code starts at q
constexpr u32 F_REF = 0x40000000u;  //          it is a reference: len | (dist - 1) << 9
constexpr u32 F_STEP = 0x20000000u; //          a step of the LZSS parse starts at q (lzss/encoder.rs:132-184: a step emits up
                                    //          to two literals and then a reference; only its first code is a step start)

struct DfBlock {
    u32 btype;     // 0 stored, 1 fixed, 2 dynamic
    u32 hdr_bits;  // BFINAL + BTYPE (+ dynamic header)
    u32 bytes;     // decompress_len
    u32 lm;        // tables that took the length-limited path
    u64 bits;      // whole block, btype 1 / 2
    u64 bit_off;   // where it starts in the stream
};

// what a part of a long stream tells the host when its kernels are over (k_df_part_keep, k_df_part_tail)
struct DfPartRes {
    u64 consumed;    // where the next part starts
    u64 last_bstart; // start of the last block written
    u32 skip, err, nb_all, end_byte;
    u32 st[5];       // blocks stored / fixed / dynamic, limited tables, dynamic blocks without distances
    u32 pad[3];
};
int df_launch_part_keep(hipStream_t st, const u64 *bstart, u32 *nb, u32 bcap, const u32 *code, u64 n, u64 guard, DfPartRes *res);
int df_launch_part_tail(hipStream_t st, const DfBlock *blocks, const u64 *bstart, const u32 *nb, const u64 *total_bits, const u8 *stream, DfPartRes *res);

u32 df_chunks(u64 n); // sort chunks of an input of n bytes
int df_launch_chains(hipStream_t st, const u8 *in, u64 n, u32 *v0, u32 *s, u16 *hs, u32 *hist, u32 *tbase, u32 *pe);
int df_launch_match(hipStream_t st, const u8 *in, const u32 *pe, u64 n, u32 *M);
int df_launch_match2(hipStream_t st, const u8 *in, u64 n, const u32 *s, u32 *M);
// the block chain taken in pieces beside the marking of the tiles (df_launch_parse with canon): a second stream,
// kCutPieces + 1 events, the arguments of df_launch_cuts and two words of device memory for the chain's state
constexpr u32 kCutPieces = 4;
struct DfPiecewiseCuts {
    hipStream_t st2;
    hipEvent_t ev[kCutPieces + 1];
    u64 *bstart;
    u32 *nb;
    u32 cap, dl0, first;
    u64 *state;
    // kdone != nullptr: piece i of the chain leaves the number of blocks it has closed in kdone[i + 1] (kdone[0] is
    // 0) and records evc[i]; the caller launches the blocks of every piece behind its event and df_launch_parse does
    // not make `st` wait for the chain
    u32 *kdone;
    hipEvent_t evc[kCutPieces];
};
u32 df_cut_pieces(u32 ntiles);
int df_launch_blocks_piece(hipStream_t st, const u8 *in, const u32 *code, u64 *bstart, u32 *nb, u32 cap, DfBlock *blocks, u8 *lens,
                           u32 *hdr, u32 *lm_scratch, u32 dl0, u32 last_is_final, const u32 *kr, u32 piece_last);
int df_launch_block_offsets(hipStream_t st, DfBlock *blocks, const u32 *nb, u64 *total_bits, u32 bit0);
// pc != nullptr (needs canon): the block starts are made on the way (no df_launch_cuts afterwards)
int df_launch_parse(hipStream_t st, const u32 *M, u64 n, u16 *step, u16 *const *tabs, u16 *const *ents, const u32 *counts,
                    u32 nlevels, u32 *code, u64 *bm, u64 *canon, const DfPiecewiseCuts *pc);
// dl0: decompress_len carried into the segment (0 unless it follows an Action::Flush); last_is_final: the
// segment ends the stream (Finish) rather than being flushed.  `in` is the segment's first byte; the dl0 bytes
// in front of it must be readable (a stored first block copies them).  The block chain (df_launch_cuts) is its
// own launch so that the host can drop the blocks at the end of a PART of a long stream (those whose cuts could
// still change with the input behind the part) before tables and offsets are made; bit0: the first block starts
// at this bit (0..7) of the output's first byte.
int df_launch_cuts(hipStream_t st, u64 n, u64 *bm, u64 *bstart, u32 *nb, u32 cap, u32 dl0, u32 first);
int df_launch_blocks(hipStream_t st, const u8 *in, const u32 *code, u64 *bstart, u32 *nb, u32 cap,
                     DfBlock *blocks, u8 *lens, u32 *hdr, u32 *lm_scratch, u64 *total_bits, u32 dl0, u32 last_is_final, u32 bit0);
int df_launch_emit(hipStream_t st, const u8 *in, const u32 *code, const u64 *bstart, const u32 *nb, u32 cap,
                   const DfBlock *blocks, const u8 *lens, const u32 *hdr, u32 *out);
struct DfCrcShifts { u32 x[8]; }; // x^(8 * 256 * 2^k) mod P, reflected: moves a CRC register over 256 * 2^k bytes
int df_launch_sums(hipStream_t st, const u8 *in, u64 n, u64 *asum, u64 *bsum, u32 *crc, u32 *last_sub, DfCrcShifts xk);
} // namespace dfgpu
