// deflate_engine.hip -- orchestration and C ABI of the Deflate / zlib / gzip encode path
// (include/bz2_mi355x.h section 4).  Kernels: k_deflate.hip.
//
// Replaces, behind the reference's Encoder surface: Inflater (deflate/encoder.rs:92-260), ZlibEncoder
// (zlib/encoder.rs:55-157), GZipEncoder (gzip/encoder.rs:50-135) driven with Action::Finish (and
// Action::Run in front of it), with or without a preset dictionary (::with_dict), and -- for Inflater -- with
// Action::Flush in the middle of a stream (a stream is then a sequence of byte-aligned SEGMENTS, df_enc_end).
#include <cstring>
#include <string>
#include <vector>

#include "engine_state.h"
#include "copy_pool.h"

#include <chrono>
#include <sys/mman.h>
#include <functional>
#include "k_deflate.h"

using namespace dfgpu;

struct DfWorkspace {
    DevBuf keys_in, keys_out, vals_in, vals_out, sort_tmp, prevd, est, segoff, concat, bitmap, canon, tabs, ents, bstart, nb, blocks, lens, hdr, lm, total,
        stream, asum, bsum, crc, part_res;
    double t_stage[6] = {0, 0, 0, 0, 0, 0}; // chains, matches, parse, blocks, emit, total
    u64 stats[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // blocks, stored, fixed, dynamic, limited tables, stream bytes, dynamic w/o distances
    hipEvent_t ev[7] = {};
    hipEvent_t evq[2 * kCutPieces + 1] = {}; // the pieces of the block chain beside the marking kernel
    bool ev_ready = false;
    // the blocks of the last call (or of its last part) for the debug entry points: fetched when one of them asks
    std::vector<DfBlock> h_blocks;
    std::vector<u64> h_bstart;
    u32 h_nb = 0;
    bool h_fetched = true;
};

void df_workspace_free(DfWorkspace *w)
{
    if (!w) return;
    DevBuf *all[] = {&w->keys_in, &w->keys_out, &w->vals_in, &w->vals_out, &w->sort_tmp, &w->prevd, &w->est, &w->segoff, &w->concat, &w->bitmap, &w->canon, &w->tabs, &w->ents,
                     &w->bstart, &w->nb, &w->blocks, &w->lens, &w->hdr, &w->lm, &w->total, &w->stream, &w->asum, &w->bsum,
                     &w->crc, &w->part_res};
    for (DevBuf *b : all) b->release();
    if (w->ev_ready) {
        for (hipEvent_t e : w->ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : w->evq) (void)hipEventDestroy(e);
    }
    delete w;
}

extern "C" size_t df_encode_bound(size_t n)
{
    // stored blocks are the worst case: 5 bytes per 0xFFFF, plus container, plus word slack for the bit writer
    return n + 5 * (n / 0xFFFF + 2) + 64;
}

// ---- reflected CRC-32 arithmetic on the host (crc32.rs:40-55): polynomials with bit 31 = x^0
static u32 gf_mul_reflected(u32 a, u32 b)
{
    u32 p = 0;
    for (u32 m = 1u << 31; m != 0 && a != 0; m >>= 1) { // a: coefficients from x^0 (bit 31) upwards
        if (a & m) {
            p ^= b;
            a &= ~m;
        }
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1; // b * x
    }
    return p;
}
static u32 gf_xpow8_reflected(u64 nbytes) // x^(8 * nbytes) mod P
{
    u32 r = 1u << 31, sq = 1u << 23; // x^0, x^8
    while (nbytes) {
        if (nbytes & 1) r = gf_mul_reflected(r, sq);
        sq = gf_mul_reflected(sq, sq);
        nbytes >>= 1;
    }
    return r;
}

// dict (host memory, may be NULL): Inflater::with_dict / ZlibEncoder::with_dict (deflate/encoder.rs:134-153,
// lzss/encoder.rs:104-130: the last 0x8000 bytes of it are history in front of the input)
// A SEGMENT of a stream (Action::Flush cuts a stream into segments: the LZSS stage is drained, the current block
// is closed WITHOUT the final bit and the bit string is padded to a byte, deflate/encoder.rs:170-195, :638-647,
// :236-246; the window and `decompress_len` live on).  d_in[0, n) is the segment; `prior` = stream bytes in
// front of it, readable at d_in[-prior, 0) (history of the match finder next to the dictionary's; a stored first
// block reaches back dl0 <= prior bytes); dl0 = decompress_len carried in; `final`: the segment ends the stream;
// head / tail: write the container's header / trailer (the trailer's checksums cover d_in[-prior, n)).
// *dl_out = decompress_len behind the segment.  A whole stream is the one segment (prior 0, dl0 0, final).
struct DfSeg {
    u64 prior = 0;
    u32 dl0 = 0;
    bool final = true, head = true, tail = true;
    // A PART of a long segment (one call handles < 2 GiB of positions): `more` = the input goes on behind
    // d_in[n) without a flush, and the n bytes handed in include look-ahead: only the blocks whose cuts and
    // codes cannot change with the bytes behind d_in[n) are written, *consumed tells where the next part
    // starts: at the STEP of the LZSS parse that holds the first block left out.  A block starts at a code,
    // but a step of the parse (lzss/encoder.rs:132-184) can emit one or two literals and then a reference, and
    // a block may be cut between them; the parse can only be taken up at the step's first byte, where the
    // decision was made.  skip = the literals of that step (0..2 bytes) that went out with the part before: the
    // first block of this part starts behind them.  bit0 / carry_byte: the bits (0..7) of the output's first
    // byte that the previous part has already filled.
    bool more = false;
    // Action::Run seen by a zlib / gzip wrapper (its container ends at the inner encoder's first None,
    // zlib/encoder.rs:146-150): write exactly the blocks the reference's Inflater has CLOSED once the iterator is dry.
    // The LZSS stage decides a step of its parse only with 258 + 3 bytes of look-ahead (lzss/encoder.rs:186-194), so
    // the codes of the steps that start at p <= n - 261 are out and are those of any longer input; a block is closed by
    // the arrival of the next block's first code (deflate/encoder.rs:585-593).  Only whole bytes leave the BitWriter.
    bool run = false;
    u32 skip = 0;
    u32 bit0 = 0;
    u8 carry_byte = 0;
    bool accumulate_stats = false;
};
struct DfPartOut {
    u64 consumed = 0;   // where the next part starts (the step that holds the first block left out)
    u32 skip = 0;       // bytes behind it that are already written (that step's literals)
    u32 end_bits = 0;   // bits (0..7) filled in the byte behind the bytes returned (more == true)
    u8 end_byte = 0;
};
// bytes at the end of a part that is not the last one whose codes or cuts may still change: a cut looks 0xFFFF
// bytes ahead, the code there another 258 + 2 + 258
constexpr u64 kPartGuard = 0x10000ull + 1024ull;
static int df_encode_core(bz_gpu_engine *g, int kind, const u8 *d_in, u64 n, const u8 *dict, size_t dict_len, u8 *d_out,
                          size_t cap, size_t *out_len, const DfSeg &seg = DfSeg(), u32 *dl_out = nullptr,
                          DfPartOut *part_out = nullptr)
{
    if (n >= (1ull << 31) - kWin) return BZ_E_PARAM; // positions and bit offsets are sized for < 2 GiB per call
    if (dict_len && kind == 2) return BZ_E_PARAM;    // GZipEncoder has no with_dict
    if (seg.dl0 > kBlockMax || seg.dl0 > seg.prior) return BZ_E_PARAM;
    HIPCHK(hipSetDevice(g->device));
    if (!g->df) g->df = new DfWorkspace();
    DfWorkspace *w = g->df;
    hipStream_t st = g->st;
    if (!w->ev_ready) {
        for (hipEvent_t &e : w->ev) HIPCHK(hipEventCreate(&e));
        for (hipEvent_t &e : w->evq) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        w->ev_ready = true;
    }
    // history in front of the segment: the last 32 KiB of (dictionary, then the stream so far)
    const u64 hist_dev = seg.prior < kWin ? seg.prior : kWin;
    const u64 hist_dict = dict_len < kWin - hist_dev ? dict_len : kWin - hist_dev;
    const u64 hist = hist_dict + hist_dev;
    const u64 nall = hist + n;                           // positions the chain and match stages see
    const u64 npad = nall + 16;
    const u32 ntiles = (u32)((n + kPTile - 1) / kPTile) + (n == 0 ? 1u : 0u);
    std::vector<u32> counts;
    counts.push_back(ntiles);
    while (counts.back() > 1) counts.push_back((counts.back() + kFan - 1) / kFan);
    const u32 nlevels = (u32)counts.size();
    size_t tab_words = 0, ent_words = 0;
    for (u32 c : counts) { tab_words += (size_t)c * kEntries; ent_words += c + 8; }
    const u32 bcap = (u32)(n / (kBlockMax - 300) + 2);

    int rc;
    const u64 nchunks = df_chunks(nall);
    const u64 nsort = nchunks * kChunkStride + 16; // words of a sorted-position array
    if ((rc = w->keys_in.ensure(npad * 2)) != BZ_OK) return rc;                          // step[]
    if ((rc = w->vals_in.ensure((nsort > npad ? nsort : npad) * 4)) != BZ_OK) return rc;  // pass 0 output, later M[]
    if ((rc = w->vals_out.ensure((nsort > npad ? nsort : npad) * 4)) != BZ_OK) return rc; // sorted positions, later code[]
    if ((rc = w->sort_tmp.ensure((nchunks * kChunkTiles + 1) * 256 * 4)) != BZ_OK) return rc; // per-tile digit counts
    if ((rc = w->keys_out.ensure(2 * (nchunks + 1) * 256 * 4)) != BZ_OK) return rc;       // per-chunk digit bases, both passes
    // BZ_DF_MATCH=walk: the chain-walking match kernel of rounds 1 and 2 (k_df_prev + k_df_match) instead of the one
    // that reads the candidates off the sorted order (k_df_match2); same match words either way
    const char *mv = getenv("BZ_DF_MATCH");
    const bool walk = mv && strcmp(mv, "walk") == 0;
    // BZ_DF_PARSE=doubling: the parse by pointer doubling of rounds 1 and 2 (k_df_tile_tab / k_df_mark) instead of
    // the canonical orbits (k_df_tile_orbit / k_df_mark2); same code words either way
    const char *pv = getenv("BZ_DF_PARSE");
    const bool doubling = pv && strcmp(pv, "doubling") == 0;
    if (!doubling && (rc = w->canon.ensure((size_t)ntiles * 64 * 8 + 64)) != BZ_OK) return rc; // a tile's canonical orbit, one bit per position
    if (walk) {
        if ((rc = w->est.ensure(nsort * 2)) != BZ_OK) return rc;  // hashes of the sorted positions
        if ((rc = w->prevd.ensure(npad * 4)) != BZ_OK) return rc; // per position: chain distance | chain length << 16
    }
    if (hist && (rc = w->concat.ensure(nall + 64)) != BZ_OK) return rc;
    if ((rc = w->tabs.ensure(tab_words * 2 + 64)) != BZ_OK) return rc;
    if ((rc = w->ents.ensure(ent_words * 2 + 64)) != BZ_OK) return rc;
    if ((rc = w->bstart.ensure(((size_t)bcap + 2) * 8)) != BZ_OK) return rc;
    if ((rc = w->bitmap.ensure((n / 64 + 8) * 8)) != BZ_OK) return rc;
    if ((rc = w->nb.ensure(64)) != BZ_OK) return rc;
    if ((rc = w->blocks.ensure((size_t)bcap * sizeof(DfBlock))) != BZ_OK) return rc;
    if ((rc = w->lens.ensure((size_t)bcap * 320)) != BZ_OK) return rc;
    if ((rc = w->hdr.ensure((size_t)bcap * kHdrWords * 4)) != BZ_OK) return rc;
    if ((rc = w->lm.ensure((size_t)bcap * kDfLmWords * 4)) != BZ_OK) return rc;
    if ((rc = w->total.ensure(128)) != BZ_OK) return rc; // the stream's bit count, the chain's state, its pieces' block counts
    if ((rc = w->part_res.ensure(sizeof(DfPartRes))) != BZ_OK) return rc;
    const size_t bound = df_encode_bound(n);
    if ((rc = w->stream.ensure(bound + 64)) != BZ_OK) return rc;
    const u64 ntot = seg.prior + n; // bytes the container's checksums cover
    const u32 npieces = (u32)((ntot + kSumPiece - 1) / kSumPiece);
    if (kind != 0) {
        if ((rc = w->asum.ensure((size_t)(npieces + 1) * 8)) != BZ_OK) return rc;
        if ((rc = w->bsum.ensure((size_t)(npieces + 1) * 8)) != BZ_OK) return rc;
        if ((rc = w->crc.ensure((size_t)(npieces + 1) * 4 + 256 * 4)) != BZ_OK) return rc; // per piece, + the last piece's sub-pieces
    }

    std::vector<u16 *> tabs(nlevels), ents(nlevels);
    {
        u16 *t = w->tabs.as<u16>(), *e = w->ents.as<u16>();
        for (u32 l = 0; l < nlevels; ++l) {
            tabs[l] = t; t += (size_t)counts[l] * kEntries;
            ents[l] = e; e += counts[l] + 8;
        }
    }
    u16 *step = w->keys_in.as<u16>();
    u32 *Mall = w->vals_in.as<u32>();
    u32 *M = Mall + hist; // match words of the input positions
    u32 *code = w->vals_out.as<u32>();
    const u8 *d_all = d_in;
    if (hist) { // history + input in one buffer for the chain and match stages
        if (hist_dict) HIPCHK(hipMemcpy(w->concat.p, dict + (dict_len - hist_dict), hist_dict, hipMemcpyHostToDevice));
        if (hist_dev + n)
            HIPCHK(hipMemcpyAsync(w->concat.as<u8>() + hist_dict, d_in - hist_dev, hist_dev + n, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemsetAsync(w->concat.as<u8>() + nall, 0, 64, st));
        d_all = w->concat.as<u8>();
    }

    HIPCHK(hipEventRecord(w->ev[0], st));
    HIPCHK(hipMemsetAsync(w->stream.p, 0, bound + 64, st));
    if (df_launch_chains(st, d_all, nall, w->vals_in.as<u32>(), w->vals_out.as<u32>(), w->est.as<u16>(), w->sort_tmp.as<u32>(), w->keys_out.as<u32>(),
                         walk ? w->prevd.as<u32>() : nullptr) != 0)
        return BZ_E_UNEXPECTED;
    HIPCHK(hipEventRecord(w->ev[1], st));
    if ((walk ? df_launch_match(st, d_all, w->prevd.as<u32>(), nall, Mall)
              : df_launch_match2(st, d_all, nall, w->vals_out.as<u32>(), Mall)) != 0)
        return BZ_E_UNEXPECTED;
    HIPCHK(hipEventRecord(w->ev[2], st));
    if (seg.skip > 2 || (seg.skip && seg.dl0)) return BZ_E_PARAM;
    // BZ_DF_CUTS=after: the chain of block starts behind the marking kernel (rounds 1 and 2) instead of in pieces
    // beside it on the engine's second stream
    const char *cv = getenv("BZ_DF_CUTS");
    const bool beside = n != 0 && !doubling && !(cv && strcmp(cv, "after") == 0); // (no tiles, no pieces)
    DfPiecewiseCuts pc;
    pc.st2 = g->st2;
    for (u32 i = 0; i <= kCutPieces; ++i) pc.ev[i] = w->evq[i];
    pc.bstart = w->bstart.as<u64>();
    pc.nb = w->nb.as<u32>();
    pc.cap = bcap; pc.dl0 = seg.dl0; pc.first = seg.skip;
    pc.state = w->total.as<u64>() + 2; // (spare words of the 64-byte buffer that takes the stream's bit count)
    // the blocks of a piece follow the chain's piece at once -- unless the host has to look at the chain first (a part
    // of a long stream, a wrapper ending under Action::Run: it drops blocks at the end)
    const bool blocks_follow = beside && !seg.more && !seg.run;
    pc.kdone = blocks_follow ? w->total.as<u32>() + 8 : nullptr;
    for (u32 i = 0; i < kCutPieces; ++i) pc.evc[i] = w->evq[kCutPieces + 1 + i];
    if (beside) HIPCHK(hipMemsetAsync(w->total.p, 0, 128, st));
    if (df_launch_parse(st, M, n, step, tabs.data(), ents.data(), counts.data(), nlevels, code, w->bitmap.as<u64>(),
                        doubling ? nullptr : w->canon.as<u64>(), beside ? &pc : nullptr) != 0)
        return BZ_E_UNEXPECTED;
    HIPCHK(hipEventRecord(w->ev[3], st));
    if (!beside && df_launch_cuts(st, n, w->bitmap.as<u64>(), w->bstart.as<u64>(), w->nb.as<u32>(), bcap, seg.dl0, seg.skip) != 0)
        return BZ_E_UNEXPECTED;
    u64 consumed = n;
    u32 next_skip = 0;
    HIPCHK(hipMemsetAsync(w->part_res.p, 0, sizeof(DfPartRes), st));
    if (seg.more) {
        // keep the blocks that START at or before n - guard: their ends (cuts) and codes are final; the next part starts at
        // the step of the parse that holds the first block left out.  Decided on the device (k_df_part_keep: rounds 1-4 read the
        // block starts and three code words back here, with the GPU idle meanwhile); the host learns it with the bit count.
        if (n <= kPartGuard) return BZ_E_UNEXPECTED;
        if (df_launch_part_keep(st, w->bstart.as<u64>(), w->nb.as<u32>(), bcap, code, n, kPartGuard, w->part_res.as<DfPartRes>()) != 0)
            return BZ_E_UNEXPECTED;
    }
    if (seg.run && !seg.more) {
        u32 nb_all = 0;
        HIPCHK(hipMemcpyAsync(&nb_all, w->nb.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (nb_all == 0xFFFFFFFFu || nb_all > bcap) return BZ_E_UNEXPECTED;
        std::vector<u64> hb((size_t)nb_all + 1);
        if (nb_all) HIPCHK(hipMemcpy(hb.data(), w->bstart.p, ((size_t)nb_all + 1) * 8, hipMemcpyDeviceToHost));
        // block j-1 is closed when the code at hb[j] has arrived: the step of the parse that holds it (at most two
        // literals in front of a reference) starts at or before n - 261.  (With a `skip` the first block starts
        // inside a step of the previous part; cuts further on are at least 0xFFFF - 257 bytes behind it.)
        u32 keep = 0;
        for (u32 j = nb_all; j-- > 1;) {
            const u64 bcut = hb[j];
            u32 cw[3] = {0, 0, 0};
            const u64 lo = bcut < 2 ? bcut : 2;
            HIPCHK(hipMemcpy(cw + (2 - lo), code + (bcut - lo), (size_t)(lo + 1) * 4, hipMemcpyDeviceToHost));
            if (!(cw[2] & F_CODE)) return BZ_E_UNEXPECTED;
            const u32 back = (cw[2] & F_STEP) ? 0u : ((cw[1] & F_STEP) ? 1u : 2u);
            if (back > lo || !(cw[2 - back] & F_STEP)) return BZ_E_UNEXPECTED;
            if (bcut - back + 261 <= n) {
                keep = j;
                break;
            }
        }
        HIPCHK(hipMemcpyAsync(w->nb.p, &keep, 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st)); // (`keep` lives on this stack frame)
    }
    if (blocks_follow) {
        const u32 np = df_cut_pieces(ntiles);
        for (u32 i = 0; i < np; ++i) {
            HIPCHK(hipStreamWaitEvent(st, pc.evc[i], 0));
            if (df_launch_blocks_piece(st, d_in, code, w->bstart.as<u64>(), w->nb.as<u32>(), bcap, w->blocks.as<DfBlock>(), w->lens.as<u8>(),
                                       w->hdr.as<u32>(), w->lm.as<u32>(), seg.dl0, seg.final ? 1u : 0u, pc.kdone + i, i + 1 == np ? 1u : 0u) != 0)
                return BZ_E_UNEXPECTED;
        }
        if (df_launch_block_offsets(st, w->blocks.as<DfBlock>(), w->nb.as<u32>(), w->total.as<u64>(), seg.bit0) != 0) return BZ_E_UNEXPECTED;
    } else if (df_launch_blocks(st, d_in, code, w->bstart.as<u64>(), w->nb.as<u32>(), bcap, w->blocks.as<DfBlock>(),
                                w->lens.as<u8>(), w->hdr.as<u32>(), w->lm.as<u32>(), w->total.as<u64>(), seg.dl0,
                                (seg.final && !seg.more && !seg.run) ? 1u : 0u, seg.bit0) != 0)
        return BZ_E_UNEXPECTED;
    if (seg.bit0) HIPCHK(hipMemcpyAsync(w->stream.p, &seg.carry_byte, 1, hipMemcpyHostToDevice, st)); // (the stream was cleared above)
    HIPCHK(hipEventRecord(w->ev[4], st));
    if (df_launch_emit(st, d_in, code, w->bstart.as<u64>(), w->nb.as<u32>(), bcap, w->blocks.as<DfBlock>(), w->lens.as<u8>(),
                       w->hdr.as<u32>(), w->stream.as<u32>()) != 0)
        return BZ_E_UNEXPECTED;
    if (kind != 0 && seg.tail) {
        DfCrcShifts xk;
        for (u32 k = 0; k < 8; ++k) xk.x[k] = gf_xpow8_reflected(256ull << k);
        if (df_launch_sums(st, d_in - seg.prior, ntot, w->asum.as<u64>(), w->bsum.as<u64>(), w->crc.as<u32>(), w->crc.as<u32>() + npieces + 1, xk) != 0)
            return BZ_E_UNEXPECTED;
    }
    HIPCHK(hipEventRecord(w->ev[5], st));
    if (df_launch_part_tail(st, w->blocks.as<DfBlock>(), w->bstart.as<u64>(), w->nb.as<u32>(), w->total.as<u64>(), w->stream.as<u8>(),
                            w->part_res.as<DfPartRes>()) != 0)
        return BZ_E_UNEXPECTED;
    u64 total_bits = 0;
    u32 nb = 0;
    DfPartRes pres;
    HIPCHK(hipMemcpyAsync(&total_bits, w->total.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nb, w->nb.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&pres, w->part_res.p, sizeof(pres), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); // the one look at the device per part
    if (seg.more && pres.err) return BZ_E_UNEXPECTED;
    if (nb == 0xFFFFFFFFu || nb > bcap) {
        fprintf(stderr, "bz2_mi355x: deflate parse produced no code start inside a block window (internal error)\n");
        return BZ_E_UNEXPECTED;
    }
    if (seg.more) {
        consumed = pres.consumed;
        next_skip = pres.skip;
    }
    // (a part that is not the last one hands on its last, partly filled byte instead of writing it)
    // ... and the bits of a partly filled byte never leave the BitWriter of a wrapper that ends under Action::Run
    const u64 body = (seg.more || seg.run) ? (total_bits >> 3) : ((total_bits + 7) >> 3);
    if (part_out) {
        part_out->consumed = consumed;
        part_out->skip = next_skip;
        part_out->end_bits = seg.more ? (u32)(total_bits & 7u) : 0u;
        part_out->end_byte = part_out->end_bits ? (u8)pres.end_byte : 0;
    }

    // container (zlib/encoder.rs:63-72,118-156; gzip/encoder.rs:62-75,88-134)
    u8 head[10], tail[8];
    size_t nhead = 0, ntail = 0;
    if (!seg.head) {
        // (a later segment of a stream: the header went out with the first one)
    } else if (kind == 1 && dict_len) { // zlib/encoder.rs:74-93: FDICT + Adler-32 of the whole dictionary
        u32 a = 1, b = 0;
        for (size_t i = 0; i < dict_len; ++i) { a = (a + dict[i]) % 65521; b = (b + a) % 65521; }
        const u32 h = (b << 16) | a;
        head[0] = 0x78; head[1] = 0xF9; head[2] = (u8)(h >> 24); head[3] = (u8)(h >> 16); head[4] = (u8)(h >> 8); head[5] = (u8)h;
        nhead = 6;
    } else if (kind == 1) {
        head[0] = 0x78; head[1] = 0xDA; nhead = 2;
    } else if (kind == 2) {
        const u8 h[10] = {0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0xFF};
        memcpy(head, h, 10); nhead = 10;
    }
    if (kind != 0 && seg.tail) {
        std::vector<u64> a(npieces), b(npieces);
        std::vector<u32> c((size_t)npieces + 1 + 256);
        if (npieces) {
            HIPCHK(hipMemcpy(a.data(), w->asum.p, (size_t)npieces * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(b.data(), w->bsum.p, (size_t)npieces * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(c.data(), w->crc.p, ((size_t)npieces + 1 + 256) * 4, hipMemcpyDeviceToHost));
        }
        if (kind == 1) { // adler32.rs:20-66
            u64 A = 1, B = 0;
            for (u32 t = 0; t < npieces; ++t) {
                const u64 len = (t + 1 == npieces) ? ntot - (u64)t * kSumPiece : kSumPiece;
                B = (B + (len % 65521) * A + b[t]) % 65521;
                A = (A + a[t]) % 65521;
            }
            const u32 h = (u32)((B << 16) | A);
            tail[0] = (u8)(h >> 24); tail[1] = (u8)(h >> 16); tail[2] = (u8)(h >> 8); tail[3] = (u8)h; ntail = 4;
        } else { // CRC-32 (reflected), little endian, then ISIZE
            const u32 xpiece = gf_xpow8_reflected(kSumPiece);
            u32 raw = 0; // register for a zero initial value over the whole input
            for (u32 t = 0; t < npieces; ++t) {
                const u64 len = (t + 1 == npieces) ? ntot - (u64)t * kSumPiece : kSumPiece;
                if (len == kSumPiece) raw = gf_mul_reflected(raw, xpiece) ^ c[t]; // folded on the device
                else
                    for (u32 s = 0; (u64)s * 256 < len; ++s) { // the last, partial piece: its sub-pieces
                        const u64 sl = (len - (u64)s * 256) < 256 ? (len - (u64)s * 256) : 256;
                        raw = gf_mul_reflected(raw, gf_xpow8_reflected(sl)) ^ c[(size_t)npieces + 1 + s];
                    }
            }
            const u32 crc = raw ^ gf_mul_reflected(0xFFFFFFFFu, gf_xpow8_reflected(ntot)) ^ 0xFFFFFFFFu;
            const u32 isz = (u32)ntot;
            for (int i = 0; i < 4; ++i) tail[i] = (u8)(crc >> (8 * i));
            for (int i = 0; i < 4; ++i) tail[4 + i] = (u8)(isz >> (8 * i));
            ntail = 8;
        }
    }
    const size_t need = nhead + body + ntail;
    *out_len = need;
    if (d_out) {
        if (need > cap) return BZ_E_CAPACITY;
        HIPCHK(hipMemcpyAsync(d_out + nhead, w->stream.p, body, hipMemcpyDeviceToDevice, st));
        if (nhead) HIPCHK(hipMemcpyAsync(d_out, head, nhead, hipMemcpyHostToDevice, st)); // (`head` and `tail` live on this stack frame: the wait below)
        if (ntail) HIPCHK(hipMemcpyAsync(d_out + nhead + body, tail, ntail, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipEventRecord(w->ev[6], st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 5; ++i) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, w->ev[i], w->ev[i + 1]);
        w->t_stage[i] = ms * 1e-3;
    }
    {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, w->ev[0], w->ev[6]);
        w->t_stage[5] = ms * 1e-3;
    }
    // statistics of the last call (tests, bench): counted by k_df_part_tail; the block list itself is fetched when asked for
    w->h_nb = nb;
    w->h_fetched = false;
    const u64 prev_bytes = seg.accumulate_stats ? w->stats[5] : 0;
    if (!seg.accumulate_stats) memset(w->stats, 0, sizeof(w->stats));
    w->stats[0] += nb;
    w->stats[1] += pres.st[0];
    w->stats[2] += pres.st[1];
    w->stats[3] += pres.st[2];
    w->stats[4] += pres.st[3];
    w->stats[6] += pres.st[4];
    w->stats[5] = prev_bytes + need;
    if (dl_out) *dl_out = nb ? (u32)((nb == 1 ? seg.dl0 : 0u) + (n - pres.last_bstart)) : seg.dl0;
    return BZ_OK;
}

// A segment of any length: parts of at most BZ_DF_PART_MIB (default 1024 MiB) positions each, so that
// positions and bit offsets stay inside 32 bits.  A part that is not the last one is handed kPartGuard bytes
// of look-ahead and writes the blocks that cannot change any more; the next part starts at the first block it
// left out (a code start: the lazy parse from a code start does not depend on what came before,
// lzss/encoder.rs:132-184), with the 32 KiB in front of it as history and in the middle of the byte its
// predecessor ended in.  The bytes are those of one pass over the whole segment.
static bool df_trace()
{
    static const bool on = getenv("BZ_DF_TRACE") != nullptr; // (diagnostics: a line per part and per one-shot call)
    return on;
}
static double df_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static u64 df_part_bytes()
{
    static const u64 v = [] {
        const char *s = getenv("BZ_DF_PART_MIB");
        long mib = s ? atol(s) : 1024;
        if (mib < 1) mib = 1;
        if (mib > 1536) mib = 1536;
        return (u64)mib << 20;
    }();
    return v;
}

// The one-shot call over host buffers runs its parts beside its own copies (df_encode_buffer_dict): a part may start once
// its bytes have been uploaded (need_input: "bytes [0, upto) of the segment must be on the device"), and its bytes of the
// stream leave for the host while the next part is encoded (part_done: "stream bytes [off, off + len) are final").
struct DfPartHooks {
    u64 part_bytes = 0; // 0: BZ_DF_PART_MIB
    u64 first_part_bytes = 0; // (if not 0) the first part's size: short, so that the kernels start behind a short upload
    std::function<int(u64 upto)> need_input;
    std::function<void(size_t off, size_t len)> part_done;
};

static int df_encode_parts(bz_gpu_engine *g, int kind, const u8 *d_in, u64 n, const u8 *dict, size_t dict_len, u8 *d_out,
                           size_t cap, size_t *out_len, const DfSeg &seg0 = DfSeg(), u32 *dl_out = nullptr,
                           const DfPartHooks *hooks = nullptr)
{
    const u64 part_all = hooks && hooks->part_bytes ? hooks->part_bytes : df_part_bytes();
    const u64 part_first = hooks && hooks->first_part_bytes ? std::min<u64>(hooks->first_part_bytes, part_all) : part_all;
    if (n <= part_first + kPartGuard) {
        if (hooks && hooks->need_input) {
            const int irc = hooks->need_input(n);
            if (irc != BZ_OK) return irc;
        }
        const int rc = df_encode_core(g, kind, d_in, n, dict, dict_len, d_out, cap, out_len, seg0, dl_out);
        if (rc == BZ_OK && hooks && hooks->part_done) hooks->part_done(0, *out_len);
        return rc;
    }
    u64 pos = 0;
    size_t written = 0;
    DfSeg seg = seg0;
    *out_len = 0;
    for (u32 k = 0;; ++k) {
        const u64 left = n - pos;
        const u64 part = k == 0 ? part_first : part_all;
        const bool last = left <= part + kPartGuard;
        const u64 len = last ? left : part + kPartGuard;
        seg.prior = seg0.prior + pos;
        seg.dl0 = pos ? 0u : seg0.dl0; // (a part starts at a block start)
        seg.head = seg0.head && pos == 0;
        seg.tail = seg0.tail && last;
        seg.final = seg0.final;
        seg.more = !last;
        seg.run = seg0.run && last;
        seg.accumulate_stats = k > 0;
        size_t got = 0;
        DfPartOut po;
        const double t_part = df_now_ms();
        if (hooks && hooks->need_input) {
            const int irc = hooks->need_input(std::min<u64>(n, pos + len + 64));
            if (irc != BZ_OK) return irc;
        }
        const double t_in = df_now_ms();
        const int rc = df_encode_core(g, kind, d_in + pos, len, dict, dict_len, d_out ? d_out + written : nullptr,
                                      d_out ? cap - written : 0, &got, seg, last ? dl_out : nullptr, &po);
        if (rc != BZ_OK) return rc;
        if (hooks && hooks->part_done && d_out) hooks->part_done(written, got);
        if (df_trace())
            fprintf(stderr, "bz2_mi355x: deflate part %u: input [%llu, +%llu) of %llu, kept %llu bytes of it, %zu stream bytes, "
                            "%u bits handed on, %llu blocks; waited for its input %.2f ms, call %.2f ms of which kernels %.2f (at %.1f)\n", k, (unsigned long long)pos, (unsigned long long)len,
                    (unsigned long long)n, (unsigned long long)(last ? len : po.consumed), got, last ? 0u : po.end_bits,
                    (unsigned long long)g->df->h_nb, t_in - t_part, df_now_ms() - t_in, g->df->t_stage[5] * 1e3, df_now_ms());
        written += got;
        *out_len = written;
        if (last) return BZ_OK;
        pos += po.consumed;
        seg.skip = po.skip;
        seg.bit0 = po.end_bits;
        seg.carry_byte = po.end_byte;
    }
}

// ---- C ABI ------------------------------------------------------------------------------------------------
extern "C" int df_gpu_encode_device(bz_gpu_engine *g, int kind, const void *d_in, size_t n, void *d_out, size_t cap,
                                    size_t *out_len)
{
    if (!g || !out_len || (!d_in && n) || kind < 0 || kind > 2) return BZ_E_PARAM;
    if (n && ((uintptr_t)d_in & 3u)) return BZ_E_PARAM;
    *out_len = 0;
    return df_encode_parts(g, kind, static_cast<const u8 *>(d_in), n, nullptr, 0, static_cast<u8 *>(d_out), cap, out_len);
}

extern "C" int df_gpu_encode_device_dict(bz_gpu_engine *g, int kind, const void *d_in, size_t n, const uint8_t *dict,
                                         size_t dict_len, void *d_out, size_t cap, size_t *out_len)
{
    if (!g || !out_len || (!d_in && n) || kind < 0 || kind > 2 || (!dict && dict_len)) return BZ_E_PARAM;
    if (n && ((uintptr_t)d_in & 3u)) return BZ_E_PARAM;
    *out_len = 0;
    return df_encode_parts(g, kind, static_cast<const u8 *>(d_in), n, dict, dict_len, static_cast<u8 *>(d_out), cap, out_len);
}

extern "C" int df_gpu_last_timings(bz_gpu_engine *g, double out_seconds[6])
{
    if (!g || !out_seconds) return BZ_E_PARAM;
    for (int i = 0; i < 6; ++i) out_seconds[i] = g->df ? g->df->t_stage[i] : 0.0;
    return BZ_OK;
}

extern "C" int df_gpu_last_stats(bz_gpu_engine *g, uint64_t out[8])
{
    if (!g || !out) return BZ_E_PARAM;
    for (int i = 0; i < 8; ++i) out[i] = g->df ? g->df->stats[i] : 0;
    return BZ_OK;
}

// test hook: the LZSS codes of the last call in stream order as (len, pos) pairs, len 0 = literal `pos`
// (what LzssEncoder::next yields, lzss/encoder.rs:203-234).  Returns the code count through *count.
extern "C" int df_gpu_debug_codes(bz_gpu_engine *g, const void *d_in, size_t n, uint32_t *out_pairs, size_t cap, size_t *count)
{
    if (!g || !g->df || !count) return BZ_E_PARAM;
    DfWorkspace *w = g->df;
    std::vector<u32> code(n);
    std::vector<u8> in(n);
    if (n) {
        HIPCHK(hipMemcpy(code.data(), w->vals_out.p, n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(in.data(), d_in, n, hipMemcpyDeviceToHost));
    }
    size_t k = 0;
    for (size_t q = 0; q < n; ++q) {
        const u32 c = code[q];
        if (!(c & F_CODE)) continue;
        if (k < cap) {
            if (c & F_REF) { out_pairs[2 * k] = c & 511u; out_pairs[2 * k + 1] = (c >> 9) & 32767u; }
            else { out_pairs[2 * k] = 0; out_pairs[2 * k + 1] = in[q]; }
        }
        ++k;
    }
    *count = k;
    return BZ_OK;
}

// test hook: (tokens are not counted here) per block: start offset, bytes, btype, bits
extern "C" int df_gpu_debug_blocks(bz_gpu_engine *g, uint64_t *out4, size_t cap, size_t *count)
{
    if (!g || !g->df || !count) return BZ_E_PARAM;
    DfWorkspace *w = g->df;
    if (!w->h_fetched) { // (the last call's blocks are still on the device)
        HIPCHK(hipSetDevice(g->device));
        w->h_blocks.resize(w->h_nb);
        w->h_bstart.resize((size_t)w->h_nb + 1);
        if (w->h_nb) HIPCHK(hipMemcpy(w->h_blocks.data(), w->blocks.p, (size_t)w->h_nb * sizeof(DfBlock), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(w->h_bstart.data(), w->bstart.p, ((size_t)w->h_nb + 1) * 8, hipMemcpyDeviceToHost));
        w->h_fetched = true;
    }
    *count = w->h_blocks.size();
    for (size_t k = 0; k < w->h_blocks.size() && k < cap; ++k) {
        const DfBlock &b = w->h_blocks[k];
        out4[4 * k] = w->h_bstart[k];
        out4[4 * k + 1] = b.bytes;
        out4[4 * k + 2] = b.btype;
        out4[4 * k + 3] = b.btype == 0 ? (((b.bit_off + 3 + 7) & ~7ull) + 32 + 8ull * b.bytes - b.bit_off) : b.bits;
    }
    return BZ_OK;
}

extern "C" int df_encode_buffer_dict(int kind, int device, const uint8_t *in, size_t in_len, const uint8_t *dict,
                                     size_t dict_len, uint8_t **out, size_t *out_len);

extern "C" int df_encode_buffer(int kind, int device, const uint8_t *in, size_t in_len, uint8_t **out, size_t *out_len)
{
    return df_encode_buffer_dict(kind, device, in, in_len, nullptr, 0, out, out_len);
}

extern "C" int df_encode_buffer_dict(int kind, int device, const uint8_t *in, size_t in_len, const uint8_t *dict,
                                     size_t dict_len, uint8_t **out, size_t *out_len)
{
    if (!out || !out_len || (!in && in_len) || kind < 0 || kind > 2 || (!dict && dict_len)) return BZ_E_PARAM;
    *out = nullptr;
    *out_len = 0;
    // (an engine per device is kept between one-shot calls, with its workspace and the two buffers below:
    // bz_release_cached_resources frees it)
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    if (df_trace()) fprintf(stderr, "bz2_mi355x: df_encode_buffer entered at %.1f\n", df_now_ms());
    bz_gpu_engine *g = dec_cache_take(device, 2);
    int rc = g ? BZ_OK : bz_gpu_engine_create(&g, device, 1);
    if (rc != BZ_OK) return rc;
    const size_t cap = df_encode_bound(in_len) + 8;
    uint8_t *h = nullptr;
    size_t n_out = 0;
    rc = hipSetDevice(device) == hipSuccess ? BZ_OK : BZ_E_UNEXPECTED;
    if (rc == BZ_OK) rc = g->dec_in.ensure(in_len + 64);
    if (rc == BZ_OK) rc = g->oneshot_out.ensure(cap);
    // Large inputs: the call's two copies run beside its kernels (copy_pool.h).  The input goes up in 32 MiB slices from one
    // thread, in order; the encode runs in parts (the bytes do not depend
    // on the parts, tests/test_gpu_deflate.py::test_many_part_seams_equal_oracle_golden), each as soon as its bytes have
    // arrived, and a part's bytes of the stream leave for the caller's buffer while the next part is encoded.  Rounds 1-4
    // uploaded everything, encoded, downloaded: 110 ms per GiB for 66 ms of kernels.
    if (rc == BZ_OK && in_len >= ((size_t)64 << 20)) {
        // Parts: a short first one (64 MiB: the kernels start 1.5 ms into the upload), then 256 MiB each -- a part's kernels cost
        // 1.4 ms more than its share of one pass (the block chain of a part stands behind its marking, every launch has its tail),
        // and the upload (50 GB/s) is 300 MiB ahead when the first part is over.  Eight parts of 128 MiB: 77 ms of kernels per
        // GiB instead of 66 (profiles/r05_host_copies.md).
        const u64 part_bytes = (u64)256 << 20, first_bytes = (u64)64 << 20;
        // the caller's buffer: 2 MiB-aligned memory that asks for huge pages (its pages are touched where stream bytes land, on
        // several threads, in front of each copy: 44 MB of 4 KiB pages took 5 ms, the last part's stood behind the call)
        {
            void *q = nullptr;
            const size_t want = (cap + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            if (posix_memalign(&q, (size_t)2 << 20, want) != 0) q = nullptr;
            if (q) (void)madvise(q, want, MADV_HUGEPAGE);
            h = static_cast<uint8_t *>(q);
        }
        if (!h) rc = BZ_E_NOMEM;
        if (rc == BZ_OK) {
            CopyPool pool(device);
            const size_t S = CopyPool::slice_bytes();
            std::vector<size_t> up;
            for (size_t off = 0; off < in_len; off += S)
                up.push_back(pool.submit(static_cast<u8 *>(g->dec_in.p) + off, in + off, std::min(S, in_len - off), hipMemcpyHostToDevice));
            size_t waited = 0;
            DfPartHooks hooks;
            hooks.part_bytes = std::min<u64>(part_bytes, df_part_bytes());
            hooks.first_part_bytes = first_bytes;
            hooks.need_input = [&](u64 upto) {
                const size_t want = (size_t)std::min<u64>((upto + S - 1) / S, up.size());
                for (; waited < want; ++waited) pool.wait(up[waited]);
                return pool.failed() ? BZ_E_UNEXPECTED : BZ_OK;
            };
            hooks.part_done = [&](size_t off, size_t len) {
                if (len) pool.submit(h + off, static_cast<const u8 *>(g->oneshot_out.p) + off, len, hipMemcpyDeviceToHost, true);
            };
            rc = df_encode_parts(g, kind, static_cast<const u8 *>(g->dec_in.p), in_len, dict, dict_len, static_cast<u8 *>(g->oneshot_out.p), cap,
                                 &n_out, DfSeg(), nullptr, &hooks);
            const double t_parts = df_now_ms();
            pool.wait_all();
            if (df_trace()) fprintf(stderr, "bz2_mi355x: df_encode_buffer: parts over at %.1f, copies over at %.1f\n", t_parts, df_now_ms());
            if (rc == BZ_OK && pool.failed()) rc = BZ_E_UNEXPECTED;
        }
        if (rc != BZ_OK) { // (the buffer is not shrunk to the stream: the pages behind it were never touched)
            free(h);
            h = nullptr;
        }
    } else {
    if (rc == BZ_OK && in_len && hipMemcpy(g->dec_in.p, in, in_len, hipMemcpyHostToDevice) != hipSuccess) rc = BZ_E_UNEXPECTED;
    if (rc == BZ_OK) rc = df_gpu_encode_device_dict(g, kind, g->dec_in.p, in_len, dict, dict_len, g->oneshot_out.p, cap, &n_out);
    if (rc == BZ_OK) {
        h = (uint8_t *)malloc(n_out ? n_out : 1);
        if (!h) rc = BZ_E_NOMEM;
        else if (hipMemcpy(h, g->oneshot_out.p, n_out, hipMemcpyDeviceToHost) != hipSuccess) {
            free(h);
            h = nullptr;
            rc = BZ_E_UNEXPECTED;
        }
    }
    }
    if (rc == BZ_OK) dec_cache_put(device, g);
    else bz_gpu_engine_destroy(g); // (an engine that met an error is not kept)
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (rc != BZ_OK) return rc;
    *out = h;
    *out_len = n_out;
    return BZ_OK;
}

// ---- streaming context: the Encoder::next contract of Inflater / ZlibEncoder / GZipEncoder ----------------
// Action::Run accumulates, Action::Finish encodes everything written so far (the stream of these encoders
// depends on all of the input: one 32 KiB window, one bit string).  Action::Flush is refused.
struct df_enc {
    int kind = 0, device = 0;
    std::vector<uint8_t> in, out, dict; // in: bytes written since the last segment went to the device
    size_t out_head = 0;
    bool finished = false;
    // the stream so far lives on the device: a segment needs the 32 KiB in front of it as history, a stored
    // block behind a flush up to 0xFFFF bytes, the container's checksums all of it
    bz_gpu_engine *g = nullptr;
    DevBuf d_data, d_out;
    size_t total = 0;   // stream bytes on the device
    size_t encoded = 0; // ... of which segments have been written for
    uint32_t dl = 0;    // InflaterInner.decompress_len behind the last segment (deflate/encoder.rs:267)
    bool wrote_head = false;
    int last_status = BZ_OK; // of the last df_enc_end: an engine that met an infrastructure error is not parked for others
};

extern "C" int df_enc_create(df_enc **out, int kind, int device)
{
    if (!out || kind < 0 || kind > 2) return BZ_E_PARAM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BZ_E_NOGPU;
    df_enc *e = new df_enc();
    e->kind = kind;
    e->device = device;
    *out = e;
    return BZ_OK;
}

extern "C" int df_enc_create_dict(df_enc **out, int kind, int device, const uint8_t *dict, size_t dict_len)
{
    if ((!dict && dict_len) || (dict_len && kind == 2)) return BZ_E_PARAM;
    const int rc = df_enc_create(out, kind, device);
    if (rc == BZ_OK && dict_len) (*out)->dict.assign(dict, dict + dict_len);
    return rc;
}

extern "C" void df_enc_destroy(df_enc *e)
{
    if (!e) return;
    if (e->g) {
        int caller_device = -1;
        (void)hipGetDevice(&caller_device);
        (void)hipSetDevice(e->device);
        e->d_data.release();
        e->d_out.release();
        // kept, with its workspace, for the next context or one-shot call on the device -- unless the context met an
        // infrastructure error (a sticky HIP error, a half-grown workspace): bz_decode_buffer, df_encode_buffer and
        // bz_dec_destroy do the same
        if (e->last_status == BZ_OK) dec_cache_put(e->device, e->g);
        else bz_gpu_engine_destroy(e->g);
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
    }
    delete e;
}

extern "C" int df_enc_write(df_enc *e, const uint8_t *in, size_t n)
{
    if (!e || (!in && n)) return BZ_E_PARAM;
    if (e->finished) return BZ_OK; // Inflater ignores input behind the final block (deflate/encoder.rs:638-660)
    e->in.insert(e->in.end(), in, in + n);
    return BZ_OK;
}

// Action::Run (Inflater): everything stays pending (the LZSS stage holds 261 bytes of look-ahead, the block stage up
// to 0xFFFF bytes; the blocks the reference would hand out early come with the next Flush / Finish: same bytes,
// coarser moments).
// Action::Flush (Inflater only): the bytes written since the last segment are encoded as one SEGMENT -- the LZSS
// stage drained (lzss/encoder.rs:196-200, :224-226), the block closed without the final bit
// (deflate/encoder.rs:177-184, :638-647), the bit string padded to a byte (:227-235); the 32 KiB window and
// decompress_len carry over.  Action::Finish: the last segment, final bit set, container trailer.
// The zlib / gzip wrappers end their container at the first None they see, whatever the action
// (zlib/encoder.rs:131-151): see `ends_container` below.
static int df_enc_end_impl(df_enc *e, int action);
extern "C" int df_enc_end(df_enc *e, int action)
{
    if (!e || action < 0 || action > 2) return BZ_E_PARAM;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    const int rc = df_enc_end_impl(e, action);
    e->last_status = rc;
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return rc;
}
static int df_enc_end_impl(df_enc *e, int action)
{
    if (action == 0 && e->kind == 0) return BZ_OK;
    if (e->finished) return BZ_OK; // flush() / finish() behind the final block do nothing (:636-660)
    // zlib / gzip: the container ends at the first None of the inner Inflater whatever the Action
    // (zlib/encoder.rs:146-150, gzip/encoder.rs:129-133): header + what the Inflater yields under this Action
    // (Run: the whole bytes of the blocks it has closed; Flush: the flushed segment) + the trailer over everything
    // the iterator handed over; afterwards the encoder yields None and leaves its caller's iterator alone.
    const bool ends_container = e->kind != 0;
    int rc;
    if (!e->g) {
        e->g = dec_cache_take(e->device, 2);
        if (!e->g && (rc = bz_gpu_engine_create(&e->g, e->device, 0)) != BZ_OK) return rc;
    }
    HIPCHK(hipSetDevice(e->device));
    const size_t add = e->in.size();
    if (e->total + add + 64 > e->d_data.cap) { // grow, keeping the stream so far
        DevBuf bigger;
        if ((rc = bigger.ensure((e->total + add) * 2 + 4096)) != BZ_OK) return rc;
        if (e->total) { // (on the engine's stream and waited for: a device-to-device copy need not be over when it returns)
            HIPCHK(hipMemcpyAsync(bigger.p, e->d_data.p, e->total, hipMemcpyDeviceToDevice, e->g->st));
            HIPCHK(hipStreamSynchronize(e->g->st));
        }
        e->d_data.release();
        e->d_data = bigger;
    }
    if (add) HIPCHK(hipMemcpy(e->d_data.as<u8>() + e->total, e->in.data(), add, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(e->d_data.as<u8>() + e->total + add, 0, 64, e->g->st)); // (ordered in front of the engine's kernels)
    e->total += add;
    e->in.clear();
    DfSeg seg;
    seg.prior = e->encoded;
    seg.dl0 = e->dl;
    seg.final = (action == 2);
    seg.run = (action == 0);
    seg.head = !e->wrote_head;
    seg.tail = (action == 2) || ends_container;
    const size_t n = e->total - e->encoded;
    const size_t cap = df_encode_bound(n) + 64;
    if ((rc = e->d_out.ensure(cap)) != BZ_OK) return rc;
    size_t k = 0;
    uint32_t dl = 0;
    rc = df_encode_parts(e->g, e->kind, e->d_data.as<u8>() + e->encoded, n, e->dict.data(), e->dict.size(), e->d_out.as<u8>(), cap,
                        &k, seg, &dl);
    if (rc != BZ_OK) return rc;
    if (e->out_head && e->out_head == e->out.size()) {
        e->out.clear();
        e->out_head = 0;
    }
    const size_t old = e->out.size();
    e->out.resize(old + k);
    if (k) HIPCHK(hipMemcpy(e->out.data() + old, e->d_out.p, k, hipMemcpyDeviceToHost));
    e->wrote_head = true;
    e->encoded = e->total;
    e->dl = dl;
    if (action == 2 || ends_container) {
        e->finished = true;
        e->d_data.release();
    }
    return BZ_OK;
}

extern "C" int df_enc_finished(const df_enc *e) { return e && e->finished ? 1 : 0; }

extern "C" size_t df_enc_pending(const df_enc *e) { return e ? e->out.size() - e->out_head : 0; }

extern "C" long df_enc_read(df_enc *e, uint8_t *out, size_t cap)
{
    if (!e || (!out && cap)) return BZ_E_PARAM;
    const size_t avail = e->out.size() - e->out_head;
    const size_t k = avail < cap ? avail : cap;
    if (k) memcpy(out, e->out.data() + e->out_head, k);
    e->out_head += k;
    return (long)k;
}
