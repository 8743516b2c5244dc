// engine_state.h -- the engine object behind `bz_gpu_engine*` (include/bz2_mi355x.h, section 2),
// shared by the encode orchestration (engine.hip) and the decode orchestration (dec_engine.hip).
#pragma once
#include "../../include/bz2_mi355x.h"
#include "bzgpu.h"

#include <cstdio>
#include <vector>

using namespace bzgpu;

struct DecWorkspace;
void dec_workspace_free(DecWorkspace *w);
// the decode path as the encoder's self-check uses it (dec_engine.hip): d_in[n] -> d_out (device memory, cap bytes);
// returns an infrastructure status, the decoder's verdict in *verdict and the bytes it produced in *produced
int dec_decode_for_verify(struct bz_gpu_engine *g, const uint8_t *d_in, uint64_t n, uint8_t *d_out, uint64_t cap,
                          uint64_t *produced, int *verdict);
// the engines the one-shot calls over host buffers keep between calls (bz_decode_buffer, df_encode_buffer; one per
// device; bz_release_cached_resources frees them)
struct bz_gpu_engine *dec_cache_take(int device, int prefer = 0);
void dec_cache_put(int device, struct bz_gpu_engine *g);
void dec_release_cached();
struct DfWorkspace;
void df_workspace_free(DfWorkspace *w);

#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "bz2_mi355x: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__,   \
                    __LINE__);                                                                        \
            return BZ_E_UNEXPECTED;                                                                   \
        }                                                                                             \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return BZ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (hipMalloc(&p, bytes) != hipSuccess) return BZ_E_NOMEM;
            want = bytes;
        }
        cap = want;
        return BZ_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct bz_gpu_engine {
    int device = 0;
    hipStream_t st = nullptr;   // rotation sort (and everything serial)
    hipStream_t st2 = nullptr;  // MTF / Huffman / emission of the previous sub-batch
    size_t max_blocks = 0;

    // constant tables
    DevBuf crc_tab, xp16, xp2;
    // partition state (sized by the input)
    DevBuf tile_last, carry_in, tile_crc, tile_count, tile_off, sub_off, sub_rs, scal, scan_part, rle, blocks_all, crc_all;
    // the cuts from tables (k_rle1.hip "kernels H")
    DevBuf cut_step_t0, cut_step_nt, cut_step_w0, cut_tab, cut_comp;
    CutPlan cut_plan = {};
    bool cut_ready = false;       // the tables of the current range are filled (or being filled on st)
    u64 h_halo_total = 0;         // sharded job: the image bytes a rank found in front of its slab (a D2H copy queued on st lands here:
                                  // the engine outlives every return path of the call, a stack frame does not)
    u64 cut_stats[2] = {0, 0};    // since creation: partitions cut from tables, partitions that fell back to the chain kernel
    std::vector<BlockDesc> h_blocks;
    std::vector<u32> h_crc;
    const u8 *d_in = nullptr;
    u64 n_in = 0;
    u64 slab_t0 = 0, slab_t1 = 0; // tiles this engine splits (the whole input on one GPU)
    int level = 9;
    // batch workspace (sized by max_blocks)
    DevBuf lblocks, lcrc, SA, R, KA, VA, KB, VB, tile_hist, count, flags, tlo, tln, nonfinal, active, per_k,
        per_shift, lin_p, lin_sig, bin_cursor, pb_gate, bin_base, newbits, sym_code, keyinfo, count2, tile_nf, L, orig_ptr, inuse_bits, summ, summ_len, init_state, rank8, ztile_last, ztile_cnt, zstate, ztick, mtf,
        mtf_freq, bout, selector, code_len, group_bitoff, lm_scratch, hglen, hpack, hrfreq, hlm, hpass, stream, error_flag, packlist, gh_tiles, gbase,
        tile_state, tickets;
    u32 sort_epoch = 0; // fused radix passes: tag of the current pass in tile_state / tickets
    bool zle_fused_broken = false; // the one-launch ZLE stage misbehaved on this engine once: it stays on the three kernels
    u32 fused_state[2] = {0, 0}; // [0] fused passes found misbehaving on THIS engine (they stay off for it), [1] fallbacks
    hipEvent_t ev_aux = nullptr; // work that runs on st2 beside st (the RLE1 image beside the cut chain)
    size_t ws_blocks = 0; // blocks the batch workspace holds now (<= max_blocks)
    // own packed buffer / assemble list for the single-GPU convenience call
    DevBuf packed, gathered, asmlist;
    unsigned long long *h_active = nullptr; // pinned
    // results of the last encode
    std::vector<BlockOut> h_out;
    std::vector<u32> h_out_nblock;
    std::vector<u32> h_out_pass; // 32 words per block: k_huff_sweep's per-pass figures (bz_gpu_debug_block_sections)
    double t_stage[6] = {0, 0, 0, 0, 0, 0};
    KernelProf prof;
    // last bz_gpu_encode_sharded: wait for the hop, hop -> hand-on, gather, assembly; entry -> ready for the hop (scan,
    // counts, the slab's image offsets, its image and the cut tables under way), the chain kernel or table look-ups alone,
    // the whole call
    double shard_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 bwt_stats[4] = {0, 0, 0, 0};
    u64 round_active[64] = {};

    struct Span {
        int stage;
        hipStream_t st;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    // self-check (bz_gpu_engine_set_verify / BZ_VERIFY=1): the blocks of every bz_gpu_encode_blocks call are decoded on
    // the device and compared with the input they cover before the call returns
    bool verify = false;
    bool debug_figures = false; // bz_gpu_profile_enable(g, 2 | ..): k_huff_sweep sums the per-pass figures of bz_gpu_debug_block_sections
    u64 verify_stats[4] = {0, 0, 0, 0}; // since creation: blocks checked, calls redone, redone calls that failed again, ns
    DevBuf vstream, vout, vseg, vmis;
    DecWorkspace *dec = nullptr; // decode workspace, created by the first decode call
    DevBuf dec_in, oneshot_out;  // bz_decode_buffer / df_encode_buffer: the caller's bytes on the device and the result (kept with the cached engine)
    DfWorkspace *df = nullptr;   // Deflate encode workspace, created by the first df_gpu_encode_device call
};

