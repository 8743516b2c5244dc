// stub_engine.cpp -- a stand-in for the device engine (csrc/engine.hip) behind the host pipeline of csrc/capi.hip, for
// the sanitizer runs of tests/test_host_pipeline_sanitize.py.  Same entry points, same contracts (include/bz2_mi355x.h
// section 2), no compression: a "block" is up to 1000 * level - 19 input bytes, its bit string is
//     magic 0xB10C (16 bits) | length (24) | checksum (32) | the bytes | 101 (3 bits)
// -- an odd number of bits, so the BitWriter carry between jobs is exercised -- and a stream is "BZh<level>" + blocks +
// the end-of-stream record with the combined checksum, exactly the framing bz_gpu_assemble writes.  The work of a call
// runs on the engine's own (shim) stream, the call returns when that stream has drained: what the real engine does.
// Test infrastructure only.
#include "../../include/bz2_mi355x.h"
#include "hip_shim.h"

#include <algorithm>
#include <cstdio>
#include <vector>

struct StubBlock {
    uint64_t in_off, len;
    uint32_t crc;
};
struct bz_gpu_engine {
    int device = 0;
    hipStream_t st = nullptr;
    int level = 9;
    const uint8_t *d_in = nullptr;
    std::vector<StubBlock> blocks;
    int verify = 0;
};

static uint32_t stub_crc(const uint8_t *p, uint64_t n)
{
    uint32_t h = 2166136261u;
    for (uint64_t i = 0; i < n; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}

extern "C" size_t bz_encode_bound(size_t n) { return n + n / 64 + (n / 900 + 2) * 16 + 64; }

extern "C" int bz_gpu_engine_create(bz_gpu_engine **out, int device, size_t)
{
    if (!out) return BZ_E_PARAM;
    *out = nullptr;
    if (device < 0 || device >= hipshim::kDevices) return BZ_E_PARAM;
    bz_gpu_engine *g = new bz_gpu_engine();
    g->device = device;
    (void)hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking);
    *out = g;
    return BZ_OK;
}
extern "C" void bz_gpu_engine_destroy(bz_gpu_engine *g)
{
    if (!g) return;
    (void)hipStreamSynchronize(g->st);
    (void)hipStreamDestroy(g->st);
    delete g;
}
extern "C" int bz_gpu_engine_reserve(bz_gpu_engine *g, size_t) { return g ? BZ_OK : BZ_E_PARAM; }
extern "C" int bz_gpu_engine_set_verify(bz_gpu_engine *g, int on)
{
    if (!g) return BZ_E_PARAM;
    g->verify = on;
    return BZ_OK;
}
extern "C" int bz_gpu_verify_stats(bz_gpu_engine *g, uint64_t out[4])
{
    if (!g || !out) return BZ_E_PARAM;
    out[0] = out[1] = out[2] = out[3] = 0;
    return BZ_OK;
}

extern "C" int bz_gpu_partition(bz_gpu_engine *g, int level, const void *d_in, size_t n, int mode, size_t *n_blocks,
                                size_t *consumed, int *tail_block)
{
    if (!g || level < 1 || level > 9) return BZ_E_PARAM;
    g->level = level;
    g->d_in = static_cast<const uint8_t *>(d_in);
    g->blocks.clear();
    const uint64_t B = 1000u * (uint64_t)level - 19u;
    std::vector<StubBlock> *dst = &g->blocks;
    const uint8_t *in = g->d_in;
    int tail = 0;
    uint64_t used = 0;
    g->st->push([=, &tail, &used] { // (on the engine's stream: the input must be complete by now, the caller says)
        // The reference collects CHUNKS -- a run of up to 255 equal bytes -- and closes a block when the next chunk would
        // not fit (src/bzip2/encoder.rs:671-697): a cut is always a chunk start, which is what lets the host pipeline
        // restart the run count at a cut.  The stub keeps that rule (its "block length" is input bytes, not RLE1 bytes).
        uint64_t pos = 0, blk = 0; // start of the chunk at hand, start of the block being filled
        while (pos < n) {
            uint64_t e = pos + 1;
            while (e < n && e - pos < 255 && in[e] == in[pos]) ++e;
            if (e - blk > B && pos > blk) { // the chunk does not fit: the block ends in front of it
                dst->push_back({blk, pos - blk, stub_crc(in + blk, pos - blk)});
                blk = pos;
            }
            pos = e;
        }
        if (mode != BZ_ACTION_RUN && blk < n) {
            dst->push_back({blk, n - blk, stub_crc(in + blk, n - blk)});
            blk = n;
            tail = 1;
        }
        used = blk;
    });
    (void)hipStreamSynchronize(g->st);
    if (n_blocks) *n_blocks = g->blocks.size();
    if (consumed) *consumed = (size_t)used;
    if (tail_block) *tail_block = tail;
    return BZ_OK;
}

namespace {
struct Bits { // MSB-first writer into host-endian u32 words whose bit 31 is the earliest bit
    uint32_t *w;
    uint64_t pos;
    void put(uint64_t v, unsigned nbits)
    {
        for (unsigned i = nbits; i-- > 0;) {
            if ((v >> i) & 1u) w[pos >> 5] |= 0x80000000u >> (pos & 31u);
            ++pos;
        }
    }
};
} // namespace

extern "C" int bz_gpu_encode_blocks(bz_gpu_engine *g, size_t first, size_t stride, void *d_packed, size_t cap_words,
                                    uint64_t *h_word_off, uint64_t *h_bit_len, uint32_t *h_crc, size_t *words_used)
{
    if (!g || stride == 0) return BZ_E_PARAM;
    if (words_used) *words_used = 0;
    uint64_t cursor = 0;
    size_t k = 0;
    for (size_t b = first; b < g->blocks.size(); b += stride, ++k) {
        const StubBlock &sb = g->blocks[b];
        const uint64_t bits = 16 + 24 + 32 + 8 * sb.len + 3;
        h_word_off[k] = cursor;
        h_bit_len[k] = bits;
        h_crc[k] = sb.crc;
        cursor += (bits + 31) / 32;
    }
    if (cursor > cap_words) return BZ_E_CAPACITY;
    const size_t count = k;
    uint32_t *words = static_cast<uint32_t *>(d_packed);
    const uint8_t *in = g->d_in;
    const std::vector<StubBlock> *blocks = &g->blocks;
    g->st->push([=] {
        for (size_t q = 0; q < count; ++q) {
            const StubBlock &sb = (*blocks)[first + q * stride];
            const uint64_t nw = (h_bit_len[q] + 31) / 32;
            memset(words + h_word_off[q], 0, nw * 4);
            Bits bw{words + h_word_off[q], 0};
            bw.put(0xB10C, 16);
            bw.put(sb.len, 24);
            bw.put(sb.crc, 32);
            for (uint64_t i = 0; i < sb.len; ++i) bw.put(in[sb.in_off + i], 8);
            bw.put(5, 3);
        }
    });
    (void)hipStreamSynchronize(g->st);
    if (words_used) *words_used = (size_t)cursor;
    return BZ_OK;
}

extern "C" int bz_gpu_assemble(bz_gpu_engine *g, int level, size_t n_blocks, const void *d_packed, const uint64_t *h_word_off,
                               const uint64_t *h_bit_len, const uint32_t *h_crc, int write_header, int write_trailer,
                               int pad_to_byte, unsigned carry_bits, unsigned carry_byte, uint32_t combined_crc_in,
                               uint32_t *combined_crc_out, void *d_out, size_t cap, size_t *out_len,
                               unsigned *out_carry_bits, unsigned *out_carry_byte)
{
    if (!g || level < 1 || level > 9 || carry_bits > 7) return BZ_E_PARAM;
    uint64_t total = carry_bits + (write_header ? 32u : 0u);
    uint32_t comb = combined_crc_in;
    for (size_t k = 0; k < n_blocks; ++k) {
        total += h_bit_len[k];
        comb = ((comb << 1) | (comb >> 31)) ^ h_crc[k];
    }
    if (write_trailer) total += 80;
    const uint64_t nwords = (total + 31) / 32;
    if (nwords * 4 > cap) return BZ_E_CAPACITY;
    uint8_t *out = static_cast<uint8_t *>(d_out);
    const uint32_t *packed = static_cast<const uint32_t *>(d_packed);
    g->st->push([=] {
        memset(out, 0, nwords * 4);
        uint64_t pos = 0;
        auto put = [&](uint64_t v, unsigned nbits) { // the OUTPUT is a byte string, most significant bit first
            for (unsigned i = nbits; i-- > 0;) {
                if ((v >> i) & 1u) out[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7u));
                ++pos;
            }
        };
        put(carry_byte >> (8 - (carry_bits ? carry_bits : 8)), carry_bits);
        if (write_header) put(0x425A6830u + (uint32_t)level, 32);
        for (size_t k = 0; k < n_blocks; ++k)
            for (uint64_t b = 0; b < h_bit_len[k]; ++b) {
                const uint32_t w = packed[h_word_off[k] + (b >> 5)];
                put((w >> (31 - (b & 31))) & 1u, 1);
            }
        if (write_trailer) {
            put(0x177245385090ull, 48);
            put(comb, 32);
        }
    });
    (void)hipStreamSynchronize(g->st);
    size_t bytes;
    unsigned ocb = 0, ocy = 0;
    if (pad_to_byte) {
        bytes = (size_t)((total + 7) / 8);
    } else {
        bytes = (size_t)(total / 8);
        ocb = (unsigned)(total & 7u);
        if (ocb) ocy = out[bytes] & (0xFFu << (8 - ocb));
    }
    if (combined_crc_out) *combined_crc_out = comb;
    if (out_len) *out_len = bytes;
    if (out_carry_bits) *out_carry_bits = ocb;
    if (out_carry_byte) *out_carry_byte = ocy;
    return BZ_OK;
}
