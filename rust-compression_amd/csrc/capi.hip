// capi.hip -- section 1 of the C ABI: the streaming encoder context (== BZip2Encoder) and the
// one-shot host-buffer call, built on the device engine (engine.hip).
//
// The context replays the control flow of the reference byte iterator
// (src/bzip2/encoder.rs:74-159 BZip2Encoder::{next_bits,next}, :718-739 EncoderInner::{flush,
// finish}, :224-291 write_block) at bulk granularity:
//   * bz_enc_write  == the input iterator yields bytes        (:80-85)
//   * bz_enc_end(a) == the input iterator returns None, and `Encoder::next` is polled until it
//                      returns None                            (:86-110, :129-146)
// State carried between calls is exactly the reference's: pending input (the reference keeps
// it as rle_buffer/rle_count + block_buf, here: the raw bytes since the last block cut), the
// combined CRC, "has a block been written" (block_no == 1, :179,245), the BitWriter carry
// (src/bitio/writer.rs:165-169) and the finished / bit_finished toggles (:45-48).
// Quirks of the reference are kept on purpose (they are observable in the bytes): Flush zero
// pads mid-stream, an empty write_block still rotates the combined CRC (:237-238) and repeats
// the "BZh" header while block_no == 1 (:245).
#include "../../include/bz2_mi355x.h"
#include "bzgpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace bzgpu;

extern "C" const char *bz_strerror(int code)
{
    switch (code) {
    case BZ_OK: return "ok";
    case BZ_E_DATA: return "data integrity error in data";   // src/error.rs:37
    case BZ_E_EOF: return "file ends unexpectedly";          // src/error.rs:38
    case BZ_E_UNEXPECTED: return "unexpected error";         // src/error.rs:39
    case BZ_E_MAGIC_FIRST: return "bad magic number (first block)";
    case BZ_E_MAGIC: return "bad magic number";
    case BZ_E_PARAM: return "invalid parameter (level must be 1..=9)";
    case BZ_E_NOGPU: return "no usable gfx950 (MI355X) device: this library has no CPU path";
    case BZ_E_NOMEM: return "out of device or host memory";
    case BZ_E_CAPACITY: return "output buffer too small";
    default: return "unknown status";
    }
}

extern "C" const char *bz_version(void) { return "bz2_mi355x 0.1 (gfx950)"; }

struct bz_enc {
    int level = 9;
    int device = 0;
    bz_gpu_engine *g = nullptr;
    std::vector<u8> in;    // input bytes since the last block cut (includes the pending run)
    std::vector<u8> out;   // encoded bytes not yet read
    size_t out_head = 0;
    // reference state
    bool finished = false;       // BZip2Encoder.finished      (encoder.rs:45)
    bool bit_finished = false;   // BZip2Encoder.bit_finished  (encoder.rs:48)
    bool inner_finished = false; // EncoderInner.finished      (encoder.rs:164)
    bool any_block = false;      // block_no > 1               (encoder.rs:168,179)
    u32 combined_crc = 0;        // encoder.rs:167
    unsigned carry_bits = 0;     // BitWriter.counter          (writer.rs:167)
    unsigned carry_byte = 0;     // BitWriter.buf              (writer.rs:166)
    // device staging
    void *d_in = nullptr;
    size_t d_in_cap = 0;
    void *d_out = nullptr;
    size_t d_out_cap = 0;
    void *d_packed = nullptr;
    size_t d_packed_cap = 0;
    size_t threshold = (size_t)64 << 20; // buffered bytes that trigger encoding of complete blocks
};

static int grow(void **p, size_t *cap, size_t want)
{
    if (want <= *cap) return BZ_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t sz = want + want / 4 + 4096;
    if (hipMalloc(p, sz) != hipSuccess) return BZ_E_NOMEM;
    *cap = sz;
    return BZ_OK;
}

extern "C" int bz_enc_create(bz_enc **out, int level, int device)
{
    if (!out) return BZ_E_PARAM;
    *out = nullptr;
    if (level < 1 || level > 9) return BZ_E_PARAM; // the reference panics (encoder.rs:59-61)
    bz_enc *e = new bz_enc();
    e->level = level;
    e->device = device;
    *out = e;
    return BZ_OK;
}

extern "C" void bz_enc_destroy(bz_enc *e)
{
    if (!e) return;
    if (e->g) {
        (void)hipSetDevice(e->device);
        if (e->d_in) (void)hipFree(e->d_in);
        if (e->d_out) (void)hipFree(e->d_out);
        if (e->d_packed) (void)hipFree(e->d_packed);
        bz_gpu_engine_destroy(e->g);
    }
    delete e;
}

// start of the pending chunk: the run that is still open at the end of the buffer, cut
// every 255 bytes from its start (encoder.rs:676-690)
static size_t pending_chunk_start(const std::vector<u8> &in)
{
    const size_t n = in.size();
    if (n == 0) return 0;
    size_t rs = n - 1;
    const u8 b = in[n - 1];
    while (rs > 0 && in[rs - 1] == b) --rs;
    const size_t q = n - 1 - rs;
    return n - 1 - (q % 255);
}

static inline u32 rotl1(u32 v) { return (v << 1) | (v >> 31); }

// One bulk replay of the write_block calls the reference would have made for `mode`.
static int process(bz_enc *e, int mode)
{
    const size_t n_all = e->in.size();
    const size_t n_eff = (mode == BZ_ACTION_FINISH) ? n_all : pending_chunk_start(e->in);
    size_t n_blocks = 0, consumed = 0;
    int tail = 0;
    std::vector<uint64_t> woff, blen;
    std::vector<uint32_t> crc;
    int rc;
    if (n_eff > 0) {
        if (!e->g) {
            rc = bz_gpu_engine_create(&e->g, e->device, 80);
            if (rc != BZ_OK) return rc;
        }
        if (hipSetDevice(e->device) != hipSuccess) return BZ_E_UNEXPECTED;
        if ((rc = grow(&e->d_in, &e->d_in_cap, n_eff + 64)) != BZ_OK) return rc;
        if (hipMemcpy(e->d_in, e->in.data(), n_eff, hipMemcpyHostToDevice) != hipSuccess) return BZ_E_UNEXPECTED;
        rc = bz_gpu_partition(e->g, e->level, e->d_in, n_eff, mode, &n_blocks, &consumed, &tail);
        if (rc != BZ_OK) return rc;
        if (n_blocks) {
            woff.resize(n_blocks);
            blen.resize(n_blocks);
            crc.resize(n_blocks);
            const size_t cap_words = bz_encode_bound(n_eff) / 4 + 2 * n_blocks + 16;
            if ((rc = grow(&e->d_packed, &e->d_packed_cap, cap_words * 4)) != BZ_OK) return rc;
            size_t used = 0;
            rc = bz_gpu_encode_blocks(e->g, 0, 1, e->d_packed, cap_words, woff.data(), blen.data(), crc.data(),
                                      &used);
            if (rc != BZ_OK) return rc;
        }
    }
    if (mode == BZ_ACTION_RUN && n_blocks == 0) return BZ_OK; // no write_block call happened

    // The flush()/finish() call itself sees an empty block_buf when every byte went into
    // blocks closed by a cut (or there was nothing at all).
    const bool final_call_empty =
        (mode == BZ_ACTION_FLUSH && !(n_blocks > 0 && tail)) || (mode == BZ_ACTION_FINISH && n_blocks == 0);
    u32 comb = e->combined_crc;
    int write_header = e->any_block ? 0 : 1; // block_no == 1 (encoder.rs:245)
    if (final_call_empty && n_blocks == 0) comb = rotl1(comb) ^ 0u; // encoder.rs:237-238, crc of nothing = 0
    const int trailer = (mode == BZ_ACTION_FINISH) ? 1 : 0;

    size_t bits_bound = 0;
    for (size_t k = 0; k < n_blocks; ++k) bits_bound += (size_t)blen[k];
    const size_t out_cap = bits_bound / 8 + 64;
    if (!e->g) { // nothing was ever encoded: frame bits only
        rc = bz_gpu_engine_create(&e->g, e->device, 80);
        if (rc != BZ_OK) return rc;
    }
    if ((rc = grow(&e->d_out, &e->d_out_cap, out_cap)) != BZ_OK) return rc;
    size_t out_len = 0;
    unsigned ocb = 0, ocy = 0;
    u32 comb_out = comb;
    rc = bz_gpu_assemble(e->g, e->level, n_blocks, e->d_packed, woff.data(), blen.data(), crc.data(), write_header,
                         trailer, 0, e->carry_bits, e->carry_byte, comb, &comb_out, e->d_out, e->d_out_cap, &out_len,
                         &ocb, &ocy);
    if (rc != BZ_OK) return rc;
    if (out_len) {
        const size_t old = e->out.size();
        e->out.resize(old + out_len);
        if (hipMemcpy(e->out.data() + old, e->d_out, out_len, hipMemcpyDeviceToHost) != hipSuccess)
            return BZ_E_UNEXPECTED;
    }
    e->carry_bits = ocb;
    e->carry_byte = ocy;
    if (final_call_empty && n_blocks > 0) comb_out = rotl1(comb_out); // the extra, empty write_block(false)
    e->combined_crc = comb_out;
    if (n_blocks > 0) e->any_block = true;
    // drop the bytes that went into blocks
    size_t drop = (mode == BZ_ACTION_RUN) ? consumed : n_eff;
    if (mode == BZ_ACTION_FINISH) drop = n_all;
    if (drop) e->in.erase(e->in.begin(), e->in.begin() + (ptrdiff_t)drop);
    if (mode == BZ_ACTION_FINISH) e->inner_finished = true;
    return BZ_OK;
}

extern "C" int bz_enc_write(bz_enc *e, const uint8_t *in, size_t n)
{
    if (!e || (!in && n)) return BZ_E_PARAM;
    e->in.insert(e->in.end(), in, in + n);
    if (!e->inner_finished && e->in.size() >= e->threshold) return process(e, BZ_ACTION_RUN);
    return BZ_OK;
}

extern "C" int bz_enc_end(bz_enc *e, int action)
{
    if (!e || action < BZ_ACTION_RUN || action > BZ_ACTION_FINISH) return BZ_E_PARAM;
    int rc;
    // blocks the reference would already have emitted while it was consuming the input
    if (!e->inner_finished && action == BZ_ACTION_RUN) {
        if ((rc = process(e, BZ_ACTION_RUN)) != BZ_OK) return rc;
    }
    for (;;) {
        // next_bits: queue empty and the iterator is exhausted (encoder.rs:86-110)
        if (!e->finished) {
            if (action == BZ_ACTION_FLUSH && !e->inner_finished) {       // :718-727
                if ((rc = process(e, BZ_ACTION_FLUSH)) != BZ_OK) return rc;
            } else if (action == BZ_ACTION_FINISH && !e->inner_finished) { // :729-739
                if ((rc = process(e, BZ_ACTION_FINISH)) != BZ_OK) return rc;
            }
            e->finished = true;
        }
        e->finished = false; // ... and the next poll returns None (:87-89)
        // Encoder::next saw None from next_bits (:129-146)
        if (e->bit_finished) {
            e->bit_finished = false;
            break;
        }
        if (action == BZ_ACTION_RUN) break;
        e->bit_finished = true;
        if (e->carry_bits == 0) break; // writer.flush() -> None (writer.rs:226-242)
        e->out.push_back((u8)e->carry_byte);
        e->carry_bits = 0;
        e->carry_byte = 0;
    }
    return BZ_OK;
}

extern "C" size_t bz_enc_pending(const bz_enc *e) { return e ? e->out.size() - e->out_head : 0; }

extern "C" long bz_enc_read(bz_enc *e, uint8_t *out, size_t cap)
{
    if (!e || (!out && cap)) return BZ_E_PARAM;
    const size_t avail = e->out.size() - e->out_head;
    const size_t k = avail < cap ? avail : cap;
    if (k) memcpy(out, e->out.data() + e->out_head, k);
    e->out_head += k;
    if (e->out_head == e->out.size()) {
        e->out.clear();
        e->out_head = 0;
    }
    return (long)k;
}

extern "C" void bz_free(void *p) { free(p); }

extern "C" int bz_encode_buffer(int level, int device, const uint8_t *in, size_t in_len, uint8_t **out,
                                size_t *out_len)
{
    if (!out || !out_len || (!in && in_len)) return BZ_E_PARAM;
    *out = nullptr;
    *out_len = 0;
    if (level < 1 || level > 9) return BZ_E_PARAM;
    const size_t est_blocks = in_len / 700000 + 2;
    bz_gpu_engine *g = nullptr;
    int rc = bz_gpu_engine_create(&g, device, est_blocks < 256 ? est_blocks : 256);
    if (rc != BZ_OK) return rc;
    void *d_in = nullptr, *d_out = nullptr;
    const size_t cap = (bz_encode_bound(in_len) + 15) & ~(size_t)15;
    uint8_t *h = nullptr;
    size_t n_out = 0;
    rc = BZ_E_NOMEM;
    if (hipMalloc(&d_in, in_len + 64) != hipSuccess) goto done;
    if (hipMalloc(&d_out, cap) != hipSuccess) goto done;
    rc = BZ_E_UNEXPECTED;
    if (in_len && hipMemcpy(d_in, in, in_len, hipMemcpyHostToDevice) != hipSuccess) goto done;
    rc = bz_gpu_encode_device(g, level, d_in, in_len, d_out, cap, &n_out);
    if (rc != BZ_OK) goto done;
    h = (uint8_t *)malloc(n_out ? n_out : 1);
    if (!h) {
        rc = BZ_E_NOMEM;
        goto done;
    }
    if (hipMemcpy(h, d_out, n_out, hipMemcpyDeviceToHost) != hipSuccess) {
        free(h);
        rc = BZ_E_UNEXPECTED;
        goto done;
    }
    *out = h;
    *out_len = n_out;
    rc = BZ_OK;
done:
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    bz_gpu_engine_destroy(g);
    return rc;
}
