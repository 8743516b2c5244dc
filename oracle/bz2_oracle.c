/* ==========================================================================
 * TEST INFRASTRUCTURE -- CPU oracle for the BZip2 block-encode hot path.
 *
 * This file is a plain-C restatement of the reference's algorithm
 * (chalharu/rust-compression 0.1.5, /root/reference).  It exists so that the
 * HIP product path can be checked bit-for-bit on a CPU.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (rust-compression_amd/csrc) never links or calls it.
 *
 * PARITY PINNING.  The reference is Rust and no Rust toolchain exists in the
 * build container, so the reference itself cannot be executed.  This oracle is
 * pinned by (tests/test_oracle_*.py):
 *   - every known-answer vector the reference's own tests hold for this path:
 *       the 39-byte `b"a\n"` level-9 stream      (src/bzip2/mod.rs:41-58)
 *       all BWT vectors                           (src/suffix_array/sais.rs:294-556)
 *       L/S/LMS classification                    (src/suffix_array/ls_type.rs:98-145)
 *       code-length vectors                       (src/huffman/cano_huff_table.rs:237-294)
 *       canonical-code vectors                    (src/huffman/encoder.rs:63-189)
 *       MSB-first bit-writer vectors              (src/bitio/writer.rs:253-444)
 *   - a differential check against system libbzip2 1.0.8: with ONLY the
 *     code-length builder swapped for libbzip2's published hbMakeCodeLengths
 *     (huffman_mode = 1) the stream must equal Python's bz2.compress byte for
 *     byte, which pins RLE1, block splitting, CRC, rotation order, MTF/ZLE,
 *     table selection, header layout and bit packing;
 *   - decodability of every produced stream by libbzip2.
 * Encoder output on data/sample[1-4].ref at level 9 is NOT pinned by any
 * reference test ("parity unpinned" beyond the vectors above).
 *
 * Each function cites the reference file:line it follows.
 * ========================================================================== */

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BZO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------
 * CRC-32/BZIP2  (src/crc32.rs:27-38, 58-72, 80-84, 128-149)
 * poly 0x04C11DB7, MSB first, init 0xFFFFFFFF, final NOT.
 * ---------------------------------------------------------------------- */
static uint32_t g_crc_table[256];
static int g_crc_ready = 0;

/* crc32.rs:58-72 make_table_normal */
static void crc_make_table_normal(uint32_t poly)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t value = i << 24;
        for (int k = 0; k < 8; k++)
            value = (value & 0x80000000u) ? ((value << 1) ^ poly) : (value << 1);
        g_crc_table[i] = value;
    }
    g_crc_ready = 1;
}

/* crc32.rs:80-84 update_normal */
static inline uint32_t crc_update_normal(uint32_t value, uint8_t byte)
{
    return g_crc_table[(uint8_t)(value >> 24) ^ byte] ^ (value << 8);
}

/* Digest: build_hasher -> 0xFFFFFFFF (crc32.rs:100-109), finish -> !value (:128-131) */
BZO_EXPORT uint32_t bzo_crc32_bzip2(const uint8_t *p, size_t n)
{
    if (!g_crc_ready) crc_make_table_normal(0x04C11DB7u);
    uint32_t v = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) v = crc_update_normal(v, p[i]);
    return ~v;
}

/* ------------------------------------------------------------------------
 * BWT = order of all cyclic rotations (src/suffix_array/sais.rs)
 * ---------------------------------------------------------------------- */

/* sais.rs:12-68 array_rotate_for_non_sentinel_bwt: start of the least rotation */
static size_t array_rotate_for_non_sentinel_bwt(const uint8_t *array, size_t count,
                                                size_t *sarray, size_t bucket_max)
{
    size_t n1 = 0;
    size_t val = bucket_max + 1;
    size_t prev_pos = 0;
    for (size_t i = 0; i < count; i++) {
        size_t j = array[i];
        if (val > j) {
            sarray[0] = i;
            val = j;
            n1 = 1;
            prev_pos = i;
        } else if (val == j) {
            prev_pos += 1;
            if (prev_pos != i) {
                sarray[n1] = i;
                n1 += 1;
            }
        }
    }

    for (size_t i = 0; i < count; i++) {
        size_t n2 = 0;
        val = bucket_max + 1;
        for (size_t j = 0; j < n1; j++) {
            size_t k = sarray[j] + 1;
            if (k >= count) k -= count;
            size_t l = array[k];
            if (val == l) {
                sarray[n2] = k;
                n2 += 1;
            } else if (val > l) {
                sarray[0] = k;
                val = l;
                n2 = 1;
            }
        }
        if (n2 == 1) {
            return (sarray[0] <= i) ? sarray[0] + count - i - 1 : sarray[0] - i - 1;
        }
        n1 = n2;
    }
    return sarray[0];
}

#define SA_T size_t
#define SA_NAME(x) sais_usize_##x
#include "sais_template.inc"

#define SA_T uint8_t
#define SA_NAME(x) sais_u8_##x
#include "sais_template.inc"

/* sais.rs:266-272 bwt(): returns rotation start indices in sorted order */
BZO_EXPORT void bzo_bwt(const uint8_t *array, size_t count, size_t *sa)
{
    if (count == 0) return;
    for (size_t i = 0; i < count; i++) sa[i] = 0;
    size_t shift = array_rotate_for_non_sentinel_bwt(array, count, sa, 255);
    sais_u8_sa_is(array, count, sa, 0, 255, shift);
}

/* exposed for the ls_type.rs:98-145 vectors */
BZO_EXPORT void bzo_ls_types(const uint8_t *array, size_t count, size_t shift,
                             uint8_t *types, uint8_t *lms)
{
    sais_u8_ls_types(array, count, shift, types, lms);
}

/* exposed so tests can check the F4 tie rule's `shift` */
BZO_EXPORT size_t bzo_bwt_shift(const uint8_t *array, size_t count)
{
    if (count == 0) return 0;
    size_t *tmp = (size_t *)calloc(count, sizeof(size_t));
    size_t s = array_rotate_for_non_sentinel_bwt(array, count, tmp, 255);
    free(tmp);
    return s;
}

/* ------------------------------------------------------------------------
 * Huffman code lengths (src/huffman/cano_huff_table.rs)
 * weight_add_fn is selected by `mode`:
 *   0: |x, y| x + y                                (cano_huff_table.rs:228-230)
 *   1: bzip2's depth-tagged combine                (bzip2/encoder.rs:647-650)
 * ---------------------------------------------------------------------- */
static inline size_t weight_add(int mode, size_t x, size_t y)
{
    if (mode == 0) return x + y;
    size_t dx = x & 0xFF, dy = y & 0xFF;
    return ((x & 0xFFFFFF00u) + (y & 0xFFFFFF00u)) | (1 + (dx > dy ? dx : dy));
}

/* cano_huff_table.rs:14-31 */
static void down_heap(size_t *buf, size_t n, size_t len)
{
    size_t tmp = buf[n];
    size_t leaf = (n << 1) + 1;
    while (leaf < len) {
        if (leaf + 1 < len && buf[buf[leaf]] > buf[buf[leaf + 1]]) leaf += 1;
        if (buf[tmp] < buf[buf[leaf]]) break;
        buf[n] = buf[leaf];
        n = leaf;
        leaf = (n << 1) + 1;
    }
    buf[n] = tmp;
}

/* cano_huff_table.rs:33-38 (buf has 2*s entries) */
static void create_heap(size_t *buf, size_t buflen)
{
    size_t s = buflen >> 1;
    for (size_t i = (s >> 1); i-- > 0;) down_heap(buf, i, s);
}

/* cano_huff_table.rs:40-55 */
static void take_package(size_t **ty, size_t *len, size_t nlen, size_t *cur, size_t i)
{
    size_t x = ty[i][cur[i]];
    if (x == nlen) {
        take_package(ty, len, nlen, cur, i + 1);
        take_package(ty, len, nlen, cur, i + 1);
    } else {
        len[x] -= 1;
    }
    cur[i] += 1;
}

/* stable insertion-free merge sort on (index,freq) by freq DESCENDING
 * (cano_huff_table.rs:64-70: Vec::sort_by is a stable sort) */
typedef struct { size_t idx, f; } freq_ent;
static void stable_sort_desc(freq_ent *a, freq_ent *tmp, size_t n)
{
    if (n < 2) return;
    size_t h = n / 2;
    stable_sort_desc(a, tmp, h);
    stable_sort_desc(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take right only when strictly greater -> stability */
        if (a[j].f > a[i].f) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(freq_ent));
}

/* cano_huff_table.rs:58-151 gen_code_lm ("reverse package merge") */
static void gen_code_lm(const size_t *freq, size_t len, size_t lim, int mode, uint8_t *out)
{
    freq_ent *fm = (freq_ent *)malloc(len * sizeof(freq_ent));
    freq_ent *tmp = (freq_ent *)malloc(len * sizeof(freq_ent));
    for (size_t i = 0; i < len; i++) { fm[i].idx = i; fm[i].f = freq[i]; }
    stable_sort_desc(fm, tmp, len);
    free(tmp);
    size_t *map = (size_t *)malloc(len * sizeof(size_t));
    size_t *sfreq = (size_t *)malloc(len * sizeof(size_t));
    for (size_t i = 0; i < len; i++) { map[i] = fm[i].idx; sfreq[i] = fm[i].f; }
    free(fm);

    size_t *max_elem = (size_t *)calloc(lim, sizeof(size_t));
    size_t *b = (size_t *)calloc(lim, sizeof(size_t));

    size_t excess = ((size_t)1 << lim) - len;      /* :75 */
    size_t half = (size_t)1 << (lim - 1);          /* :76 */
    max_elem[lim - 1] = len;                       /* :77 */

    for (size_t j = 0; j < lim; j++) {             /* :79-88 */
        if (excess >= half) {
            b[j] = 1;
            excess -= half;
        }
        excess <<= 1;
        if (lim >= 2 + j) max_elem[lim - 2 - j] = max_elem[lim - 1 - j] / 2 + len;
    }

    max_elem[0] = b[0];                            /* :90-95 */
    for (size_t j = 1; j < lim; j++) {
        if (max_elem[j] > 2 * max_elem[j - 1] + b[j]) max_elem[j] = 2 * max_elem[j - 1] + b[j];
    }

    size_t **val = (size_t **)malloc(lim * sizeof(size_t *));
    size_t **ty = (size_t **)malloc(lim * sizeof(size_t *));
    for (size_t i = 0; i < lim; i++) {             /* :97-98 */
        val[i] = (size_t *)calloc(max_elem[i] ? max_elem[i] : 1, sizeof(size_t));
        ty[i] = (size_t *)calloc(max_elem[i] ? max_elem[i] : 1, sizeof(size_t));
    }
    size_t *c = (size_t *)malloc(len * sizeof(size_t));
    for (size_t i = 0; i < len; i++) c[i] = lim;   /* :99 */

    for (size_t t = 0; t < len && t < max_elem[lim - 1]; t++) { /* :101-104 */
        val[lim - 1][t] = sfreq[t];
        ty[lim - 1][t] = t;
    }

    size_t *cur = (size_t *)calloc(lim, sizeof(size_t));
    if (b[lim - 1] == 1) {                         /* :107-110 */
        c[0] -= 1;
        cur[lim - 1] += 1;
    }

    size_t j = lim - 1;
    while (j > 0) {                                /* :112-142 */
        size_t i = 0;
        size_t next = cur[j];
        for (size_t t = 0; t < max_elem[j - 1]; t++) {
            size_t weight = (next + 1 < max_elem[j])
                                ? weight_add(mode, val[j][next], val[j][next + 1])
                                : 0;
            if (weight > sfreq[i]) {
                val[j - 1][t] = weight;
                ty[j - 1][t] = len;
                next += 2;
            } else {
                val[j - 1][t] = sfreq[i];
                ty[j - 1][t] = i;
                i += 1;
                if (i >= len) break;
            }
        }
        j -= 1;
        cur[j] = 0;
        if (b[j] == 1) take_package(ty, c, len, cur, j);
    }

    for (size_t i = 0; i < len; i++) out[map[i]] = (uint8_t)c[i]; /* :144-150 */

    for (size_t i = 0; i < lim; i++) { free(val[i]); free(ty[i]); }
    free(val); free(ty); free(c); free(cur); free(b); free(max_elem);
    free(map); free(sfreq);
}

/* cano_huff_table.rs:153-196 gen_code */
static int gen_code(const size_t *freq, size_t n, size_t lim, int mode, uint8_t *out)
{
    if (n == 1) {
        out[0] = 1;
        return 0;
    }
    size_t *buf = (size_t *)malloc(2 * n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) buf[i] = n + i;
    for (size_t i = 0; i < n; i++) buf[n + i] = freq[i];

    create_heap(buf, 2 * n);

    for (size_t i = n - 1; i >= 1; i--) {          /* :168-178 */
        size_t m1 = buf[0];
        buf[0] = buf[i];
        down_heap(buf, 0, i);
        size_t m2 = buf[0];
        buf[i] = weight_add(mode, buf[m1], buf[m2]);
        buf[0] = i;
        buf[m1] = i;
        buf[m2] = i;
        down_heap(buf, 0, i);
    }

    buf[1] = 0;                                    /* :181-184 */
    for (size_t i = 2; i < n; i++) buf[i] = buf[buf[i]] + 1;

    int too_long = 0;
    for (size_t i = 0; i < n; i++) {               /* :186-188 */
        out[i] = (uint8_t)(buf[buf[i + n]] + 1);
        if ((size_t)out[i] > lim) too_long = 1;
    }
    free(buf);
    if (too_long) {                                /* :190-194 */
        gen_code_lm(freq, n, lim, mode, out);
        return 1;
    }
    return 0;
}

/* cano_huff_table.rs:198-225 make_tab_with_fn.  Returns 1 when the
 * length-limited path (gen_code_lm) was taken, 0 otherwise.
 * NOTE: the reference returns a Vec truncated after the last non-zero
 * frequency; `*out_n` receives that length, out[] must hold n entries. */
BZO_EXPORT int bzo_make_tab_with_fn(const size_t *freq, size_t n, size_t lim, int mode,
                                    uint8_t *out, size_t *out_n)
{
    size_t *s = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    size_t *l = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (freq[i] != 0) { s[k] = i; l[k] = freq[i]; k++; }
    }
    int lm = 0;
    size_t outlen = 0;
    if (k > 0) {
        uint8_t *v = (uint8_t *)malloc(k);
        lm = gen_code(l, k, lim, mode, v);
        size_t c = 0;
        for (size_t t = 0; t < k; t++) {
            while (c < s[t]) out[c++] = 0;
            out[c++] = v[t];
        }
        outlen = c;
        free(v);
    }
    if (out_n) *out_n = outlen;
    free(s); free(l);
    return lm;
}

/* libbzip2 1.0.8 huffman.c BZ2_hbMakeCodeLengths, restated from the published
 * algorithm.  Used ONLY by huffman_mode 1 (the differential check against the
 * system libbzip2); never part of a parity claim by itself. */
static void libbz2_hb_make_code_lengths(uint8_t *len, const int32_t *freq,
                                        int32_t alphaSize, int32_t maxLen)
{
    enum { MAXA = 258 };
    int32_t heap[MAXA + 2], weight[MAXA * 2], parent[MAXA * 2];
    int32_t nNodes, nHeap, n1 = 0, n2 = 0, i, j, k;
    for (i = 0; i < alphaSize; i++) weight[i + 1] = (freq[i] == 0 ? 1 : freq[i]) << 8;
    for (;;) {
        nNodes = alphaSize;
        nHeap = 0;
        heap[0] = 0; weight[0] = 0; parent[0] = -2;
        for (i = 1; i <= alphaSize; i++) {
            parent[i] = -1;
            nHeap++;
            heap[nHeap] = i;
            { int32_t zz = nHeap, tmp = heap[zz];
              while (weight[tmp] < weight[heap[zz >> 1]]) { heap[zz] = heap[zz >> 1]; zz >>= 1; }
              heap[zz] = tmp; }
        }
        while (nHeap > 1) {
            for (int rep = 0; rep < 2; rep++) {
                int32_t top = heap[1];
                if (rep == 0) n1 = top; else n2 = top;
                heap[1] = heap[nHeap];
                nHeap--;
                { int32_t zz = 1, yy, tmp = heap[zz];
                  for (;;) {
                      yy = zz << 1;
                      if (yy > nHeap) break;
                      if (yy < nHeap && weight[heap[yy + 1]] < weight[heap[yy]]) yy++;
                      if (weight[tmp] < weight[heap[yy]]) break;
                      heap[zz] = heap[yy];
                      zz = yy;
                  }
                  heap[zz] = tmp; }
            }
            nNodes++;
            parent[n1] = parent[n2] = nNodes;
            {
                uint32_t w1 = (uint32_t)weight[n1], w2 = (uint32_t)weight[n2];
                uint32_t d1 = w1 & 0xff, d2 = w2 & 0xff;
                weight[nNodes] = (int32_t)(((w1 & 0xffffff00u) + (w2 & 0xffffff00u)) |
                                           (1 + (d1 > d2 ? d1 : d2)));
            }
            parent[nNodes] = -1;
            nHeap++;
            heap[nHeap] = nNodes;
            { int32_t zz = nHeap, tmp = heap[zz];
              while (weight[tmp] < weight[heap[zz >> 1]]) { heap[zz] = heap[zz >> 1]; zz >>= 1; }
              heap[zz] = tmp; }
        }
        int tooLong = 0;
        for (i = 1; i <= alphaSize; i++) {
            j = 0; k = i;
            while (parent[k] >= 0) { k = parent[k]; j++; }
            len[i - 1] = (uint8_t)j;
            if (j > maxLen) tooLong = 1;
        }
        if (!tooLong) break;
        for (i = 1; i <= alphaSize; i++) {
            j = weight[i] >> 8;
            j = 1 + (j / 2);
            weight[i] = j << 8;
        }
    }
}

/* ------------------------------------------------------------------------
 * Canonical codes (src/huffman/mod.rs:22-67, src/bucket_sort.rs:45-74,
 * src/huffman/encoder.rs:35-55).  Left direction => no bit reversal.
 * code[] gets the code value, only for len != 0.
 * ---------------------------------------------------------------------- */
BZO_EXPORT void bzo_canonical_codes(const uint8_t *symb_len, size_t n, uint32_t *code)
{
    /* stable sort of the used symbols by length == iterate lengths ascending,
     * symbols ascending inside a length (bucket_sort_all_by_key is stable) */
    uint32_t c_len = 0, c_code = 0;
    for (unsigned l = 1; l <= 255; l++) {
        for (size_t s = 0; s < n; s++) {
            if (symb_len[s] != l) continue;
            uint32_t cd = c_code << ((c_len < l) ? (l - c_len) : 0); /* mod.rs:43 */
            c_len = l;
            c_code = cd + 1;                                         /* mod.rs:44 */
            code[s] = cd;
        }
    }
}

/* ------------------------------------------------------------------------
 * Bit writer (src/bitio/writer.rs:165-243, direction/left.rs:17-62) with T=u32
 * ---------------------------------------------------------------------- */
typedef struct { uint8_t buf; size_t counter; } bit_writer;

static inline uint32_t left_forward(uint32_t v, size_t c) { return c >= 32 ? 0 : v << c; }
static inline uint32_t left_backward(uint32_t v, size_t c) { return c >= 32 ? 0 : v >> c; }
static inline uint32_t left_convert(uint32_t v, size_t src_cap, size_t dst_cap)
{
    return (src_cap > dst_cap) ? v >> (src_cap - dst_cap) : v << (dst_cap - src_cap);
}

/* writer.rs:186-224: returns the word (left aligned) and the number of whole bytes */
static void bw_write_bits(bit_writer *w, uint32_t data, size_t len, uint32_t *word, size_t *nbytes)
{
    if (len == 0) { *word = 0; *nbytes = 0; return; }
    uint32_t d = left_convert(data, len, 32);
    size_t clen = len + w->counter;
    uint32_t wdata = left_convert((uint32_t)w->buf, 8, 32) | left_backward(d, w->counter);
    size_t wlen = clen >> 3;
    w->buf = (uint8_t)left_convert(
        (wlen == 0) ? wdata : left_forward(d, (wlen << 3) - w->counter), 32, 8);
    w->counter = clen - (wlen << 3);
    *word = wdata;
    *nbytes = wlen;
}

/* writer.rs:226-242 */
static int bw_flush(bit_writer *w, uint32_t *word, size_t *nbytes)
{
    if (w->counter > 0) {
        size_t c = 8 - w->counter;
        bw_write_bits(w, 0, c, word, nbytes);
        return 1;
    }
    return 0;
}

/* test hook for the writer.rs vectors: packs (value,len) pairs, then flushes */
BZO_EXPORT size_t bzo_bitwriter_pack(const uint32_t *vals, const uint8_t *lens, size_t n,
                                     uint8_t *out, size_t cap)
{
    bit_writer w = {0, 0};
    size_t o = 0;
    uint32_t word; size_t nb;
    for (size_t i = 0; i < n; i++) {
        bw_write_bits(&w, vals[i], lens[i], &word, &nb);
        for (size_t k = 0; k < nb && o < cap; k++) { out[o++] = (uint8_t)(word >> 24); word <<= 8; }
    }
    if (bw_flush(&w, &word, &nb))
        for (size_t k = 0; k < nb && o < cap; k++) { out[o++] = (uint8_t)(word >> 24); word <<= 8; }
    return o;
}

/* ------------------------------------------------------------------------
 * MTF (src/bzip2/mtf.rs:12-39)
 * ---------------------------------------------------------------------- */
typedef struct { size_t data[256]; size_t count; } mtf_position;

static void mtf_new(mtf_position *m, size_t count)
{
    m->count = count;
    for (size_t i = 0; i < count; i++) m->data[i] = i;
}

static size_t mtf_pop(mtf_position *m, size_t value)
{
    if (value == m->data[0]) return 0;
    size_t t = m->data[0];
    m->data[0] = value;
    for (size_t i = 1; i < m->count; i++) {
        size_t d = m->data[i];
        m->data[i] = t;
        t = d;
        if (t == value) return i;
    }
    fprintf(stderr, "oracle: mtf unreachable\n");
    abort();
}

/* ------------------------------------------------------------------------
 * BZip2 encoder (src/bzip2/encoder.rs)
 * ---------------------------------------------------------------------- */
#define BZ_G_SIZE 50                       /* bzip2/mod.rs:20 */
#define BZ_N_ITERS 4                       /* encoder.rs:294 */
#define BZ_MAX_SELECTORS (2 + (900000 / BZ_G_SIZE)) /* encoder.rs:295 */
#define BZ_LESSER_ICOST 0                  /* encoder.rs:297 */
#define BZ_GREATER_ICOST 15                /* encoder.rs:298 */

enum { ACTION_RUN = 0, ACTION_FLUSH = 1, ACTION_FINISH = 2 }; /* action.rs:8-13 */

typedef struct { uint32_t data; uint8_t len; } small_bit_vec;

typedef struct {
    small_bit_vec *q;
    size_t head, tail, cap;
} bit_queue;

static void q_push(bit_queue *q, uint32_t data, size_t len)
{
    if (q->head == q->tail) q->head = q->tail = 0;
    if (q->tail == q->cap) {
        if (q->head > 0) {
            memmove(q->q, q->q + q->head, (q->tail - q->head) * sizeof(small_bit_vec));
            q->tail -= q->head;
            q->head = 0;
        }
        if (q->tail == q->cap) {
            q->cap = q->cap ? q->cap * 2 : 4096;
            q->q = (small_bit_vec *)realloc(q->q, q->cap * sizeof(small_bit_vec));
        }
    }
    q->q[q->tail].data = data;
    q->q[q->tail].len = (uint8_t)len;
    q->tail++;
}

/* per-block statistics (the reference's log::debug! lines) */
typedef struct {
    uint32_t nblock, block_crc, orig_ptr, mtf_count, in_use_count, group_num, n_selectors;
    uint32_t max_len, lm_tables; /* lm_tables: how many tables took gen_code_lm */
    uint64_t bits;
    /* the figures of the two other debug lines of write_blockdata: "pass k: size is {totc / 8}, grp uses are {fave}"
       (encoder.rs:483-498) and "bits: mapping .., selectors .., code lengths .., codes .." (:556-636) */
    uint32_t pass_size[4], fave[4][6];
    uint32_t bits_mapping, bits_selectors, bits_lengths, bits_codes;
} bzo_block_stats;

typedef struct {
    /* EncoderInner, encoder.rs:162-175 */
    uint8_t *block_buf; size_t block_len, block_cap;
    int inner_finished;
    size_t block_size_100k, block_max_len;
    uint32_t combined_crc;
    size_t block_no;
    uint32_t block_crc; /* raw running value (Digest.value) */
    uint8_t rle_buffer; size_t rle_count;
    uint8_t in_use[256];
    uint16_t *mtf_buffer;
    uint64_t num_z;
    /* BZip2Encoder, encoder.rs:40-49 */
    bit_writer writer;
    bit_queue queue;
    int finished;
    uint32_t bitbuf; size_t bitbuflen;
    int bit_finished;
    /* oracle controls */
    int huffman_mode; /* 0 reference, 1 libbzip2 differential */
    bzo_block_stats *stats; size_t n_stats, cap_stats;
    int want_stats;
} bzo_enc;

static void enc_write(bzo_enc *e, uint32_t data, size_t len) /* encoder.rs:203-210 */
{
    e->num_z += len;
    q_push(&e->queue, data, len);
}
static void enc_write_u8(bzo_enc *e, uint8_t v) { enc_write(e, v, 8); }
static void enc_write_u16(bzo_enc *e, uint16_t v) { enc_write(e, v, 16); }
static void enc_write_u32(bzo_enc *e, uint32_t v) { enc_write(e, v, 32); }

static void prepare_new_block(bzo_enc *e) /* encoder.rs:178-183 */
{
    e->block_no += 1;
    e->block_crc = 0xFFFFFFFFu;
    e->block_len = 0;
    memset(e->in_use, 0, sizeof(e->in_use));
}

/* encoder.rs:641-651 create_huffman */
static void create_huffman(bzo_enc *e, const size_t *freq, size_t alpha, size_t lim,
                           uint8_t *out, bzo_block_stats *st)
{
    if (e->huffman_mode == 1) {
        int32_t f32[258];
        for (size_t i = 0; i < alpha; i++) f32[i] = (int32_t)freq[i];
        libbz2_hb_make_code_lengths(out, f32, (int32_t)alpha, (int32_t)lim);
        return;
    }
    size_t weight[258];
    for (size_t i = 0; i < alpha; i++) weight[i] = (freq[i] > 1 ? freq[i] : 1) << 8;
    size_t outn = 0;
    int lm = bzo_make_tab_with_fn(weight, alpha, lim, 1, out, &outn);
    if (lm && st) st->lm_tables += 1;
}

/* encoder.rs:653-669 zle_write */
static void zle_write(bzo_enc *e, size_t zero_count, uint32_t *mtf_freq, size_t *mtf_count)
{
    if (zero_count != 0) {
        zero_count += 1;
        while (zero_count > 1) {
            uint16_t run = (uint16_t)(zero_count & 1);
            e->mtf_buffer[*mtf_count] = run;
            *mtf_count += 1;
            mtf_freq[run] += 1;
            zero_count >>= 1;
        }
    }
}

/* encoder.rs:300-639 write_blockdata */
static int write_blockdata(bzo_enc *e, bzo_block_stats *st)
{
    size_t in_use_count = 0;
    uint8_t unseq2seq[256];
    memset(unseq2seq, 0, sizeof(unseq2seq));
    for (size_t i = 0; i < 256; i++) {              /* :307-314 */
        if (e->in_use[i]) {
            unseq2seq[i] = (uint8_t)in_use_count;
            in_use_count += 1;
        }
    }
    size_t eob = in_use_count + 1;                  /* :316 */

    mtf_position mtf_table;
    mtf_new(&mtf_table, in_use_count);

    size_t zero_count = 0;
    uint32_t mtf_freq[258];
    memset(mtf_freq, 0, sizeof(mtf_freq));
    size_t mtf_count = 0;
    size_t nblock = e->block_len;

    size_t *sa = (size_t *)malloc((nblock ? nblock : 1) * sizeof(size_t));
    bzo_bwt(e->block_buf, nblock, sa);              /* :324 */

    for (size_t i = 0; i < nblock; i++) {           /* :324-353 */
        size_t s = sa[i];
        size_t j;
        if (s == 0) {
            enc_write(e, (uint32_t)i, 24);          /* origPtr, :333 */
            if (st) st->orig_ptr = (uint32_t)i;
            j = nblock - 1;
        } else {
            j = s - 1;
        }
        size_t val = unseq2seq[e->block_buf[j]];
        uint16_t c = (uint16_t)(mtf_pop(&mtf_table, val) + 1);
        if (c == 1) {
            zero_count += 1;
        } else {
            zle_write(e, zero_count, mtf_freq, &mtf_count);
            zero_count = 0;
            e->mtf_buffer[mtf_count] = c;
            mtf_count += 1;
            mtf_freq[c] += 1;
        }
    }
    free(sa);

    zle_write(e, zero_count, mtf_freq, &mtf_count); /* :355-358 */
    e->mtf_buffer[mtf_count] = (uint16_t)eob;
    mtf_count += 1;
    mtf_freq[eob] += 1;

    size_t alpha_size = in_use_count + 2;           /* :367 */

    size_t group_num;                               /* :370-376 */
    if (mtf_count < 200) group_num = 2;
    else if (mtf_count < 600) group_num = 3;
    else if (mtf_count < 1200) group_num = 4;
    else if (mtf_count < 2400) group_num = 5;
    else group_num = 6;

    /* initial coding tables, :379-426.  `len` is stored in the order the
     * reference's scan produces it: len[k] belongs to n_part = group_num-k,
     * i.e. to libbzip2's table (group_num-1-k)... the reference then always
     * walks it with .rev(), so table t (selector value) == len[group_num-1-t]. */
    uint8_t len[6][258];
    {
        uint32_t rem_freq = (uint32_t)mtf_count;
        long gs = 0;
        for (size_t k = 0; k < group_num; k++) {
            size_t n_part = group_num - k;
            uint32_t t_freq = rem_freq / (uint32_t)n_part;
            long ge = gs - 1;
            uint32_t a_freq = 0;
            while (a_freq < t_freq && ge < (long)alpha_size - 1) {
                ge += 1;
                a_freq += mtf_freq[ge];
            }
            if (ge > gs && n_part != group_num && n_part != 1 &&
                (((group_num - n_part) & 1) == 1)) {
                a_freq -= mtf_freq[ge];
                ge -= 1;
            }
            for (long i = 0; i < (long)alpha_size; i++)
                len[k][i] = (i >= gs && i <= ge) ? BZ_LESSER_ICOST : BZ_GREATER_ICOST;
            rem_freq -= a_freq;
            gs = ge + 1;
        }
    }

    size_t n_selectors = 0;
    static __thread uint8_t selector[BZ_MAX_SELECTORS];

    for (int iter = 0; iter < BZ_N_ITERS; iter++) { /* :433-509 */
        size_t rfreq[6][258];
        memset(rfreq, 0, sizeof(rfreq));
        n_selectors = 0;
        size_t gs = 0;
        uint32_t totc = 0, fave[6] = {0, 0, 0, 0, 0, 0}; /* :435-438 */
        while (gs < mtf_count) {
            size_t ge = gs + BZ_G_SIZE < mtf_count ? gs + BZ_G_SIZE : mtf_count;
            /* len.iter().rev().map(cost).enumerate().min_by(): first minimum wins */
            size_t bt = 0;
            uint16_t bc = 0;
            for (size_t t = 0; t < group_num; t++) {
                const uint8_t *li = len[group_num - 1 - t];
                uint16_t cost = 0;
                for (size_t i = gs; i < ge; i++) cost = (uint16_t)(cost + li[e->mtf_buffer[i]]);
                if (t == 0 || cost < bc) { bc = cost; bt = t; }
            }
            selector[n_selectors] = (uint8_t)bt;
            n_selectors += 1;
            totc += bc;    /* :469 */
            fave[bt] += 1; /* :470 */
            for (size_t i = gs; i < ge; i++) rfreq[bt][e->mtf_buffer[i]] += 1;
            gs = ge;
        }
        if (st) { /* the debug line :483-498 */
            st->pass_size[iter] = totc / 8;
            for (size_t t = 0; t < 6; t++) st->fave[iter][t] = fave[t];
        }
        /* len = rfreq.iter().rev().map(create_huffman), :504-508 */
        for (size_t k = 0; k < group_num; k++)
            create_huffman(e, rfreq[group_num - 1 - k], alpha_size, 17, len[k], st);
    }

    /* selector MTF, :511-517 */
    mtf_position sel_tab;
    mtf_new(&sel_tab, group_num);
    uint8_t *selector_mtf = (uint8_t *)malloc(n_selectors ? n_selectors : 1);
    for (size_t i = 0; i < n_selectors; i++) selector_mtf[i] = (uint8_t)mtf_pop(&sel_tab, selector[i]);

    /* code = len.iter().rev().map(HuffmanEncoder::new), :519-524 */
    uint32_t code[6][258];
    for (size_t t = 0; t < group_num; t++)
        bzo_canonical_codes(len[group_num - 1 - t], alpha_size, code[t]);

    /* mapping table, :527-565 */
    const uint64_t z_map = e->num_z; /* :532 */
    {
        uint16_t in_use16 = 0;
        for (size_t i = 0; i < 16; i++) {
            int any = 0;
            for (size_t j = 0; j < 16; j++) any |= e->in_use[i * 16 + j];
            in_use16 = (uint16_t)((in_use16 << 1) + (any ? 1 : 0));
        }
        enc_write_u16(e, in_use16);
        for (size_t i = 0; i < 16; i++) {
            if (in_use16 & (0x8000 >> i))
                for (size_t j = 0; j < 16; j++) enc_write(e, e->in_use[i * 16 + j] ? 1 : 0, 1);
        }
    }

    /* selectors, :567-574 */
    const uint64_t z_sel = e->num_z; /* :568 */
    enc_write(e, (uint32_t)group_num, 3);
    enc_write(e, (uint32_t)n_selectors, 15);
    for (size_t i = 0; i < n_selectors; i++) {
        size_t s = selector_mtf[i];
        enc_write(e, (1u << (s + 1)) - 2, s + 1);
    }
    free(selector_mtf);

    /* coding tables, :583-601 */
    const uint64_t z_len = e->num_z; /* :584 */
    uint32_t max_len = 0;
    for (size_t t = 0; t < group_num; t++) {
        const uint8_t *l = len[group_num - 1 - t];
        uint8_t curr = l[0];
        enc_write(e, curr, 5);
        for (size_t i = 0; i < alpha_size; i++) {
            uint8_t li = l[i];
            if (li > max_len) max_len = li;
            while (curr < li) { enc_write(e, 2, 2); curr += 1; }
            while (curr > li) { enc_write(e, 3, 2); curr -= 1; }
            enc_write(e, 0, 1);
        }
    }

    /* block data, :609-629 */
    const uint64_t z_cod = e->num_z; /* :610 */
    {
        size_t sel_ctr = 0, gs = 0;
        while (gs < mtf_count) {
            size_t ge = gs + BZ_G_SIZE < mtf_count ? gs + BZ_G_SIZE : mtf_count;
            size_t t = selector[sel_ctr];
            const uint8_t *l = len[group_num - 1 - t];
            for (size_t i = gs; i < ge; i++) {
                uint16_t b = e->mtf_buffer[i];
                enc_write(e, code[t][b], l[b]);
            }
            gs = ge;
            sel_ctr += 1;
        }
    }

    if (st) {
        st->mtf_count = (uint32_t)mtf_count;
        st->in_use_count = (uint32_t)in_use_count;
        st->group_num = (uint32_t)group_num;
        st->n_selectors = (uint32_t)n_selectors;
        st->max_len = max_len;
        st->bits_mapping = (uint32_t)(z_sel - z_map);
        st->bits_selectors = (uint32_t)(z_len - z_sel);
        st->bits_lengths = (uint32_t)(z_cod - z_len);
        st->bits_codes = (uint32_t)(e->num_z - z_cod);
    }
    return 0;
}

/* encoder.rs:699-716 write_rle */
static void write_rle(bzo_enc *e)
{
    for (size_t i = 0; i < e->rle_count; i++) e->block_crc = crc_update_normal(e->block_crc, e->rle_buffer);
    size_t ret_count = e->rle_count < 4 ? e->rle_count : 4;
    for (size_t i = 0; i < ret_count; i++) {
        e->in_use[e->rle_buffer] = 1;
        e->block_buf[e->block_len++] = e->rle_buffer;
    }
    if (ret_count == 4) {
        uint8_t v = (uint8_t)(e->rle_count - 4);
        e->in_use[v] = 1;
        e->block_buf[e->block_len++] = v;
    }
}

/* encoder.rs:224-291 write_block */
static int write_block(bzo_enc *e, int is_final)
{
    if (is_final) {
        write_rle(e);
        e->rle_count = 0;
    }
    size_t nblock = e->block_len;
    uint32_t block_crc = ~e->block_crc;

    e->combined_crc = ((e->combined_crc << 1) | (e->combined_crc >> 31)) ^ block_crc; /* :237-238 */

    if (e->block_no == 1) {                          /* :245-251 */
        enc_write_u8(e, 0x42);
        enc_write_u8(e, 0x5a);
        enc_write_u8(e, 0x68);
        enc_write_u8(e, (uint8_t)(0x30 + e->block_size_100k));
    }

    if (nblock > 0) {                                /* :253-277 */
        uint64_t z0 = e->num_z;
        enc_write_u8(e, 0x31); enc_write_u8(e, 0x41); enc_write_u8(e, 0x59);
        enc_write_u8(e, 0x26); enc_write_u8(e, 0x53); enc_write_u8(e, 0x59);
        enc_write_u32(e, block_crc);
        enc_write(e, 0, 1);
        bzo_block_stats *st = NULL;
        if (e->want_stats) {
            if (e->n_stats == e->cap_stats) {
                e->cap_stats = e->cap_stats ? e->cap_stats * 2 : 16;
                e->stats = (bzo_block_stats *)realloc(e->stats, e->cap_stats * sizeof(bzo_block_stats));
            }
            st = &e->stats[e->n_stats++];
            memset(st, 0, sizeof(*st));
            st->nblock = (uint32_t)nblock;
            st->block_crc = block_crc;
        }
        int rc = write_blockdata(e, st);
        if (rc) return rc;
        if (st) st->bits = e->num_z - z0;
        prepare_new_block(e);
    }
    if (is_final) {                                  /* :279-289 */
        enc_write_u8(e, 0x17); enc_write_u8(e, 0x72); enc_write_u8(e, 0x45);
        enc_write_u8(e, 0x38); enc_write_u8(e, 0x50); enc_write_u8(e, 0x90);
        enc_write_u32(e, e->combined_crc);
    }
    return 0;
}

/* encoder.rs:671-697 EncoderInner::next */
static int inner_next(bzo_enc *e, uint8_t buf)
{
    if (e->rle_count == 0) {
        e->rle_buffer = buf;
        e->rle_count = 1;
        return 0;
    }
    if (e->rle_buffer == buf && e->rle_count < 255) {
        e->rle_count += 1;
        return 0;
    }
    write_rle(e);
    e->rle_count = 1;
    e->rle_buffer = buf;
    if (e->block_len >= e->block_max_len) return write_block(e, 0);
    return 0;
}

/* encoder.rs:718-727 / :729-739 */
static int inner_flush(bzo_enc *e) { return e->inner_finished ? 0 : write_block(e, 0); }
static int inner_finish(bzo_enc *e)
{
    if (!e->inner_finished) {
        e->inner_finished = 1;
        return write_block(e, 1);
    }
    return 0;
}

/* BZip2Encoder::new, encoder.rs:58-72; returns NULL where the reference panics */
BZO_EXPORT bzo_enc *bzo_enc_new(int level)
{
    if (level < 1 || level > 9) return NULL;
    if (!g_crc_ready) crc_make_table_normal(0x04C11DB7u);
    bzo_enc *e = (bzo_enc *)calloc(1, sizeof(bzo_enc));
    e->block_size_100k = (size_t)level;
    e->block_max_len = (size_t)level * 100000 - 19;   /* :186 */
    e->block_cap = (size_t)level * 100000 + 16;
    e->block_buf = (uint8_t *)malloc(e->block_cap);
    e->mtf_buffer = (uint16_t *)calloc((size_t)level * 100000 + 1, sizeof(uint16_t)); /* :198 */
    e->block_crc = 0xFFFFFFFFu;
    e->block_no = 1;                                  /* :195 */
    return e;
}

BZO_EXPORT void bzo_enc_free(bzo_enc *e)
{
    if (!e) return;
    free(e->block_buf); free(e->mtf_buffer); free(e->queue.q); free(e->stats);
    free(e);
}

BZO_EXPORT void bzo_enc_set_huffman_mode(bzo_enc *e, int mode) { e->huffman_mode = mode; }
BZO_EXPORT void bzo_enc_enable_stats(bzo_enc *e, int on) { e->want_stats = on; }
BZO_EXPORT size_t bzo_enc_stats(bzo_enc *e, bzo_block_stats *out, size_t cap)
{
    size_t n = e->n_stats < cap ? e->n_stats : cap;
    if (out) memcpy(out, e->stats, n * sizeof(bzo_block_stats));
    return e->n_stats;
}

/* input iterator: returns 0..255, or -1 for None */
typedef int (*bzo_pull_fn)(void *ctx);

/* encoder.rs:74-114 next_bits.  Returns 1 = Some(Ok), 0 = None, <0 = Some(Err) */
static int next_bits(bzo_enc *e, bzo_pull_fn pull, void *ctx, int action, small_bit_vec *out)
{
    while (e->queue.head == e->queue.tail) {
        int s = pull(ctx);
        if (s >= 0) {
            int rc = inner_next(e, (uint8_t)s);
            if (rc) return rc;
        } else {
            if (e->finished) {
                e->finished = 0;
                return 0;
            } else {
                int rc = 0;
                if (action == ACTION_FLUSH) rc = inner_flush(e);
                else if (action == ACTION_FINISH) rc = inner_finish(e);
                if (rc) return rc;
                e->finished = 1;
            }
        }
    }
    *out = e->queue.q[e->queue.head++];
    return 1;
}

/* Encoder::next for BZip2Encoder, encoder.rs:116-159.
 * Returns 1 and *out_byte = Some(Ok(byte)); 0 = None; -3 = Some(Err(Unexpected)) */
BZO_EXPORT int bzo_enc_next(bzo_enc *e, bzo_pull_fn pull, void *ctx, int action, uint8_t *out_byte)
{
    while (e->bitbuflen == 0) {
        small_bit_vec s = {0, 0};
        uint32_t word = 0; size_t nb = 0;
        int r = next_bits(e, pull, ctx, action, &s);
        if (r < 0) return r;
        if (r == 1) {
            bw_write_bits(&e->writer, s.data, s.len, &word, &nb);
        } else {
            if (e->bit_finished) {
                e->bit_finished = 0;
                return 0;
            } else if (action == ACTION_FINISH || action == ACTION_FLUSH) {
                e->bit_finished = 1;
                if (!bw_flush(&e->writer, &word, &nb) || nb == 0) return 0;
            } else {
                return 0;
            }
        }
        e->bitbuf = word;
        e->bitbuflen = nb;
    }
    *out_byte = (uint8_t)left_convert(e->bitbuf, 32, 8);
    e->bitbuf = left_forward(e->bitbuf, 8);
    e->bitbuflen -= 1;
    return 1;
}

/* ---- bulk conveniences built on the iterator semantics above ------------ */
typedef struct { const uint8_t *p; size_t n, i; } buf_iter;
static int buf_pull(void *ctx)
{
    buf_iter *b = (buf_iter *)ctx;
    return b->i < b->n ? (int)b->p[b->i++] : -1;
}

/* `data.iter().encode(&mut enc, action).collect()` on an existing encoder:
 * pulls until Encoder::next returns None.  Returns bytes written or <0. */
BZO_EXPORT long bzo_enc_encode_iter(bzo_enc *e, const uint8_t *in, size_t n, int action,
                                    uint8_t *out, size_t cap)
{
    buf_iter it = {in, n, 0};
    size_t o = 0;
    for (;;) {
        uint8_t b;
        int r = bzo_enc_next(e, buf_pull, &it, action, &b);
        if (r < 0) return r;
        if (r == 0) break;
        if (o >= cap) return -100;
        out[o++] = b;
    }
    return (long)o;
}

/* one-shot: BZip2Encoder::new(level) + encode(Action::Finish).collect() */
BZO_EXPORT long bzo_encode_buffer(int level, int huffman_mode, const uint8_t *in, size_t n,
                                  uint8_t *out, size_t cap,
                                  bzo_block_stats *stats, size_t stats_cap, size_t *n_stats)
{
    bzo_enc *e = bzo_enc_new(level);
    if (!e) return -200;
    e->huffman_mode = huffman_mode;
    e->want_stats = stats != NULL;
    long r = bzo_enc_encode_iter(e, in, n, ACTION_FINISH, out, cap);
    if (stats) {
        size_t k = bzo_enc_stats(e, stats, stats_cap);
        if (n_stats) *n_stats = k;
    }
    bzo_enc_free(e);
    return r;
}

/* upper bound for an output buffer */
BZO_EXPORT size_t bzo_encode_bound(size_t n)
{
    /* RLE1 can expand by 5/4, Huffman worst case 17 bits/symbol + tables */
    return n * 3 + 8192 + (n / 800000 + 2) * 40000;
}

/* ---- stage-level entry points used by the GPU parity tests --------------- */

/* RLE1 + block split only (encoder.rs:671-716, :692): writes the concatenated
 * block buffers to `rle`, block end offsets (exclusive, into rle) to
 * `block_ends`, the exclusive END of the input bytes covered by each block
 * to `in_ends`, per-block CRC to `crcs`.  Returns the number of blocks.
 * Action::Finish semantics (the pending run goes into the last block). */
BZO_EXPORT size_t bzo_rle1_blocks(int level, const uint8_t *in, size_t n,
                                  uint8_t *rle, size_t rle_cap,
                                  uint64_t *block_ends, uint64_t *in_ends, uint32_t *crcs,
                                  size_t max_blocks)
{
    if (!g_crc_ready) crc_make_table_normal(0x04C11DB7u);
    size_t block_max_len = (size_t)level * 100000 - 19;
    size_t nb = 0, o = 0, blk_start = 0;
    uint32_t crc = 0xFFFFFFFFu;
    uint8_t rb = 0; size_t rc = 0;
    size_t consumed = 0; /* input bytes already flushed into blocks */
    for (size_t i = 0; i <= n; i++) {
        int have = i < n;
        uint8_t b = have ? in[i] : 0;
        if (have && rc == 0) { rb = b; rc = 1; continue; }
        if (have && rb == b && rc < 255) { rc++; continue; }
        if (rc > 0) {
            for (size_t k = 0; k < rc; k++) crc = crc_update_normal(crc, rb);
            size_t m = rc < 4 ? rc : 4;
            for (size_t k = 0; k < m && o < rle_cap; k++) rle[o++] = rb;
            if (m == 4 && o < rle_cap) rle[o++] = (uint8_t)(rc - 4);
            consumed += rc;
        }
        if (have) { rc = 1; rb = b; } else { rc = 0; }
        if ((have && o - blk_start >= block_max_len) || (!have && o > blk_start)) {
            if (nb < max_blocks) {
                block_ends[nb] = o;
                in_ends[nb] = consumed;
                crcs[nb] = ~crc;
            }
            nb++;
            blk_start = o;
            crc = 0xFFFFFFFFu;
        }
    }
    return nb;
}

/* MTF + ZLE of one block given its rotation order (encoder.rs:304-358).
 * Returns mtf_count; writes symbols (incl. EOB) and mtf_freq[258]. */
BZO_EXPORT size_t bzo_mtf_zle(const uint8_t *block, size_t n, const size_t *sa,
                              uint16_t *mtf_out, uint32_t *mtf_freq, uint32_t *orig_ptr,
                              uint32_t *in_use_count_out)
{
    uint8_t in_use[256];
    memset(in_use, 0, sizeof(in_use));
    for (size_t i = 0; i < n; i++) in_use[block[i]] = 1;
    size_t in_use_count = 0;
    uint8_t unseq2seq[256];
    memset(unseq2seq, 0, 256);
    for (size_t i = 0; i < 256; i++) if (in_use[i]) unseq2seq[i] = (uint8_t)in_use_count++;
    mtf_position t;
    mtf_new(&t, in_use_count);
    memset(mtf_freq, 0, 258 * sizeof(uint32_t));
    size_t zero_count = 0, mtf_count = 0;
    for (size_t i = 0; i < n; i++) {
        size_t s = sa[i], j;
        if (s == 0) { *orig_ptr = (uint32_t)i; j = n - 1; } else j = s - 1;
        uint16_t c = (uint16_t)(mtf_pop(&t, unseq2seq[block[j]]) + 1);
        if (c == 1) { zero_count++; continue; }
        if (zero_count) {
            zero_count += 1;
            while (zero_count > 1) { uint16_t r = zero_count & 1; mtf_out[mtf_count++] = r; mtf_freq[r]++; zero_count >>= 1; }
            zero_count = 0;
        }
        mtf_out[mtf_count++] = c;
        mtf_freq[c]++;
    }
    if (zero_count) {
        zero_count += 1;
        while (zero_count > 1) { uint16_t r = zero_count & 1; mtf_out[mtf_count++] = r; mtf_freq[r]++; zero_count >>= 1; }
    }
    mtf_out[mtf_count++] = (uint16_t)(in_use_count + 1);
    mtf_freq[in_use_count + 1]++;
    *in_use_count_out = (uint32_t)in_use_count;
    return mtf_count;
}

/* ==========================================================================
 * BZip2 DECODER restatement (src/bzip2/decoder.rs, src/huffman/decoder.rs,
 * src/bitio/reader.rs, src/bzip2/mtf.rs:45-65).  Same rules as above: test
 * infrastructure only.
 *
 * Pinned by: data/sample1-4.bz2 -> sample1-4.ref (src/bzip2/mod.rs:84-148, incl.
 * the two-stream sample4), round trips of the encoder restatement, and libbzip2
 * streams.  Known deviation: where the reference PANICS on malformed code-length
 * tables (unreachable!() / out-of-bounds in huffman/decoder.rs:150-206) this
 * restatement reports DataError.
 * ========================================================================== */
#include "bz2_rnums.h"

/* BZip2Error, src/bzip2/error.rs:5-11 (as negative status codes of the C ABI) */
#define BZE_DATA (-1)
#define BZE_EOF (-2)
#define BZE_UNEXPECTED (-3)
#define BZE_MAGIC_FIRST (-4)
#define BZE_MAGIC (-5)

/* BitReader<Left> over a byte buffer (bitio/reader.rs:70-186).  At the end of the input a read
 * does not fail: it returns the bits that are left (possibly none) as a shorter number. */
typedef struct { const uint8_t *p; uint64_t nbits, pos; } bit_reader;

static uint32_t br_peek(const bit_reader *r, unsigned len, unsigned *got)
{
    uint64_t avail = r->nbits - r->pos;
    unsigned k = (avail < len) ? (unsigned)avail : len;
    uint32_t v = 0;
    for (unsigned i = 0; i < k; i++) {
        uint64_t b = r->pos + i;
        v = (v << 1) | ((r->p[b >> 3] >> (7 - (b & 7))) & 1u);
    }
    *got = k;
    return v;
}
static void br_skip(bit_reader *r, unsigned len)
{
    uint64_t avail = r->nbits - r->pos;
    r->pos += (avail < len) ? avail : len;
}
static uint32_t br_read(bit_reader *r, unsigned len)
{
    unsigned got;
    uint32_t v = br_peek(r, len, &got);
    br_skip(r, got);
    return v;
}

/* HuffmanDecoder::<Left>::new(l, 12) + dec (huffman/decoder.rs:98-233), as canonical first-code
 * tables: equivalent for every table the reference accepts. */
typedef struct {
    uint8_t len[258];
    uint32_t code[258];
    unsigned alpha, max_len, stab_bits;
    int valid;
} huff_dec;

static int huff_dec_new(huff_dec *h, const uint8_t *len, unsigned alpha)
{
    memset(h, 0, sizeof(*h));
    h->alpha = alpha;
    unsigned max_len = 0;
    for (unsigned i = 0; i < alpha; i++) { h->len[i] = len[i]; if (len[i] > max_len) max_len = len[i]; }
    h->max_len = max_len;
    h->stab_bits = max_len < 12 ? max_len : 12;
    if (max_len >= 32) return -1;                       /* "length error", :113-115 */
    bzo_canonical_codes(len, alpha, h->code);           /* create_huffman_table, huffman/mod.rs:22-67 */
    /* the reference fills a table / tree; an over-subscribed set of lengths indexes out of bounds
     * (panic), a Leaf/Branch clash is Err -> DataError.  Both become "invalid" here. */
    for (unsigned i = 0; i < alpha; i++)
        if (len[i] && (h->code[i] >> len[i]) != 0) return -1;
    h->valid = 1;
    return 0;
}

/* returns symbol, or -1 = Ok(None) (no bits left), -2 = Err / unreachable */
static int huff_dec_sym(const huff_dec *h, bit_reader *r)
{
    unsigned got;
    uint32_t c = br_peek(r, h->stab_bits, &got);
    if (got == 0) return -1;                            /* :204-207 */
    (void)c;
    /* walk the canonical code: read up to max_len bits (zero padded past the end, like the
     * reference's shifted peek) and find the symbol whose code matches */
    uint32_t acc = 0;
    for (unsigned l = 1; l <= h->max_len; l++) {
        unsigned g;
        uint64_t save = r->pos;
        r->pos += l - 1;
        uint32_t bit = (r->pos < r->nbits) ? br_peek(r, 1, &g) : 0;
        r->pos = save;
        acc = (acc << 1) | bit;
        for (unsigned s = 0; s < h->alpha; s++) {
            if (h->len[s] == l && h->code[s] == acc) {
                /* bits beyond the table width are consumed one by one by the tree walk, which
                 * fails at the end of the input (:221-224) */
                if (l > h->stab_bits && save + l > r->nbits) return -2;
                br_skip(r, l);
                return (int)s;
            }
        }
    }
    return -2; /* incomplete code hit: unreachable!() in the reference */
}

typedef struct {
    bit_reader rd;
    /* BZip2DecoderBase, decoder.rs:93-108 */
    size_t block_no, block_size_100k;
    uint32_t combined_crc, block_crc, crc_value;
    uint32_t *tt; size_t tt_len;
    size_t n_block_used;
    uint32_t t_pos;
    size_t rnd_n2go, rnd_tpos; int block_randomised;
    size_t result_count, result_wrote_count; uint8_t result_charactor;
    size_t stream_no;
} bzo_dec;

/* MtfPositionDecoder::pop, mtf.rs:51-64 */
static size_t mtfd_pop(size_t *data, size_t value)
{
    if (value == 0) return data[0];
    size_t t = data[value];
    for (size_t i = value; i-- > 0;) data[i + 1] = data[i];
    data[0] = t;
    return t;
}

/* decoder.rs:163-525.  1 = Ok(true), 0 = Ok(false), <0 = Err */
static int dec_init_block(bzo_dec *d)
{
    bit_reader *r = &d->rd;
    for (;;) {
        if (d->block_no == 0) {
            int magic_err = d->stream_no == 1 ? BZE_MAGIC_FIRST : BZE_MAGIC;
            (void)br_read(r, 8); (void)br_read(r, 8); (void)br_read(r, 8); /* 'B','Z','h': read, not compared (:175-180) */
            uint32_t b = br_read(r, 8);
            if (b < 1 + 0x30 || b > 9 + 0x30) return magic_err;            /* :184-186 */
            d->block_size_100k = b - 0x30;
        } else {
            uint32_t data_block_crc = ~d->crc_value;                         /* :189-201 */
            if (data_block_crc != d->block_crc) return BZE_DATA;
            d->combined_crc = ((d->combined_crc << 1) | (d->combined_crc >> 31)) ^ d->block_crc;
            d->crc_value = 0xFFFFFFFFu;
        }
        uint32_t head = br_read(r, 8);
        if (head == 0x31) {
            for (int k = 0; k < 5; k++) (void)br_read(r, 8);                 /* :207-221: read, not compared */
            d->block_no += 1;
            d->block_crc = br_read(r, 32);
            d->block_randomised = br_read(r, 1) == 1;
            size_t orig_pos = br_read(r, 24);
            if (orig_pos > 10 + 100000 * d->block_size_100k) return BZE_DATA; /* :238 */
            size_t seq2unseq[256], n_in_use = 0;
            {
                int in_use16[16];
                for (int i = 0; i < 16; i++) in_use16[i] = br_read(r, 1) == 1;
                for (int i = 0; i < 16; i++)
                    if (in_use16[i])
                        for (int j = 0; j < 16; j++)
                            if (br_read(r, 1) == 1) seq2unseq[n_in_use++] = (size_t)(i * 16 + j);
            }
            if (n_in_use == 0) return BZE_DATA;                               /* :273-275 */
            size_t alpha_size = n_in_use + 2;
            size_t n_groups = br_read(r, 3);
            if (n_groups < 2 || n_groups > 6) return BZE_DATA;
            size_t n_selectors = br_read(r, 15);
            if (n_selectors < 1) return BZE_DATA;
            uint8_t *selector = (uint8_t *)malloc(n_selectors);
            {
                size_t lst[6];
                for (size_t i = 0; i < n_groups; i++) lst[i] = i;
                for (size_t s = 0; s < n_selectors; s++) {
                    size_t j = 0;
                    while (br_read(r, 1) != 0) {
                        j += 1;
                        if (j >= n_groups) { free(selector); return BZE_DATA; }
                    }
                    selector[s] = (uint8_t)mtfd_pop(lst, j);
                }
            }
            uint8_t len[6][258];
            for (size_t t = 0; t < n_groups; t++) {                            /* :318-348 */
                uint32_t curr = br_read(r, 5);
                for (size_t i = 0; i < alpha_size; i++) {
                    while (br_read(r, 1) != 0) {
                        if (curr < 1 || curr > 20) { free(selector); return BZE_DATA; }
                        if (br_read(r, 1) == 0) curr += 1; else curr -= 1;
                    }
                    len[t][i] = (uint8_t)curr;
                }
            }
            huff_dec *code = (huff_dec *)malloc(n_groups * sizeof(huff_dec));
            for (size_t t = 0; t < n_groups; t++)
                if (huff_dec_new(&code[t], len[t], (unsigned)alpha_size) != 0) { free(code); free(selector); return BZE_DATA; }
            uint16_t eob = (uint16_t)(alpha_size - 1);
            size_t nblock_max = 100000 * d->block_size_100k;
            size_t unzftab[257];
            memset(unzftab, 0, sizeof(unzftab));
            d->tt = (uint32_t *)realloc(d->tt, (nblock_max + 1) * sizeof(uint32_t));
            d->tt_len = 0;
            int err = 0;
            {
                size_t group_no = 0, group_pos = 0, n = 1, es = 0;
                size_t mtf[256];
                for (size_t i = 0; i < n_in_use; i++) mtf[i] = i;
                for (;;) {
                    if (group_pos == 0) {
                        group_no += 1;
                        if (group_no > n_selectors) { err = BZE_DATA; break; }
                        group_pos = BZ_G_SIZE;
                    }
                    group_pos -= 1;
                    int sym = huff_dec_sym(&code[selector[group_no - 1]], r);
                    if (sym < 0) { err = BZE_DATA; break; }
                    if (es > 0 && sym != 0 && sym != 1) {
                        size_t uc = seq2unseq[mtfd_pop(mtf, 0)];
                        unzftab[uc + 1] += es;
                        if (d->tt_len + es > nblock_max) { /* Vec::push would grow; the check below rejects it */
                            err = BZE_DATA; break;
                        }
                        for (size_t k = 0; k < es; k++) d->tt[d->tt_len++] = (uint32_t)uc;
                        if (d->tt_len >= nblock_max) { err = BZE_DATA; break; }
                        n = 1; es = 0;
                    }
                    if ((uint16_t)sym == eob) break;
                    if (n >= 2 * 1024 * 1024) { err = BZE_DATA; break; }
                    if (sym == 0) { es += n; n <<= 1; }
                    else if (sym == 1) { n <<= 1; es += n; }
                    else {
                        if (d->tt_len >= nblock_max) { err = BZE_DATA; break; }
                        size_t uc = seq2unseq[mtfd_pop(mtf, (size_t)sym - 1)];
                        unzftab[uc + 1] += 1;
                        d->tt[d->tt_len++] = (uint32_t)uc;
                    }
                }
            }
            free(code); free(selector);
            if (err) return err;
            if (orig_pos >= d->tt_len) return BZE_DATA;                        /* :441-443 */
            if (unzftab[0] != 0) return BZE_DATA;
            for (size_t i = 1; i < 257; i++) {
                unzftab[i] += unzftab[i - 1];
                if (unzftab[i - 1] > unzftab[i]) return BZE_DATA;
            }
            if (unzftab[256] != d->tt_len) return BZE_DATA;
            for (size_t i = 0; i < d->tt_len; i++) {                           /* :468-473 */
                size_t uc = d->tt[i] & 0xFF;
                d->tt[unzftab[uc]] |= (uint32_t)i << 8;
                unzftab[uc] += 1;
            }
            d->t_pos = d->tt[orig_pos] >> 8;
            d->n_block_used = 0;
            if (d->block_randomised) { d->rnd_n2go = 0; d->rnd_tpos = 0; }
            d->result_count = 0;
            d->result_wrote_count = 0;
            return 1;
        } else if (head == 0x17) {
            for (int k = 0; k < 5; k++) (void)br_read(r, 8);
            uint32_t stored = br_read(r, 32);
            if (stored != d->combined_crc) return BZE_DATA;
            r->pos = (r->pos + 7) & ~(uint64_t)7;                              /* skip_to_next_byte */
            if (r->pos > r->nbits) r->pos = r->nbits;
            unsigned got;
            (void)br_peek(r, 8, &got);
            if (got == 8) {
                d->block_no = 0;
                d->combined_crc = 0;
                d->stream_no += 1;
            } else {
                return 0;
            }
        } else {
            return BZE_DATA;
        }
    }
}

/* decoder.rs:527-542 */
static int dec_next_lfm(bzo_dec *d, uint8_t *out)
{
    uint32_t position = d->t_pos;
    if (position >= 100000 * (uint32_t)d->block_size_100k) return BZE_DATA;
    position = d->tt[position];
    uint8_t k0 = (uint8_t)position;
    d->t_pos = position >> 8;
    d->n_block_used += 1;
    if (d->block_randomised) {
        if (d->rnd_n2go == 0) {
            d->rnd_n2go = kBz2RNums[d->rnd_tpos];
            d->rnd_tpos += 1;
            if (d->rnd_tpos == 512) d->rnd_tpos = 0;
        }
        d->rnd_n2go -= 1;
        k0 ^= (d->rnd_n2go == 1) ? 1 : 0;
    }
    *out = k0;
    return 0;
}

/* BitDecodeService::next, decoder.rs:545-581.  1 = byte, 0 = None, <0 error */
static int dec_next(bzo_dec *d, uint8_t *out)
{
    if (d->result_count == d->result_wrote_count) {
        if (d->n_block_used == d->tt_len) {
            int rc = dec_init_block(d);
            if (rc <= 0) return rc;
        }
        uint8_t buffer;
        int rc = dec_next_lfm(d, &buffer);
        if (rc) return rc;
        if (buffer == d->result_charactor && d->result_count < 4) {
            d->result_count += 1;
            d->result_wrote_count += 1;
        } else {
            d->result_charactor = buffer;
            d->result_count = 1;
            d->result_wrote_count = 1;
        }
        if (d->result_count == 4) {
            uint8_t c;
            rc = dec_next_lfm(d, &c);
            if (rc) return rc;
            d->result_count += c;
        }
    } else {
        d->result_wrote_count += 1;
    }
    d->crc_value = crc_update_normal(d->crc_value, d->result_charactor);
    *out = d->result_charactor;
    return 1;
}

/* `bytes.decode(&mut BZip2Decoder::new()).collect()`: decodes until None or the first error.
 * Returns the number of bytes produced (those before an error are kept, as the iterator would have
 * yielded them); *status = 0 or the BZip2Error code. */
BZO_EXPORT size_t bzo_decode_buffer(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int *status)
{
    if (!g_crc_ready) crc_make_table_normal(0x04C11DB7u);
    bzo_dec d;
    memset(&d, 0, sizeof(d));
    d.rd.p = in; d.rd.nbits = (uint64_t)n * 8; d.rd.pos = 0;
    d.crc_value = 0xFFFFFFFFu;
    d.stream_no = 1;
    size_t o = 0;
    int st = 0;
    for (;;) {
        uint8_t b = 0;
        int rc = dec_next(&d, &b);
        if (rc == 0) break;
        if (rc < 0) { st = rc; break; }
        if (o >= cap) { st = -100; break; }
        out[o++] = b;
    }
    free(d.tt);
    if (status) *status = st;
    return o;
}
