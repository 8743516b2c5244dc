/* ==========================================================================
 * TEST INFRASTRUCTURE -- CPU oracle for the Deflate encode path.
 *
 * Plain-C restatement of the reference's algorithm (chalharu/rust-compression 0.1.5,
 * /root/reference/src): LzssEncoder + SlideDict (lzss/encoder.rs, lzss/slidedict.rs), Inflater /
 * InflaterInner (deflate/encoder.rs, deflate/mod.rs), the LSB-first bit writer
 * (bitio/writer.rs, bitio/direction/right.rs), and the zlib / gzip containers
 * (zlib/encoder.rs, gzip/encoder.rs, adler32.rs, crc32.rs).  The Huffman length builder is
 * the one bz2_oracle.c already restates (bzo_make_tab_with_fn, mode 0 = plain x + y).
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the benches may load it; the
 * product library never links or calls it.
 *
 * PARITY PINNING (tests/test_oracle_deflate.py): every known-answer vector the reference's own
 * tests hold for this path --
 *     LZSS token vectors              lzss/encoder.rs:246-598 (13 tests, incl. with_dict)
 *     Deflate byte / bit vectors      deflate/encoder.rs:660-1027 (empty, unit, arr, arr2, arr3
 *                                     stored, arr4 dynamic Huffman)
 *     length / distance code tables   deflate/encoder.rs:1029-1210
 *     zlib (plain and with_dict), gzip  zlib/encoder.rs:161-192, gzip/encoder.rs:147-164
 *     Adler-32 / CRC-32 known answers adler32.rs tests, crc32.rs:158-163
 * plus decodability of every produced stream by zlib (python zlib.decompressobj).  Output on large
 * inputs is not pinned by any reference test beyond these.
 * ========================================================================== */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DFO_EXPORT __attribute__((visibility("default")))

extern int bzo_make_tab_with_fn(const size_t *freq, size_t n, size_t lim, int mode, uint8_t *out, size_t *out_n);

/* ---------------------------------------------------------------- growable byte vector */
typedef struct { uint8_t *p; size_t n, cap; } bytes;
static void bytes_push(bytes *b, uint8_t v)
{
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->p = (uint8_t *)realloc(b->p, b->cap); }
    b->p[b->n++] = v;
}

/* ---------------------------------------------------------------- HashTab (lzss/slidedict.rs:24-115) */
#define TAB_LEN 65536u
typedef struct { uint16_t *search_tab; uint8_t *flag_tab; size_t len; } hashtab;

static size_t get_hash(const uint8_t *d, size_t n) /* slidedict.rs:80-87 (64-bit usize) */
{
    uint64_t hash = 0;
    for (size_t i = 0; i < n; i++) hash = (hash << 8) | ((hash >> 24) ^ (uint64_t)d[i]);
    return (size_t)((hash * 0x7A7C4F9F7A7C4F9Full) >> 48);
}

static void gen_change(hashtab *h) /* :72-77 */
{
    for (size_t i = 0; i < TAB_LEN / 4; i++) h->flag_tab[i] = (uint8_t)((h->flag_tab[i] & 0x55) << 1);
    h->len = 0;
}

/* :99-114 push: distance to the previous trigram with this hash, or 0 = none */
static size_t hash_push(hashtab *h, const uint8_t *d, size_t n, int *found)
{
    const size_t hash = get_hash(d, n);
    const unsigned f = (h->flag_tab[hash >> 2] >> ((hash & 3) << 1)) & 3;
    size_t ret = 0;
    *found = 0;
    if (f != 0) {
        const size_t p = h->search_tab[hash];
        ret = (f & 1) ? h->len - p : TAB_LEN + h->len - p;
        *found = 1;
    }
    h->search_tab[hash] = (uint16_t)h->len; /* push_tab :90-97 */
    h->flag_tab[hash >> 2] |= (uint8_t)(1u << ((hash & 3) << 1));
    h->len += 1;
    if (h->len >= TAB_LEN) gen_change(h);
    return ret;
}

/* ---------------------------------------------------------------- SlideDict + LzssEncoder
 * The reference keeps bytes and chain distances in circular buffers of window + max_match +
 * lazy_level + 1 entries; everything it can still address is kept here in flat arrays
 * (data[i], posd[i] = chain distance of the trigram starting at i), which changes indices only. */
typedef struct { uint32_t len, pos; } lz_info; /* MatchInfo; len == 0 with pos == ~0: None */

typedef struct dfo_lzss {
    bytes data;
    size_t *posd; size_t posd_n, posd_cap;
    hashtab ht;
    size_t max_pos, min_match, max_match, lazy_level;
    size_t offset;
    int cmp_mode; /* 0: deflate/encoder.rs:34-51, 1: lzss/mod.rs:96-112 (the lzss tests' comparison) */
    void (*sink)(void *, int is_ref, size_t len, size_t pos);
    void *sink_ctx;
} dfo_lzss;

/* comp(lhs, rhs) == Ordering::Less for two references */
static int cmp_less(int mode, lz_info l, lz_info r)
{
    const size_t a = mode ? ((size_t)l.len << 3) + r.pos : ((size_t)l.len << 3) + l.pos;
    const size_t b = mode ? ((size_t)r.len << 3) + l.pos : ((size_t)r.len << 3) + r.pos;
    return a > b; /* .cmp().reverse() == Less */
}

static void lz_init(dfo_lzss *z, int cmp_mode, size_t window, size_t max_match, size_t min_match, size_t lazy)
{
    memset(z, 0, sizeof(*z));
    z->ht.search_tab = (uint16_t *)calloc(TAB_LEN, 2);
    z->ht.flag_tab = (uint8_t *)calloc(TAB_LEN / 4, 1);
    z->max_pos = window; z->min_match = min_match; z->max_match = max_match; z->lazy_level = lazy;
    z->cmp_mode = cmp_mode;
}

static void lz_free(dfo_lzss *z)
{
    free(z->data.p); free(z->posd); free(z->ht.search_tab); free(z->ht.flag_tab);
}

/* SlideDict::append (slidedict.rs:192-214), one byte or a dictionary at a time: every trigram that
 * becomes complete is pushed (push_pos :151-156) */
static void slide_append(dfo_lzss *z, const uint8_t *d, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        bytes_push(&z->data, d[i]);
        if (z->data.n >= z->min_match) {
            const size_t start = z->data.n - z->min_match;
            int found;
            const size_t dist = hash_push(&z->ht, z->data.p + start, z->min_match, &found);
            if (z->posd_n == z->posd_cap) {
                z->posd_cap = z->posd_cap ? z->posd_cap * 2 : 4096;
                z->posd = (size_t *)realloc(z->posd, z->posd_cap * sizeof(size_t));
            }
            z->posd[z->posd_n++] = found ? dist : z->max_pos + 1;
        }
    }
}

/* slidedict.rs:158-190 */
static size_t check_match(const dfo_lzss *z, size_t cur, size_t cand, size_t max_match)
{
    size_t l = 0;
    while (l < max_match && z->data.p[cand + l] == z->data.p[cur + l]) l++;
    return l;
}

/* slidedict.rs:216-267.  Returns 0 for None. */
static int search_dic(const dfo_lzss *z, size_t offset, size_t max_match, lz_info *out)
{
    if (offset < z->min_match) return 0;
    const size_t cur = z->data.n - offset;
    size_t pos = z->posd[cur];
    if (max_match > offset) max_match = offset;
    int have = 0;
    lz_info info = {0, 0};
    size_t pos_count = 256 - 1;
    while (pos <= z->max_pos && pos_count > 0) {
        const size_t nlen = check_match(z, cur, cur - pos, max_match);
        const lz_info nw = {(uint32_t)nlen, (uint32_t)((pos - 1) & 0xFFFF)};
        if (!(have && (info.len >= nlen || cmp_less(z->cmp_mode, info, nw)))) { info = nw; have = 1; }
        if (nlen == max_match) pos_count = 0; else pos_count -= 1;
        pos += z->posd[cur - pos];
    }
    if (have) *out = info;
    return have;
}

/* lzss/encoder.rs:132-184 */
static void lz_encode(dfo_lzss *z)
{
    lz_info info;
    if (search_dic(z, z->offset, z->max_match, &info) && info.len >= z->min_match) {
        const size_t lazy_level = info.len < z->lazy_level ? info.len : z->lazy_level;
        lz_info out = info;
        size_t lazy_index = 0;
        for (size_t i = 1; i < lazy_level; i++) {
            if (out.len >= z->max_match) break;
            lz_info item;
            if (search_dic(z, z->offset - i, z->max_match, &item)) {
                if (item.len > z->min_match && cmp_less(z->cmp_mode, item, out)) { out = item; lazy_index = i; }
            }
        }
        if (lazy_index < z->min_match) {
            for (size_t i = 1; i <= lazy_index; i++) z->sink(z->sink_ctx, 0, 0, z->data.p[z->data.n - (z->offset - i) - 1]);
        } else {
            z->sink(z->sink_ctx, 1, lazy_index, info.pos);
        }
        z->sink(z->sink_ctx, 1, out.len, out.pos);
        z->offset -= out.len + lazy_index;
    } else {
        z->sink(z->sink_ctx, 0, 0, z->data.p[z->data.n - z->offset]);
        z->offset -= 1;
    }
}

static void lz_next_in(dfo_lzss *z, uint8_t b) /* lzss/encoder.rs:186-194 */
{
    if (z->max_match + z->lazy_level > z->offset) { slide_append(z, &b, 1); z->offset += 1; }
    while (z->offset >= z->max_match + z->lazy_level) lz_encode(z);
}

static void lz_flush(dfo_lzss *z) { while (z->offset > 0) lz_encode(z); } /* :196-200 */

/* ---------------------------------------------------------------- LZSS-only entry (token vectors) */
typedef struct { uint32_t *out; size_t n, cap; } tok_sink;
static void tok_push(void *c, int is_ref, size_t len, size_t pos)
{
    tok_sink *t = (tok_sink *)c;
    if (t->n < t->cap) { t->out[2 * t->n] = is_ref ? (uint32_t)len : 0; t->out[2 * t->n + 1] = (uint32_t)pos; }
    t->n++;
}

/* tokens as (len, pos) pairs; len == 0: literal pos.  Returns the token count (may exceed cap). */
DFO_EXPORT size_t dfo_lzss_tokens(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_n, int cmp_mode,
                                  size_t window, size_t max_match, size_t min_match, size_t lazy, uint32_t *out,
                                  size_t cap)
{
    dfo_lzss z;
    tok_sink t = {out, 0, cap};
    lz_init(&z, cmp_mode, window, max_match, min_match, lazy);
    z.sink = tok_push; z.sink_ctx = &t;
    if (dict_n) { /* with_dict, lzss/encoder.rs:104-130 */
        const size_t start = dict_n - (window < dict_n ? window : dict_n);
        slide_append(&z, dict + start, dict_n - start);
    }
    for (size_t i = 0; i < n; i++) lz_next_in(&z, in[i]);
    lz_flush(&z);
    lz_free(&z);
    return t.n;
}

/* ---------------------------------------------------------------- code tables (deflate/mod.rs:27-125) */
typedef struct { uint8_t *codes; uint16_t offsets[32]; uint8_t ext_bits[32]; } code_table;
static code_table g_len_tab, g_off_tab;
static int g_tabs_ready = 0;

static uint8_t *gen_codes(size_t len, const uint16_t *offsets) /* :64-74 */
{
    uint8_t *codes = (uint8_t *)malloc(len);
    size_t j = 0;
    for (size_t i = 0; i < len; i++) {
        while (offsets[j + 1] <= i) j++;
        codes[i] = (uint8_t)j;
    }
    return codes;
}

static void tabs_init(void)
{
    if (g_tabs_ready) return;
    size_t k = 0;
    for (unsigned i = 0; i < 8; i++) { g_len_tab.offsets[k] = (uint16_t)i; g_len_tab.ext_bits[k++] = 0; } /* :76-104 */
    for (unsigned i = 8; i < 28; i++) { const unsigned n = (i >> 2) - 1; g_len_tab.offsets[k] = (uint16_t)(((i & 3) | 4) << n); g_len_tab.ext_bits[k++] = (uint8_t)n; }
    g_len_tab.offsets[k] = 255; g_len_tab.ext_bits[k++] = 0;
    g_len_tab.offsets[k] = 0xFFFF;
    g_len_tab.codes = gen_codes(256, g_len_tab.offsets);
    k = 0;
    for (unsigned i = 0; i < 4; i++) { g_off_tab.offsets[k] = (uint16_t)i; g_off_tab.ext_bits[k++] = 0; } /* :106-125 */
    for (unsigned i = 4; i < 30; i++) { const unsigned n = (i >> 1) - 1; g_off_tab.offsets[k] = (uint16_t)(((i & 1) | 2) << n); g_off_tab.ext_bits[k++] = (uint8_t)n; }
    g_off_tab.offsets[k] = 0xFFFF;
    g_off_tab.codes = gen_codes(0x8000, g_off_tab.offsets);
    g_tabs_ready = 1;
}

/* (code, extra value, extra bits) of a length - 3 / a distance - 1: CodeTable::convert :40-48 */
DFO_EXPORT void dfo_convert(int which, unsigned value, unsigned *code, unsigned *ext, unsigned *ext_bits)
{
    tabs_init();
    const code_table *t = which ? &g_off_tab : &g_len_tab;
    const unsigned c = t->codes[value];
    *code = c; *ext = value - t->offsets[c]; *ext_bits = t->ext_bits[c];
}

/* ---------------------------------------------------------------- bit writer (bitio/writer.rs, Right) */
typedef struct { bytes out; uint32_t acc; unsigned cnt; } bitw;
static void bw_bits(bitw *w, uint32_t v, unsigned len) /* write_bits :172-205: LSB first */
{
    if (!len) return;
    w->acc |= (v & ((len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1))) << w->cnt;
    w->cnt += len;
    while (w->cnt >= 8) { bytes_push(&w->out, (uint8_t)w->acc); w->acc >>= 8; w->cnt -= 8; }
}
static void bw_pad(bitw *w) { if (w->cnt) bw_bits(w, 0, 8 - w->cnt); } /* flush :207-223 */

/* canonical codes, bit-reversed for the LSB-first stream (huffman/mod.rs:16-63, is_reverse) */
static void make_codes(const uint8_t *len, size_t n, uint16_t *code)
{
    unsigned cur = 0, last = 0;
    for (unsigned l = 1; l <= 15; l++)
        for (size_t s = 0; s < n; s++)
            if (len[s] == l) {
                cur <<= (last < l ? l - last : 0);
                last = l;
                unsigned r = 0;
                for (unsigned b = 0; b < l; b++) r |= ((cur >> b) & 1u) << (l - 1 - b);
                code[s] = (uint16_t)r;
                cur += 1;
            }
}

/* ---------------------------------------------------------------- InflaterInner (deflate/encoder.rs:262-660) */
#define MAX_BLOCK 0xFFFFu
typedef struct { uint16_t sym; uint16_t len_ext; uint8_t len_bits; uint8_t off_code; uint16_t off_ext; uint8_t off_bits; } dcode;

typedef struct dfo_block_info { uint64_t tokens, bytes, btype, bits; } dfo_block_info;

typedef struct {
    dcode *buf; size_t buf_n;
    size_t decompress_len;
    size_t sym_freq[286], off_freq[30];
    bytes hist;          /* stands in for nocomp_buf: all decoded bytes so far (the reference keeps 65535) */
    int finished;
    bitw w;
    dfo_block_info *blocks; size_t blocks_n, blocks_cap; /* test hook: what each block looked like */
} inner;

static void init_block(inner *s) /* :274-279 */
{
    s->buf_n = 0;
    memset(s->sym_freq, 0, sizeof(s->sym_freq));
    memset(s->off_freq, 0, sizeof(s->off_freq));
    s->sym_freq[256] = 1;
}

typedef struct { uint8_t s; uint16_t e; } tab_item;

/* enc_tab_to_freq :318-376 */
static size_t enc_tab_to_freq(const uint8_t *tab, size_t n, tab_item *list, size_t *freq)
{
    size_t k = 0;
    unsigned old = 255;
    size_t len = 0;
    for (size_t i = 0; i <= n; i++) {
        const unsigned d = i < n ? tab[i] : 255;
        if (old != d) {
            if (old == 0) {
                if (len >= 11) { freq[18] += 1; list[k].s = 18; list[k++].e = (uint16_t)(len - 11); }
                else if (len >= 3) { freq[17] += 1; list[k].s = 17; list[k++].e = (uint16_t)(len - 3); }
                else { for (size_t t = 0; t < len; t++) { list[k].s = 0; list[k++].e = 0; } freq[0] += len; }
            } else if (len >= 3) { freq[16] += 1; list[k].s = 16; list[k++].e = (uint16_t)(len - 3); }
            else if (len > 0) { for (size_t t = 0; t < len; t++) { list[k].s = (uint8_t)old; list[k++].e = 0; } freq[old] += len; }
            if (d != 0 && d != 255) { list[k].s = (uint8_t)d; list[k++].e = 0; freq[d] += 1; len = 0; }
            else len = 1;
            old = d;
        } else {
            len += 1;
            if (old == 0 && len == 138) { freq[18] += 1; list[k].s = 18; list[k++].e = 127; len = 0; }
            else if (old != 0 && len == 6) { freq[16] += 1; list[k].s = 16; list[k++].e = 3; len = 0; }
        }
    }
    return k;
}

typedef struct { uint32_t v; uint8_t n; } bitvec;

/* create_custom_huffman_table :395-452.  Returns the number of (value, bits) items. */
static size_t custom_header(const uint8_t *sym_tab, size_t sym_n, const uint8_t *off_tab, size_t off_n, bitvec *ret)
{
    static const unsigned len_map[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};
    tab_item symlist[300], offlist[40];
    size_t lenfreq[19] = {0};
    const size_t symk = enc_tab_to_freq(sym_tab, sym_n, symlist, lenfreq);
    const size_t offk = enc_tab_to_freq(off_tab, off_n, offlist, lenfreq);
    uint8_t len_enc[19] = {0};
    size_t len_enc_n = 0;
    bzo_make_tab_with_fn(lenfreq, 19, 7, 0, len_enc, &len_enc_n);
    uint8_t len_tab[19] = {0};
    size_t len_count = 3;
    for (size_t i = 0; i < len_enc_n; i++)
        if (len_enc[i] != 0) { len_tab[len_map[i]] = len_enc[i]; if (len_map[i] > len_count) len_count = len_map[i]; }
    size_t hlit = 0, hdist = 0;
    for (size_t i = 0; i < sym_n; i++) if (sym_tab[i]) hlit = i;
    hlit -= 256;
    for (size_t i = 0; i < off_n; i++) if (off_tab[i]) hdist = i;
    size_t k = 0;
    ret[k].v = 2; ret[k++].n = 2;
    ret[k].v = (uint32_t)hlit; ret[k++].n = 5;
    ret[k].v = (uint32_t)hdist; ret[k++].n = 5;
    ret[k].v = (uint32_t)(len_count - 3); ret[k++].n = 4;
    for (size_t i = 0; i <= len_count; i++) { ret[k].v = len_tab[i]; ret[k++].n = 3; }
    uint16_t lcode[19] = {0};
    make_codes(len_enc, len_enc_n, lcode);
    for (int which = 0; which < 2; which++) { /* conv_tab :378-393 */
        const tab_item *list = which ? offlist : symlist;
        const size_t n = which ? offk : symk;
        for (size_t i = 0; i < n; i++) {
            ret[k].v = lcode[list[i].s]; ret[k++].n = len_enc[list[i].s];
            if (list[i].s == 16) { ret[k].v = list[i].e; ret[k++].n = 2; }
            else if (list[i].s == 17) { ret[k].v = list[i].e; ret[k++].n = 3; }
            else if (list[i].s == 18) { ret[k].v = list[i].e; ret[k++].n = 7; }
        }
    }
    return k;
}

/* cals_comp_len :549-575 */
static uint64_t comp_len(const inner *s, const uint8_t *sym_tab, size_t sym_n, const uint8_t *off_tab, size_t off_n)
{
    uint64_t t = 0;
    for (size_t i = 0; i < sym_n && i < 286; i++)
        t += (uint64_t)s->sym_freq[i] * ((uint64_t)sym_tab[i] + (i >= 257 ? g_len_tab.ext_bits[i - 257] : 0));
    for (size_t i = 0; i < off_n && i < 30; i++)
        t += (uint64_t)s->off_freq[i] * ((uint64_t)off_tab[i] + g_off_tab.ext_bits[i]);
    return t;
}

/* write_block :454-547 */
static void write_block(inner *s, int is_final)
{
    const size_t bits0 = s->w.out.n * 8 + s->w.cnt;
    if (is_final) s->finished = 1;
    bw_bits(&s->w, is_final ? 1 : 0, 1);
    uint8_t sym_tab[286] = {0}, off_tab[30] = {0};
    size_t sym_n = 0, off_n = 0;
    bzo_make_tab_with_fn(s->sym_freq, 286, 15, 0, sym_tab, &sym_n);
    bzo_make_tab_with_fn(s->off_freq, 30, 15, 0, off_tab, &off_n);
    static bitvec hdr[1024];
    const size_t hdr_n = custom_header(sym_tab, sym_n, off_tab, off_n, hdr);
    uint64_t custom = comp_len(s, sym_tab, sym_n, off_tab, off_n);
    for (size_t i = 0; i < hdr_n; i++) custom += hdr[i].n;
    uint8_t fix_sym[288], fix_off[32]; /* deflate/mod.rs:15-25 */
    for (int i = 0; i < 288; i++) fix_sym[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    memset(fix_off, 5, sizeof(fix_off));
    const uint64_t fixed = comp_len(s, fix_sym, 288, fix_off, 32) + 2;
    const uint64_t original = ((uint64_t)s->decompress_len << 3) + 2 + 16 + 16;
    unsigned btype;
    if (original <= custom && original <= fixed) {
        btype = 0;
        bw_bits(&s->w, 0, 2);
        bw_pad(&s->w);
        bw_bits(&s->w, (uint16_t)s->decompress_len, 16);
        bw_bits(&s->w, (uint16_t)s->decompress_len ^ 0xFFFF, 16);
        for (size_t i = 1; i <= s->decompress_len; i++) /* nocomp_buf[decompress_len - i]: oldest first */
            bytes_push(&s->w.out, s->hist.p[s->hist.n - 1 - (s->decompress_len - i)]);
    } else {
        uint16_t sc[288] = {0}, oc[32] = {0};
        const uint8_t *sl, *ol;
        if (fixed <= custom) {
            btype = 1;
            bw_bits(&s->w, 1, 2);
            make_codes(fix_sym, 288, sc); make_codes(fix_off, 32, oc);
            sl = fix_sym; ol = fix_off;
        } else {
            btype = 2;
            for (size_t i = 0; i < hdr_n; i++) bw_bits(&s->w, hdr[i].v, hdr[i].n);
            make_codes(sym_tab, sym_n, sc); make_codes(off_tab, off_n, oc);
            sl = sym_tab; ol = off_tab;
        }
        for (size_t i = 0; i < s->buf_n; i++) {
            const dcode *c = &s->buf[i];
            bw_bits(&s->w, sc[c->sym], sl[c->sym]);
            if (c->sym >= 257) {
                bw_bits(&s->w, c->len_ext, c->len_bits);
                bw_bits(&s->w, oc[c->off_code], ol[c->off_code]);
                bw_bits(&s->w, c->off_ext, c->off_bits);
            }
        }
        bw_bits(&s->w, sc[256], sl[256]);
    }
    if (s->blocks_n == s->blocks_cap) {
        s->blocks_cap = s->blocks_cap ? s->blocks_cap * 2 : 64;
        s->blocks = (dfo_block_info *)realloc(s->blocks, s->blocks_cap * sizeof(dfo_block_info));
    }
    dfo_block_info *bi = &s->blocks[s->blocks_n++];
    bi->tokens = s->buf_n; bi->bytes = s->decompress_len; bi->btype = btype;
    bi->bits = s->w.out.n * 8 + s->w.cnt - bits0;
    init_block(s);
}

/* InflaterInner::next :577-636 */
static void inner_next(void *ctx, int is_ref, size_t len, size_t pos)
{
    inner *s = (inner *)ctx;
    const size_t next_len = is_ref ? len : 1;
    const size_t new_len = s->decompress_len + next_len;
    if ((new_len > MAX_BLOCK && s->decompress_len <= MAX_BLOCK && s->decompress_len != 0) || s->buf_n == MAX_BLOCK) {
        write_block(s, 0);
        s->decompress_len = next_len;
    } else {
        s->decompress_len = new_len;
    }
    dcode c;
    memset(&c, 0, sizeof(c));
    if (!is_ref) {
        bytes_push(&s->hist, (uint8_t)pos);
        c.sym = (uint16_t)pos;
        s->sym_freq[c.sym] += 1;
    } else {
        for (size_t i = 0; i < len; i++) bytes_push(&s->hist, s->hist.p[s->hist.n - 1 - pos]);
        unsigned code, ext, eb; /* from_with_codetab :64-82 */
        dfo_convert(0, (unsigned)(len - 3), &code, &ext, &eb);
        c.sym = (uint16_t)(code + 257); c.len_ext = (uint16_t)ext; c.len_bits = (uint8_t)eb;
        dfo_convert(1, (unsigned)pos, &code, &ext, &eb);
        c.off_code = (uint8_t)code; c.off_ext = (uint16_t)ext; c.off_bits = (uint8_t)eb;
        s->sym_freq[c.sym] += 1;
        s->off_freq[c.off_code] += 1;
    }
    s->buf[s->buf_n++] = c;
}

/* ---------------------------------------------------------------- Inflater (deflate/encoder.rs:92-260) */
typedef struct dfo_enc {
    dfo_lzss lz;
    inner in;
} dfo_enc;

DFO_EXPORT dfo_enc *dfo_enc_new(const uint8_t *dict, size_t dict_n)
{
    tabs_init();
    dfo_enc *e = (dfo_enc *)calloc(1, sizeof(dfo_enc));
    lz_init(&e->lz, 0, 0x8000, 258, 3, 3); /* :114-132 */
    e->lz.sink = inner_next; e->lz.sink_ctx = &e->in;
    e->in.buf = (dcode *)malloc(sizeof(dcode) * (MAX_BLOCK + 1));
    init_block(&e->in);
    if (dict_n) { /* with_dict :134-153, InflaterInner::with_dict :297-316 */
        const size_t start = dict_n - (0x8000 < dict_n ? 0x8000 : dict_n);
        slide_append(&e->lz, dict + start, dict_n - start);
        for (size_t i = 0; i < dict_n; i++) bytes_push(&e->in.hist, dict[i]);
    }
    return e;
}

/* feeds `n` bytes, then applies `action` at the end of this iterator: 0 Run, 1 Flush, 2 Finish */
DFO_EXPORT void dfo_enc_feed(dfo_enc *e, const uint8_t *in, size_t n, int action)
{
    for (size_t i = 0; i < n; i++) lz_next_in(&e->lz, in[i]);
    if (action == 1 || action == 2) {
        lz_flush(&e->lz);                                   /* lzss/encoder.rs:224-226 */
        if (!e->in.finished) write_block(&e->in, action == 2); /* flush / finish :638-660 */
        bw_pad(&e->in.w);                                   /* Inflater::next :236-246 */
    }
}

DFO_EXPORT size_t dfo_enc_output(dfo_enc *e, const uint8_t **p) { *p = e->in.w.out.p; return e->in.w.out.n; }
DFO_EXPORT size_t dfo_enc_blocks(dfo_enc *e, const dfo_block_info **p) { *p = e->in.blocks; return e->in.blocks_n; }

DFO_EXPORT void dfo_enc_free(dfo_enc *e)
{
    if (!e) return;
    lz_free(&e->lz);
    free(e->in.buf); free(e->in.hist.p); free(e->in.w.out.p); free(e->in.blocks);
    free(e);
}

/* ---------------------------------------------------------------- checksums */
DFO_EXPORT uint32_t dfo_adler32(const uint8_t *p, size_t n) /* adler32.rs:20-66 */
{
    uint32_t a = 1, b = 0;
    uint16_t t = 5549;
    for (size_t i = 0; i < n; i++) {
        a += p[i]; b += a;
        if (t == 0) { t = 5549; a %= 0xFFF1; b %= 0xFFF1; } else t -= 1;
    }
    return ((b % 0xFFF1) << 16) | (a % 0xFFF1);
}

DFO_EXPORT uint32_t dfo_crc32(const uint8_t *p, size_t n) /* crc32.rs:16-24, 40-55, 74-78: reflected 0xEDB88320 */
{
    static uint32_t tab[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1; tab[i] = c; }
        ready = 1;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}

/* ---------------------------------------------------------------- one-shot (Action::Finish) entries
 * kind 0: raw Deflate (Inflater), 1: zlib (ZlibEncoder, zlib/encoder.rs:55-157), 2: gzip
 * (GZipEncoder, gzip/encoder.rs:50-135).  Returns the stream length, or the needed capacity
 * negated when `cap` is too small. */
DFO_EXPORT long dfo_encode(int kind, const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_n, uint8_t *out, size_t cap)
{
    dfo_enc *e = dfo_enc_new(kind == 2 ? NULL : dict, kind == 2 ? 0 : dict_n);
    dfo_enc_feed(e, in, n, 2);
    const uint8_t *p;
    const size_t m = dfo_enc_output(e, &p);
    bytes o = {0, 0, 0};
    if (kind == 1) {
        if (dict_n) {
            const uint32_t h = dfo_adler32(dict, dict_n);
            bytes_push(&o, 0x78); bytes_push(&o, 0xF9);
            bytes_push(&o, (uint8_t)(h >> 24)); bytes_push(&o, (uint8_t)(h >> 16)); bytes_push(&o, (uint8_t)(h >> 8)); bytes_push(&o, (uint8_t)h);
        } else { bytes_push(&o, 0x78); bytes_push(&o, 0xDA); }
    } else if (kind == 2) {
        static const uint8_t hdr[10] = {0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0xFF};
        for (int i = 0; i < 10; i++) bytes_push(&o, hdr[i]);
    }
    for (size_t i = 0; i < m; i++) bytes_push(&o, p[i]);
    if (kind == 1) {
        const uint32_t h = dfo_adler32(in, n);
        bytes_push(&o, (uint8_t)(h >> 24)); bytes_push(&o, (uint8_t)(h >> 16)); bytes_push(&o, (uint8_t)(h >> 8)); bytes_push(&o, (uint8_t)h);
    } else if (kind == 2) {
        const uint32_t h = dfo_crc32(in, n);
        const uint32_t sz = (uint32_t)n;
        for (int i = 0; i < 4; i++) bytes_push(&o, (uint8_t)(h >> (8 * i)));
        for (int i = 0; i < 4; i++) bytes_push(&o, (uint8_t)(sz >> (8 * i)));
    }
    long ret;
    if (o.n <= cap) { memcpy(out, o.p, o.n); ret = (long)o.n; } else ret = -(long)o.n;
    free(o.p);
    dfo_enc_free(e);
    return ret;
}

/* ---------------------------------------------------------------- ZlibEncoder / GZipEncoder at the iterator level
 * ZlibEncoder::next (zlib/encoder.rs:118-152) and GZipEncoder::next (gzip/encoder.rs:88-135), byte by byte: the
 * header first; then the inner Inflater's bytes, pulled through a ScanIterator that feeds every input byte it
 * passes to the checksum; at the FIRST None of the inner encoder -- whatever the Action -- the checksum is
 * finished and the trailer follows (zlib: Adler-32 big endian; gzip: CRC-32 then ISIZE, little endian); from then
 * on the encoder yields None without touching the caller's iterator.  So Action::Run gives header + the whole
 * bytes of the blocks the Inflater has closed so far + trailer, Action::Flush header + the flushed segment +
 * trailer, and every later call nothing.  No reference test drives a wrapper with Run or Flush: this part of the
 * oracle is a restatement without a pin (like the mid-stream Flush of Inflater). */
typedef struct dfo_wrap {
    int kind;              /* 1 zlib, 2 gzip */
    dfo_enc *inner;        /* Inflater */
    size_t inner_pos;      /* bytes of the inner encoder's output already handed on */
    uint8_t header[10];
    unsigned header_n, header_len; /* header.len(), header_len (zlib/encoder.rs:58,126-129) */
    int has_hash;          /* hash: Option<u32> */
    uint32_t hash;
    unsigned hashlen, i_size_len;
    uint32_t i_size;
    bytes seen;            /* what the ScanIterator's closure has been shown (adler32.write_u8 / crc32.write_u8) */
} dfo_wrap;

DFO_EXPORT dfo_wrap *dfo_wrap_new(int kind, const uint8_t *dict, size_t dict_n)
{
    if (kind != 1 && kind != 2) return NULL;
    dfo_wrap *w = (dfo_wrap *)calloc(1, sizeof(dfo_wrap));
    w->kind = kind;
    if (kind == 1 && dict_n) { /* ZlibEncoder::with_dict zlib/encoder.rs:88-113 */
        const uint32_t h = dfo_adler32(dict, dict_n);
        const uint8_t hd[6] = {0x78, 0xF9, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h};
        memcpy(w->header, hd, 6); w->header_n = w->header_len = 6;
        w->inner = dfo_enc_new(dict, dict_n);
    } else if (kind == 1) {    /* ZlibEncoder::new :70-86 */
        w->header[0] = 0x78; w->header[1] = 0xDA; w->header_n = w->header_len = 2;
        w->inner = dfo_enc_new(NULL, 0);
    } else {                   /* GZipEncoder::new gzip/encoder.rs:66-80 */
        static const uint8_t hd[10] = {0x1F, 0x8B, 0x08, 0, 0, 0, 0, 0, 0, 0xFF};
        memcpy(w->header, hd, 10); w->header_n = w->header_len = 10;
        w->inner = dfo_enc_new(NULL, 0);
    }
    w->hashlen = 3; w->i_size_len = 4;
    return w;
}

DFO_EXPORT void dfo_wrap_free(dfo_wrap *w)
{
    if (!w) return;
    dfo_enc_free(w->inner);
    free(w->seen.p);
    free(w);
}

/* `in.iter().cloned().encode(&mut wrapper, action).collect()`: the bytes until next() returns None.  *pulled =
 * how many of the n bytes the encoder took from the iterator (all of them, or none once it is finished).
 * Returns the byte count, or the needed capacity negated. */
DFO_EXPORT long dfo_wrap_encode_iter(dfo_wrap *w, const uint8_t *in, size_t n, int action, uint8_t *out, size_t cap,
                                     size_t *pulled)
{
    bytes o = {0, 0, 0};
    int fed = 0;
    if (pulled) *pulled = 0;
    for (;;) {
        if (w->header_len > 0) {                       /* :125-129 */
            bytes_push(&o, w->header[w->header_n - w->header_len]);
            w->header_len -= 1;
        } else if (w->has_hash) {                      /* :130-136 / gzip :103-118 */
            if (w->hashlen == 0) {
                if (w->kind == 1 || w->i_size_len == 0) break; /* None */
                w->i_size_len -= 1;
                bytes_push(&o, (uint8_t)w->i_size);
                w->i_size >>= 8;
            } else {
                w->hashlen -= 1;
                if (w->kind == 1) bytes_push(&o, (uint8_t)(w->hash >> (w->hashlen << 3)));
                else { bytes_push(&o, (uint8_t)w->hash); w->hash >>= 8; }
            }
        } else {
            /* self.inflater.next(&mut ScanIterator::new(iter, ..), action) :138-145: the Inflater pulls the iterator
             * dry before it can return None (deflate/encoder.rs:206-259), the closure sees every byte on the way */
            if (!fed) {
                for (size_t i = 0; i < n; i++) bytes_push(&w->seen, in[i]);
                if (w->kind == 2) w->i_size += (uint32_t)n;
                dfo_enc_feed(w->inner, in, n, action);
                fed = 1;
                if (pulled) *pulled = n;
            }
            const uint8_t *p;
            const size_t m = dfo_enc_output(w->inner, &p);
            if (w->inner_pos < m) {
                bytes_push(&o, p[w->inner_pos++]);
            } else {                                   /* ret.is_none() :146-150 / gzip :129-133 */
                w->has_hash = 1;
                if (w->kind == 1) {
                    w->hash = dfo_adler32(w->seen.p, w->seen.n);
                    bytes_push(&o, (uint8_t)(w->hash >> 24));
                } else {
                    const uint32_t h = dfo_crc32(w->seen.p, w->seen.n);
                    bytes_push(&o, (uint8_t)h);
                    w->hash = h >> 8;
                }
            }
        }
    }
    long ret;
    if (o.n <= cap) { if (o.n) memcpy(out, o.p, o.n); ret = (long)o.n; } else ret = -(long)o.n;
    free(o.p);
    return ret;
}
