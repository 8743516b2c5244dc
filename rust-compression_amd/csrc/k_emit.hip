// k_emit.hip -- stream assembly: bit-granular concatenation of block bit strings,
// stream header and trailer.
//
// Reference being replaced: BitWriter<Left>::write_bits/flush (src/bitio/writer.rs:186-243,
// MSB-first, blocks are NOT byte aligned), the framing of write_block
// (src/bzip2/encoder.rs:245-251 "BZh<level>", :279-288 trailer + combined CRC) and the
// byte draining of BZip2Encoder::next (:149-157).
//
// A block bit string is an array of logical 32-bit words (bit 31 = first bit).  The output
// is assembled as logical words too and byte-swapped when stored, so the bytes in memory
// are the stream.
#include "bzgpu.h"
#include <chrono>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace bzgpu {

// grid: (ceil(max_words_per_block / 256), n_blocks)
__global__ __launch_bounds__(256) void k_assemble(const u32 *__restrict__ packed,
                                                   const AsmBlock *__restrict__ blocks,
                                                   u32 *__restrict__ out_words)
{
    const AsmBlock b = blocks[blockIdx.y];
    if (b.bit_len == 0) return;
    const u64 first_w = b.dst_bit >> 5;
    const u64 last_w = (b.dst_bit + b.bit_len - 1) >> 5;
    const u64 W = first_w + (u64)blockIdx.x * 256u + threadIdx.x;
    if (W > last_w) return;
    const u32 *src = packed + b.src_word;
    const long long sb = (long long)(W << 5) - (long long)b.dst_bit; // block bit held by this word's MSB
    u32 v;
    if (sb >= 0) {
        const u64 wi = (u64)sb >> 5;
        const u32 sh = (u32)sb & 31u;
        const u64 nwords = (b.bit_len + 31) >> 5;
        const u32 hi = src[wi];
        const u32 lo = (sh && wi + 1 < nwords) ? src[wi + 1] : 0u;
        v = sh ? ((hi << sh) | (lo >> (32u - sh))) : hi;
        const u64 remain = b.bit_len - (u64)sb; // valid bits from sb on
        if (remain < 32) v &= ~0u << (32u - (u32)remain);
    } else {
        const u32 s = (u32)(-sb); // 1..31 leading bits belong to whatever precedes the block
        v = src[0] >> s;
        const u64 room = 32u - s;
        if (b.bit_len < room) v &= (~0u << (32u - (u32)b.bit_len)) >> s;
    }
    const u32 stored = __builtin_bswap32(v);
    if (W == first_w || W == last_w) atomicOr(&out_words[W], stored);
    else out_words[W] = stored;
}

// stream header / carry-in / trailer: a handful of bits, one lane
__global__ void k_frame(u32 *__restrict__ out_words, int write_header, u32 level, u32 carry_bits,
                        u32 carry_byte, int write_trailer, u64 trailer_bit, u32 combined_crc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // put `nbits` of v at stream bit position pos
    auto put = [&](u64 pos, u32 v, u32 nbits) {
        for (u32 i = 0; i < nbits; ++i) {
            const u32 bit = (v >> (nbits - 1 - i)) & 1u;
            if (bit) {
                const u64 p = pos + i;
                atomicOr(&out_words[p >> 5], __builtin_bswap32(1u << (31u - (u32)(p & 31u))));
            }
        }
    };
    u64 pos = 0;
    if (carry_bits) {
        put(0, carry_byte >> (8u - carry_bits), carry_bits);
        pos = carry_bits;
    }
    if (write_header) {
        put(pos, 0x425A68u, 24);             // "BZh", encoder.rs:246-248
        put(pos + 24, 0x30u + level, 8);     // :249-250
    }
    if (write_trailer) {
        put(trailer_bit, 0x177245u, 24);     // :280-285
        put(trailer_bit + 24, 0x385090u, 24);
        put(trailer_bit + 48, combined_crc, 32); // :286-287
    }
}

__global__ __launch_bounds__(256) void k_pack(const u32 *__restrict__ src, const PackBlock *__restrict__ pb,
                                               u32 *__restrict__ dst)
{
    const PackBlock b = pb[blockIdx.y];
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < b.nwords; i += (u64)gridDim.x * 256u)
        dst[b.dst_word + i] = src[b.src_word + i];
}

void launch_assemble(hipStream_t st, const u32 *packed, const AsmBlock *d_blocks, u32 n_blocks,
                     u64 max_words, u32 *out_words)
{
    if (n_blocks == 0) return;
    const u32 gx = (u32)((max_words + 1 + 255) / 256);
    hipLaunchKernelGGL(k_assemble, dim3(gx, n_blocks), dim3(256), 0, st, packed, d_blocks, out_words);
}

void launch_frame(hipStream_t st, u32 *out_words, int write_header, u32 level, u32 carry_bits, u32 carry_byte,
                  int write_trailer, u64 trailer_bit, u32 combined_crc)
{
    hipLaunchKernelGGL(k_frame, dim3(1), dim3(64), 0, st, out_words, write_header, level, carry_bits,
                       carry_byte, write_trailer, trailer_bit, combined_crc);
}

void launch_pack(hipStream_t st, const u32 *src, const PackBlock *d_pb, u32 n_blocks, u32 *dst)
{
    if (n_blocks == 0) return;
    hipLaunchKernelGGL(k_pack, dim3(64, n_blocks), dim3(256), 0, st, src, d_pb, dst);
}

// ---------------------------------------------------------------------------------- a few bytes for the host
// The host side of a batch reads small results between launches -- how many blocks the cuts made, how many rotations a
// sort round left, the bit counts of the blocks -- and one block alone is a chain of ~25 such round trips.  What one
// costs (tools/ubench/roundtrip.hip, a 64-workgroup kernel in front): hipMemcpyAsync into pageable memory +
// hipStreamSynchronize 24.6 us, into pinned memory 16.0, a one-workgroup kernel that writes the bytes into host-mapped
// memory and raises a sequence word the host polls 10.1.  mail_fetch is the third: per stream a 16 KB mailbox in
// coherent pinned memory, up to four device ranges per trip.  When the word has arrived, everything in front of the
// kernel on that stream is done (the same promise hipStreamSynchronize makes); ranges that do not fit go the old way.
constexpr size_t kMailCap = 16384;
struct MailSegs {
    const u8 *d[4];
    u32 off[4], n[4];
    u32 nseg;
};
__global__ __launch_bounds__(256) void k_mail(u8 *mail, u32 *seq, MailSegs sg, u32 s)
{
    for (u32 k = 0; k < sg.nseg; ++k) {
        const u8 *src = sg.d[k];
        u8 *dst = mail + sg.off[k];
        const u32 n = sg.n[k];
        if ((((size_t)src | n) & 3u) == 0) {
            for (u32 i = threadIdx.x; i < n / 4u; i += 256u) reinterpret_cast<u32 *>(dst)[i] = reinterpret_cast<const u32 *>(src)[i];
        } else {
            for (u32 i = threadIdx.x; i < n; i += 256u) dst[i] = src[i];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(seq, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ... and a few bytes FROM the host: hipMemcpyAsync out of pageable memory stages and waits (16 us in front of the next
// launch for a 32-byte block record); up to 1 KB travel as a kernel's argument instead.
struct PokeBytes {
    u32 w[256];
};
__global__ __launch_bounds__(256) void k_poke(u8 *dst, PokeBytes pb, u32 n)
{
    const u32 i = threadIdx.x;
    if ((((size_t)dst | n) & 3u) == 0) {
        if (i < n / 4u) reinterpret_cast<u32 *>(dst)[i] = pb.w[i];
    } else {
        for (u32 b = i * 4u; b < n && b < i * 4u + 4u; ++b) dst[b] = (u8)(pb.w[i] >> (8u * (b & 3u)));
    }
}
static bool mailbox_off() // BZ_MAILBOX=0 (tests): hipMemcpyAsync + hipStreamSynchronize as in rounds 1-5, same bytes
{
    static const bool off = getenv("BZ_MAILBOX") && atoi(getenv("BZ_MAILBOX")) == 0;
    return off;
}
int mail_poke(hipStream_t st, void *d_dst, const void *h_src, size_t n)
{
    const bool off = mailbox_off();
    if (n == 0) return 0;
    if (n > sizeof(PokeBytes) || off) return hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, st) == hipSuccess ? 0 : -1;
    PokeBytes pb;
    memset(&pb, 0, sizeof(pb));
    memcpy(&pb, h_src, n);
    hipLaunchKernelGGL(k_poke, dim3(1), dim3(256), 0, st, (u8 *)d_dst, pb, (u32)n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
struct Mailbox {
    u8 *h = nullptr, *d = nullptr;   // the same memory, host and device address
    u32 *hseq = nullptr, *dseq = nullptr;
    u32 seq = 0;
};
static Mailbox *mailbox_of(hipStream_t st)
{
    static std::mutex mu;
    static std::unordered_map<hipStream_t, Mailbox *> boxes; // (a stream is driven by one host thread at a time)
    std::lock_guard<std::mutex> lk(mu);
    auto it = boxes.find(st);
    if (it != boxes.end()) return it->second;
    Mailbox *m = new Mailbox;
    void *p = nullptr;
    if (hipHostMalloc(&p, kMailCap + 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
        delete m;
        return nullptr;
    }
    m->h = (u8 *)p;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
        (void)hipHostFree(p);
        delete m;
        return nullptr;
    }
    m->d = (u8 *)dp;
    m->hseq = reinterpret_cast<u32 *>(m->h + kMailCap);
    m->dseq = reinterpret_cast<u32 *>(m->d + kMailCap);
    *m->hseq = 0;
    boxes[st] = m;
    return m;
}
int mail_fetch(hipStream_t st, const MailSeg *segs, int nseg)
{
    const bool off = mailbox_off();
    size_t total = 0;
    for (int k = 0; k < nseg; ++k) total += (segs[k].n + 15) & ~(size_t)15;
    Mailbox *m = (nseg <= 4 && total <= kMailCap && !off) ? mailbox_of(st) : nullptr;
    if (!m) {
        for (int k = 0; k < nseg; ++k)
            if (segs[k].n && hipMemcpyAsync(segs[k].h, segs[k].d, segs[k].n, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
        return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
    }
    MailSegs sg;
    sg.nseg = (u32)nseg;
    u32 o = 0;
    for (int k = 0; k < 4; ++k) {
        sg.d[k] = k < nseg ? (const u8 *)segs[k].d : nullptr;
        sg.n[k] = k < nseg ? (u32)segs[k].n : 0u;
        sg.off[k] = o;
        o += (sg.n[k] + 15u) & ~15u;
    }
    const u32 s = ++m->seq;
    hipLaunchKernelGGL(k_mail, dim3(1), dim3(256), 0, st, m->d, m->dseq, sg, s);
    if (hipGetLastError() != hipSuccess) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    u32 spins = 0;
    while (__atomic_load_n(m->hseq, __ATOMIC_ACQUIRE) != s) {
#if defined(__x86_64__)
        __builtin_ia32_pause(); // (the sibling hardware thread may be the caller's copy thread)
#endif
        if ((++spins & 0x3FFFFu) == 0) { // every ~ms: a stream that has failed or gone idle without the word will never raise it
            const hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) return -1;
            if (q == hipSuccess && __atomic_load_n(m->hseq, __ATOMIC_ACQUIRE) != s) return -1;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) return -1;
        }
    }
    for (int k = 0; k < nseg; ++k)
        if (segs[k].n) memcpy(segs[k].h, m->h + sg.off[k], segs[k].n);
    return 0;
}

} // namespace bzgpu
