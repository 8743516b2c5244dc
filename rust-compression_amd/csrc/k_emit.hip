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

} // namespace bzgpu
