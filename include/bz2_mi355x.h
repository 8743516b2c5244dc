/*
 * bz2_mi355x.h -- C ABI of the MI355X-native BZip2 block-encode path (sections
 * 1-2) and of the matching decode path (section 3).
 *
 * This is the drop-in boundary for ONE hot path of chalharu/rust-compression:
 *   `iter.encode(&mut BZip2Encoder::new(level), action)`  (BZip2 encode),
 * widened by its inverse `iter.decode(&mut BZip2Decoder::new())`.
 * The reference is pure Rust and has no FFI of its own; each entry point below
 * names the reference interface it replaces (paths relative to the reference
 * repo).  A Rust shim re-implements `Encoder::next` on top
 * of these calls; host/compression.hpp is the same shim in C++.
 *
 * Plain C types only: pointers, sizes, ints.  No torch / HIP types appear in
 * any signature; device pointers are `void*`/`const void*` and a HIP stream is
 * an opaque `void*` (0 = the engine's own stream).
 *
 * Stream ordering: an engine launches its kernels on its OWN HIP stream, and every device-pointer
 * call returns after that stream has drained.  The caller makes sure that the input behind a device
 * pointer is complete before the call (synchronise the stream that produced it), and does not hand
 * the engine a buffer that a stream-ordered allocator may still be reading on another stream (a
 * PyTorch tensor allocated right after asynchronous work that freed its inputs is such a buffer:
 * torch.cuda.synchronize() first).
 *
 * Results are bit-identical to the reference encoder (checked against the CPU
 * oracle in oracle/).  There is NO CPU fallback: without a gfx950 device every
 * compute entry point returns BZ_E_NOGPU.
 */
#ifndef BZ2_MI355X_H
#define BZ2_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------
 * 0 = ok.  -1..-3 mirror CompressionError::{DataError,UnexpectedEof,Unexpected}
 * (src/error.rs:10-15); -4/-5 are BZip2Error::{DataErrorMagicFirst,DataErrorMagic}
 * (src/bzip2/error.rs:5-11, produced by the decoder entry points of section 3).  The encoder can
 * only fail with BZ_E_UNEXPECTED in the reference (src/bzip2/encoder.rs:623);
 * HIP failures map to it too. */
#define BZ_OK 0
#define BZ_E_DATA (-1)
#define BZ_E_EOF (-2)
#define BZ_E_UNEXPECTED (-3)
#define BZ_E_MAGIC_FIRST (-4)
#define BZ_E_MAGIC (-5)
#define BZ_E_PARAM (-6)  /* invalid level: BZip2Encoder::new panics (src/bzip2/encoder.rs:59-61) */
#define BZ_E_NOGPU (-7)  /* no gfx950 device / HIP runtime unusable: the path fails loudly */
#define BZ_E_NOMEM (-8)
#define BZ_E_CAPACITY (-9) /* caller's output buffer too small */

/* src/action.rs:8-13 */
#define BZ_ACTION_RUN 0
#define BZ_ACTION_FLUSH 1
#define BZ_ACTION_FINISH 2

const char *bz_strerror(int code);
const char *bz_version(void);
/* number of usable gfx950 devices (0 when none; never initialises a context) */
int bz_device_count(void);

/* ========================================================================
 * 1. Streaming encoder context  ==  `BZip2Encoder`  (src/bzip2/encoder.rs:40-49)
 * ======================================================================== */
typedef struct bz_enc bz_enc;

/* BZip2Encoder::new(level) (src/bzip2/encoder.rs:58-72).  level 1..=9 else
 * BZ_E_PARAM (the shim turns that back into the reference's panic).
 * `device` is the HIP device ordinal.  The GPU is first touched lazily, at the
 * first call that has work for it. */
int bz_enc_create(bz_enc **out, int level, int device);

/* The same encoder over SEVERAL devices of one process (SURVEY.md 8(b) "Surface"; in the shims
 * `BZip2Encoder::with_devices(level, &[0, 1, ..])`, still src/bzip2/encoder.rs:58-72 for the caller, who drives it
 * through the unchanged Encoder::next of src/traits/encoder.rs:41-79).  Every entry of `devices` (HIP ordinals; an
 * ordinal may be listed more than once) gets two LANES (BZ_ENC_LANES in the environment: 2 .. 8) -- an engine with its
 * own streams and buffers each; the input is cut into chunks (BZ_ENC_CHUNK_MIB, default 192), chunk q is uploaded to and
 * encoded on lane q mod (2 n_devices), the unconsumed tail of a
 * chunk's input crosses to the next lane's device (hipMemcpyPeerAsync: xGMI between the GPUs of a node), and the
 * block bit strings are concatenated in chunk order with the BitWriter carry, the combined CRC and "a block has
 * been written" handed from chunk to chunk on the host.  The bytes are those of bz_enc_create for any device list
 * and any chunk size.  bz_enc_create(out, level, d) == bz_enc_create_multi(out, level, &d, 1). */
int bz_enc_create_multi(bz_enc **out, int level, const int *devices, int n_devices);

/* The input iterator yielded `n` more bytes (src/bzip2/encoder.rs:80-85,
 * EncoderInner::next :671-697).  Bytes are copied; complete blocks may be
 * encoded immediately. */
int bz_enc_write(bz_enc *e, const uint8_t *in, size_t n);

/* The input iterator returned None while the caller's Action is `action`
 * (src/bzip2/encoder.rs:86-110 and :129-146): Run = nothing is flushed,
 * Flush = current block without the pending run, then zero-pad to a byte,
 * Finish = pending run, last block, trailer, pad.  Runs the reference's
 * state machine up to the point where `Encoder::next` would return None. */
int bz_enc_end(bz_enc *e, int action);

/* Output bytes in stream order (the bytes `Encoder::next` hands out,
 * src/bzip2/encoder.rs:149-157).  Returns the number of bytes copied (0 = none
 * ready) or a negative status. */
long bz_enc_read(bz_enc *e, uint8_t *out, size_t cap);
/* bytes currently readable */
size_t bz_enc_pending(const bz_enc *e);

void bz_enc_destroy(bz_enc *e);

/* Self-check (opt-in; BZ_VERIFY=1 in the environment turns it on for every context and engine of the process).  The
 * reference encoder is sequential code that cannot emit a stream which does not decode to its input
 * (src/bzip2/encoder.rs:224-291); a GPU pipeline with look-back words, ticket counters and stream-ordered clears can,
 * if one of them is ever wrong.  With the check on, the blocks of every job are decoded again on the same device (the
 * decode path of section 3) and compared byte for byte with the input they cover BEFORE their bytes can be read; a job
 * that fails is encoded again without any look-back pass and checked again, and if that fails too bz_enc_write /
 * bz_enc_end return BZ_E_UNEXPECTED.  Costs about one decode per encode (encode 12.6 GB/s, decode 22 GB/s).
 * stats: [0] blocks checked, [1] jobs that failed the check and were redone, [2] redone jobs that failed again (the
 * context is in error), [3] nanoseconds spent checking. */
int bz_enc_set_verify(bz_enc *e, int on);
int bz_enc_verify_stats(bz_enc *e, uint64_t out[4]);

/* Where the time of a stream went (milliseconds, summed over its jobs; a diagnostic -- the bytes do not depend on it):
 * [0] the caller's copies into pinned staging memory  [1] the caller waiting for a free staging buffer
 * [2] SPLIT sections (compose + RLE1 + block cuts; serial from job to job)  [3] ENCODE (jobs side by side on the lanes)
 * [4] ASSEMBLE sections (serial)  [5] downloads  [6] number of jobs  [7] workers waiting for their turn in the two serial
 * sections.  bz_encode_buffer_last_phases: the same for the last one-shot call of the process. */
int bz_enc_phase_stats(bz_enc *e, double out_ms[8]);
int bz_encode_buffer_last_phases(double out_ms[8]);

/* One-shot over host buffers:
 * `in.iter().cloned().encode(&mut BZip2Encoder::new(level), Action::Finish).collect()`.
 * *out is malloc'ed by the library; release with bz_free. */
int bz_encode_buffer(int level, int device, const uint8_t *in, size_t in_len,
                     uint8_t **out, size_t *out_len);
/* ... over several devices (see bz_enc_create_multi).  A one-shot call knows its length: it cuts the input into
 * equal chunks, a whole number of rounds over the lanes. */
int bz_encode_buffer_multi(int level, const int *devices, int n_devices, const uint8_t *in, size_t in_len,
                           uint8_t **out, size_t *out_len);
void bz_free(void *p);
/* Contexts and one-shot calls park their engines (batch workspace: about 31.5 MB of HBM per block of the largest
 * chunk seen, i.e. up to ~8 GB per lane with the default 192 MiB chunks, two lanes per listed device), device staging
 * buffers and 2 x BZ_ENC_CHUNK_MIB of pinned host memory in a per-process cache (two device lists at most) when they
 * end, so that the next one does not pay hipMalloc / hipHostMalloc again (fresh device memory costs about 40 ms per
 * GiB on this platform: the FIRST 1 GiB call of a process takes 0.35 s, a later one 0.1 s).  This call releases what
 * is parked (call it between contexts to run without the cache).  The decode and Deflate entry points over host
 * buffers (bz_decode_buffer, bz_dec_*, df_encode_buffer, df_enc_*) park up to TWO engines per device the same way (the
 * lanes of a streaming decoder; a decode workspace is about 13 MB of HBM per block of the largest stream seen; a call
 * takes an engine that holds its kind of workspace, else the one parked last), and bz_dec_* contexts leave up to four of
 * their chunk buffers (64 MiB of host memory each by default): released here too. */
void bz_release_cached_resources(void);
/* Diagnostic for hosts with several devices (no reference counterpart: the reference has no devices).  What a context
 * over `devices` does when a job's unconsumed tail changes device -- hipMemcpyPeerAsync, across xGMI where peer access
 * can be enabled, through the host where not -- done once per neighbour pair devices[i] -> devices[i + 1 mod n] with
 * `bytes` (1 ... 2^30) of a known pattern, there and back, compared on the host.  peer_access[i] (may be NULL): 1 direct
 * access enabled, 0 not offered, -1 both on one device (nothing copied); out_ms[i] (may be NULL): the round trip's wall
 * time, -1 if the pair failed.  BZ_OK when every pair's bytes came back intact; a failing pair is named on stderr. */
int bz_peer_copy_selftest(const int *devices, int n_devices, size_t bytes, int *peer_access, double *out_ms);

/* ========================================================================
 * 2. Device-resident engine (what bz_enc drives; also the bench / multi-GPU
 *    surface: input and output stay in HBM).
 *    Stages == src/bzip2/encoder.rs: RLE1+split+CRC (:671-716), write_blockdata
 *    (:300-639: BWT src/suffix_array/sais.rs:266, MTF src/bzip2/mtf.rs:16-39,
 *    ZLE :653-669, table selection :370-509, code lengths
 *    src/huffman/cano_huff_table.rs:198-225, canonical codes
 *    src/huffman/mod.rs:22-67, emission :527-629) and stream framing
 *    write_block (:224-291) + BitWriter<Left> (src/bitio/writer.rs:186-243).
 * ======================================================================== */
typedef struct bz_gpu_engine bz_gpu_engine;

/* max_blocks_in_flight bounds the workspace (about 26 MB of HBM per 900 KB
 * block); inputs with more blocks are processed in several batches. */
int bz_gpu_engine_create(bz_gpu_engine **out, int device, size_t max_blocks_in_flight);
void bz_gpu_engine_destroy(bz_gpu_engine *g);
/* The batch workspace (about 31.5 MB per block in flight) is made by the first call that needs it and grows with the
 * largest call seen; a caller who knows the size of the calls to come reserves it once (fresh device memory costs about
 * 40 ms per GiB here: growing means freeing and paying again). */
int bz_gpu_engine_reserve(bz_gpu_engine *g, size_t blocks);

/* The self-check of section 1 for the device-resident calls: the blocks of every bz_gpu_encode_blocks (and so of
 * bz_gpu_encode_device / bz_gpu_encode_sharded on each rank) are decoded on the device and compared with the input they
 * cover before the call returns; stats as for bz_enc_verify_stats, since the engine's creation. */
int bz_gpu_engine_set_verify(bz_gpu_engine *g, int on);
int bz_gpu_verify_stats(bz_gpu_engine *g, uint64_t out[4]);

/* Upper bound of the .bz2 size for n input bytes (for sizing d_out). */
size_t bz_encode_bound(size_t n);

/* Whole stream on one GPU: d_in[n] (HBM, 16-byte aligned) -> d_out (HBM,
 * 4-byte aligned, cap bytes).  *out_len receives the stream length.
 * Equivalent to the one-shot above minus the PCIe copies. */
int bz_gpu_encode_device(bz_gpu_engine *g, int level, const void *d_in, size_t n,
                         void *d_out, size_t cap, size_t *out_len);

/* ---- the same, split into the three steps a multi-GPU job needs ---------- */

/* (a) RLE1 + block split over the whole input (every rank runs it: it is
 * <2 % of the work and keeps block boundaries a pure function of the input).
 * mode: BZ_ACTION_FINISH = all chunks, tail block emitted;
 *       BZ_ACTION_FLUSH  = like FINISH over the bytes given (the caller has
 *                          already removed the pending run);
 *       BZ_ACTION_RUN    = only blocks closed by a cut; *consumed tells how
 *                          many input bytes they cover.
 * Returns the number of blocks in *n_blocks; *tail_block = 1 when the last of
 * them is the unfinished tail rather than a block closed by a cut (the
 * streaming context needs that to replay write_block's call sequence). */
int bz_gpu_partition(bz_gpu_engine *g, int level, const void *d_in, size_t n, int mode,
                     size_t *n_blocks, size_t *consumed, int *tail_block);

/* (a') The same split sharded over ranks by SLABS of 4 KiB input tiles (rank r owns tiles
 * [tile0, tile1); every rank sees the whole input d_in[n] but touches only its slab plus the tail
 * of the block that straddles its left edge).  Three steps with two tiny exchanges in between:
 *   _slab_begin  run starts + tile CRCs of the slab; *slab_last_start = last run start inside it
 *                (-1: none).              [exchange: all-gather of that one value]
 *   _slab_count  carry_run = the last run start BEFORE the slab (max over lower ranks, -1: none);
 *                RLE1 byte counts of the slab.
 *   _slab_finish start_in = first input byte of this rank's first block (rank 0: 0; else the
 *                next_in handed over by rank-1: a serial chain of one 8-byte message per rank);
 *                is_last: the last rank also emits the unfinished tail block.  Produces this rank's
 *                blocks (stream order = rank order) for bz_gpu_encode_blocks(g, 0, 1, ...).
 * RLE1 restarted at a block cut equals RLE1 continued (a cut is a chunk start, encoder.rs:689-693),
 * so a rank codes its first block afresh from the cut it is handed.
 * bz_gpu_partition == these three calls over all tiles. */
int bz_gpu_partition_slab_begin(bz_gpu_engine *g, int level, const void *d_in, size_t n,
                                uint64_t tile0, uint64_t tile1, int64_t *slab_last_start);
int bz_gpu_partition_slab_count(bz_gpu_engine *g, int64_t carry_run);
int bz_gpu_partition_slab_finish(bz_gpu_engine *g, uint64_t start_in, int is_last, size_t *n_blocks,
                                 uint64_t *next_in, int *tail_block);

/* Blocks of the last partition (what *n_blocks returned): sizes the host arrays of (b). */
size_t bz_gpu_block_count(const bz_gpu_engine *g);

/* (b) Encode blocks first, first+stride, ... (< n_blocks) of the last
 * partition.  Each block's bit string (block magic .. last payload bit,
 * MSB-first) is appended to d_packed as host-endian uint32 words whose bit 31
 * is the earliest bit, zero padded to a whole word (cap_words words).  For the k-th local block:
 * h_word_off[k] = first word in d_packed, h_bit_len[k] = length in bits,
 * h_crc[k] = block CRC.  Arrays are HOST arrays with room for the local
 * block count ceil((n_blocks-first)/stride). *words_used = words written. */
int bz_gpu_encode_blocks(bz_gpu_engine *g, size_t first, size_t stride,
                         void *d_packed, size_t cap_words,
                         uint64_t *h_word_off, uint64_t *h_bit_len, uint32_t *h_crc,
                         size_t *words_used);

/* (c) Assemble a stream from block bit strings held in d_packed (in STREAM
 * order k = 0..n_blocks-1, described by the three host arrays): optional
 * "BZh<level>" header, blocks back to back at bit granularity, optional
 * trailer (0x177245385090 + combined CRC) and zero padding to a byte.
 * `carry_bits` (0..7) bits of `carry_byte` (left aligned) precede the
 * output -- the BitWriter carry of a previous call; *out_carry_* return the
 * new carry when no padding is requested.  combined_crc_in/out carry
 * src/bzip2/encoder.rs:237-238 across calls. */
int bz_gpu_assemble(bz_gpu_engine *g, int level, size_t n_blocks,
                    const void *d_packed, const uint64_t *h_word_off,
                    const uint64_t *h_bit_len, const uint32_t *h_crc,
                    int write_header, int write_trailer, int pad_to_byte,
                    unsigned carry_bits, unsigned carry_byte,
                    uint32_t combined_crc_in, uint32_t *combined_crc_out,
                    void *d_out, size_t cap, size_t *out_len,
                    unsigned *out_carry_bits, unsigned *out_carry_byte);

/* ---- the whole stream over several GPUs (BASELINE.json configs[2]) ----------------
 * One process (one engine) per GPU; every rank calls bz_gpu_encode_sharded with the same level and n.
 * Rank r splits its SLAB of the input -- a contiguous range of the 4 KiB tiles, bz_shard_slab_tiles: nearly equal
 * shares, shrinking a little from rank to rank (BZ_SHARD_SKEW) because rank r starts its blocks r links of
 * the cut chain later than rank 0 --, encodes the blocks that END in its slab
 * (stream order == rank order) and the block bit strings are gathered to rank 0, which assembles the
 * serial stream (bz_gpu_assemble).  The cuts (encoder.rs:692) are the only serial dependency between the slabs, and a
 * rank prepares its link of that chain before the cut of the rank in front arrives: with the slabs' image sizes
 * all-gathered every rank knows where its slab lies in the RLE1 image of the whole input, the k-th block of the whole
 * input can only start at one of 4 k + 1 offsets of that image, and the rank resolves the cut behind every such start
 * inside its slab in parallel (tables; bz_gpu_cut_stats).  The link itself is then about a hundred table look-ups.  d_in is addressed as the whole input, but a rank only reads its
 * slab and, in front of it, the input bytes of the block that straddles its left edge (a level-9
 * block covers at most 900000 * 255 / 5 = 45.9 MB of input), so the rest need not be backed by memory.
 * The transport is the caller's: RCCL (ncclAllGather / ncclSend / ncclRecv over xGMI), MPI or
 * torch.distributed behind four C callbacks, each returning 0 on success.  Every rank takes part in
 * every exchange; a rank-local error is carried in the exchanged status words and returned by ALL
 * ranks (no rank is left waiting in a collective).
 *   allgather(ctx, send, bytes, recv)   HOST memory: `bytes` bytes of every rank, in rank order, into recv
 *   send(ctx, dst, buf, bytes) / recv(ctx, src, buf, bytes)   HOST memory, 32 bytes along the cut chain
 *   gatherv(ctx, d_send, send_bytes, d_recv, recv_off, recv_bytes)   DEVICE memory: rank r's send_bytes
 *       bytes land at d_recv + recv_off[r] on rank 0 (recv_bytes[r] == that rank's send_bytes; the
 *       two arrays are valid on every rank, d_recv only on rank 0; rank 0's own part included).  The transfer
 *       must be OVER when the callback returns: the library reads d_recv (rank 0) and overwrites d_send (every
 *       rank, at its next call) on its own stream right away; a transport that queues the transfer on a stream
 *       synchronises that stream first (an RCCL request's wait only orders streams).
 * d_packed / d_gather: optional caller-owned device buffers (e.g. registered with the transport) for
 * this rank's bit strings and, on rank 0, everybody's; NULL = the engine's own.  On rank 0 *out_len
 * receives the stream length and d_out the stream; the other ranks get *out_len = 0. */
typedef struct bz_shard_comm {
    void *ctx;
    int rank, world;
    int (*allgather)(void *ctx, const void *send, size_t bytes, void *recv);
    int (*send)(void *ctx, int dst, const void *buf, size_t bytes);
    int (*recv)(void *ctx, int src, void *buf, size_t bytes);
    int (*gatherv)(void *ctx, const void *d_send, size_t send_bytes, void *d_recv,
                   const uint64_t *recv_off, const uint64_t *recv_bytes);
} bz_shard_comm;
int bz_gpu_encode_sharded(bz_gpu_engine *g, int level, const void *d_in, size_t n,
                          const bz_shard_comm *comm, void *d_packed, size_t packed_cap_words,
                          void *d_gather, size_t gather_cap_words, void *d_out, size_t cap,
                          size_t *out_len);

/* The same call for a rank that holds only a WINDOW of the input (a rank of a large job need not back the bytes it
 * never reads): d_window[0 .. window_bytes) are the input bytes [window_off, window_off + window_bytes) of the n-byte
 * input (d_window 16-byte aligned, window_off a multiple of 16).  The window must hold the rank's slab, the bytes of
 * the block that straddles its left edge (from the start of their 4 KiB tile on) and the 4 KiB tile behind the slab (or
 * all there is): bz_shard_window returns one that always does -- the slab plus bz_shard_halo_bytes(level) in front (a
 * level-9 block covers at most 899981 * 255 / 5 = 45.9 MB of input) and one tile behind.  A window that turns out
 * too short (the cut handed over by the rank before lies in front of it) makes every rank return BZ_E_CAPACITY.
 * bz_gpu_encode_sharded(g, level, d_in, n, ...) == bz_gpu_encode_sharded_window(g, level, d_in, 0, n, n, ...). */
int bz_shard_slab_tiles(size_t n, int rank, int world, uint64_t *tile0, uint64_t *tile1); /* the rank's tiles [tile0, tile1) */
size_t bz_shard_halo_bytes(int level);
int bz_shard_window(int level, size_t n, int rank, int world, uint64_t *window_off, size_t *window_bytes);
int bz_gpu_encode_sharded_window(bz_gpu_engine *g, int level, const void *d_window, uint64_t window_off,
                                 size_t window_bytes, size_t n, const bz_shard_comm *comm, void *d_packed,
                                 size_t packed_cap_words, void *d_gather, size_t gather_cap_words,
                                 void *d_out, size_t cap, size_t *out_len);

/* Where the last bz_gpu_encode_sharded[_window] call of this rank spent its wall time around the exchanges (ms):
 * [0] waiting for the cut from the rank before (rank 0: 0)  [1] from that cut's arrival to the hand-on of this rank's own
 * cut -- its LINK of the one serial chain across the ranks (left halo, tile offsets, the cut chain over its slab)
 * [2] the gather of the bit strings  [3] the assembly (rank 0). */
int bz_gpu_last_shard_timings(bz_gpu_engine *g, double out_ms[4]);
/* The same four and: [4] entry -> ready for the cut of the rank in front (scan, counts, the slab's image offsets, its
 * image and the cut tables under way: nothing of it waits for another rank's cuts), [5] the cuts themselves once the
 * hop is there (table look-ups, or the chain kernel), [6] unused, [7] the whole call. */
int bz_gpu_last_shard_phases(bz_gpu_engine *g, double out_ms[8]);

/* A ready-made transport: the four callbacks over RCCL (xGMI inside a node).  These three entry points are
 * exported by a SECOND library, libbz2_mi355x_rccl.so (it links librccl; the codec library does not):
 * rank 0 draws an id and ships its BZ_RCCL_ID_BYTES bytes to the other ranks by any means; every rank then
 * creates its communicator (collective: ncclCommInitRank) on the device its engine lives on.  Small host-byte
 * exchanges go through a pinned + a device staging buffer, the bit strings travel as one group of
 * ncclSend / ncclRecv between the callers' device buffers. */
#define BZ_RCCL_ID_BYTES 128
int bz_rccl_unique_id(uint8_t id[BZ_RCCL_ID_BYTES]);
int bz_rccl_comm_create(bz_shard_comm **out, const uint8_t id[BZ_RCCL_ID_BYTES], int rank, int world, int device);
void bz_rccl_comm_destroy(bz_shard_comm *comm);
/* ranks of the communicator as RCCL counts them (ncclCommCount), or a negative status */
int bz_rccl_comm_count(const bz_shard_comm *comm);
/* No exchange of this transport waits for ever: each one polls its stream, the communicator's asynchronous
 * error state and a deadline (BZ_RCCL_TIMEOUT_S in the environment, default 300 s); on an error or a timeout
 * the communicator is aborted (ncclCommAbort), the callback fails and bz_gpu_encode_sharded returns
 * BZ_E_UNEXPECTED on this rank. */

/* Runs known patterns through a transport's four callbacks, shaped like the exchanges above (every
 * rank calls it): BZ_OK, BZ_E_DATA (bytes arrived wrong somewhere; the same verdict on every rank) or
 * BZ_E_UNEXPECTED (a callback failed).  host_memory != 0: the buffers handed to gatherv are host
 * memory (a CPU transport under test; needs no GPU), else device memory. */
int bz_shard_comm_selftest(const bz_shard_comm *comm, int host_memory);

/* Seconds of GPU time spent in the kernels of the last bz_gpu_encode_device /
 * bz_gpu_encode_blocks call, by stage (HIP events on the engine's stream):
 * [0] rle1+crc+split [1] bwt [2] mtf+zle [3] huffman [4] emit+assemble [5] total. */
int bz_gpu_last_timings(bz_gpu_engine *g, double out_seconds[6]);
/* [0] BWT rounds executed (prefix doubling), [1] total sorted elements, [2] batches of the last call; [3] sorts of
 * this engine since its creation that fell back from the fused radix passes to the three-kernel passes (a
 * look-back gave up or tile tickets were not handed out evenly: the engine then stays on the three-kernel passes) */
int bz_gpu_last_bwt_stats(bz_gpu_engine *g, uint64_t out[4]);
/* The block cuts (encoder.rs:692), since the engine's creation: [0] partitions whose cuts came from the tables of
 * candidate cuts (every cut a block start can lead to, resolved in parallel; BZ_CUT_TABLES=0 turns them off),
 * [1] partitions that took the chain kernel instead although the tables were asked (never seen). */
int bz_gpu_cut_stats(bz_gpu_engine *g, uint64_t out[2]);
/* rotations still unordered after the initial 4-byte sort (out[0]) and after each doubling
 * round (out[1..]), summed over the blocks of the last encode */
int bz_gpu_last_bwt_rounds(bz_gpu_engine *g, uint64_t out[64]);

/* Per-kernel timing of the BWT kernels (HIP events around every launch, on the engine's
 * stream).  Enable, run encodes, then read: kernel name (all template instances of one
 * kernel are pooled), launches, summed duration, summed ALGORITHMIC bytes.
 * bz_gpu_profile_enable also clears the counters.  `on`: bit 0 the kernel timing, bit 1 the per-pass figures of
 * bz_gpu_debug_block_sections (a few ballots per group in the Huffman sweeps: off by default). */
int bz_gpu_profile_enable(bz_gpu_engine *g, int on);
int bz_gpu_profile_kernels(bz_gpu_engine *g);
int bz_gpu_profile_get(bz_gpu_engine *g, int idx, const char **name, uint64_t *launches,
                       double *seconds, uint64_t *algorithmic_bytes);

/* ---- stage probes for the parity tests (device results copied to host) ---- */

/* Rotation order of ONE block (src/suffix_array/sais.rs:266 `bwt`). */
int bz_gpu_debug_bwt(bz_gpu_engine *g, const uint8_t *h_block, size_t n, uint32_t *h_sa);
/* Code lengths of one table (EncoderInner::create_huffman,
 * src/bzip2/encoder.rs:641-651) through the device code.  alpha <= 258 and a total of
 * less than 2^20 occurrences (a block holds 900 001 symbols at most; the device's weights
 * are 32-bit, the reference's usize: equal below 2^24 occurrences per package), else
 * BZ_E_PARAM. */
int bz_gpu_debug_code_lengths(bz_gpu_engine *g, const uint32_t *h_freq, size_t alpha,
                              uint8_t *h_len, int *took_length_limited_path);
/* Per-block statistics of the last bz_gpu_encode_blocks / encode_device call,
 * the numbers behind the reference's log::debug! lines
 * (src/bzip2/encoder.rs:240-243, :360-365): 8 x uint32 per block:
 * nblock, crc, origPtr, mtf_count, in_use_count, group_num, n_selectors, max_len */
int bz_gpu_debug_block_stats(bz_gpu_engine *g, uint32_t *h_stats, size_t cap_blocks,
                             size_t *n_blocks);
/* ... and the figures behind the other two debug lines of write_blockdata
 * (src/bzip2/encoder.rs:483-498 "pass k: size is .., grp uses are ..", :556-636
 * "bits: mapping .., selectors .., code lengths .., codes .."): 32 x uint32 per
 * block: [0..3] totc / 8 of the four refinement passes, [4 + 6 k + t] groups that
 * chose table t in pass k, [28] bits of the mapping table, [29] of the
 * selectors, [30] of the code lengths, [31] of the symbols.  The per-pass
 * figures need bz_gpu_profile_enable(g, 2) in force during the encode (else 0). */
int bz_gpu_debug_block_sections(bz_gpu_engine *g, uint32_t *h_sections, size_t cap_blocks,
                                size_t *n_blocks);

/* ========================================================================
 * 3. Decoder  ==  `BZip2Decoder`  (src/bzip2/decoder.rs:583-612; the work is
 *    BZip2DecoderBase::init_block / next, src/bzip2/decoder.rs:163-581)
 *
 * The decode calls return the DECODER'S verdict: BZ_OK, BZ_E_DATA
 * (BZip2Error::DataError), BZ_E_MAGIC_FIRST / BZ_E_MAGIC (bad level digit of
 * the first / a later stream header), or an infrastructure status
 * (BZ_E_NOGPU, BZ_E_NOMEM, BZ_E_CAPACITY, BZ_E_UNEXPECTED).  With a decoder
 * error the bytes of all blocks in front of the failing record are still
 * produced -- exactly the items the reference's iterator yields before its
 * Err (a block whose CRC is wrong is handed out first, src/bzip2/decoder.rs:189-201).
 * Multi-stream files decode to the concatenation (src/bzip2/decoder.rs:503-516).
 * ======================================================================== */

/* Whole file on one GPU: d_in[n] (HBM, 4-byte aligned) -> d_out (HBM, cap
 * bytes).  *out_len = bytes decoded.  d_out == NULL: sizes only (nothing is
 * written, CRCs are not checked), to learn the capacity a second call needs. */
int bz_gpu_decode_device(bz_gpu_engine *g, const void *d_in, size_t n,
                         void *d_out, size_t cap, size_t *out_len);
/* The same over several GPUs (one process per GPU, every rank holds the whole
 * compressed file): rank r decodes the r-th contiguous share of the blocks
 * into ITS OWN d_out.  *out_len = bytes of this rank's slice, *out_offset =
 * where the slice sits in the decoded file, *total_len = decoded bytes over all
 * ranks; the verdict is the same on every rank.  `allgather` is the only
 * collective the path needs (called twice, with a few bytes per block): it must
 * copy `bytes` bytes from `send` of every rank, in rank order, into `recv`
 * (world * bytes) -- RCCL, MPI or torch.distributed behind a C callback; return
 * 0 on success.  Blocks must carry their full 48-bit magic (every encoder's do). */
typedef int (*bz_allgather_fn)(void *ctx, const void *send, size_t bytes, void *recv);
int bz_gpu_decode_device_sharded(bz_gpu_engine *g, const void *d_in, size_t n,
                                 void *d_out, size_t cap, int rank, int world,
                                 bz_allgather_fn allgather, void *ctx,
                                 size_t *out_len, size_t *out_offset, size_t *total_len);
/* Seconds of GPU time of the last decode by stage (HIP events):
 * [0] magic scan + Huffman [1] zero runs + inverse MTF [2] inverse BWT
 * [3] RLE1 undo + CRC [4] total. */
int bz_gpu_last_decode_timings(bz_gpu_engine *g, double out_seconds[5]);
/* [0] block-magic candidates [1] blocks decoded [2] streams [3] blocks that
 * started without the full 48-bit magic */
int bz_gpu_last_decode_stats(bz_gpu_engine *g, uint64_t out[4]);

/* One-shot over host buffers: `in.iter().cloned().decode(&mut BZip2Decoder::new())`
 * collected until None or the first Err.  *out (malloc'ed, release with
 * bz_free) holds the bytes yielded before the verdict, also when that is an error.
 * 1 GiB of decoded bytes in 0.068 s (15.6 GB/s; rounds 1-4: 0.11 s) from the second call of a process on: the engine
 * and its workspace are kept between calls (bz_release_cached_resources), the bytes land once, in huge-page-backed memory,
 * sub-batch by sub-batch beside the kernels of the next one. */
int bz_decode_buffer(int device, const uint8_t *in, size_t in_len,
                     uint8_t **out, size_t *out_len);

/* Streaming context == BZip2Decoder as the DecodeIterator drives it
 * (src/traits/decoder.rs:73-86): compressed bytes in (bz_dec_write), end of
 * the input iterator (bz_dec_end), decoded bytes out in order (bz_dec_read:
 * > 0 bytes copied; 0 = nothing ready yet; once the verdict is final and
 * nothing is left, the verdict -- 0 = `None`, negative = the `Err` item).
 * Decoding is incremental and runs BESIDE the caller, on a thread of the
 * context: whenever BZ_DEC_CHUNK bytes (environment; default 64 MiB, the first
 * chunk of a stream 16 MiB) have been written, the records that are wholly
 * there are decoded and their bytes queued as they land in host memory, and the
 * chain state (bit position, stream number, level, combined CRC) is carried to
 * the next chunk; chunks of less than 4 MiB are decoded before bz_dec_write
 * returns.  bz_dec_end hands over the rest and does not wait for the decode --
 * it returns the verdict if it is already final, else BZ_OK; like bz_dec_write
 * it waits only while two chunks are already queued, and for a last chunk of
 * less than 4 MiB, which is decoded inside the call -- and from then on
 * bz_dec_read WAITS for the next bytes or the final verdict instead of
 * answering "nothing yet" (a caller that polls between other work must not
 * call it behind bz_dec_end until it can afford to block): a consumer reads the
 * head of the file while its tail is being decoded (the verdict follows the
 * last byte, as the reference's iterator yields it).  INPUT memory is bounded
 * by the chunks in flight (three); decoded bytes wait in host segments until
 * they are read, without a limit -- a caller that writes a whole file before
 * its first read holds the whole decoded output (back-pressure on the worker
 * would dead-lock exactly that caller: its writes wait for the worker).
 * The reference decodes lazily block by block -- same items, coarser moments.
 * 1 GiB written in 1 MiB pieces and read in 4 MiB pieces: see bench.py
 * extra.decode.end_to_end.streaming (rounds 1-4, which decoded inside
 * bz_dec_write: 4.2 GB/s). */
typedef struct bz_dec bz_dec;
int bz_dec_create(bz_dec **out, int device);        /* BZip2Decoder::new, src/bzip2/decoder.rs:588-594 */
int bz_dec_write(bz_dec *d, const uint8_t *data, size_t n);
int bz_dec_end(bz_dec *d);
long bz_dec_read(bz_dec *d, uint8_t *out, size_t cap);
size_t bz_dec_pending(const bz_dec *d);
void bz_dec_destroy(bz_dec *d);

/* ========================================================================
 * 4. Deflate / zlib / gzip ENCODE
 *
 * Replaces `Inflater` (src/deflate/encoder.rs:92-260: LzssEncoder with window
 * 0x8000, matches 3..258, lazy level 3, hash chains of 255, src/lzss/encoder.rs,
 * src/lzss/slidedict.rs; blocks of <= 0xFFFF bytes, stored / fixed / dynamic
 * chosen by size, src/deflate/encoder.rs:454-547), `ZlibEncoder`
 * (src/zlib/encoder.rs:55-157: 78 DA, Adler-32 big endian) and `GZipEncoder`
 * (src/gzip/encoder.rs:50-135: 10-byte header, CRC-32 and ISIZE little endian),
 * driven with Action::Finish.  The stream is the reference's, bit for bit.
 * `kind`: 0 raw Deflate, 1 zlib, 2 gzip.
 * Action::Flush inside a stream is offered for Inflater through the streaming context (df_enc_end):
 * the stream becomes a sequence of byte-aligned segments (src/deflate/encoder.rs:170-195, :227-235,
 * :638-647).  The zlib / gzip wrappers end their container at the first None of their inner Inflater whatever
 * the Action (src/zlib/encoder.rs:138-150, src/gzip/encoder.rs:120-133): df_enc_end(Run) / (Flush) on kinds 1 / 2
 * write header + what the Inflater yields under that Action + trailer, and the context is finished.  Inputs of
 * any length: a call works through a long segment in parts of BZ_DF_PART_MIB (default 1024) MiB.
 * ======================================================================== */
#define DF_KIND_DEFLATE 0
#define DF_KIND_ZLIB 1
#define DF_KIND_GZIP 2

/* Upper bound of the stream length for n input bytes (every block stored). */
size_t df_encode_bound(size_t n);
/* d_in[n] (HBM, 4-byte aligned) -> d_out (HBM, cap bytes); *out_len = stream
 * bytes.  d_out == NULL: length only.  Uses the engine's stream; workspace
 * (about 20 bytes per input byte) is created by the first call and kept. */
int df_gpu_encode_device(bz_gpu_engine *g, int kind, const void *d_in, size_t n,
                         void *d_out, size_t cap, size_t *out_len);
/* The same with a preset dictionary == Inflater::with_dict / ZlibEncoder::with_dict
 * (src/deflate/encoder.rs:134-153, src/zlib/encoder.rs:74-93): `dict` is HOST
 * memory; its last 0x8000 bytes are the window in front of the input
 * (src/lzss/encoder.rs:104-130), the zlib header becomes 78 F9 + Adler-32 of the
 * whole dictionary.  kind 2 (gzip) has no with_dict: BZ_E_PARAM. */
int df_gpu_encode_device_dict(bz_gpu_engine *g, int kind, const void *d_in, size_t n,
                              const uint8_t *dict, size_t dict_len,
                              void *d_out, size_t cap, size_t *out_len);
/* Seconds of GPU time of the last call by stage (HIP events): [0] hash chains
 * (sort) [1] matches [2] parse [3] blocks + tables [4] emission + checksums [5] total. */
int df_gpu_last_timings(bz_gpu_engine *g, double out_seconds[6]);
/* [0] blocks [1] stored [2] fixed [3] dynamic [4] tables that took the
 * length-limited path [5] stream bytes [6] dynamic blocks without any match: for
 * those the reference writes HDIST = 0 and no distance code length
 * (src/deflate/encoder.rs:431-436, 449-451), which RFC 1951 decoders reject; the
 * stream is reproduced as the reference writes it and counted here */
int df_gpu_last_stats(bz_gpu_engine *g, uint64_t out[8]);
/* Test hooks: the LZSS codes of the last call in stream order as (len, pos)
 * pairs, len 0 = the literal `pos` (what LzssEncoder::next yields,
 * src/lzss/encoder.rs:203-234); per block (start offset, bytes, BTYPE, bits). */
int df_gpu_debug_codes(bz_gpu_engine *g, const void *d_in, size_t n,
                       uint32_t *out_pairs, size_t cap, size_t *count);
int df_gpu_debug_blocks(bz_gpu_engine *g, uint64_t *out4, size_t cap, size_t *count);

/* One-shot over host buffers: `in.iter().cloned().encode(&mut Inflater::new(),
 * Action::Finish)` collected (or ZlibEncoder / GZipEncoder).  *out is malloc'ed,
 * release with bz_free.  Inputs of 64 MiB and more are uploaded, encoded (in parts: 64 MiB, then 256 MiB each) and
 * downloaded side by side: 1 GiB in 0.079 s (13.6 GB/s); *out is then a buffer of the stream's upper bound in
 * huge-page-backed memory of which only the pages the stream fills have been touched. */
int df_encode_buffer(int kind, int device, const uint8_t *in, size_t in_len,
                     uint8_t **out, size_t *out_len);
int df_encode_buffer_dict(int kind, int device, const uint8_t *in, size_t in_len,
                          const uint8_t *dict, size_t dict_len,
                          uint8_t **out, size_t *out_len);

/* Streaming context == the Encoder::next contract of the three encoders:
 * df_enc_write feeds bytes, df_enc_end(action) marks the end of an input
 * iterator (0 Run, kind 0: nothing comes out yet -- the blocks the reference hands out while it
 * consumes input come with the next Flush / Finish, the bytes are the same;
 * 1 Flush: the bytes written since the last segment come out as one segment:
 * LZSS stage drained, current block closed without the final bit, padded to a byte, window and
 * decompress_len carried over; 2 Finish: the last segment, final bit, container trailer;
 * kinds 1 / 2 (ZlibEncoder / GZipEncoder): ANY action ends the container -- Run: header, the whole bytes of
 * the blocks the Inflater has closed with 261 bytes of look-ahead held back, trailer; Flush: header, the
 * flushed segment, trailer -- and the encoder is finished: later input is not even pulled
 * (src/zlib/encoder.rs:130-136)), df_enc_read drains.  The stream so far stays in device memory until the
 * context is finished.  df_enc_finished: 1 once the final block (kind 0) / the trailer (kinds 1, 2) is out. */
typedef struct df_enc df_enc;
int df_enc_create(df_enc **out, int kind, int device);
int df_enc_create_dict(df_enc **out, int kind, int device, const uint8_t *dict, size_t dict_len); /* ::with_dict */
int df_enc_write(df_enc *e, const uint8_t *in, size_t n);
int df_enc_end(df_enc *e, int action);
int df_enc_finished(const df_enc *e);
long df_enc_read(df_enc *e, uint8_t *out, size_t cap);
size_t df_enc_pending(const df_enc *e);
void df_enc_destroy(df_enc *e);

#ifdef __cplusplus
}
#endif
#endif /* BZ2_MI355X_H */
