// snapmi_kernels.hpp -- kernel argument blocks and kernel declarations shared
// between the .hip translation units and the host API.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snapmi.h"
#include "snapmi_tiny.hpp"
#include "snapmi_profile.hpp"

namespace snapmi {

// Batch of raw streams to compress.  All pointers are device memory.
struct CompressArgs {
    const void *const *in_ptrs; // [n] stream i input
    const uint64_t *in_lens;    // [n]
    void *const *out_ptrs;      // [n] stream i output
    const uint64_t *out_caps;   // [n] or nullptr
    uint64_t *out_lens;         // [n]
    snapmi_error *errs;         // [n] or nullptr
    uint32_t *blk_first;        // [n+1] first global block index of stream i
    uint32_t *slot_first;       // [n+1] first scratch slot of stream i
    uint32_t *blk_size;         // [blocks] compressed bytes of each block
    uint64_t *blk_off;          // [blocks+1] exclusive scan of blk_size
    uint8_t *scratch;           // [slots * kSlotBytes]
    uint32_t n_streams;
    uint32_t host_blocks; // launch geometry computed from the host lengths
    uint32_t host_slots;
    uint32_t *ticket; // device-wide block ticket counter, zeroed per launch
    // k_compress_spans alone on a batch of several blocks per wavefront:
    // the order of the blocks is chosen as the launch goes (SpanSched in
    // snapmi_compress.hip); nullptr: blocks in ticket order
    uint32_t *sched;
    // The token path (k_match_* -> k_encode_tokens): a block's tokens lie in
    // PAGES of kTokPage tokens that its match finder takes from one pool as
    // it goes (tok_ctl[0]), its exceptions in pages of kExcPage from the same
    // pool; tok_pages says which: kPageTabStride entries per block of the
    // launch, token pages first, exception pages from kTokPagesPerBlock on.
    // The pool holds tok_pool_pages pages and one more, the DUMP: a block
    // that asks for a page when none is left is SPILLED - it goes on
    // (its encoded size is still counted), writes what follows to the dump,
    // puts itself on the list behind tok_ctl and leaves kTokSpilled in ntok;
    // k_encode_tokens skips it and k_redo_spilled compresses it once more
    // with the window kernel, straight to where the encoder would have put
    // it.  Results cannot depend on who spills: both kernels compute the
    // reference's bytes (snapmi_api.hip sizes the pool; option
    // token_pool_pct).
    uint32_t *tok_pool;
    uint32_t *tok_pages;
    // [0] pages asked for, [1] blocks spilled, [2] k_redo_spilled's ticket,
    // from kTokCtlList on: the spilled blocks
    uint32_t *tok_ctl;
    uint32_t tok_pool_pages;
    // staging arrays of the window wavefronts of a token-path launch
    // (TokenWriter): kTokStageWords u32 per wavefront - kMaxTokens tokens,
    // then 1 026 exceptions of 8 bytes
    uint32_t *tok_stage;
    // ... wavefront w of workgroup g has array g * tok_stage_waves + w -
    // tok_stage_wave0 (k_match_both: its window wavefronts come behind the
    // lanes')
    uint32_t tok_stage_waves, tok_stage_wave0;
    uint32_t *ntok;             // [blocks]
    uint32_t blk_lo, blk_hi;    // blocks this lane/encode launch covers
    uint32_t tok_base;          // block whose tokens lie at tokens[0]
    uint32_t direct; // k_encode_tokens writes final positions (no slots)
    unsigned long long *lane_tables; // lane g: 16-byte entries from g * lane_stride
    unsigned long long lane_stride;  // >= kMaxTable (tables are spread out)
    // tables made of mapped chunks (place_lane_tables): lane g's table is
    // slot (g % lane_chunks) * lane_per_chunk + g / lane_chunks, so that a
    // launch of FEWER lanes than the context has tables still uses every
    // chunk - the spread over the device's memory is what the chunks are for
    // (0: slot g)
    uint32_t lane_chunks, lane_per_chunk;
    uint32_t *lane_epochs;      // [lanes]
    uint32_t n_lanes;
    // experiment builds (-DSNAPMI_PROFILE) only: 16 u64 cycle counters
    unsigned long long *prof;
    // partial sums of the many-workgroup scans (k_plan_compress_*,
    // k_scan_sizes_*): one per 1024 streams / blocks, + 1
    uint2 *plan_part;
    unsigned long long *scan_part;
    // streams of 1 .. small_limit - 1 bytes get no blocks: k_compress_tiny
    // (under 256 bytes, one per lane) and k_compress_small (under 2 KiB, a
    // few per wavefront) compress them (0: every stream goes through blocks)
    uint32_t small_limit;
    // block-length class of this launch: only blocks of cls_lo < n <= cls_hi
    // bytes are its work (the match finders of the token path; a launch of
    // the whole batch has 0, kMaxBlock).  Round 5: blocks of at most 8 KiB -
    // pages, short frame chunks, tails - go to a window kernel whose tables
    // are as small as the reference makes them for such blocks
    // (src/compress.rs:491-518), twice as many per CU.
    uint32_t cls_lo, cls_hi;
};

// wavefronts (= hash tables) per persistent compress workgroup: 5 x 32 KiB
// is all of a CU's LDS
constexpr uint32_t kCompressWaves = 5;
// ... of the window kernel for blocks of at most 8 KiB (16 KiB tables)
constexpr uint32_t kSmallTableWaves = 10;
// k_match_both: wavefronts per CU, and how many of them are the lane kernel's
#define SNAPMI_BOTH_WAVES 6
#define SNAPMI_BOTH_LANE_WAVES 4
constexpr uint32_t kBothWaves = SNAPMI_BOTH_WAVES,
                   kBothLaneWaves = SNAPMI_BOTH_LANE_WAVES;
// Tokens (what a match finder hands k_encode_tokens): FOUR bytes each since
// round 6 - offset << 16 | field << 10 | literal length (0..1023), field =
// copy length - 4 (copies of 4..64 bytes), kTokLiteral (no copy: the block's
// last literal) or kTokException: a token that does not fit - a literal of
// 1 024 bytes or more, a copy of more than 64 - whose three numbers lie, in
// round 5's 8-byte form (literal | copy << 17 | offset << 33), in the block's
// exception list, in the order of their tokens.  Such a token covers 65
// bytes of input or more, so a block has at most 1 008 of them.
// A block has at most 16 385 tokens (every token but the last ends in a copy
// of >= 4 bytes); a lane writes its tokens a 128-byte group of 32 at a time.
// Pages (round 6): 512 tokens or 256 exceptions = 2 KiB, taken from a pool as
// a block needs them - the corpus round needs 5 900 tokens a block, 0.4 of
// its input in bytes, where a slot for the worst case of every block was 1.13
// (round 5: 2.0).
constexpr uint32_t kMaxTokens = 16416;
constexpr uint32_t kTokPage = 512, kExcPage = 256;
constexpr uint32_t kTokPagesPerBlock = (kMaxTokens + kTokPage - 1) / kTokPage;
constexpr uint32_t kExcPagesPerBlock = 4; // 1 008 exceptions at most
constexpr uint32_t kPageTabStride = 40;
static_assert(kTokPagesPerBlock + kExcPagesPerBlock <= kPageTabStride, "");
// ... of a block of at most 8 KiB: 2 049 tokens, 126 exceptions
constexpr uint32_t kPagesPerSmallBlock = (8192 / 4 + 64 + kTokPage - 1) / kTokPage + 1;
constexpr uint32_t kTokCtlList = 16;
constexpr uint32_t kTokStageWords = kMaxTokens + 2 * (kMaxTokens / 16) + 28; // 128-byte multiple
constexpr uint32_t kTokSpilled = 0xFFFFFFFEu; // in ntok: see tok_pool
constexpr uint32_t kTokLiteral = 61, kTokException = 62;
#if defined(__HIPCC__)
__device__ __forceinline__ bool tok_fits(uint32_t lit, uint32_t copy)
{
    return lit <= 1023u && copy <= 64u;
}
// (copy == 0: a literal alone; a copy is 4 bytes or more)
__device__ __forceinline__ uint32_t tok_pack(uint32_t lit, uint32_t copy,
                                             uint32_t offset)
{
    const uint32_t field = copy ? copy - 4u : kTokLiteral;
    return tok_fits(lit, copy) ? (offset << 16) | (field << 10) | lit
                               : kTokException << 10;
}
__device__ __forceinline__ unsigned long long tok_pack64(uint32_t lit,
                                                         uint32_t copy,
                                                         uint32_t offset)
{
    return (unsigned long long)lit | ((unsigned long long)copy << 17) |
           ((unsigned long long)offset << 33);
}
#endif

// Batch of raw streams to decompress.
struct DecompressArgs {
    const void *const *in_ptrs;
    const uint64_t *in_lens;
    void *const *out_ptrs;    // nullptr for "lengths only"
    const uint64_t *out_caps; // [n] or nullptr when out_ptrs is nullptr
    uint64_t *out_lens;
    snapmi_error *errs; // [n] or nullptr
    // optional [n]: 1 = stored chunk (frame type 0x01): plain copy of the
    // input; 2 = headerless piece of a long stream (k_stream_*): elements only,
    // out_caps[i] is the exact output length; 3 = not this launch's (a long
    // stream of a batch, decoded through its pieces): nothing is read, nothing
    // written
    const uint8_t *modes;
    // optional: the launch does nothing unless *gate == gate_value
    const unsigned long long *gate;
    unsigned long long gate_value;
    uint32_t n_streams;
    // [n] stream indices, longest compressed stream first (k_plan_decompress)
    uint32_t *order;
    uint32_t *bucket_pos; // [64] scratch of k_plan_decompress
    // experiment builds (-DSNAPMI_PROFILE) only: 16 u64 cycle counters
    unsigned long long *prof;
};

__global__ void k_probe_lds_order(uint32_t *bad);
__global__ void k_zero16(unsigned long long *p, unsigned long long n);
__global__ void k_seam_compress_tiny(const uint8_t *in, uint32_t n,
                                     uint8_t *out,
                                     unsigned long long *out_len,
                                     uint32_t *done, uint32_t seq);
__global__ void k_seam_decompress_tiny(DecompressArgs a, uint32_t *done,
                                       uint32_t seq);
__global__ void k_probe_tables(unsigned long long *tables,
                               unsigned long long stride, uint32_t steps,
                               uint32_t chunks, uint32_t per_chunk);
__global__ void k_plan_compress(CompressArgs a);
__global__ void k_plan_compress_a(CompressArgs a);
__global__ void k_plan_compress_b(CompressArgs a, uint32_t nparts);
__global__ void k_plan_compress_c(CompressArgs a);
__global__ void k_scan_sizes_a(CompressArgs a, uint32_t nparts);
__global__ void k_scan_sizes_b(CompressArgs a, uint32_t nparts);
__global__ void k_scan_sizes_c(CompressArgs a);
__global__ void k_compress_blocks(CompressArgs a);
__global__ void k_compress_block_lds(CompressArgs a);
__global__ void k_compress_spans(CompressArgs a);    // window steps, 5 tables/CU
__global__ void k_match_spans(CompressArgs a); // ... as the token path's match finder
__global__ void k_redo_spilled(CompressArgs a, uint32_t *h_stat, uint32_t seq);
__global__ void k_match_spans_8k(CompressArgs a); // ... blocks <= 8 KiB: 10 tables / CU
__global__ void k_post_ratio(uint32_t *host_mapped, const uint64_t *blk_off,
                             uint32_t blocks, const uint64_t *in_lens,
                             uint32_t n_streams, uint32_t seq);
__global__ void k_compress_span_lds(CompressArgs a); // ... one block per CU
__global__ void k_compress_tiny(CompressArgs a);
__global__ void k_compress_small512(CompressArgs a); // [256, 512) bytes
__global__ void k_compress_small1k(CompressArgs a);  // [512, 1024)
__global__ void k_compress_small2k(CompressArgs a);  // [1024, 2048)
__global__ void k_match_blocks(CompressArgs a);
__global__ void k_match_blocks_spec(CompressArgs a); // launches with blocks <= lanes
__global__ void k_match_both(CompressArgs a); // 3 lane + 2 window wavefronts per CU
__global__ void k_encode_tokens(CompressArgs a);
__global__ void k_scan_sizes(CompressArgs a);
__global__ void k_compact(CompressArgs a);
__global__ void k_stream_lens(CompressArgs a);

// One long raw stream decoded by many wavefronts (snapmi_decompress_stream).
// The element chain is sequential, so it is resolved hierarchically first:
// per 4 KiB segment and per 256 KiB super-segment, "if an element starts at
// offset o (< kEntry) of this piece, where does the chain leave it and how many
// bytes has it produced".
constexpr uint32_t kSeg = 4096;           // bytes of compressed input
// Entry offsets tabulated per segment / child: a chain is followed from the
// first kEntry bytes behind a boundary.  8 instead of a full wavefront of 64:
// the 64 chains of a segment merge within a few elements, so most of the
// scan's hops were duplicates; a wavefront now scans eight segments (one raw
// stream of 3 GiB: 52 -> 119 GiB/s; 16 entries gave 103, 4 gave 123 within
// noise of 8).  A literal of 9-60 bytes that straddles a boundary jumps over
// the landing zone and costs one more segment of hops: rare next to the 8x.
constexpr uint32_t kEntry = 8;
constexpr uint32_t kSegPerSuper = 64;
constexpr uint32_t kCutSegs = 512;       // segments per wavefront of k_stream_cuts
constexpr uint32_t kScanSegs = 64;       // segments per wavefront of k_stream_scan, at most
// wavefronts a scan launch should have before its wavefronts own that many
// segments each (StreamArgs::scan_segs)
constexpr uint32_t kScanFill = 1024;
constexpr uint32_t kStreamChunk = 65536;  // output bytes per piece: the
                                          // encoders' block size, so pieces
                                          // of their streams are independent
struct StreamArgs {
    const uint8_t *in;
    unsigned long long in_len;
    uint8_t *out;
    unsigned long long out_cap;
    unsigned long long *out_len; // [1]
    snapmi_error *err;           // [1]
    // meta[0] header bytes, [1] decoded length, [2] 0 = pieces decode it,
    // 1 = the sequential decoder must (error, or a stream whose pieces are
    // not independent), [3] pieces
    unsigned long long *meta;
    // per level (4 KiB, 256 KiB, 16 MiB blocks): tables of (exit, produced)
    // per entry offset < kEntry - level 1 [segments * kEntry], levels 2 and 3
    // [blocks * 64 children * 64] - and entries [blocks] of (position,
    // produced) where the chain enters each block (~0 = it does not)
    unsigned long long *s1, *s2, *s3;
    unsigned long long *e1, *e2, *e3;
    unsigned long long *cuts;     // [(kmax + 1) * 2] (src, dst) of piece k
    uint32_t nseg, nsuper, nsuper3, kmax;
    // log2 of the segment size of this call (12 = kSeg; 10 for streams that
    // are short enough for the scan to be a wait for its longest walk: a
    // quarter of the hops per lane, four times the lanes - round 5)
    uint32_t seg_log2;
    // segments a wavefront of k_stream_scan owns: kScanSegs when that still
    // makes kScanFill wavefronts, fewer (a power of two from 8) when it does
    // not - a scan wavefront is as slow as the chain of walks its lanes are
    // handed, ~630 cycles a hop whoever else is on the chip
    // (stream_scan_segs in snapmi_api.hip)
    uint32_t scan_segs;
    // piece descriptors for k_decompress_streams
    const void **c_in;
    unsigned long long *c_inlen;
    void **c_out;
    unsigned long long *c_cap;
    unsigned long long *c_outlen;
    snapmi_error *c_err;
    uint8_t *c_mode;
    // a long stream of a batch: where k_stream_finish says whether the
    // launch behind it must decode the stream (0) or not (3); else nullptr
    uint8_t *fb_mode;
};
// the long streams of a batch: their descriptors and, per kernel, the first
// workgroup of every stream (nullptr: one workgroup each)
struct BatchStreams {
    const StreamArgs *descs;
    const uint32_t *pre; // [n + 1]
    uint32_t n;
};
// Is a raw stream of `len` compressed bytes that announces `dl` bytes of
// output worth cutting into pieces (ten small launches, ~0.5 ms, instead of
// one wavefront at 0.14 GiB/s of elements or 0.85 GB/s of literal bytes)?
// From min_len (32 KiB) on when it expands by half and fills a piece and a
// half; from 256 KiB on whatever it holds (profiles/r4_scalar_latency.txt).
__host__ __device__ inline bool long_stream_rule(unsigned long long len,
                                                 unsigned long long dl,
                                                 unsigned long long min_len)
{
    const bool sane = dl / 22 <= len && 2 * dl >= 3ull * kStreamChunk;
    return sane && len >= min_len &&
           (2 * dl >= 3 * len || len >= (256ull << 10));
}
struct LongItem { // what k_long_plan hands to the host
    uint32_t idx, pad;
    unsigned long long in_len, dlen;
    const void *in;
    void *out;
    unsigned long long out_cap;
};
#define SNAPMI_STREAM_KERNEL_DECL(name)                                       \
    __global__ void k_stream_##name(StreamArgs a);                            \
    __global__ void k_bstream_##name(BatchStreams b);
SNAPMI_STREAM_KERNEL_DECL(head)
SNAPMI_STREAM_KERNEL_DECL(scan)
SNAPMI_STREAM_KERNEL_DECL(super)
SNAPMI_STREAM_KERNEL_DECL(super3)
SNAPMI_STREAM_KERNEL_DECL(chain)
SNAPMI_STREAM_KERNEL_DECL(spread3)
SNAPMI_STREAM_KERNEL_DECL(spread2)
SNAPMI_STREAM_KERNEL_DECL(cuts)
SNAPMI_STREAM_KERNEL_DECL(pieces)
SNAPMI_STREAM_KERNEL_DECL(finish)
__global__ void k_long_plan(const void *const *in_ptrs, const uint64_t *in_lens,
                            void *const *out_ptrs, const uint64_t *out_caps,
                            uint32_t n, uint64_t min_len, uint8_t *modes,
                            LongItem *list, uint32_t cap, uint32_t *count);

__global__ void k_plan_decompress(DecompressArgs a);
__global__ void k_plan_decompress_a(DecompressArgs a);
__global__ void k_plan_decompress_b(DecompressArgs a);
__global__ void k_plan_decompress_c(DecompressArgs a);
__global__ void k_decompress_streams3(DecompressArgs a);
__global__ void k_decompress_streams2(DecompressArgs a);
__global__ void k_decompress_sequential(DecompressArgs a);
__global__ void k_decompress_tiny(DecompressArgs a);
__global__ void k_decompress_small(DecompressArgs a); // ... under 512 bytes, 32 per wavefront
// streams per workgroup of k_decompress_streams3_many (batches of more than
// snapmi_ctx::decode_many_min streams)
constexpr uint32_t kManyStreams = 16;
__global__ void k_decompress_streams3_many(DecompressArgs a);
__global__ void k_decompress_len(DecompressArgs a);

} // namespace snapmi
