/*
 * snapmi.h -- C ABI of libsnapmi.so, the MI355X (gfx950) Snappy raw block codec.
 *
 * This is the drop-in boundary for the hot path of BurntSushi/rust-snappy:
 *   snap::raw::Encoder::compress   (reference src/compress.rs:99-154)
 *   snap::raw::Decoder::decompress (reference src/decompress.rs:75-95)
 * and the functions beside them.  Groups of entry points:
 *
 *  1. The libsnappy C API (snappy-c.h) -- exactly the symbols the reference's
 *     own native seam binds (snappy-cpp/src/lib.rs:66-88, linked by
 *     snappy-cpp/build.rs:2 as dylib=snappy).  Pointing that link line at
 *     libsnapmi.so makes the reference's `--features cpp` tests and benches
 *     run against the GPU codec unchanged.
 *  2. Scalar mirrors of snap::raw::* that carry the full snap::Error
 *     (variant + fields, reference src/error.rs:72-180) across the ABI.
 *  3. The batched, device-resident API a GPU pipeline calls: arrays of
 *     independent raw streams in HBM in, arrays of compressed streams in HBM
 *     out.  This is the form that is benchmarked.
 *  4. The Snappy frame format (reference src/frame.rs, src/crc32.rs,
 *     src/write.rs, src/read.rs): on device buffers (asynchronous), and on
 *     host buffers one batch of chunks per call - what a host-language
 *     FrameEncoder / FrameDecoder (shim/, rust-snappy_amd/frame.py,
 *     tools/szip.cpp) makes per batch.
 *  5. The gather of framed parts across the GPUs of a node (RCCL).
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions are
 * blocking unless stated otherwise.  Every compute entry point runs HIP
 * kernels on the GPU; there is no CPU fallback -- without a usable device
 * they fail with SNAPMI_E_DEVICE.
 */
#ifndef SNAPMI_H
#define SNAPMI_H

#include <stddef.h>
#include <stdint.h>

/* Every function declared here (and nothing else) is exported by
 * libsnapmi.so: the library is built with -fvisibility=hidden and the export
 * list rust-snappy_amd/csrc/snapmi.map, which csrc/gen_exports.py writes
 * from the SNAPMI_API lines of this header (tests/test_abi_cpu.py compares
 * `nm -D` of the built library with it, name for name). */
#if defined(__GNUC__)
#define SNAPMI_API __attribute__((visibility("default")))
#else
#define SNAPMI_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* 1. libsnappy C API (snappy-c.h:46-122; bound by the reference at     */
/*    snappy-cpp/src/lib.rs:66-88).  Host pointers.                     */
/* ------------------------------------------------------------------ */
typedef enum {
    SNAPPY_OK = 0,
    SNAPPY_INVALID_INPUT = 1,
    SNAPPY_BUFFER_TOO_SMALL = 2
} snappy_status;

/* replaces snappy_compress (snappy-cpp/src/lib.rs:67-72).
 * *compressed_length: in = capacity, out = bytes written. */
SNAPMI_API snappy_status snappy_compress(const char *input, size_t input_length,
                              char *compressed, size_t *compressed_length);
/* replaces snappy_uncompress (snappy-cpp/src/lib.rs:74-79) */
SNAPMI_API snappy_status snappy_uncompress(const char *compressed,
                                size_t compressed_length, char *uncompressed,
                                size_t *uncompressed_length);
/* replaces snappy_max_compressed_length (snappy-cpp/src/lib.rs:81) */
SNAPMI_API size_t snappy_max_compressed_length(size_t source_length);
/* replaces snappy_uncompressed_length (snappy-cpp/src/lib.rs:83-87) */
SNAPMI_API snappy_status snappy_uncompressed_length(const char *compressed,
                                         size_t compressed_length,
                                         size_t *result);
/* snappy-c.h:120-122; not bound by the reference, kept for completeness */
SNAPMI_API snappy_status snappy_validate_compressed_buffer(const char *compressed,
                                                size_t compressed_length);

/* ------------------------------------------------------------------ */
/* Errors: snap::Error (reference src/error.rs:72-180), same order.     */
/* ------------------------------------------------------------------ */
enum snapmi_kind {
    SNAPMI_OK = 0,
    SNAPMI_TOO_BIG = 1,                  /* a=given        b=max            */
    SNAPMI_BUFFER_TOO_SMALL = 2,         /* a=given        b=min            */
    SNAPMI_EMPTY = 3,
    SNAPMI_HEADER = 4,
    SNAPMI_HEADER_MISMATCH = 5,          /* a=expected_len b=got_len        */
    SNAPMI_LITERAL = 6,                  /* a=len  b=src_len  c=dst_len     */
    SNAPMI_COPY_READ = 7,                /* a=len  b=src_len                */
    SNAPMI_COPY_WRITE = 8,               /* a=len  b=dst_len                */
    SNAPMI_OFFSET = 9,                   /* a=offset  b=dst_pos             */
    SNAPMI_STREAM_HEADER = 10,           /* a=byte                          */
    SNAPMI_STREAM_HEADER_MISMATCH = 11,  /* a=first 6 body bytes, LE packed */
    SNAPMI_UNSUPPORTED_CHUNK_TYPE = 12,  /* a=byte                          */
    SNAPMI_UNSUPPORTED_CHUNK_LENGTH = 13,/* a=len  b=header (0/1)           */
    SNAPMI_CHECKSUM = 14,                /* a=expected  b=got               */
    /* not snap::Error variants: */
    SNAPMI_E_UNEXPECTED_EOF = 64, /* io::ErrorKind::UnexpectedEof (frame)   */
    SNAPMI_E_DEVICE = 100,  /* no GPU / HIP call failed; see last_error     */
    SNAPMI_E_ARGUMENT = 101 /* NULL pointer, bad context                    */
};

typedef struct snapmi_error {
    int32_t kind; /* enum snapmi_kind */
    uint32_t reserved;
    uint64_t a, b, c;
} snapmi_error;

/* impl fmt::Display for snap::Error (reference src/error.rs:249-335): the
 * text the reference prints for this error - "snappy: corrupt input (expected
 * copy write of length 11; remaining dst: 4)" - into buf (NUL-terminated,
 * truncated to cap).  Returns the length the whole text has.  Host code. */
SNAPMI_API size_t snapmi_error_string(const snapmi_error *err, char *buf, size_t cap);

/* ------------------------------------------------------------------ */
/* Context: one HIP device + stream + device scratch.  Maps to          */
/* snap::raw::Encoder (exclusive &mut self, owns its scratch table --   */
/* reference src/compress.rs:67-70): one context per thread.            */
/* ------------------------------------------------------------------ */
typedef struct snapmi_ctx snapmi_ctx;

/* device: HIP ordinal.  hip_stream: a hipStream_t the kernels are
 * launched on, or NULL for a stream owned by the context. */
SNAPMI_API int snapmi_ctx_create(int device, void *hip_stream, snapmi_ctx **out);
SNAPMI_API void snapmi_ctx_destroy(snapmi_ctx *ctx);
/* Message for the last SNAPMI_E_DEVICE / SNAPMI_E_ARGUMENT on this ctx. */
SNAPMI_API const char *snapmi_last_error(const snapmi_ctx *ctx);
/* Page-locked host memory for the host-buffer entry points (their H2D / D2H
 * copies run at PCIe speed from such buffers, at about a third of it from
 * pageable memory).  NULL when it cannot be had. */
SNAPMI_API void *snapmi_host_alloc(size_t bytes);
SNAPMI_API void snapmi_host_free(void *p);
/* "ms(stride) ms(stride) ... | held at most B of budget B | kept S KiB apart,
 * L lanes, B bytes | placement T ms | free B -> B": the placement probe's
 * time for every candidate region of the last lane-table allocation (none
 * when lane_table_tries is 1), the most device memory it held at once, what
 * it kept, how long it took and hipMemGetInfo's free bytes around it. */
SNAPMI_API const char *snapmi_table_probe_log(const snapmi_ctx *ctx);
/* Name of the dominant kernel of the last batch call - the one
 * snapmi_timing.dominant_ms times: "k_match_both", "k_match_blocks",
 * "k_match_spans", "k_compress_spans", "k_decompress_streams3" ... ("" before
 * the first call).  For profiles: bench.py names its roofline by it. */
SNAPMI_API const char *snapmi_last_kernel(const snapmi_ctx *ctx);
/* hipStream_t the context launches on (for event timing by the caller). */
SNAPMI_API void *snapmi_ctx_stream(const snapmi_ctx *ctx);
/* "snapmi <version> gfx950" */
SNAPMI_API const char *snapmi_version(void);
/*
 * Options (results never depend on them, only speed and memory):
 *   "compress_mode"        0 wavefront-per-block kernel only (k_compress_spans:
 *                          no scratch beyond the caller's buffers), 1
 *                          lane-per-block kernel on large batches (default)
 *                          [2, both at once, is a cross-check of the test
 *                          build: snapmi_test.h]
 *   "window_tokens"        1: a batch of more than two blocks per CU and
 *                          fewer than lane_min_blocks runs the window kernel
 *                          as a match finder and encodes with a wide kernel
 *                          of its own (128 KiB of token scratch per block of
 *                          the batch, at most 1 GiB); 0 (default): the window
 *                          kernel encodes while it matches (no scratch but
 *                          the block slots) - the two measure within 3 %
 *   "small_table_kernel"   1 (default): blocks of at most 8 KiB - pages,
 *                          short frame chunks, tails - are matched by a
 *                          window kernel with the 16 KiB table the reference
 *                          gives them (src/compress.rs:491-518), ten
 *                          wavefronts per CU instead of five, when a batch
 *                          has "small_table_min_blocks" (default 256) of
 *                          them; 0: they are blocks like any other
 *   "small_batch_kernel"   1 (default): batches of at most two blocks per
 *                          CU run one block per CU with table AND input block
 *                          in LDS; 0 never; 2 whenever the wavefront kernel
 *                          would run
 *   "tiny_stream_kernel"   1 (default): streams of fewer than 256 bytes are
 *                          compressed one per LANE, input, table and output
 *                          in LDS (k_compress_tiny); 0: they are one-block
 *                          streams of the block kernels (and so are the
 *                          streams of the next option)
 *   "small_stream_kernel"  1 (default): streams of 256 .. 1023 bytes are
 *                          compressed a few per wavefront, one per lane, with
 *                          input and table in LDS (k_compress_small); 2: up
 *                          to 2047 bytes (slower than the block kernels from
 *                          1 KiB on); 0: they are one-block streams of the
 *                          block kernels
 *   "span_schedule"        1 (default): a window-kernel launch of more blocks
 *                          than it has wavefronts (1 280) chooses the order
 *                          of its blocks as it goes - the first block of
 *                          every stream first, then the blocks of streams
 *                          that proved heavy, the light ones last - so that
 *                          it ends with small jobs (256 MiB of mixed files:
 *                          a quarter less time); 0: ticket order; 2: also
 *                          for fewer blocks
 *   "lane_min_blocks"      batches with at least this many 64 KiB blocks use
 *                          the lane-per-block kernel (default 20480 = 1.25 GiB)
 *   "lane_speculate"       1 (default): a lane-kernel launch of at most 24 576
 *                          blocks (1.5 GiB) also fetches, in a probe's round,
 *                          the table entry of the probe that follows a miss -
 *                          fewer dependent rounds per block where a block's
 *                          latency is what is waited for; 0: never
 *   "lane_segment_blocks"  most blocks per lane-kernel launch (default
 *                          262144 = 16 GiB of input).  A larger batch is
 *                          matched and encoded in equal launches of at most
 *                          this many blocks, and the token scratch - 72 KiB a
 *                          block, 1.13x the input - is one launch's: at cfg2
 *                          (146 700 blocks) 73 350 makes the context hold
 *                          23.3 GB instead of 29.1 and costs 8 % of the
 *                          compress rate, 48 900 21.2 GB (tokens 0.42x the
 *                          input) and 25 % (profiles/r6_token_segments.txt)
 *   "token_pool_pct"       39 (default): the token pool - where the match
 *                          finders of a large batch leave their tokens for
 *                          the encoder, in pages of 2 KiB taken as a block
 *                          needs them - is this share of what the worst case
 *                          of every block would take (1.16x the input: a
 *                          token of 4 bytes per 4 bytes of input, rounded to
 *                          pages), + a page and a half per lane in flight:
 *                          0.49x the input at cfg2 with everything, of which
 *                          cfg2 uses 0.46x; English text needs 0.73x.  A block
 *                          that finds no page is compressed a second time by
 *                          the window kernel (same bytes; costs that block
 *                          twice), and the context's next batch gets a pool
 *                          half as large again when more than 1 % of a batch
 *                          did (a sixth larger when it was under 10 %).
 *                          100: no block ever spills.  Never under
 *   "token_pool_min_pages" 32768 (default; 64 MiB): batches of up to 750
 *                          blocks never spill
 *   "lane_table_spread"    1 (default): the lane kernel's hash tables are
 *                          spread over up to 4x their size, as far as the
 *                          budget below allows (HBM sustains up to 30 % more
 *                          random accesses on tables that are not packed
 *                          into the memory a process is handed first,
 *                          DESIGN 4.1); 0: packed (17-25 GB)
 *   "lane_table_budget_pct"  percent of the device memory that is free when
 *                          a context first needs its lane tables that the
 *                          tables - and, while a placement is being chosen,
 *                          its candidates together - may hold (default 33,
 *                          1..90).  No call holds more at any moment, except
 *                          snapmi_ctx_prepare with SNAPMI_PREPARE_TOP_OF_MEMORY
 *   "lane_table_tries"     placements of the lane tables that are timed
 *                          (k_probe_tables, 3 ms each) before the fastest is
 *                          kept - at most this many (default 2: spread, then
 *                          packed behind that), fewer when one probes at the
 *                          fast rate or the budget has no room for another.
 *                          1: no probing, one region
 *   "release_scratch"      1: the compressor's per-batch scratch (token
 *                          arrays of the lane kernel: 128 KiB per block of a
 *                          launch, up to 34 GB) is freed by
 *                          snapmi_ctx_synchronize instead of being kept for
 *                          the next batch (default 0)
 *   "batch_long_streams"   1 (default): snapmi_decompress_batch looks at a
 *                          batch of at most 16 384 streams first (one small
 *                          kernel, one synchronisation of the context's
 *                          stream, ~30 us) and decodes up to 4 096 long
 *                          streams in it - 32 KiB compressed or more that
 *                          expand by half, 256 KiB or more of anything -
 *                          through their 64 KiB pieces, like
 *                          snapmi_decompress_stream, instead of one wavefront
 *                          each (a batch otherwise waits 3-5 ms for a 700 KB
 *                          stream); 0: never, the call only enqueues
 *   "decode_kernel"        3 (default) k_decompress_streams3; 0 one element
 *                          at a time [2, the second generation alone, is a
 *                          cross-check of the test build]
 *   "frame_parallel_walk_min"  framed streams of at least this many bytes
 *                          decoded without a side index get their chunk
 *                          headers found in parallel (default 4 MiB)
 *   "host_encode_slice"    input bytes per slice of snapmi_frame_encode_host
 *                          (default 2 GiB: the match finder wants few, large
 *                          launches)
 *   "host_decode_slice_chunks"  data chunks per slice of
 *                          snapmi_frame_decode_host (default 8192)
 *   "host_copy_kernel"     bit 0 / bit 1: decoded / encoded results go home
 *                          by a copy kernel instead of hipMemcpyAsync when
 *                          the caller's buffer is pinned (default 1)
 * The knobs of the test suite and of the experiment drivers are declared in
 * snapmi_test.h (snapmi_ctx_set_test_option).
 * Returns SNAPMI_E_ARGUMENT for an unknown name.
 */
SNAPMI_API int snapmi_ctx_set_option(snapmi_ctx *ctx, const char *name, int64_t value);

/* Allocates and places, NOW, the device memory a batch of `blocks` 64 KiB
 * blocks (input bytes / 65 536, rounded up per stream) would make the first
 * snapmi_compress_batch allocate: the lane kernel's hash tables (256 KiB per
 * lane in flight, 17-26 GB for a batch that fills the chip; nothing for a
 * batch of fewer than lane_min_blocks blocks).  Optional - a compress call
 * does the same on demand, within lane_table_budget_pct.
 * flags:
 *   SNAPMI_PREPARE_TOP_OF_MEMORY  place the tables at the far end of the
 *     device's memory, where HBM sustains 30 % more random accesses than in
 *     the part a process is handed first (cfg2: 72 instead of 57-66 GiB/s,
 *     DESIGN 4.1).  The only way there is through everything in front of it:
 *     for the duration of two hipMalloc calls this call holds ALL free device
 *     memory (any other allocation on the device fails meanwhile - other
 *     processes, torch in this process, other contexts), and the driver then
 *     wipes what was given back, in the background, for a few seconds.  For a
 *     process that owns the GPU, once, at start-up; never taken by default.
 * Like the reference's Encoder::new (src/compress.rs:80-82: an encoder's
 * scratch is its own and allocated once), made explicit because here it is
 * gigabytes. */
#define SNAPMI_PREPARE_TOP_OF_MEMORY 1u
SNAPMI_API int snapmi_ctx_prepare(snapmi_ctx *ctx, uint64_t blocks,
                                  uint32_t flags);

/* What a context holds and what its last batch did, by name:
 *   "scratch_bytes"         device memory the context's grow-only buffers
 *                           hold now (lane tables, token pool, plans, ...)
 *   "token_scratch_bytes"   ... the token pool, its page tables, the counts
 *   "token_pool_pages"      pages (2 KiB) of the last token-path launch's pool
 *   "token_pool_pct_now"    what token_pool_pct has grown to on this context
 *   "token_pages_asked"     pages the last token-path launch asked for, and
 *   "token_blocks_spilled"  blocks of it that found none and were compressed
 *                           a second time (both wait for the launch)
 * SNAPMI_E_ARGUMENT for a name that is not in this list. */
SNAPMI_API int snapmi_ctx_get_info(snapmi_ctx *ctx, const char *name,
                                   int64_t *value);

/* ------------------------------------------------------------------ */
/* 2. Scalar mirrors of snap::raw (host buffers; H2D + kernels + D2H).  */
/*    Return enum snapmi_kind; *err (may be NULL) gets the fields.      */
/* ------------------------------------------------------------------ */
/* snap::raw::max_compress_len, reference src/compress.rs:42-53
 * (0 when the input or the bound exceeds 2^32-1). */
SNAPMI_API size_t snapmi_max_compress_len(size_t input_len);
/* snap::raw::decompress_len, reference src/decompress.rs:30-35.
 * Pure header parse on the host. */
SNAPMI_API int snapmi_decompress_len(const uint8_t *input, size_t input_len,
                          size_t *result, snapmi_error *err);
/* snap::raw::Encoder::compress, reference src/compress.rs:99-154 */
SNAPMI_API int snapmi_raw_compress(snapmi_ctx *ctx, const uint8_t *input,
                        size_t input_len, uint8_t *output, size_t output_cap,
                        size_t *written, snapmi_error *err);
/* snap::raw::Decoder::decompress, reference src/decompress.rs:75-95 */
SNAPMI_API int snapmi_raw_decompress(snapmi_ctx *ctx, const uint8_t *input,
                          size_t input_len, uint8_t *output,
                          size_t output_cap, size_t *written,
                          snapmi_error *err);

/* ------------------------------------------------------------------ */
/* 3. Batched device-resident API.  Every d_* pointer is device memory  */
/*    on the context's device; stream i is an independent raw stream    */
/*    (its own varint header), exactly one Encoder::compress /          */
/*    Decoder::decompress call of the reference.                        */
/*    Asynchronous: work is enqueued on the context's stream; results   */
/*    are valid after snapmi_ctx_synchronize (or a caller-side wait on  */
/*    that stream).  Return value reports enqueue-time failures only;   */
/*    per-stream results land in d_out_lens / d_errs.                   */
/* ------------------------------------------------------------------ */

/*
 * Compress n streams.
 *   d_in_ptrs[i], d_in_lens[i] : input bytes of stream i
 *   h_in_lens                  : host copy of d_in_lens (needed to size the
 *                                launch); NULL = fetched with a blocking D2H
 *   d_out_ptrs[i], d_out_caps[i]: output buffer; capacity must be >=
 *                                snapmi_max_compress_len(len) or stream i
 *                                fails with BufferTooSmall (reference
 *                                src/compress.rs:111-116); d_out_caps NULL =
 *                                capacities are not checked
 *   d_out_lens[i]              : bytes written (0 on error)
 *   d_errs[i]                  : per-stream snapmi_error (may be NULL)
 */
SNAPMI_API int snapmi_compress_batch(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                          const uint64_t *d_in_lens,
                          const uint64_t *h_in_lens, void *const *d_out_ptrs,
                          const uint64_t *d_out_caps, uint64_t *d_out_lens,
                          snapmi_error *d_errs, size_t n);

/*
 * Decompress n streams.  d_out_caps[i] is the capacity of d_out_ptrs[i]
 * (reference: output.len(), src/decompress.rs:84-89); d_out_lens[i] gets the
 * decompressed length on success, 0 on error.
 * A stream is decoded by one wavefront - except the long streams of a batch
 * of at most 16 384 streams, which are cut into pieces like the one stream of
 * snapmi_decompress_stream (option "batch_long_streams": that look at the
 * batch waits once for the context's stream, so with it the call is NOT
 * enqueue-only.  A host that pipelines many small batches from one thread
 * sets the option to 0; while the context's stream is being captured into a
 * hipGraph the look is skipped by itself and the call only enqueues).
 */
SNAPMI_API int snapmi_decompress_batch(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                            const uint64_t *d_in_lens,
                            void *const *d_out_ptrs,
                            const uint64_t *d_out_caps, uint64_t *d_out_lens,
                            snapmi_error *d_errs, size_t n);

/*
 * ONE long raw stream, device resident, decoded by many wavefronts.
 * A raw stream has no index (reference src/decompress.rs:130-148 walks it
 * element by element) and snapmi_decompress_batch gives a stream to a single
 * wavefront; here the element chain is first resolved by a hierarchical scan
 * (4 KiB segments, 256 KiB super-segments, one short sequential pass), the
 * stream is cut at the element boundaries next to every 64 KiB of output,
 * and the pieces are decoded like independent streams.  Streams of this
 * encoder, the reference and libsnappy (64 KiB blocks) always take that
 * path; a stream whose pieces depend on each other (a copy reaching across a
 * cut), or with any error, is decoded by the sequential path, so results and
 * errors are those of snapmi_decompress_batch with n = 1.  Asynchronous on
 * the context's stream; d_out_len[0] / d_err[0] as in the batch call.
 * The scalar entry points use it for inputs of 32 KiB and more that announce
 * half as much output again (and 96 KiB or more), and from 256 KiB on for
 * any input - the rule snapmi_decompress_batch applies to the long streams
 * of a batch.
 */
SNAPMI_API int snapmi_decompress_stream(snapmi_ctx *ctx, const void *d_in,
                             uint64_t in_len, void *d_out, uint64_t out_cap,
                             uint64_t *d_out_len, snapmi_error *d_err);
/* Which way the last snapmi_decompress_stream on this context went (waits
 * for it): 0 = pieces on many wavefronts, 1 = the sequential path, -1 = no
 * such call yet.  For tests and benchmarks. */
SNAPMI_API int snapmi_stream_decode_path(snapmi_ctx *ctx);

/* decompress_len for n streams on the device (reference
 * src/decompress.rs:30-35): d_out_lens[i] = header value, d_errs[i] as the
 * reference would return. */
SNAPMI_API int snapmi_decompress_len_batch(snapmi_ctx *ctx,
                                const void *const *d_in_ptrs,
                                const uint64_t *d_in_lens,
                                uint64_t *d_out_lens, snapmi_error *d_errs,
                                size_t n);

/* Wait for everything enqueued on the context's stream. */
SNAPMI_API int snapmi_ctx_synchronize(snapmi_ctx *ctx);

/*
 * Timing of the last batch call, measured with HIP events recorded on the
 * context's stream around each kernel group.  Valid after synchronize.
 * Times in milliseconds; a group that did not run reports 0.
 */
typedef struct snapmi_timing {
    float plan_ms;     /* descriptor scan / block table kernels            */
    float codec_ms;    /* the dominant kernel: compress or decompress      */
    float compact_ms;  /* compress only: gather of blocks 1.. into place   */
    float total_ms;    /* first event to last event                        */
    uint64_t codec_launches; /* launches of the dominant kernel            */
    float dominant_ms; /* the single dominant kernel alone: k_match_blocks,
                          k_compress_blocks or k_decompress_streams         */
    float reserved;
} snapmi_timing;
SNAPMI_API int snapmi_last_timing(snapmi_ctx *ctx, snapmi_timing *out);

/* ------------------------------------------------------------------ */
/* 4. Snappy frame format on the device (reference src/frame.rs,        */
/*    src/crc32.rs, src/write.rs, src/read.rs).  One framed stream per  */
/*    call; every <=64 KiB chunk is an independent raw stream, so the   */
/*    chunk is the parallel unit.  All d_* pointers are device memory.  */
/*    Asynchronous like the batch calls.                                */
/* ------------------------------------------------------------------ */

/* Upper bound of snapmi_frame_compress output for n input bytes:
 * stream identifier + per chunk (8-byte header + at most the chunk itself,
 * because of the uncompressed fallback of reference src/frame.rs:85). */
SNAPMI_API size_t snapmi_frame_max_len(size_t n);

/*
 * What write::FrameEncoder::write_all(input) followed by into_inner()
 * produces (reference src/write.rs:123-192 chunking, src/frame.rs:62-104
 * compress_frame, src/crc32.rs:35-38 masked CRC32C of the uncompressed chunk):
 *   d_out_len[0]        : framed length (0 for empty input, as the reference)
 *   d_chunk_offsets     : optional [chunks+1] offsets of every chunk header
 *                         in d_out (a side index; not part of the stream)
 */
SNAPMI_API int snapmi_frame_compress(snapmi_ctx *ctx, const void *d_in, uint64_t in_len,
                          void *d_out, uint64_t out_cap, uint64_t *d_out_len,
                          uint64_t *d_chunk_offsets);

/*
 * read::FrameDecoder over the whole stream (reference src/read.rs:105-238):
 * stream identifier, chunk types, length limits, raw decode, CRC check.
 *   d_chunk_offsets/n_chunks : optional side index of the chunk headers
 *                         (as written by snapmi_frame_compress); without it
 *                         the headers are walked on the device, one after
 *                         the other (the format has no index)
 *   d_out == NULL       : only compute the decompressed length
 *   d_out_len[0]        : decompressed length; on an error the bytes in
 *                         front of the failing chunk (see _ex below)
 *   d_err[0]            : first error in stream order (kind 0 = ok);
 *                         an io::ErrorKind::UnexpectedEof is reported as
 *                         SNAPMI_E_UNEXPECTED_EOF
 */
SNAPMI_API int snapmi_frame_decompress(snapmi_ctx *ctx, const void *d_in,
                            uint64_t in_len, void *d_out, uint64_t out_cap,
                            uint64_t *d_out_len, snapmi_error *d_err,
                            const uint64_t *d_chunk_offsets,
                            uint64_t n_chunks);

/* flags of the frame entry points below */
#define SNAPMI_FRAME_NO_IDENT 1u     /* compress: do not emit the identifier */
#define SNAPMI_FRAME_CONTINUATION 1u /* decode: identifier already seen      */

/*
 * write::FrameEncoder with the chunk boundaries chosen by the caller: the
 * reference cuts a chunk wherever a flush, a direct write larger than its
 * 64 KiB buffer, or a short read of read::FrameEncoder ends it
 * (src/write.rs:123-192, src/read.rs:365-409), so chunks of a stream are not
 * always 65536 bytes.  Chunk i is the next h_chunk_lens[i] (1..65536) bytes
 * of d_in; each goes through compress_frame (src/frame.rs:62-104).  With
 * SNAPMI_FRAME_NO_IDENT the 10-byte stream identifier is not written (a
 * later batch of the same stream, src/write.rs:167-170).  out_cap >=
 * 10 + sum(lens) + 8 n.  h_chunk_lens is host memory, read before the call
 * returns; the rest is asynchronous like snapmi_frame_compress.
 */
SNAPMI_API int snapmi_frame_compress_chunks(snapmi_ctx *ctx, const void *d_in,
                                 const uint32_t *h_chunk_lens, size_t n,
                                 uint32_t flags, void *d_out, uint64_t out_cap,
                                 uint64_t *d_out_len,
                                 uint64_t *d_chunk_offsets);

/*
 * snapmi_frame_decompress for one BATCH of a stream a host reader delivers
 * piecewise (read::FrameDecoder decodes chunk by chunk, src/read.rs:105-238):
 *   flags    SNAPMI_FRAME_CONTINUATION: the stream identifier was seen in an
 *            earlier batch (read.rs:123-128)
 *   stale10  NULL, or the first 10 bytes of the reference reader's scratch
 *            buffer as earlier batches left them (snapmi_frame_scan_host
 *            maintains them): the reference parses a compressed chunk's
 *            length header from that whole buffer (read.rs:216), so a payload
 *            of fewer than 10 bytes without a varint terminator is judged
 *            together with those bytes.  NULL = a fresh reader (zeros).
 * On an error d_out_len[0] is the number of output bytes IN FRONT of the
 * failing chunk: they are decoded and CRC-checked, and the reference's
 * reader has returned them before it reports the error.  The side index is
 * a hint: if it does not tile [identifier, in_len) with plain data chunks,
 * or a chunk needs the stale-buffer rule, the headers are walked instead.
 */
SNAPMI_API int snapmi_frame_decompress_ex(snapmi_ctx *ctx, const void *d_in,
                               uint64_t in_len, void *d_out, uint64_t out_cap,
                               uint64_t *d_out_len, snapmi_error *d_err,
                               const uint64_t *d_chunk_offsets,
                               uint64_t n_chunks, uint32_t flags,
                               const uint8_t *stale10);

/*
 * Host-side scan for a reader that delivers the stream piecewise: walks the
 * chunk headers of h_in[0, in_len) as far as chunks are complete and
 * well-formed.  *consumed = end of the last such chunk, *n_chunks = data
 * chunks in front of it, h_offsets (optional, cap >= n + 1) = their header
 * offsets followed by *consumed, stale10 (optional, 10 bytes, in/out) = the
 * reference reader's src[0..10) after those chunks.  Returns 0 when
 * consumed == in_len, 2 when the next chunk is cut off by in_len (read more,
 * or at end of input hand the rest to the device: UnexpectedEof), 1 when the
 * next chunk's header is one the decoder rejects (hand [consumed, ...) to
 * the device without an index: it reports the reference's error), or
 * SNAPMI_E_ARGUMENT.  No GPU work.
 */
SNAPMI_API int snapmi_frame_scan_host(const void *h_in, uint64_t in_len, uint32_t flags,
                           uint8_t *stale10, uint64_t *h_offsets,
                           uint64_t cap, uint64_t *n_chunks,
                           uint64_t *consumed);

/*
 * Host-buffer forms of the two calls above (blocking: H2D, kernels, D2H
 * through the context's staging buffers) - what a host-language
 * FrameEncoder / FrameDecoder calls once per batch of chunks (shim/src/
 * write.rs, read.rs; rust-snappy_amd/frame.py; tools/szip.cpp).
 *
 * snapmi_frame_encode_host: chunks as in snapmi_frame_compress_chunks, from
 * h_in back to back; out_cap >= snapmi_frame_encode_bound(sum, n).
 *
 * snapmi_frame_decode_host: decodes the complete, well-formed chunks at the
 * start of h_in[0, in_len) - as many as out_cap / 65536 allows - and reports
 * how far it got: *consumed input bytes, *written output bytes.  Call it again
 * with the rest (plus more input) and SNAPMI_FRAME_CONTINUATION.
 *   consumed == 0 and kind OK : not one whole chunk yet, supply more input
 *   SNAPMI_FRAME_FINAL        : no more input will follow: a cut-off chunk is
 *                               UnexpectedEof instead of "supply more"
 *   on an error (return value = *err's kind) *written bytes in front of the
 *   failing chunk are valid output and must be delivered first
 *   (src/read.rs:111-118); *consumed stays 0.
 *   stale10: 10 bytes of decoder state kept by the caller between calls
 *   (zeros for a new stream), see snapmi_frame_decompress_ex.
 */
#define SNAPMI_FRAME_FINAL 2u
SNAPMI_API size_t snapmi_frame_encode_bound(size_t total_bytes, size_t n_chunks);
SNAPMI_API int snapmi_frame_encode_host(snapmi_ctx *ctx, const uint8_t *h_in,
                             const uint32_t *h_chunk_lens, size_t n,
                             uint32_t flags, uint8_t *h_out, size_t out_cap,
                             size_t *written);
SNAPMI_API int snapmi_frame_decode_host(snapmi_ctx *ctx, const uint8_t *h_in,
                             size_t in_len, uint32_t flags, uint8_t *stale10,
                             uint8_t *h_out, size_t out_cap, size_t *written,
                             size_t *consumed, snapmi_error *err);

/* Host-side chunk scan of a framed stream that is still in HOST memory: the
 * hops FrameDecoder::read makes while it reads (src/read.rs:105-172).  The
 * format is a linked list of chunk headers; on the device every hop is a
 * dependent HBM access (~0.7 us), on the host it is free while the bytes are
 * being staged.  Writes the offsets of the DATA chunk headers (types 0x00 and
 * 0x01) and, last, in_len: n + 1 values; copy them to the device and pass
 * them as d_chunk_offsets.  h_offsets may be NULL to count only.
 * Returns 0 and *n_chunks for a structurally regular stream (identifier
 * first; data, skippable, padding and repeated identifier chunks; ends on a
 * chunk boundary; every length within the format's limits), 1 otherwise or
 * when `cap` < n + 1: decode such a stream WITHOUT an index and the device
 * walk reports the reference's error.  No GPU work. */
SNAPMI_API int snapmi_frame_index_host(const void *h_in, uint64_t in_len,
                            uint64_t *h_offsets, uint64_t cap,
                            uint64_t *n_chunks);

/* Masked CRC32C (reference CheckSummer::crc32c_masked, src/crc32.rs:35-38)
 * of n buffers of at most 65536 bytes each. */
SNAPMI_API int snapmi_crc32c_masked_batch(snapmi_ctx *ctx, const void *const *d_ptrs,
                               const uint64_t *d_lens, uint32_t *d_out,
                               size_t n);

/* ------------------------------------------------------------------ */
/* 5. Multi-GPU: the one exchange step of the path.         */
/*                                                                      */
/* Raw streams and frame chunks are independent, so N GPUs shard a job  */
/* with no exchange during compute: rank r frames a contiguous range of */
/* chunks with snapmi_frame_compress[_chunks] (ranks behind the first   */
/* pass SNAPMI_FRAME_NO_IDENT, or drop the 10-byte identifier), and the */
/* concatenation of the parts in rank order is the single-stream        */
/* framing the reference's writer produces (src/write.rs:165-192).      */
/* snapmi_gatherv assembles it on one rank over RCCL (xGMI inside a     */
/* node): sizes first, then ONE group of point-to-point transfers       */
/* straight into the root's buffer at every rank's prefix offset, so    */
/* the root's links receive in parallel.  librccl is opened at run time */
/* (dlopen); a process that already holds one (PyTorch) shares it.      */
/* ------------------------------------------------------------------ */
#define SNAPMI_COMM_ID_BYTES 128
typedef struct snapmi_comm snapmi_comm;

/* A fresh rendezvous id (ncclGetUniqueId): call on ONE rank, hand the 128
 * bytes to the others by any means (file, socket, MPI, torch store). */
SNAPMI_API int snapmi_comm_unique_id(uint8_t id_out[SNAPMI_COMM_ID_BYTES]);
/* Collective: every rank calls it with the same id and world, its own rank,
 * and a context on the GPU it drives (one process per GPU). */
SNAPMI_API int snapmi_comm_init(snapmi_ctx *ctx, const uint8_t id[SNAPMI_COMM_ID_BYTES],
                     int rank, int world, snapmi_comm **out);
/* Or use a communicator the host already has (an ncclComm_t, as void *);
 * it is not destroyed by snapmi_comm_destroy. */
SNAPMI_API int snapmi_comm_wrap(snapmi_ctx *ctx, void *nccl_comm, int rank, int world,
                     snapmi_comm **out);
SNAPMI_API void snapmi_comm_destroy(snapmi_comm *comm);
/* Collective, blocking.  Every rank contributes d_send[0, send_bytes) (device
 * memory, may be empty); on `root`, d_recv[0, *total) receives the parts in
 * rank order.  h_sizes (host, [world], may be NULL) and *total are filled on
 * EVERY rank.  recv_cap matters on the root only; if the parts do not fit,
 * every rank returns SNAPMI_E_ARGUMENT and nothing is exchanged. */
SNAPMI_API int snapmi_gatherv(snapmi_ctx *ctx, snapmi_comm *comm, int root,
                   const void *d_send, uint64_t send_bytes, void *d_recv,
                   uint64_t recv_cap, uint64_t *h_sizes, uint64_t *total);

#ifdef __cplusplus
}
#endif
#endif /* SNAPMI_H */
