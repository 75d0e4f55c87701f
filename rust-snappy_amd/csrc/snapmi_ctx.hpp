// snapmi_ctx.hpp -- private: the context object and helpers shared by the
// host-side translation units (snapmi_api.hip, snapmi_frame.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "snapmi.h"

namespace snapmi {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

} // namespace snapmi

struct snapmi_host_pipe; // snapmi_frame.hip: staging of the host-buffer calls

struct snapmi_ctx {
    int device = 0;
    snapmi_host_pipe *pipe = nullptr;
    // 256 bytes of pinned, device-mapped host memory: kernels post small
    // results here (no copy-engine round trip behind a bulk copy)
    volatile uint32_t *h_mail = nullptr;
    // pinned host staging of the scalar (host-pointer) entry points
    void *pin_in = nullptr, *pin_out = nullptr, *pin_desc = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0, pin_desc_cap = 0;
    // slices of the host-buffer frame calls: input bytes per encode slice,
    // data chunks per decode slice
    // (measured, profiles/r3_host_pipeline.txt: the match finder's latency
    // floor wants two slices for 4 GiB, the decoder is happy from 0.5 GiB)
    uint64_t host_encode_slice = 2048ull << 20;
    uint64_t host_decode_slice_chunks = 8192;
    // results go home by a copy kernel (bit 0: decode, bit 1: encode) or by
    // hipMemcpyAsync: see k_to_host.  The kernel is used only when the
    // caller's buffer is pinned, device-mapped host memory
    // (snapmi_host_alloc).  Encode: off - the copy kernel would wait for the
    // match finder's persistent workgroups.
    int host_copy_kernel = 1;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    // grow-only device scratch of the raw codec
    snapmi::DevBuf blk_first, slot_first, blk_size, blk_off, slots, plan_part;
    // lane-per-block match finder: tokens, token counts, HBM hash tables
    snapmi::DevBuf tokens, tok_pages, tok_stage, ntok, lane_tables,
        lane_epochs;
    // lane_tables made of physical chunks (hipMemCreate) mapped into one
    // address range (place_lane_tables, snapmi_api.hip): the chunks and the
    // bytes of the range; empty / 0 when lane_tables.p came from hipMalloc
    std::vector<hipMemGenericAllocationHandle_t> lane_chunks;
    size_t lane_chunk_bytes = 0, lane_va_bytes = 0;
    uint32_t lane_chunk_count = 0, lane_per_chunk = 0; // lane -> table map
    uint32_t n_lanes = 0;
    uint64_t lane_stride = 0;      // 16-byte entries between two lanes' tables
    bool lane_table_spread = true; // spread the tables over free memory
    bool lane_tables_top = false;  // placed by snapmi_ctx_prepare(TOP_OF_MEMORY)
    // SNAPMI_COMPRESS=waves|lanes|both: 0 = wavefront kernel only, 1 = lane
    // kernel on large batches and the wavefront kernel on small ones
    // (default), 2 = on large batches both kernels at once, sharing one
    // two-ended ticket (no faster: the wavefront kernel takes whole CUs' LDS
    // away from the lanes' input windows; kept as a cross-check)
    int compress_mode = 1;
    // 1: batches between two blocks per CU and lane_min_blocks run the window
    // kernel as a match finder (k_match_spans) in front of k_encode_tokens;
    // 0 (default): k_compress_spans, which encodes while it matches (no token
    // scratch).  Measured equal within 3 % either way on 64 MiB .. 1 GiB
    // (profiles/r5_span_sweep.txt), so the path without scratch is the default
    int window_tokens = 0;
    // k_compress_spans on more blocks than it has wavefronts: 1 (default) the
    // order of the blocks is chosen as the launch goes (SpanSched,
    // snapmi_compress.hip), 0 ticket order, 2 scheduled whatever the count
    int span_schedule = 1;
    snapmi::DevBuf sched;
    // 1: lane-kernel launches of at least lane_coresident_min_blocks blocks
    // run k_match_both - three lane wavefronts and two window wavefronts on
    // every CU, one two-ended ticket
    // (cfg2, 146 700 blocks: 115.6 -> 108.4 ms for the match finder; at 4 GiB
    // a draw, below that the lanes have taken every block before a window
    // wavefront has finished one: profiles/r5_coresident.txt)
    int lane_coresident = 1;
    uint64_t lane_coresident_min_blocks = 98304;
    const char *last_kernel = "";
    // 1: blocks of at most 8 KiB go to the window kernel with 16 KiB tables
    // (k_match_spans_8k: 10 wavefronts per CU) when the batch has at least
    // small_table_min_blocks of them
    int small_table_kernel = 1;
    uint64_t small_table_min_blocks = 256;
    // k_compress_block_lds (one block per CU, input block in LDS as well):
    // 1 = for batches of at most two blocks per CU (default), 0 = never,
    // 2 = whenever the wavefront kernel would run (tests)
    int small_batch_kernel = 1;
    // the wavefront-per-block kernels' step: 1 (default) = a window of 63
    // consecutive positions per step, every parse event inside it resolved
    // by a walk over registers (k_compress_spans / k_compress_span_lds);
    // 0 = one copy per step (k_compress_blocks / k_compress_block_lds, the
    // kernels of rounds 1-3: kept as the cross-check)
    int span_kernel = 1;
    // compress_mode 2 ("both"): CUs the wavefront kernel's persistent
    // workgroups take (each owns a CU's whole LDS, so the lane kernel's
    // wavefronts run on the others); 0 = half of the CUs
    uint32_t both_wave_cus = 0;
    // 1: the per-batch scratch of the compressor (token arrays: 128 KiB per
    // block of a lane-kernel segment, up to 34 GB; block slots of the
    // wavefront kernels) is given back when the batch's results are waited
    // for (snapmi_ctx_synchronize; the scalar and libsnappy entry points,
    // which wait for their own result) instead of being kept for the next
    // batch.  snapmi_last_timing waits for events only and the host frame
    // calls keep their pipeline's scratch: neither frees.  For a host
    // that compresses now and then and shares the GPU; costs a hipFree /
    // hipMalloc pair per batch.  The lane tables are not scratch: they are
    // bounded by lane_table_budget_pct.
    int release_scratch = 0;
    // the match finder of the token path (large batches, compress_mode 1):
    // 0 = k_match_blocks (a lane per block, hash tables in HBM), 1 =
    // k_match_spans (a wavefront per block, table in LDS, no tables in HBM
    // at all), 2 (default) = by what this context's last batch compressed
    // to: data that does not compress (ratio >= match_spans_ratio_pct) costs
    // a lane three HBM transactions per probe for nothing
    int match_kernel = 2;
    uint32_t match_spans_ratio_pct = 90;
    // the hint: pinned words the last token-path batch posted (compressed
    // bytes, input bytes, its number)
    volatile uint32_t *h_ratio = nullptr;
    uint32_t ratio_seq = 0; // batches posted so far
    // the token pool (CompressArgs::tok_pool): per cent of the worst case it
    // is sized to (option token_pool_pct), what that has grown to behind
    // batches that spilled, the floor in pages (64 MiB: small batches never
    // spill; test option token_pool_min_pages), and what k_redo_spilled
    // posted of the last launch it has finished: pages asked for | blocks
    // spilled | blocks | seq
    uint32_t token_pool_pct = 39, token_pool_now = 0;
    uint32_t token_pool_min_pages = 32768;
    volatile uint32_t *h_tokstat = nullptr;
    uint32_t tokstat_seq = 0, tokstat_seen = 0;
    uint32_t tok_pages_asked = 0, tok_blocks_spilled = 0;
    uint32_t tok_pool_pages_last = 0; // pages of the last launch's pool
    // 1 (default): a lane-kernel launch of at most lane_speculate_max_blocks
    // blocks (and no more blocks than lanes) runs k_match_blocks_spec (a
    // probe's round also fetches the entry of the probe that follows a
    // miss); 0: always the plain kernel.  Measured: -10..15 % at 2 048 ..
    // 16 384 blocks of text, nothing at 32 768 (test option to move the
    // limit: lane_speculate_max_blocks).
    int lane_speculate = 1;
    uint64_t lane_speculate_max_blocks = 24576;
    // k_compress_tiny (streams of fewer than 256 bytes, one per LANE, all of
    // their state in LDS): 1 = on (default), 0 = such streams are one-block
    // streams of the block kernels (cross-check, and what round 2 measured)
    int tiny_stream_kernel = 1;
    // k_compress_small (streams of 256 bytes and more, a few per wavefront,
    // state in LDS; needs tiny_stream_kernel): 1 = up to 1 023 bytes
    // (default), 2 = up to 2 047 (measured slower than the lane kernel from
    // 1 KiB on), 0 = off
    int small_stream_kernel = 1;
    // batches of more streams than this are decoded by
    // k_decompress_streams3_many (kManyStreams streams per workgroup)
    uint64_t decode_many_min = 1u << 20;
    // the lane kernel's segment is matched in two halves and the first
    // half's tokens are encoded on the side stream meanwhile: 1 = when the
    // segment has at least 1.4 blocks per lane, 0 = never (default: measured
    // SLOWER, 121.6 -> 135 ms at cfg2 - the encoder's streaming traffic under
    // the match finder costs its random accesses far more than the 3.3 ms
    // it hides), 2 = whenever the segment has two blocks (tests)
    int lane_overlap_encode = 0;
    // 1: the lane kernel's encoder writes every block at its final position
    // (sizes are known after matching); 0: scratch slots + k_compact
    int lane_direct_encode = 1;
    // 1: the lane tables come from hipExtMallocWithFlags(hipDeviceMallocUncached)
    int lane_tables_uncached = 0;
    // 3: k_decompress_streams3 (element per lane, 128-byte windows; default);
    // 2: k_decompress_streams2 (64-byte windows), kept as a cross-check;
    // 0: the sequential decoder alone
    int decode_kernel = 3;
    hipStream_t stream2 = nullptr; // the wavefront kernel's side stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_crc[2] = {nullptr, nullptr}; // frame encode: CRC on stream2
    bool frame_crc_side_stream = true;
    // staging for the host-pointer (scalar) entry points
    snapmi::DevBuf st_in, st_out, st_desc, st_prof, ticket, order;
    // long-stream decode scratch (snapmi_decompress_stream)
    snapmi::DevBuf sd_tables, sd_desc;
    // the long streams of a small batch get their pieces
    // (snapmi_decompress_batch, option batch_long_streams): modes of the
    // batch's own launch and of the one behind, what k_long_plan found, the
    // streams' descriptors and workgroup prefixes; pinned staging of both
    int batch_long_streams = 1;
    snapmi::DevBuf bl_modes, bl_list, bl_descs, bl_order;
    void *pin_bl = nullptr;
    size_t pin_bl_cap = 0;
    void *pin_bl2 = nullptr; // descriptors of the long streams of a batch
    // segment size of the long-stream scan: 0 = by size (1 KiB under 256 MiB
    // of long streams, 4 KiB from there), 10 / 12 forced (test option)
    uint32_t stream_seg_log2 = 0;
    uint32_t stream_scan_segs = 0; // test option: 0 = by size
    size_t pin_bl2_cap = 0;
    // frame layer scratch (snapmi_frame.hip)
    snapmi::DevBuf fr_tables, fr_desc, fr_meta, fr_scan, fr_slots, fr_chunk_off;
    bool fr_tables_ready = false;
    // framed streams of at least this many bytes without a side index get
    // their chunk headers found in parallel (k_fw_*); shorter ones are walked
    uint64_t frame_parallel_walk_min = 4ull << 20;
    uint64_t frame_walk_segment = 32ull << 20; // >= 128 KiB (test knob)
    int num_cus = 0;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // measured crossover (profiles/r5_crossover.txt): the corpus round 1.7
    // GiB, alice29.txt 1.15, html 0.55 - 8 192 until the window kernel got
    // its lane-parallel walk in round 5
    uint32_t lane_min_blocks = 20480;
    // blocks per lane-kernel launch: bounds the token scratch (34 GB here)
    uint32_t lane_segment_blocks = 262144;
    uint32_t lane_waves_per_cu = 6; // 24 KiB of LDS per wave
    uint32_t lane_max_waves = 0;    // test knob: cap on lane-kernel waves (0 = none)
    // placements of the lane tables that are timed before one is kept
    // (place_lane_tables in snapmi_api.hip: spread over the budget, then
    // packed behind that, then spread again; a candidate costs one hipMalloc
    // and a 3 ms probe, and the driver wipes what a loser gives back)
    uint32_t lane_table_tries = 2;
    // percent of the free device memory the lane tables (and, while a
    // placement is chosen, their candidates) may hold: the GPU may be shared
    uint32_t lane_table_budget_pct = 33;
    bool lane_table_probe = false; // experiment knob: probe even with 1 try
    uint32_t lane_table_stride_kib = 0; // experiment knob: bytes between tables
    std::string probe_log;          // k_probe_tables ms of every candidate
    // test knob: every lane's table epoch is set to this value before the
    // next lane-kernel launch (-1 = leave the epochs alone); lets a test
    // reach the 16-bit epoch wrap without 65 535 blocks per lane
    int64_t lane_epoch_preset = -1;
    // ds_mskor_rtn_b32 applies same-address lanes in ascending lane order on
    // this device (checked by k_probe_lds_order at context creation); the
    // wavefront-per-block kernel is only used when this holds
    bool lds_order_ok = false;
    bool lds_order_hw = false;
    // overlapping lanes of one plain DS store land in ascending lane order
    // (same self-check): k_decompress_streams2 needs it
    bool lds_store_order_ok = true; // what the self-check found (the option can only lower it)
    bool timing_valid = false;
    bool timing_is_compress = false;
    bool dominant_split = false; // ev[4]/ev[5] bracket k_match_blocks
    uint64_t codec_launches = 0;
    uint32_t seam_seq = 0; // the seam's single-launch path: its last ticket
};

namespace snapmi {

inline int fail_ctx(snapmi_ctx *ctx, int kind, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->last_error = buf;
    return kind;
}

#define HIP_TRY(ctx, expr)                                                    \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess)                                                 \
            return snapmi::fail_ctx((ctx), SNAPMI_E_DEVICE, "%s failed: %s",  \
                                    #expr, hipGetErrorString(_e));            \
    } while (0)

// (slack: an eighth more than asked for, so that batches that grow a little
// do not reallocate every time - or none, for the token pool, whose size is a
// stated share of the input)
inline int reserve(snapmi_ctx *ctx, DevBuf &b, size_t bytes,
                   bool slack = true)
{
    if (bytes <= b.cap)
        return SNAPMI_OK;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + (slack ? bytes / 8 : 0) + 256;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return SNAPMI_OK;
}

} // namespace snapmi

// internal launchers (snapmi_api.hip), shared with the frame layer
namespace snapmi {
void host_pipe_destroy(snapmi_ctx *ctx); // snapmi_frame.hip
// raw compress of n streams; blocks/slots = launch geometry computed from
// the (host-known) stream lengths
int launch_compress(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                    const uint64_t *d_in_lens, void *const *d_out_ptrs,
                    const uint64_t *d_out_caps, uint64_t *d_out_lens,
                    snapmi_error *d_errs, size_t n, uint64_t blocks,
                    uint64_t slots, uint32_t small_classes = 0xF,
                    uint64_t cnt8 = 0, uint64_t block_bytes = 0);
// raw decompress; d_modes optional (1 = stored chunk, plain copy)
int launch_decompress(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                      const uint64_t *d_in_lens, void *const *d_out_ptrs,
                      const uint64_t *d_out_caps, uint64_t *d_out_lens,
                      snapmi_error *d_errs, const uint8_t *d_modes, size_t n,
                      const unsigned long long *d_gate = nullptr,
                      unsigned long long gate_value = 0,
                      // a launch beside the context's stream: its stream and
                      // its own dispatch-order scratch (no timing events)
                      hipStream_t side = nullptr, DevBuf *side_order = nullptr);
} // namespace snapmi
