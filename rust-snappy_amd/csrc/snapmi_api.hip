// snapmi_api.hip -- host side of the C ABI declared in include/snapmi.h.
//
// Nothing here computes Snappy on the CPU: the only host-side arithmetic is
// the header varint parse (decompress_len) and max_compress_len, which the
// reference also treats as free-standing helpers (src/compress.rs:42-53,
// src/decompress.rs:30-35).  Every compress/decompress entry point launches
// the HIP kernels and fails with SNAPMI_E_DEVICE when there is no GPU.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <deque>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "snapmi.h"
#include "snapmi_test.h"
#include "snapmi_ctx.hpp"
#include "snapmi_pool.hpp"
#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"

using namespace snapmi;



namespace {

void set_err(snapmi_error *e, int kind, uint64_t a = 0, uint64_t b = 0,
             uint64_t c = 0)
{
    if (e) {
        e->kind = kind;
        e->reserved = 0;
        e->a = a;
        e->b = b;
        e->c = c;
    }
}

// reference bytes::read_varu64, src/bytes.rs:73-90
size_t host_varint(const uint8_t *p, size_t n, uint64_t *value)
{
    uint64_t acc = 0;
    unsigned shift = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t b = p[i];
        if (shift >= 64)
            return 0;
        if (b < 0x80) {
            *value = acc | ((uint64_t)b << shift);
            return i + 1;
        }
        acc |= (uint64_t)(b & 0x7F) << shift;
        shift += 7;
    }
    return 0;
}

} // namespace

namespace snapmi {
// the lane tables back to the device: a hipMalloc region, or physical chunks
// mapped into one address range (place_lane_tables)
static void free_lane_tables(snapmi_ctx *ctx)
{
    if (!ctx->lane_tables.p)
        return;
    if (ctx->lane_va_bytes) {
        (void)hipMemUnmap(ctx->lane_tables.p, ctx->lane_va_bytes);
        for (auto h : ctx->lane_chunks)
            (void)hipMemRelease(h);
        (void)hipMemAddressFree(ctx->lane_tables.p, ctx->lane_va_bytes);
        ctx->lane_chunks.clear();
        ctx->lane_va_bytes = 0;
    } else {
        (void)hipFree(ctx->lane_tables.p);
    }
    ctx->lane_tables.p = nullptr;
    ctx->lane_tables.cap = 0;
    ctx->n_lanes = 0;
}
} // namespace snapmi

namespace snapmi {
// option release_scratch: the compressor's per-batch scratch goes back to
// the allocator.  Only behind a synchronisation of ctx->stream (nothing of a
// batch is in flight): snapmi_ctx_synchronize and the scalar / libsnappy
// entry points, which wait for their result anyway.
int release_batch_scratch(snapmi_ctx *ctx)
{
    if (!ctx->release_scratch)
        return SNAPMI_OK;
    for (DevBuf *b : {&ctx->tokens, &ctx->tok_pages, &ctx->tok_stage,
                      &ctx->slots}) {
        if (b->p) {
            HIP_TRY(ctx, hipFree(b->p));
            b->p = nullptr;
            b->cap = 0;
        }
    }
    return SNAPMI_OK;
}
} // namespace snapmi

extern "C" {

const char *snapmi_version(void) { return "snapmi 0.1.0 gfx950"; }

int snapmi_ctx_create(int device, void *hip_stream, snapmi_ctx **out)
{
    if (!out)
        return SNAPMI_E_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device < 0 || device >= count) {
        fprintf(stderr,
                "snapmi: no usable HIP device (requested %d, %d visible: "
                "%s); the codec has no CPU fallback\n",
                device, count, hipGetErrorString(e));
        return SNAPMI_E_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess)
        return SNAPMI_E_DEVICE;
    snapmi_ctx *ctx = new (std::nothrow) snapmi_ctx();
    if (!ctx)
        return SNAPMI_E_DEVICE;
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
            delete ctx;
            return SNAPMI_E_DEVICE;
        }
        ctx->num_cus = prop.multiProcessorCount;
    }
    // experiment knobs from the environment (tests/hw drivers): only for a
    // process that says SNAPMI_TESTING=1 - a production host sets what it
    // needs through snapmi_ctx_set_option
#ifdef SNAPMI_TESTING // (the test build, libsnapmi_test.so: Makefile)
    if (getenv("SNAPMI_TESTING")) {
        if (const char *e = getenv("SNAPMI_LANE_WAVES")) {
            const int v = atoi(e);
            ctx->lane_waves_per_cu = (uint32_t)(v < 1 ? 1 : (v > 32 ? 32 : v));
        }
        if (const char *e = getenv("SNAPMI_LANE_SEGMENT_BLOCKS")) {
            const long v = atol(e);
            ctx->lane_segment_blocks = (uint32_t)(v < 64 ? 64 : v);
        }
        if (const char *e = getenv("SNAPMI_LANE_TABLE_SPREAD"))
            ctx->lane_table_spread = atoi(e) != 0;
        if (const char *e = getenv("SNAPMI_LANE_MIN_BLOCKS")) {
            const long v = atol(e);
            ctx->lane_min_blocks = (uint32_t)(v < 1 ? 1 : v);
        }
        if (const char *e = getenv("SNAPMI_FRAME_CRC_SIDE"))
            ctx->frame_crc_side_stream = atoi(e) != 0;
        if (const char *e = getenv("SNAPMI_LANE_UNCACHED"))
            ctx->lane_tables_uncached = atoi(e) != 0;
        if (const char *e = getenv("SNAPMI_LANE_DIRECT"))
            ctx->lane_direct_encode = atoi(e) != 0;
        if (const char *e = getenv("SNAPMI_HOST_COPY_KERNEL"))
            ctx->host_copy_kernel = atoi(e) & 3;
        if (const char *e = getenv("SNAPMI_HOST_ENCODE_SLICE"))
            ctx->host_encode_slice = (uint64_t)atoll(e) < 65536
                                         ? 65536
                                         : (uint64_t)atoll(e);
        if (const char *e = getenv("SNAPMI_HOST_DECODE_CHUNKS"))
            ctx->host_decode_slice_chunks =
                atoll(e) < 1 ? 1 : (uint64_t)atoll(e);
        if (const char *e = getenv("SNAPMI_DECODE_KERNEL"))
            ctx->decode_kernel = atoi(e) == 0 ? 0 : (atoi(e) == 2 ? 2 : 3);
        if (const char *e = getenv("SNAPMI_SPAN_KERNEL"))
            ctx->span_kernel = atoi(e) != 0;
        if (const char *m = getenv("SNAPMI_COMPRESS"))
            ctx->compress_mode = strcmp(m, "waves") == 0
                                     ? 0
                                     : (strcmp(m, "lanes") == 0 ? 1 : 2);
    }
#endif
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) !=
            hipSuccess) {
            delete ctx;
            return SNAPMI_E_DEVICE;
        }
        ctx->owns_stream = true;
    }
    for (auto &ev : ctx->ev) {
        if (hipEventCreate(&ev) != hipSuccess) {
            snapmi_ctx_destroy(ctx);
            return SNAPMI_E_DEVICE;
        }
    }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) !=
            hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) !=
            hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) !=
            hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_crc[0], hipEventDisableTiming) !=
            hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_crc[1], hipEventDisableTiming) !=
            hipSuccess) {
        snapmi_ctx_destroy(ctx);
        return SNAPMI_E_DEVICE;
    }
    // Hardware self-check the wavefront-per-block compressor rests on (one
    // tiny kernel): without the ascending-lane order of DS atomics that
    // kernel would silently emit non-reference streams, so a context on such
    // a device compresses with the lane-per-block kernel only.
    {
        uint32_t bad = 1;
        if (reserve(ctx, ctx->ticket, 64) != SNAPMI_OK ||
            hipMemsetAsync(ctx->ticket.p, 0, 64, ctx->stream) != hipSuccess) {
            snapmi_ctx_destroy(ctx);
            return SNAPMI_E_DEVICE;
        }
        // (a launch that fills the chip: the properties are checked under
        // the LDS contention of the real kernels)
        hipLaunchKernelGGL(k_probe_lds_order,
                           dim3((uint32_t)(ctx->num_cus > 0 ? ctx->num_cus * 32
                                                             : 32)),
                           dim3(64), 0, ctx->stream,
                           (uint32_t *)ctx->ticket.p);
        if (hipMemcpyAsync(&bad, ctx->ticket.p, 4, hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            fprintf(stderr, "snapmi: LDS order self-check did not run: %s\n",
                    hipGetErrorString(hipGetLastError()));
            snapmi_ctx_destroy(ctx);
            return SNAPMI_E_DEVICE;
        }
        ctx->lds_order_ok = ctx->lds_order_hw = (bad & 1) == 0;
        if (bad & 2) { // the element-major decoder needs ordered DS stores
            ctx->lds_store_order_ok = false;
            ctx->decode_kernel = 0;
            fprintf(stderr,
                    "snapmi: this device does not apply overlapping lanes of "
                    "one DS store in ascending lane order; streams are "
                    "decoded one element at a time\n");
        }
        if (!ctx->lds_order_ok)
            fprintf(stderr,
                    "snapmi: this device does not apply the lanes of one DS "
                    "atomic in ascending lane order; the wavefront-per-block "
                    "compressor is disabled (lane-per-block kernel only)\n");
    }
    *out = ctx;
    return SNAPMI_OK;
}

void snapmi_ctx_destroy(snapmi_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    host_pipe_destroy(ctx);
    for (void *p : {ctx->pin_in, ctx->pin_out, ctx->pin_desc, ctx->pin_bl,
                    ctx->pin_bl2})
        if (p)
            (void)hipHostFree(p);
    if (ctx->h_mail)
        (void)hipHostFree((void *)ctx->h_mail);
    if (ctx->h_ratio)
        (void)hipHostFree((void *)ctx->h_ratio);
    if (ctx->h_tokstat)
        (void)hipHostFree((void *)ctx->h_tokstat);
    snapmi::free_lane_tables(ctx);
    for (DevBuf *b : {&ctx->blk_first, &ctx->slot_first, &ctx->blk_size,
                      &ctx->blk_off, &ctx->slots, &ctx->plan_part,
                      &ctx->st_in, &ctx->st_out,
                      &ctx->st_desc, &ctx->st_prof, &ctx->ticket,
                      &ctx->order, &ctx->fr_tables, &ctx->fr_desc,
                      &ctx->fr_meta, &ctx->fr_scan, &ctx->fr_slots,
                      &ctx->fr_chunk_off,
                      &ctx->tokens, &ctx->tok_pages, &ctx->tok_stage,
                      &ctx->ntok, &ctx->sched,
                      &ctx->lane_epochs, &ctx->sd_tables, &ctx->sd_desc,
                      &ctx->bl_modes, &ctx->bl_list, &ctx->bl_descs,
                      &ctx->bl_order})
        if (b->p)
            (void)hipFree(b->p);
    for (auto &ev : ctx->ev)
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    if (ctx->ev_fork)
        (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join)
        (void)hipEventDestroy(ctx->ev_join);
    for (auto &ev : ctx->ev_crc)
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->owns_stream && ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int snapmi_ctx_set_option(snapmi_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name)
        return SNAPMI_E_ARGUMENT;
#ifdef SNAPMI_TESTING
    // cross-checks of the test build: both compressors at once on one
    // ticket (compress_mode 2), the one-copy-per-step wavefront kernels of
    // rounds 1-3 (span_kernel 0), the second-generation decoder alone
    // (decode_kernel 2)
    constexpr int64_t kModeMax = 2, kSpanMin = 0;
    constexpr bool kDec2 = true;
#else
    constexpr int64_t kModeMax = 1, kSpanMin = 1;
    constexpr bool kDec2 = false;
#endif
    if (strcmp(name, "compress_mode") == 0 && value >= 0 && value <= kModeMax)
        ctx->compress_mode = (int)value;
    else if (strcmp(name, "lane_min_blocks") == 0 && value >= 1)
        ctx->lane_min_blocks = (uint32_t)value;
    else if (strcmp(name, "lane_segment_blocks") == 0 && value >= 64 &&
             value <= 0x7FFFFFFF)
        ctx->lane_segment_blocks = (uint32_t)value;
    else if (strcmp(name, "lane_table_spread") == 0 && value >= 0 &&
             value <= 1)
        ctx->lane_table_spread = value != 0;
    else if (strcmp(name, "lane_table_tries") == 0 && value >= 1 &&
             value <= 16)
        ctx->lane_table_tries = (uint32_t)value;
    else if (strcmp(name, "lane_table_budget_pct") == 0 && value >= 1 &&
             value <= 90)
        ctx->lane_table_budget_pct = (uint32_t)value;
    else if (strcmp(name, "window_tokens") == 0 && value >= 0 && value <= 1)
        ctx->window_tokens = (int)value;
    else if (strcmp(name, "token_pool_pct") == 0 && value >= 1 &&
             value <= 100) {
        ctx->token_pool_pct = (uint32_t)value;
        ctx->token_pool_now = 0; // (what it had grown to is forgotten)
    } else if (strcmp(name, "token_pool_min_pages") == 0 && value >= 0 &&
               value <= 0x7FFFFFFF)
        ctx->token_pool_min_pages = (uint32_t)value;
    else if (strcmp(name, "span_schedule") == 0 && value >= 0 && value <= 2)
        ctx->span_schedule = (int)value;
    else if (strcmp(name, "lane_coresident") == 0 && value >= 0 && value <= 1)
        ctx->lane_coresident = (int)value;
    else if (strcmp(name, "lane_coresident_min_blocks") == 0 && value >= 1)
        ctx->lane_coresident_min_blocks = (uint64_t)value;
    else if (strcmp(name, "small_table_kernel") == 0 && value >= 0 &&
             value <= 1)
        ctx->small_table_kernel = (int)value;
    else if (strcmp(name, "small_table_min_blocks") == 0 && value >= 1)
        ctx->small_table_min_blocks = (uint64_t)value;
    else if (strcmp(name, "small_batch_kernel") == 0 && value >= 0 &&
             value <= 2)
        ctx->small_batch_kernel = (int)value;
    else if (strcmp(name, "lane_speculate") == 0 && value >= 0 && value <= 1)
        ctx->lane_speculate = (int)value;
    else if (strcmp(name, "span_kernel") == 0 && value >= kSpanMin &&
             value <= 1)
        ctx->span_kernel = (int)value;
    else if (strcmp(name, "match_kernel") == 0 && value >= 0 && value <= 2)
        ctx->match_kernel = (int)value;
    else if (strcmp(name, "match_spans_ratio_pct") == 0 && value >= 1 &&
             value <= 200)
        ctx->match_spans_ratio_pct = (uint32_t)value;
    else if (strcmp(name, "release_scratch") == 0 && value >= 0 && value <= 1)
        ctx->release_scratch = (int)value;
    else if (strcmp(name, "batch_long_streams") == 0 && value >= 0 &&
             value <= 1)
        ctx->batch_long_streams = (int)value;
    else if (strcmp(name, "both_wave_cus") == 0 && value >= 0 &&
             value <= 4096)
        ctx->both_wave_cus = (uint32_t)value;
    else if (strcmp(name, "tiny_stream_kernel") == 0 && value >= 0 &&
             value <= 1)
        ctx->tiny_stream_kernel = (int)value;
    else if (strcmp(name, "small_stream_kernel") == 0 && value >= 0 &&
             value <= 2)
        ctx->small_stream_kernel = (int)value;
    else if (strcmp(name, "frame_parallel_walk_min") == 0 && value >= 0)
        ctx->frame_parallel_walk_min = (uint64_t)value;
    else if (strcmp(name, "host_copy_kernel") == 0 && value >= 0 && value <= 3)
        ctx->host_copy_kernel = (int)value;
    else if (strcmp(name, "host_encode_slice") == 0 && value >= (1 << 16))
        ctx->host_encode_slice = (uint64_t)value;
    else if (strcmp(name, "host_decode_slice_chunks") == 0 && value >= 1)
        ctx->host_decode_slice_chunks = (uint64_t)value;
    else if (strcmp(name, "decode_kernel") == 0 &&
             (value == 0 || (value == 2 && kDec2) || value == 3))
        ctx->decode_kernel = ctx->lds_store_order_ok ? (int)value : 0;
    else
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "unknown option %s", name);
    return SNAPMI_OK;
}

#ifdef SNAPMI_TESTING
// include/snapmi_test.h: knobs of the test suite and the experiment drivers
// (libsnapmi_test.so only: the product library does not export it)
int snapmi_ctx_set_test_option(snapmi_ctx *ctx, const char *name,
                               int64_t value)
{
    if (!ctx || !name)
        return SNAPMI_E_ARGUMENT;
    if (strcmp(name, "lane_waves_per_cu") == 0 && value >= 1 && value <= 32)
        ctx->lane_waves_per_cu = (uint32_t)value;
    else if (strcmp(name, "decode_many_min") == 0 && value >= 0)
        ctx->decode_many_min = (uint64_t)value;
    else if (strcmp(name, "lane_speculate_max_blocks") == 0 && value >= 0)
        ctx->lane_speculate_max_blocks = (uint64_t)value;
    else if (strcmp(name, "lane_max_waves") == 0 && value >= 0 &&
             value <= 0x7FFFFFFF)
        ctx->lane_max_waves = (uint32_t)value;
    else if (strcmp(name, "frame_crc_side_stream") == 0 && value >= 0 &&
             value <= 1)
        ctx->frame_crc_side_stream = value != 0;
    else if (strcmp(name, "lane_tables_uncached") == 0 && value >= 0 &&
             value <= 1)
        ctx->lane_tables_uncached = (int)value;
    else if (strcmp(name, "lane_direct_encode") == 0 && value >= 0 &&
             value <= 1)
        ctx->lane_direct_encode = (int)value;
    else if (strcmp(name, "lane_overlap_encode") == 0 && value >= 0 &&
             value <= 2)
        ctx->lane_overlap_encode = (int)value;
    else if (strcmp(name, "frame_walk_segment") == 0 && value >= (128 << 10))
        ctx->frame_walk_segment = (uint64_t)value;
    else if (strcmp(name, "lane_table_stride_kib") == 0 &&
             (value == 0 || (value >= 256 && value <= 4096 && value % 4 == 0)))
        ctx->lane_table_stride_kib = (uint32_t)value; // 0: by the budget
    else if (strcmp(name, "lane_table_probe") == 0 && value >= 0 &&
             value <= 1)
        ctx->lane_table_probe = value != 0; // time the placement even if 1 try
    else if (strcmp(name, "lane_tables_renew") == 0 && value == 1) {
        // drop the tables so the next launch places new ones
        if (ctx->lane_tables.p) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            snapmi::free_lane_tables(ctx);
        }
    } else if (strcmp(name, "lane_epoch_preset") == 0 && value >= -1 &&
             value <= 0xFFFF)
        ctx->lane_epoch_preset = value;
    else if (strcmp(name, "stream_seg_log2") == 0 &&
             (value == 0 || value == 10 || value == 12))
        ctx->stream_seg_log2 = (uint32_t)value; // 0: by size
    else if (strcmp(name, "stream_scan_segs") == 0 &&
             (value == 0 || value == 8 || value == 16 || value == 32 ||
              value == 64))
        ctx->stream_scan_segs = (uint32_t)value; // 0: by size
    else if (strcmp(name, "lds_order_ok") == 0 && value >= 0 && value <= 1)
        ctx->lds_order_ok = ctx->lds_order_hw && value != 0; // can only lower
    else
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "unknown test option %s",
                        name);
    return SNAPMI_OK;
}
#endif // SNAPMI_TESTING

void *snapmi_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void snapmi_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

const char *snapmi_table_probe_log(const snapmi_ctx *ctx)
{
    return ctx ? ctx->probe_log.c_str() : "";
}

size_t snapmi_error_string(const snapmi_error *e, char *buf, size_t cap)
{
    // reference src/error.rs:249-335, variant by variant
    char tmp[256];
    const unsigned long long a = e ? e->a : 0, b = e ? e->b : 0,
                             c = e ? e->c : 0;
    int n = 0;
    switch (e ? e->kind : -1) {
    case SNAPMI_TOO_BIG:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: input buffer (size = %llu) is larger than "
                     "allowed (size = %llu)", a, b);
        break;
    case SNAPMI_BUFFER_TOO_SMALL:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: output buffer (size = %llu) is smaller than "
                     "required (size = %llu)", a, b);
        break;
    case SNAPMI_EMPTY:
        n = snprintf(tmp, sizeof tmp, "snappy: corrupt input (empty)");
        break;
    case SNAPMI_HEADER:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (invalid header)");
        break;
    case SNAPMI_HEADER_MISMATCH:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (header mismatch; expected %llu "
                     "decompressed bytes but got %llu)", a, b);
        break;
    case SNAPMI_LITERAL:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected literal read of length "
                     "%llu; remaining src: %llu; remaining dst: %llu)", a, b,
                     c);
        break;
    case SNAPMI_COPY_READ:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected copy read of length "
                     "%llu; remaining src: %llu)", a, b);
        break;
    case SNAPMI_COPY_WRITE:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected copy write of length "
                     "%llu; remaining dst: %llu)", a, b);
        break;
    case SNAPMI_OFFSET:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected valid offset but got "
                     "offset %llu; dst position: %llu)", a, b);
        break;
    case SNAPMI_STREAM_HEADER:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected stream header but got "
                     "unexpected chunk type byte %llu)", a);
        break;
    case SNAPMI_STREAM_HEADER_MISMATCH: {
        // (six bytes, little-endian in `a`; std::ascii::escape_default)
        char esc[6 * 4 + 1];
        int k = 0;
        for (int i = 0; i < 6; i++) {
            const unsigned ch = (unsigned)(a >> (8 * i)) & 0xFF;
            if (ch == 9 || ch == 10 || ch == 13) {
                esc[k++] = '\\';
                esc[k++] = ch == 9 ? 't' : (ch == 10 ? 'n' : 'r');
            } else if (ch == 39 || ch == 34 || ch == 92) {
                esc[k++] = '\\';
                esc[k++] = (char)ch;
            } else if (ch >= 0x20 && ch <= 0x7E) {
                esc[k++] = (char)ch;
            } else {
                k += snprintf(esc + k, 5, "\\x%02x", ch);
            }
        }
        esc[k] = 0;
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (expected sNaPpY stream header "
                     "but got %s)", esc);
        break;
    }
    case SNAPMI_UNSUPPORTED_CHUNK_TYPE:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (unsupported chunk type: %llu)",
                     a);
        break;
    case SNAPMI_UNSUPPORTED_CHUNK_LENGTH:
        n = snprintf(tmp, sizeof tmp,
                     b ? "snappy: corrupt input (invalid stream header "
                         "length: %llu)"
                       : "snappy: corrupt input (unsupported chunk length: "
                         "%llu)", a);
        break;
    case SNAPMI_CHECKSUM:
        n = snprintf(tmp, sizeof tmp,
                     "snappy: corrupt input (bad checksum; expected: %llu, "
                     "got: %llu)", a, b);
        break;
    case SNAPMI_E_UNEXPECTED_EOF:
        // (not a snap::Error: the io::Error of read_exact that the
        // reference's reader passes on, src/read.rs:105-172 - std's text)
        n = snprintf(tmp, sizeof tmp, "failed to fill whole buffer");
        break;
    case SNAPMI_OK:
        n = snprintf(tmp, sizeof tmp, "ok");
        break;
    default:
        n = snprintf(tmp, sizeof tmp, "snapmi: device or argument error %d",
                     e ? e->kind : -1);
    }
    if (buf && cap) {
        const size_t m = (size_t)n < cap - 1 ? (size_t)n : cap - 1;
        memcpy(buf, tmp, m);
        buf[m] = 0;
    }
    return (size_t)n;
}

const char *snapmi_last_kernel(const snapmi_ctx *ctx)
{
    return ctx ? ctx->last_kernel : "";
}

const char *snapmi_last_error(const snapmi_ctx *ctx)
{
    return ctx ? ctx->last_error.c_str() : "null context";
}

void *snapmi_ctx_stream(const snapmi_ctx *ctx)
{
    return ctx ? (void *)ctx->stream : nullptr;
}

int snapmi_ctx_synchronize(snapmi_ctx *ctx)
{
    if (!ctx)
        return SNAPMI_E_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return snapmi::release_batch_scratch(ctx);
}

size_t snapmi_max_compress_len(size_t input_len)
{
    return (size_t)max_compress_len_u64(input_len);
}

int snapmi_decompress_len(const uint8_t *input, size_t input_len,
                          size_t *result, snapmi_error *err)
{
    if (!result || (!input && input_len))
        return SNAPMI_E_ARGUMENT;
    *result = 0;
    set_err(err, SNAPMI_OK);
    if (input_len == 0)
        return SNAPMI_OK;
    uint64_t v = 0;
    if (host_varint(input, input_len, &v) == 0) {
        set_err(err, SNAPMI_HEADER);
        return SNAPMI_HEADER;
    }
    if (v > kMaxInput) {
        set_err(err, SNAPMI_TOO_BIG, v, kMaxInput);
        return SNAPMI_TOO_BIG;
    }
    *result = (size_t)v;
    return SNAPMI_OK;
}

} // extern "C"
namespace snapmi {
// streams shorter than this are compressed by the lane-per-stream kernels
// (k_compress_tiny under 256 bytes, k_compress_small under 1 KiB - under
// 2 KiB with small_stream_kernel = 2) and get no blocks; 0: every stream goes
// through the block kernels
int prepare_lane_tables(snapmi_ctx *ctx, uint64_t blocks, bool top);
static uint64_t small_stream_limit(const snapmi_ctx *ctx)
{
    if (!ctx->tiny_stream_kernel)
        return 0;
    return ctx->small_stream_kernel == 2
               ? kSmallCompress
               : (ctx->small_stream_kernel ? kSmallCompress / 2
                                           : kTinyCompress);
}
} // namespace snapmi
extern "C" {

// ----------------------------------------------------------------------
// batched device-resident API
// ----------------------------------------------------------------------
int snapmi_compress_batch(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                          const uint64_t *d_in_lens,
                          const uint64_t *h_in_lens, void *const *d_out_ptrs,
                          const uint64_t *d_out_caps, uint64_t *d_out_lens,
                          snapmi_error *d_errs, size_t n)
{
    if (!ctx)
        return SNAPMI_E_ARGUMENT;
    if (n == 0)
        return SNAPMI_OK;
    if (!d_in_ptrs || !d_in_lens || !d_out_ptrs || !d_out_lens ||
        n > 0x7FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "compress_batch: bad args");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    std::vector<uint64_t> fetched;
    if (!h_in_lens) {
        fetched.resize(n);
        HIP_TRY(ctx, hipMemcpyAsync(fetched.data(), d_in_lens,
                                    n * sizeof(uint64_t),
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        h_in_lens = fetched.data();
    }
    uint64_t blocks = 0, slots = 0, cnt8 = 0, block_bytes = 0;
    // streams under this length are the lane-per-stream kernels': no block
    const uint64_t small = small_stream_limit(ctx);
    uint32_t classes = 0; // which of those kernels have anything to do
    for (size_t i = 0; i < n; i++) {
        const uint64_t len = h_in_lens[i];
        // (first the cheap test: a batch of ten million tiny streams is
        // walked here once per call)
        if (len < small) {
            classes |= len < kTinyCompress ? (len ? 1u : 0u)
                                           : (len < 512 ? 2u
                                                        : (len < 1024 ? 4u : 8u));
            continue;
        }
        if (len == 0 || max_compress_len_u64(len) == 0)
            continue;
        const uint64_t nb = (len + kMaxBlock - 1) / kMaxBlock;
        blocks += nb;
        block_bytes += len;
        slots += nb - 1;
        // the stream's last block: a page, a short chunk, a tail?
        const uint64_t last = len - (nb - 1) * kMaxBlock;
        cnt8 += last <= 8192;
    }
    return launch_compress(ctx, d_in_ptrs, d_in_lens, d_out_ptrs, d_out_caps,
                           d_out_lens, d_errs, n, blocks, slots, classes,
                           cnt8, block_bytes);
}

int snapmi_ctx_prepare(snapmi_ctx *ctx, uint64_t blocks, uint32_t flags)
{
    if (!ctx || (flags & ~(uint32_t)SNAPMI_PREPARE_TOP_OF_MEMORY))
        return SNAPMI_E_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return snapmi::prepare_lane_tables(
        ctx, blocks, (flags & SNAPMI_PREPARE_TOP_OF_MEMORY) != 0);
}

int snapmi_ctx_get_info(snapmi_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value)
        return SNAPMI_E_ARGUMENT;
    if (strcmp(name, "scratch_bytes") == 0) {
        uint64_t sum = ctx->lane_tables.cap;
        for (const DevBuf *b :
             {&ctx->blk_first, &ctx->slot_first, &ctx->blk_size,
              &ctx->blk_off, &ctx->slots, &ctx->plan_part, &ctx->st_in,
              &ctx->st_out, &ctx->st_desc, &ctx->st_prof, &ctx->ticket,
              &ctx->order, &ctx->fr_tables, &ctx->fr_desc, &ctx->fr_meta,
              &ctx->fr_scan, &ctx->fr_slots, &ctx->fr_chunk_off,
              &ctx->tokens, &ctx->tok_pages, &ctx->tok_stage, &ctx->ntok,
              &ctx->sched,
              &ctx->lane_epochs, &ctx->sd_tables, &ctx->sd_desc,
              &ctx->bl_modes, &ctx->bl_list, &ctx->bl_descs, &ctx->bl_order})
            sum += b->cap;
        *value = (int64_t)sum;
    } else if (strcmp(name, "token_scratch_bytes") == 0) {
        *value = (int64_t)(ctx->tokens.cap + ctx->tok_pages.cap +
                           ctx->tok_stage.cap + ctx->ntok.cap);
    } else if (strcmp(name, "token_pool_pages") == 0) {
        *value = ctx->tok_pool_pages_last;
    } else if (strcmp(name, "token_pool_pct_now") == 0) {
        *value = ctx->token_pool_now;
    } else if (strcmp(name, "token_pages_asked") == 0 ||
               strcmp(name, "token_blocks_spilled") == 0) {
        // of the last token-path launch: wait for it
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const volatile uint32_t *t = ctx->h_tokstat;
        *value = !t ? 0 : name[6] == 'p' ? t[0] : t[1];
    } else {
        ctx->last_error = std::string("unknown info: ") + name;
        return SNAPMI_E_ARGUMENT;
    }
    return SNAPMI_OK;
}

} // extern "C"

namespace snapmi {

// batches of up to this many streams / blocks are planned and scanned by one
// workgroup (one launch instead of three)
constexpr size_t kPlanOneWg = 16384;

// ---------------------------------------------------------------------
// The lane kernel's hash tables: one 256 KiB table of 16-byte entries per lane
// in flight, allocated when a launch first needs more lanes than the context
// has tables for.
//
// WHERE the tables lie decides 10-25 % of the match finder's duration: HBM
// sustains 2.0e10 dependent random read + write pairs per second on tables
// packed into the memory a fresh process is handed first, and 2.6e10 on
// tables that lie in the last third of the device's memory or are spread
// over enough of it (tests/hw/zone_map.hip, addr_bits.hip, vmm_layouts.hip;
// profiles/r6_table_placement.txt).  The placement cannot be requested, but
// it can be measured - k_probe_tables is the kernel's own access pattern -
// so at most lane_table_tries candidate regions are allocated and timed:
//   0. the tables SPREAD over as much memory as the budget allows (up to a
//      MiB per 256 KiB table);
//   1. the tables PACKED, allocated while candidate 0 is still held (so it
//      lies behind it);
//   2+ spread again, behind what is held.
// The search stops at the first candidate that probes at the fast rate; the
// best one is kept, the others are freed.  At NO moment does the context
// hold more than lane_table_budget_pct of the memory that was free when the
// placement began (tests/test_gpu_parity.py polls hipMemGetInfo from a second
// thread meanwhile).
//
// top_of_memory (snapmi_ctx_prepare with SNAPMI_PREPARE_TOP_OF_MEMORY, never
// taken by a compress call on its own): ONE packed candidate allocated while
// a filler holds everything else that is free, which is given back at once -
// the tables then lie at the far end of the device's memory, the fast part.
// For the duration of two hipMalloc calls the process holds the whole device
// (another allocation on it fails meanwhile), and the driver wipes what the
// filler gives back in the background (seconds for 250 GB, during which large
// allocations wait: round 5 did this once per candidate inside a compress
// call, ten times over - 72 s for a context's first 4 GiB batch,
// profiles/r6_sweep_repro_head.txt).  That is why it is a call of its own.
// ---------------------------------------------------------------------
// (snapmi_pool.hpp: equal launches of at most lane_segment_blocks blocks)
static uint64_t segment_blocks(const snapmi_ctx *ctx, uint64_t blocks)
{
    return snapmi::segment_blocks(blocks, ctx->lane_segment_blocks);
}
static_assert(snapmi::kPoolTokPage == kTokPage &&
                  snapmi::kPoolExcPage == kExcPage &&
                  snapmi::kPoolPagesPerBlock ==
                      kTokPagesPerBlock + kExcPagesPerBlock,
              "snapmi_pool.hpp and snapmi_kernels.hpp disagree");

static uint32_t lane_count(const snapmi_ctx *ctx, uint64_t seg_blocks,
                           bool both_cores)
{
    // waves of the lane-per-block match finder: a few per CU saturate the
    // random-access rate of HBM; never more lanes than blocks
    uint64_t waves = (uint64_t)ctx->num_cus * ctx->lane_waves_per_cu;
    const uint64_t need = (seg_blocks + 63) / 64;
    if (waves > need)
        waves = need ? need : 1;
    if (ctx->lane_max_waves && waves > ctx->lane_max_waves)
        waves = ctx->lane_max_waves;
    // k_match_both: its lane wavefronts on every CU, beside two of the
    // window kernel
    if (both_cores)
        waves = (uint64_t)ctx->num_cus * kBothLaneWaves;
    return (uint32_t)waves * 64;
}

static int place_lane_tables(snapmi_ctx *ctx, uint32_t lanes,
                             bool top_of_memory)
{
    int rc;
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t tbytes = (size_t)kMaxTable * 16;
    if (ctx->lane_tables.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        free_lane_tables(ctx);
    }
    if ((rc = reserve(ctx, ctx->lane_epochs, (size_t)lanes * sizeof(uint32_t))))
        return rc;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    const size_t free_before = free_b;
    // The GPU may be shared: what this context holds while it chooses, and
    // afterwards, stays within lane_table_budget_pct of what is free now (a
    // third by default).
    const size_t budget = free_b / 100 * ctx->lane_table_budget_pct;
    // (a handful of tables has no measurable placement: only a launch that
    // fills the chip is spread or probed)
    const bool full = lanes >= 16384;
    size_t spread = tbytes;
    if (ctx->lane_table_spread && full) {
        spread = budget / lanes / 4096 * 4096;
        if (spread > 4 * tbytes)
            spread = 4 * tbytes;
        if (spread < tbytes)
            spread = tbytes;
    }
    if (ctx->lane_table_stride_kib) // test option
        spread = (size_t)ctx->lane_table_stride_kib << 10;
    uint32_t tries = full && ctx->lane_table_tries ? ctx->lane_table_tries : 1;
    if (spread == tbytes && tries > 1 && !ctx->lane_table_stride_kib)
        tries = 1; // (the budget holds packed tables only: one region)
    if (top_of_memory)
        tries = 1;
    // what the probe takes at the fast rate: 768 dependent read + write
    // pairs per lane at 2.6e10 pairs/s (tests/hw/random_rw16.hip), 3 % on top
    const float fast_ms = (float)((double)lanes * 768 / 2.6e10 * 1e3 * 1.03);
    struct Cand {
        void *p = nullptr;
        size_t stride = 0, bytes = 0;
        float ms = 0;
    };
    // (an error on the way out frees what was allocated here)
    struct Held {
        Cand best, other;
        ~Held()
        {
            for (void *p : {best.p, other.p})
                if (p)
                    (void)hipFree(p);
        }
    } held;
    ctx->probe_log.clear();
    size_t held_peak = 0;
    auto alloc = [&](Cand &c) {
        const bool ok =
            (ctx->lane_tables_uncached
                 ? hipExtMallocWithFlags(&c.p, c.bytes, hipDeviceMallocUncached)
                 : hipMalloc(&c.p, c.bytes)) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            c.p = nullptr;
        }
        return ok;
    };
    // ---- candidate 0 of a chip-filling launch: CHUNKS.  Spreading pays for
    // what lies between the tables with memory; here that memory is given
    // back.  Physical chunks (hipMemCreate) are created one after the other,
    // `pitch` times as many as the tables need; every pitch-th is mapped into
    // one address range and the others are released at once: the tables lie
    // packed in their range and spread over the device's memory, and the
    // context HOLDS what the tables fill (17 GB for 65 536 lanes) - within
    // the budget all the while (pitch x the tables for a moment).  Measured
    // equal to a MiB per table over memory that stays held
    // (tests/hw/vmm_layouts.hip, vmm_spread.hip: chunks of a GiB at every
    // fourth, of 256 MiB at every fourth, against 64 GiB held).  Any call of
    // the virtual-memory API that fails sends the placement to the plain
    // hipMalloc candidates below.
    if (full && ctx->lane_table_spread && !top_of_memory &&
        !ctx->lane_table_stride_kib && !ctx->lane_tables_uncached) {
        const size_t bytes = (size_t)lanes * tbytes;
        const size_t CH = bytes >= ((size_t)8 << 30) ? (size_t)1 << 30
                                                     : (size_t)256 << 20;
        const size_t need = (bytes + CH - 1) / CH;
        size_t pitch = budget / (need * CH);
        if (pitch > 4)
            pitch = 4;
        if (pitch >= 2) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = ctx->device;
            std::vector<hipMemGenericAllocationHandle_t> all;
            all.reserve(need * pitch);
            bool ok = true;
            for (size_t i = 0; ok && i < need * pitch; i++) {
                hipMemGenericAllocationHandle_t h;
                ok = hipMemCreate(&h, CH, &prop, 0) == hipSuccess;
                if (ok)
                    all.push_back(h);
            }
            void *va = nullptr;
            size_t mapped = 0;
            if (ok)
                ok = hipMemAddressReserve(&va, need * CH, 0, nullptr, 0) ==
                     hipSuccess;
            if (ok) {
                // (the last of every group of `pitch`: the farthest in)
                for (size_t i = 0; ok && i < need; i++) {
                    ok = hipMemMap((char *)va + i * CH, CH, 0,
                                   all[i * pitch + pitch - 1], 0) == hipSuccess;
                    if (ok)
                        mapped = i + 1;
                }
            }
            if (ok) {
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                ok = hipMemSetAccess(va, need * CH, &acc, 1) == hipSuccess;
            }
            held_peak = all.size() * CH;
            // give back what lies between - on failure everything: the
            // mappings first, every chunk once, the address range
            if (!ok) {
                (void)hipGetLastError();
                if (mapped)
                    (void)hipMemUnmap(va, mapped * CH);
            }
            std::vector<hipMemGenericAllocationHandle_t> kept;
            for (size_t i = 0; i < all.size(); i++) {
                if (ok && i % pitch == pitch - 1)
                    kept.push_back(all[i]);
                else
                    (void)hipMemRelease(all[i]);
            }
            if (!ok) {
                if (va)
                    (void)hipMemAddressFree(va, need * CH);
                held_peak = 0;
                ctx->probe_log += "chunks: the virtual-memory calls failed ";
            } else {
                ctx->lane_tables.p = va;
                ctx->lane_tables.cap = need * CH;
                ctx->lane_chunks = kept;
                ctx->lane_chunk_bytes = CH;
                ctx->lane_va_bytes = need * CH;
                ctx->lane_stride = tbytes / 16;
                ctx->lane_chunk_count = (uint32_t)need;
                ctx->lane_per_chunk = (uint32_t)(CH / tbytes);
                float ms = 0;
                const uint32_t per_chunk = (uint32_t)(CH / tbytes);
                hipLaunchKernelGGL(k_probe_tables, dim3(lanes / 64), dim3(64),
                                   0, ctx->stream, (unsigned long long *)va,
                                   (unsigned long long)(tbytes / 16), 64u,
                                   (uint32_t)need, per_chunk);
                HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
                hipLaunchKernelGGL(k_probe_tables, dim3(lanes / 64), dim3(64),
                                   0, ctx->stream, (unsigned long long *)va,
                                   (unsigned long long)(tbytes / 16), 768u,
                                   (uint32_t)need, per_chunk);
                HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
                HIP_TRY(ctx, hipEventSynchronize(ctx->ev[5]));
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
                // tables start as "never used": epoch 0 in every entry
                // (every chunk in full: the lanes' tables are dealt out
                // over all of them)
                hipLaunchKernelGGL(k_zero16, dim3(ctx->num_cus * 8), dim3(256),
                                   0, ctx->stream, (unsigned long long *)va,
                                   (unsigned long long)(need * CH / 16));
                HIP_TRY(ctx, hipMemsetAsync(ctx->lane_epochs.p, 0,
                                            (size_t)lanes * 4, ctx->stream));
                ctx->n_lanes = lanes;
                size_t free_after = 0;
                (void)hipMemGetInfo(&free_after, &total_b);
                const double t_ms =
                    std::chrono::duration<double, std::milli>(
                        std::chrono::steady_clock::now() - t_begin).count();
                char buf[320];
                snprintf(buf, sizeof buf,
                         "%.2f(chunks: %zu of %zu x %zu MiB) | held at most "
                         "%zu of budget %zu | kept %zu KiB apart, %u lanes, "
                         "%zu bytes | placement %.1f ms | free %zu -> %zu",
                         ms, need, need * pitch, CH >> 20, held_peak, budget,
                         tbytes >> 10, lanes, ctx->lane_tables.cap, t_ms,
                         free_before, free_after);
                ctx->probe_log += buf;
                return SNAPMI_OK;
            }
        }
    }
    for (uint32_t t = 0; t < tries; t++) {
        Cand c;
        c.stride = top_of_memory || (t & 1) ? tbytes : spread;
        c.bytes = (size_t)lanes * c.stride;
        // a loser is freed before the next candidate comes unless the budget
        // has room for all three (then the new one cannot be the loser's
        // memory again)
        const size_t alive = held.best.bytes + held.other.bytes;
        if (held.other.p && alive + c.bytes > budget) {
            HIP_TRY(ctx, hipFree(held.other.p));
            held.other = Cand();
        }
        if (t && held.best.bytes + held.other.bytes + c.bytes > budget)
            break; // no room for another candidate within the budget
        void *filler = nullptr;
        if (top_of_memory) {
            // everything that is free but the region itself and a GiB
            // beside it (with less left free the region is pieced together
            // from what is free elsewhere, profiles/r5_table_budget.txt)
            const size_t spare = c.bytes + ((size_t)1 << 30);
            size_t want = free_b > spare ? free_b - spare : 0;
            for (int k = 0; k < 3 && want >= ((size_t)8 << 30); k++) {
                if (hipMalloc(&filler, want) == hipSuccess)
                    break;
                (void)hipGetLastError();
                filler = nullptr;
                want = want / 16 * 15;
            }
        }
        bool got = alloc(c);
        if (filler) {
            (void)hipFree(filler);
            if (!got) // (not behind the filler: the plain way)
                got = alloc(c);
        }
        if (!got)
            break; // keep the best so far
        {
            const size_t now = held.best.bytes + held.other.bytes + c.bytes;
            held_peak = now > held_peak ? now : held_peak;
        }
        // (the probe runs on the memory as it comes: only the region that is
        // kept gets zeroed)
        if (tries > 1 || ctx->lane_table_probe || top_of_memory) {
            hipLaunchKernelGGL(k_probe_tables, dim3(lanes / 64), dim3(64), 0,
                               ctx->stream, (unsigned long long *)c.p,
                               (unsigned long long)(c.stride / 16), 64u, 0u,
                               0u);
            HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
            hipLaunchKernelGGL(k_probe_tables, dim3(lanes / 64), dim3(64), 0,
                               ctx->stream, (unsigned long long *)c.p,
                               (unsigned long long)(c.stride / 16), 768u, 0u,
                               0u);
            HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev[5]));
            HIP_TRY(ctx, hipEventElapsedTime(&c.ms, ctx->ev[4], ctx->ev[5]));
            char buf[48];
            snprintf(buf, sizeof buf, "%s%.2f(%zuK)", t ? " " : "", c.ms,
                     c.stride >> 10);
            ctx->probe_log += buf;
        }
        if (!held.best.p || c.ms < held.best.ms) {
            if (held.other.p) {
                HIP_TRY(ctx, hipFree(held.other.p));
                held.other = Cand();
            }
            held.other = held.best;
            held.best = c;
        } else {
            if (held.other.p)
                HIP_TRY(ctx, hipFree(held.other.p));
            held.other = c;
        }
        if (tries > 1 && held.best.ms <= fast_ms)
            break;
    }
    if (held.other.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(held.other.p));
        held.other = Cand();
    }
    if (!held.best.p)
        return fail_ctx(ctx, SNAPMI_E_DEVICE,
                        "hipMalloc of %zu bytes of lane tables failed",
                        (size_t)lanes * tbytes);
    // tables start as "never used": epoch 0 in every entry
    HIP_TRY(ctx, hipMemset2DAsync(held.best.p, held.best.stride, 0, tbytes,
                                  lanes, ctx->stream));
    ctx->lane_tables.p = held.best.p;
    ctx->lane_tables.cap = held.best.bytes;
    ctx->lane_chunk_count = 0;
    ctx->lane_per_chunk = 0;
    ctx->lane_stride = held.best.stride / 16;
    held.best = Cand(); // the context owns it now
    HIP_TRY(ctx, hipMemsetAsync(ctx->lane_epochs.p, 0, (size_t)lanes * 4,
                                ctx->stream));
    ctx->n_lanes = lanes;
    {
        size_t free_after = 0;
        (void)hipMemGetInfo(&free_after, &total_b);
        const double ms =
            std::chrono::duration<double, std::milli>(
                std::chrono::steady_clock::now() - t_begin).count();
        char buf[256];
        snprintf(buf, sizeof buf,
                 " | held at most %zu of budget %zu | kept %zu KiB apart, "
                 "%u lanes, %zu bytes%s | placement %.1f ms | free %zu -> %zu",
                 held_peak, budget, (size_t)(ctx->lane_stride * 16) >> 10,
                 lanes, ctx->lane_tables.cap,
                 top_of_memory ? " (top of memory)" : "", ms, free_before,
                 free_after);
        ctx->probe_log += buf;
    }
    return SNAPMI_OK;
}


// snapmi_ctx_prepare: the tables a batch of `blocks` blocks would make the
// first compress call allocate, now (the same lane count launch_compress
// derives); nothing when such a batch does not run the lane kernel or the
// context already has that many tables - unless the far end of the memory is
// asked for and the tables are not there yet.
int prepare_lane_tables(snapmi_ctx *ctx, uint64_t blocks, bool top)
{
    if (ctx->compress_mode == 0 || blocks < ctx->lane_min_blocks)
        return SNAPMI_OK;
    const uint64_t seg_blocks = segment_blocks(ctx, blocks);
    const bool both = ctx->compress_mode == 1 && ctx->lds_order_ok &&
                      ctx->lane_coresident &&
                      blocks >= ctx->lane_coresident_min_blocks;
    const uint32_t lanes = lane_count(ctx, seg_blocks, both);
    if (lanes <= ctx->n_lanes && !(top && !ctx->lane_tables_top))
        return SNAPMI_OK;
    const int rc = place_lane_tables(ctx, lanes > ctx->n_lanes ? lanes
                                                               : ctx->n_lanes,
                                     top);
    if (rc == SNAPMI_OK)
        ctx->lane_tables_top = top;
    return rc;
}


// k_scan_sizes over the blocks [a.blk_lo, min(a.blk_hi, host_blocks))
static void launch_scan_sizes(const CompressArgs &a, hipStream_t s)
{
    const uint32_t hi = a.blk_hi < a.host_blocks ? a.blk_hi : a.host_blocks;
    const uint32_t cnt = hi > a.blk_lo ? hi - a.blk_lo : 0;
    if (cnt > kPlanOneWg) {
        const uint32_t parts = (cnt + 1023) / 1024;
        hipLaunchKernelGGL(k_scan_sizes_a, dim3(parts), dim3(1024), 0, s, a,
                           parts);
        hipLaunchKernelGGL(k_scan_sizes_b, dim3(1), dim3(1024), 0, s, a,
                           parts);
        hipLaunchKernelGGL(k_scan_sizes_c, dim3(parts), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, s, a);
    }
}

int launch_compress(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                    const uint64_t *d_in_lens, void *const *d_out_ptrs,
                    const uint64_t *d_out_caps, uint64_t *d_out_lens,
                    snapmi_error *d_errs, size_t n, uint64_t blocks,
                    uint64_t slots, uint32_t small_classes, uint64_t cnt8,
                    uint64_t block_bytes)
{
    if (blocks > 0x7FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                        "compress_batch: %llu blocks in one batch",
                        (unsigned long long)blocks);
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    int rc;
    if ((rc = reserve(ctx, ctx->blk_first, (n + 1) * sizeof(uint32_t))) ||
        (rc = reserve(ctx, ctx->slot_first, (n + 1) * sizeof(uint32_t))) ||
        (rc = reserve(ctx, ctx->blk_size, (blocks + 1) * sizeof(uint32_t))) ||
        (rc = reserve(ctx, ctx->blk_off, (blocks + 2) * sizeof(uint64_t))) ||
        (rc = reserve(ctx, ctx->plan_part,
                      ((n > blocks ? n : blocks) / 1024 + 4) * 16)) ||
        (rc = reserve(ctx, ctx->ticket, 64)))
        return rc;

    CompressArgs a;
    a.in_ptrs = d_in_ptrs;
    a.in_lens = d_in_lens;
    a.out_ptrs = d_out_ptrs;
    a.out_caps = d_out_caps;
    a.out_lens = d_out_lens;
    a.errs = d_errs;
    a.blk_first = (uint32_t *)ctx->blk_first.p;
    a.slot_first = (uint32_t *)ctx->slot_first.p;
    a.blk_size = (uint32_t *)ctx->blk_size.p;
    a.blk_off = (uint64_t *)ctx->blk_off.p;
    // (the two scans never run at the same time: one buffer for both)
    a.plan_part = (uint2 *)ctx->plan_part.p;
    a.scan_part = (unsigned long long *)ctx->plan_part.p;
    a.n_streams = (uint32_t)n;
    a.host_blocks = (uint32_t)blocks;
    a.host_slots = (uint32_t)slots;
    a.ticket = (uint32_t *)ctx->ticket.p;
    a.tok_pool = nullptr;
    a.tok_pages = nullptr;
    a.tok_ctl = nullptr;
    a.tok_pool_pages = 0;
    a.tok_stage = nullptr;
    a.tok_stage_waves = 0;
    a.tok_stage_wave0 = 0;
    a.sched = nullptr;
    a.ntok = nullptr;
    a.lane_tables = nullptr;
    a.lane_epochs = nullptr;
    a.lane_stride = kMaxTable;
    a.lane_chunks = 0;
    a.lane_per_chunk = 0;
    a.n_lanes = 0;
    a.tok_base = 0;
    a.small_limit = (uint32_t)small_stream_limit(ctx);
    a.cls_lo = 0;
    a.cls_hi = kMaxBlock;
    // Small batches are latency-bound: the wavefront kernel finishes a block
    // in ~2 ms, a lane needs tens of ms.  Large batches are throughput-bound
    // and go to the lane-per-block kernel.
    // Blocks of at most 8 KiB (cnt8 of them: the caller counted one-block
    // streams and tails): a window kernel with the table the reference gives
    // such blocks, ten per CU instead of five (k_match_spans_8k), whatever
    // the size of the batch - every probe of the lane kernel into a block's
    // fresh table is an HBM transaction, and the 64 KiB window kernel keeps a
    // CU's issue slots four fifths idle.  (Twenty tables of 8 KiB per CU for
    // blocks of at most 4 KiB were built too and measured the same: at ten
    // wavefronts the CU's one scalar unit is 70 % busy.)
    const uint64_t nb_small = cnt8 < blocks ? cnt8 : blocks;
    const bool use_small = blocks > 0 && ctx->lds_order_ok &&
                           ctx->compress_mode == 1 &&
                           ctx->small_table_kernel &&
                           nb_small >= ctx->small_table_min_blocks;
    const uint64_t nb_big = use_small ? blocks - nb_small : blocks;
    const bool big = nb_big >= ctx->lane_min_blocks;
    // (a device that failed the LDS order self-check only has the lane kernel)
    // The token path with the WINDOW kernel as its match finder
    // (k_match_spans): every block at its final position, no slots, no
    // k_compact, the encoder a wide kernel of its own.  What the small-block
    // kernel runs on, and - option window_tokens - a mid-size batch (more
    // than two blocks per CU, fewer than lane_min_blocks).
    const bool win_tok =
        blocks > 0 && ctx->lds_order_ok && ctx->compress_mode == 1 && !big &&
        (use_small ||
         (ctx->window_tokens &&
          !(ctx->small_batch_kernel == 2 ||
            (ctx->small_batch_kernel == 1 &&
             blocks <= 2 * (uint64_t)ctx->num_cus))));
    const bool lanes_mode =
        blocks > 0 &&
        (!ctx->lds_order_ok || (ctx->compress_mode != 0 && big) || win_tok);
    // (segment_blocks: equal launches that bound the token scratch; "both at
    // once" needs the whole list in one segment)
    const uint64_t seg_blocks = segment_blocks(ctx, blocks);
    const bool waves_mode =
        blocks > 0 && ctx->lds_order_ok &&
        (!lanes_mode || ctx->compress_mode == 0 ||
         (ctx->compress_mode == 2 && seg_blocks == blocks));
    // The lane kernel knows every block's encoded size before a byte of it
    // is written, so its encoder puts the blocks where they belong; only the
    // wavefront kernel (which encodes while it matches) needs a scratch slot
    // per block and the k_compact pass.
    const bool direct = lanes_mode && !waves_mode && ctx->lane_direct_encode &&
                        ctx->lane_overlap_encode == 0;
    a.direct = direct ? 1 : 0;
    if (!direct &&
        (rc = reserve(ctx, ctx->slots, (slots + 1) * (size_t)kSlotBytes)))
        return rc;
    a.scratch = (uint8_t *)ctx->slots.p;
    a.blk_lo = 0;
    a.blk_hi = (uint32_t)blocks;
    // The token path's match finder: the lane kernel, or - option
    // match_kernel - the window kernel (k_match_spans: table in LDS, no
    // tables in HBM).  By default the context's last batch decides: data that
    // does not compress costs a lane three HBM transactions per probe for
    // nothing (cfg5: 12 ms of lane kernel for 32 GiB against ~4 of windows).
    bool span_match = false;
    if (lanes_mode && !waves_mode && ctx->lds_order_ok) {
        if (ctx->match_kernel == 1 || win_tok) {
            span_match = true;
        } else if (ctx->match_kernel == 2 && ctx->h_ratio) {
            // the slot of the latest batch that has finished (a slot reads
            // 0 in word 4 while a kernel is writing it)
            const volatile uint32_t *r = ctx->h_ratio;
            const uint32_t s0 = r[4], s1 = r[12];
            const volatile uint32_t *slot = s1 > s0 ? r + 8 : r;
            const uint32_t seq = slot[4];
            const uint64_t c = ((uint64_t)slot[1] << 32) | slot[0];
            const uint64_t u = ((uint64_t)slot[3] << 32) | slot[2];
            span_match = seq && slot[4] == seq && u &&
                         c * 100 >= u * ctx->match_spans_ratio_pct;
        }
    }
    // both match finders on every CU (k_match_both): launches that fill the
    // chip with lanes anyway
    const bool both_cores = lanes_mode && !waves_mode && !span_match &&
                            ctx->lds_order_ok && ctx->lane_coresident &&
                            nb_big >= ctx->lane_coresident_min_blocks;
    // The token pool (CompressArgs::tok_pool): pages of 2 KiB for the tokens
    // of a launch's blocks - token_pool_pct per cent of what the worst case
    // of every block would take, and what the launch keeps in hand on top;
    // never fewer than 32 768 pages (64 MiB: a small batch does not spill);
    // grown for the batch behind one of which more than a hundredth spilled
    // - by half, or by a sixth when it was less than a tenth (k_redo_spilled
    // posts the counts; read without waiting, like the ratio above).
    if (ctx->h_tokstat) {
        const volatile uint32_t *t = ctx->h_tokstat;
        const uint32_t seq = t[3];
        if (seq != ctx->tokstat_seen) {
            ctx->tokstat_seen = seq;
            const uint32_t asked = t[0], spilled = t[1], of = t[2];
            if (t[3] == seq && of) {
                ctx->tok_pages_asked = asked;
                ctx->tok_blocks_spilled = spilled;
                ctx->token_pool_now =
                    snapmi::pool_grow(ctx->token_pool_now, spilled, of);
            }
        }
    }
    if (ctx->token_pool_now < ctx->token_pool_pct)
        ctx->token_pool_now = ctx->token_pool_pct;
    // (snapmi_pool.hpp: the share of the worst case, what the launch keeps in
    // hand, the floor, and "100 means never")
    const uint32_t pool_lanes =
        lanes_mode && !span_match ? lane_count(ctx, seg_blocks, both_cores)
                                  : 0;
    const uint64_t pool_pages =
        snapmi::pool_pages(block_bytes, blocks, seg_blocks, pool_lanes,
                           ctx->token_pool_now, ctx->token_pool_min_pages);
    // (+ the dump page of the lanes)
    const size_t pool_bytes = (size_t)(pool_pages + 1) * kTokPage * 4;
    // the window wavefronts' staging arrays (TokenWriter): one per wavefront
    // of the largest workgroup this call launches
    const uint32_t stage_waves =
        use_small ? kSmallTableWaves
                  : both_cores ? kBothWaves - kBothLaneWaves
                               : span_match ? kCompressWaves : 0;
    const size_t stage_bytes =
        (size_t)ctx->num_cus * stage_waves * kTokStageWords * 4;
    a.tok_stage_waves = stage_waves;
    a.tok_stage_wave0 = 0;
    // ... and behind its control words and the list of the spilled blocks,
    // the blocks' page tables
    const size_t tab_off =
        ((size_t)(kTokCtlList + seg_blocks) * 4 + 255) & ~(size_t)255;
    const size_t tab_bytes =
        tab_off + (size_t)seg_blocks * kPageTabStride * 4;
    if (lanes_mode && span_match) {
        if ((rc = reserve(ctx, ctx->tokens, pool_bytes, /*slack=*/false)) ||
            (rc = reserve(ctx, ctx->tok_pages, tab_bytes)) ||
            (rc = reserve(ctx, ctx->tok_stage, stage_bytes + 256)) ||
            (rc = reserve(ctx, ctx->ntok, (size_t)blocks * sizeof(uint32_t))))
            return rc;
        a.tok_pool = (uint32_t *)ctx->tokens.p;
        a.tok_ctl = (uint32_t *)ctx->tok_pages.p;
        a.tok_pages = (uint32_t *)((uint8_t *)ctx->tok_pages.p + tab_off);
        a.tok_pool_pages = (uint32_t)pool_pages;
        a.tok_stage = (uint32_t *)ctx->tok_stage.p;
        ctx->tok_pool_pages_last = (uint32_t)pool_pages;
        a.ntok = (uint32_t *)ctx->ntok.p;
    } else if (lanes_mode) {
        const uint32_t lanes = lane_count(ctx, seg_blocks, both_cores);
        if (lanes > ctx->n_lanes) {
            if ((rc = place_lane_tables(ctx, lanes, /*top_of_memory=*/false)))
                return rc;
            ctx->lane_tables_top = false;
        }
        if ((rc = reserve(ctx, ctx->tokens, pool_bytes, /*slack=*/false)) ||
            (rc = reserve(ctx, ctx->tok_pages, tab_bytes)) ||
            (rc = reserve(ctx, ctx->tok_stage, stage_bytes + 256)) ||
            (rc = reserve(ctx, ctx->ntok, (size_t)blocks * sizeof(uint32_t))))
            return rc;
        if (ctx->lane_epoch_preset >= 0) { // test knob, see snapmi_ctx.hpp
            HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->lane_epochs.p,
                                           (int)ctx->lane_epoch_preset,
                                           ctx->n_lanes, ctx->stream));
            ctx->lane_epoch_preset = -1;
        }
        a.tok_pool = (uint32_t *)ctx->tokens.p;
        a.tok_ctl = (uint32_t *)ctx->tok_pages.p;
        a.tok_pages = (uint32_t *)((uint8_t *)ctx->tok_pages.p + tab_off);
        a.tok_pool_pages = (uint32_t)pool_pages;
        a.tok_stage = (uint32_t *)ctx->tok_stage.p;
        ctx->tok_pool_pages_last = (uint32_t)pool_pages;
        a.ntok = (uint32_t *)ctx->ntok.p;
        a.lane_tables = (unsigned long long *)ctx->lane_tables.p;
        a.lane_epochs = (uint32_t *)ctx->lane_epochs.p;
        a.lane_stride = ctx->lane_stride;
        a.lane_chunks = ctx->lane_chunk_count;
        a.lane_per_chunk = ctx->lane_per_chunk;
        a.n_lanes = lanes;
    }
    a.prof = nullptr;
    PROF(
    if ((rc = reserve(ctx, ctx->st_prof, 16 * sizeof(uint64_t))))
        return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->st_prof.p, 0, 16 * sizeof(uint64_t),
                                ctx->stream));
    a.prof = (unsigned long long *)ctx->st_prof.p;
    )

    hipStream_t s = ctx->stream;
    ctx->timing_valid = false;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], s));
    if (n > kPlanOneWg) {
        const uint32_t parts = (uint32_t)((n + 1023) / 1024);
        hipLaunchKernelGGL(k_plan_compress_a, dim3(parts), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(k_plan_compress_b, dim3(1), dim3(1024), 0, s, a,
                           parts);
        hipLaunchKernelGGL(k_plan_compress_c, dim3(parts), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_plan_compress, dim3(1), dim3(1024), 0, s, a);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], s));
    // the lane-per-stream kernels: streams under 256 bytes one per lane,
    // streams under 1 KiB a few per wavefront (a wavefront of streams that
    // are all larger returns at once: 16 384 idle wavefronts for a million
    // 64 KiB chunks).  small_classes: what the caller knows of the lengths -
    // a class without a stream is not launched.
    {
        const dim3 grid((uint32_t)((n + 63) / 64));
        if (a.small_limit && (small_classes & 1))
            hipLaunchKernelGGL(k_compress_tiny, grid, dim3(64), 0, s, a);
        if (a.small_limit > kTinyCompress) {
            if (small_classes & 2)
                hipLaunchKernelGGL(k_compress_small512, grid, dim3(64), 0, s,
                                   a);
            if (small_classes & 4)
                hipLaunchKernelGGL(k_compress_small1k, grid, dim3(64), 0, s,
                                   a);
            if (small_classes & 8)
                hipLaunchKernelGGL(k_compress_small2k, grid, dim3(64), 0, s,
                                   a);
        }
    }
    if (blocks) {
        hipStream_t ws = s; // stream of the wavefront kernel
        if (waves_mode) {
            HIP_TRY(ctx, hipMemsetAsync(ctx->ticket.p, 0, 64, s));
            if (lanes_mode) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
                HIP_TRY(ctx,
                        hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                ws = ctx->stream2;
            }
            // persistent: one 5-wave workgroup per CU (all of its LDS), each
            // wavefront pulls blocks from the back of the ticket
            // the smallest batches (no more than two blocks per CU: scalar
            // calls, short frames) run one block per CU with the input block
            // in LDS too - half the time per block, a fifth of the blocks in
            // flight; five tables per CU from there
            const bool lds_input =
                ctx->small_batch_kernel == 2 ||
                (ctx->small_batch_kernel == 1 &&
                 blocks <= 2 * (uint64_t)ctx->num_cus);
            if (lds_input) {
                const uint32_t wgs = (uint32_t)(
                    blocks < (uint64_t)ctx->num_cus ? blocks : ctx->num_cus);
#ifdef SNAPMI_TESTING
                hipLaunchKernelGGL(ctx->span_kernel ? k_compress_span_lds
                                                    : k_compress_block_lds,
                                   dim3(wgs), dim3(64), 0, ws, a);
#else
                hipLaunchKernelGGL(k_compress_span_lds, dim3(wgs), dim3(64),
                                   0, ws, a);
#endif
            } else {
                const uint64_t want =
                    (blocks + kCompressWaves - 1) / kCompressWaves;
                // (beside the lane kernel it takes a share of the CUs only:
                // a persistent workgroup owns its CU's whole LDS)
                const uint64_t cus =
                    lanes_mode ? (ctx->both_wave_cus ? ctx->both_wave_cus
                                                     : ctx->num_cus / 2)
                               : (uint64_t)ctx->num_cus;
                const uint32_t wgs = (uint32_t)(want < cus ? want : cus);
                // several blocks per wavefront, the window kernel alone:
                // the order of the blocks is chosen as the launch goes
                // (SpanSched, snapmi_compress.hip: heavy streams first,
                // light ones last - a launch ends with its small jobs)
                bool sched = !lanes_mode && ctx->span_kernel &&
                             (ctx->span_schedule == 2 ||
                              (ctx->span_schedule == 1 &&
                               blocks > (uint64_t)wgs * kCompressWaves));
                if (sched) {
                    const size_t head = (size_t)(16 + n) * 4;
                    const size_t lists = (size_t)2 * (slots + 1) * 4;
                    if ((rc = reserve(ctx, ctx->sched, head + lists)))
                        return rc;
                    HIP_TRY(ctx, hipMemsetAsync(ctx->sched.p, 0, head, ws));
                    HIP_TRY(ctx, hipMemsetAsync((uint8_t *)ctx->sched.p + head,
                                                0xFF, lists, ws));
                    a.sched = (uint32_t *)ctx->sched.p;
                }
#ifdef SNAPMI_TESTING
                hipLaunchKernelGGL(ctx->span_kernel ? k_compress_spans
                                                    : k_compress_blocks,
                                   dim3(wgs), dim3(kCompressWaves * 64), 0, ws,
                                   a);
#else
                hipLaunchKernelGGL(k_compress_spans, dim3(wgs),
                                   dim3(kCompressWaves * 64), 0, ws, a);
#endif
                a.sched = nullptr;
            }
        }
        if (lanes_mode) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev[4], s));
            for (uint64_t lo = 0; lo < blocks; lo += seg_blocks) {
                const uint64_t hi =
                    lo + seg_blocks < blocks ? lo + seg_blocks : blocks;
                // Option lane_overlap_encode (off by default): the segment is
                // matched in two halves and the first half's tokens are
                // encoded on the side stream while the second half is
                // matched.  Two halves alone cost the match finder nothing
                // (114.2 vs 115 ms), but the encoder's streaming traffic under
                // it does: 121.6 -> 135 ms at cfg2.  Kept as a measured dead
                // end that the encoder tests still run through.
                uint64_t mid = hi;
                if (!waves_mode && !use_small &&
                    (ctx->lane_overlap_encode == 2
                         ? hi - lo >= 2
                         : (ctx->lane_overlap_encode == 1 &&
                            (hi - lo) * 10 >= (uint64_t)a.n_lanes * 14)))
                    mid = lo + (hi - lo) / 2;
                a.tok_base = (uint32_t)lo;
                a.blk_lo = (uint32_t)lo;
                a.blk_hi = (uint32_t)mid;
                // the pool is the segment's: no page handed out, no block
                // spilled, k_redo_spilled's ticket at 0
                HIP_TRY(ctx, hipMemsetAsync(ctx->tok_pages.p, 0,
                                            kTokCtlList * 4, s));
                if (!waves_mode) // (shared with the wavefront kernel if on)
                    HIP_TRY(ctx, hipMemsetAsync(ctx->ticket.p, 0, 64, s));
                // A launch of few blocks waits for the latency of its
                // rounds with the memory system idle: the kernel that also
                // fetches the next probe's entry (k_match_blocks_spec) takes
                // 10-15 % off 2 048 .. 16 384 blocks of text.  From 32 768
                // blocks on the launch is at the random-access rate of HBM
                // even with one block per lane (1.6e10 rounds a second, as at
                // 146 700 blocks) and the extra reads buy nothing
                // (profiles/r3_lane_speculation.txt).
                const bool spec = ctx->lane_speculate &&
                                  hi - lo <= a.n_lanes &&
                                  hi - lo <= ctx->lane_speculate_max_blocks;
                // (with the small-block kernels on, this launch's class is
                // the blocks of more than 8 KiB - if the batch has any)
                a.cls_lo = use_small ? 8192 : 0;
                if (use_small && nb_big == 0) {
                } else if (span_match) {
                    const uint64_t mine =
                        use_small && nb_big < mid - lo ? nb_big : mid - lo;
                    const uint64_t want =
                        (mine + kCompressWaves - 1) / kCompressWaves;
                    hipLaunchKernelGGL(
                        k_match_spans,
                        dim3((uint32_t)(want < (uint64_t)ctx->num_cus
                                            ? want : ctx->num_cus)),
                        dim3(kCompressWaves * 64), 0, s, a);
                } else if (both_cores) {
                    a.tok_stage_wave0 = kBothLaneWaves;
                    hipLaunchKernelGGL(k_match_both, dim3(ctx->num_cus),
                                       dim3(kBothWaves * 64), 0, s, a);
                    a.tok_stage_wave0 = 0;
                } else
                hipLaunchKernelGGL(spec ? k_match_blocks_spec : k_match_blocks,
                                   dim3(a.n_lanes / 64), dim3(64), 0, s, a);
                if (use_small) {
                    // one 640-thread workgroup per CU: ten 16 KiB tables
                    HIP_TRY(ctx, hipMemsetAsync(ctx->ticket.p, 0, 64, s));
                    a.cls_lo = 0;
                    a.cls_hi = 8192;
                    const uint64_t want =
                        (nb_small + kSmallTableWaves - 1) / kSmallTableWaves;
                    const uint64_t room = ctx->num_cus;
                    hipLaunchKernelGGL(
                        k_match_spans_8k,
                        dim3((uint32_t)(want < room ? want : room)),
                        dim3(kSmallTableWaves * 64), 0, s, a);
                }
                a.cls_lo = 0;
                a.cls_hi = kMaxBlock;
                if (mid < hi) {
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2,
                                                    ctx->ev_fork, 0));
                    hipLaunchKernelGGL(k_encode_tokens,
                                       dim3((uint32_t)(mid - lo)), dim3(64),
                                       0, ctx->stream2, a);
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                    a.blk_lo = (uint32_t)mid;
                    a.blk_hi = (uint32_t)hi;
                    HIP_TRY(ctx, hipMemsetAsync(ctx->ticket.p, 0, 64, s));
                    if (span_match) {
                        const uint64_t want =
                            (hi - mid + kCompressWaves - 1) / kCompressWaves;
                        hipLaunchKernelGGL(
                            k_match_spans,
                            dim3((uint32_t)(want < (uint64_t)ctx->num_cus
                                                ? want : ctx->num_cus)),
                            dim3(kCompressWaves * 64), 0, s, a);
                    } else if (both_cores) {
                        a.tok_stage_wave0 = kBothLaneWaves;
                        hipLaunchKernelGGL(k_match_both, dim3(ctx->num_cus),
                                           dim3(kBothWaves * 64), 0, s, a);
                        a.tok_stage_wave0 = 0;
                    } else
                    hipLaunchKernelGGL(
                        spec ? k_match_blocks_spec : k_match_blocks,
                        dim3(a.n_lanes / 64), dim3(64), 0, s, a);
                }
                if (hi == blocks) // dominant_ms: first match start .. last end
                    HIP_TRY(ctx, hipEventRecord(ctx->ev[5], s));
                if (waves_mode) {
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                    HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                }
                if (direct)
                    launch_scan_sizes(a, s);
                hipLaunchKernelGGL(k_encode_tokens,
                                   dim3((uint32_t)(hi - a.blk_lo)), dim3(64),
                                   0, s, a);
                if (mid < hi) // the side stream's half is done as well
                    HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                {
                    // the blocks whose tokens found no page: once more, by
                    // the window kernel, to where the encoder would have put
                    // them (CompressArgs::tok_pool)
                    if (!ctx->h_tokstat) {
                        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_tokstat,
                                                   64, hipHostMallocDefault));
                        memset((void *)ctx->h_tokstat, 0, 64);
                    }
                    CompressArgs r = a;
                    r.blk_lo = (uint32_t)lo;
                    r.blk_hi = (uint32_t)hi;
                    r.cls_lo = 0;
                    r.cls_hi = kMaxBlock;
                    const uint64_t want =
                        (hi - lo + kCompressWaves - 1) / kCompressWaves;
                    hipLaunchKernelGGL(
                        k_redo_spilled,
                        dim3((uint32_t)(want < (uint64_t)ctx->num_cus
                                            ? want : ctx->num_cus)),
                        dim3(kCompressWaves * 64), 0, s, r,
                        (uint32_t *)ctx->h_tokstat, ++ctx->tokstat_seq);
                }
            }
            a.blk_lo = 0;
            a.tok_base = 0;
            a.blk_hi = (uint32_t)blocks;
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], s));
    if (blocks && direct) {
        hipLaunchKernelGGL(k_stream_lens, dim3((uint32_t)((n + 255) / 256)),
                           dim3(256), 0, s, a);
        // what this batch compressed to, for the next batch's choice of
        // match finder (read without waiting: a hint)
        if (ctx->match_kernel == 2 && blocks >= 2 * ctx->lane_min_blocks) {
            if (!ctx->h_ratio) {
                HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_ratio, 64,
                                           hipHostMallocDefault));
                memset((void *)ctx->h_ratio, 0, 64);
            }
            hipLaunchKernelGGL(k_post_ratio, dim3(1), dim3(1024), 0, s,
                               (uint32_t *)ctx->h_ratio, a.blk_off,
                               (uint32_t)blocks, a.in_lens, a.n_streams,
                               ++ctx->ratio_seq);
        }
    } else if (blocks) {
        launch_scan_sizes(a, s);
        hipLaunchKernelGGL(k_compact, dim3((uint32_t)blocks), dim3(256), 0,
                           s, a);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], s));
    HIP_TRY(ctx, hipGetLastError());
    ctx->timing_valid = true;
    ctx->timing_is_compress = true;
    ctx->last_kernel =
        !blocks ? "k_compress_tiny"
        : !lanes_mode ? "k_compress_spans"
        : span_match ? (use_small && nb_big == 0 ? "k_match_spans_8k"
                                                 : "k_match_spans")
        : both_cores ? "k_match_both" : "k_match_blocks";
    ctx->dominant_split = lanes_mode;
    ctx->codec_launches = blocks ? 1 : 0;
    return SNAPMI_OK;
}

int launch_decompress(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                      const uint64_t *d_in_lens, void *const *d_out_ptrs,
                      const uint64_t *d_out_caps, uint64_t *d_out_lens,
                      snapmi_error *d_errs, const uint8_t *d_modes, size_t n,
                      const unsigned long long *d_gate,
                      unsigned long long gate_value, hipStream_t side,
                      DevBuf *side_order)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf &order = side_order ? *side_order : ctx->order;
    DecompressArgs a;
    a.gate = d_gate;
    a.gate_value = gate_value;
    a.in_ptrs = d_in_ptrs;
    a.in_lens = d_in_lens;
    a.out_ptrs = d_out_ptrs;
    a.out_caps = d_out_caps;
    a.out_lens = d_out_lens;
    a.errs = d_errs;
    a.modes = d_modes;
    a.n_streams = (uint32_t)n;
    {
        // (+ 64 bucket counters of the many-workgroup sort behind the order,
        // + the number of streams that are not tiny)
        int rc = reserve(ctx, order, (n + 72) * sizeof(uint32_t));
        if (rc)
            return rc;
    }
    a.order = (uint32_t *)order.p;
    a.bucket_pos = a.order + n;
    a.prof = nullptr;
    PROF(
    {
        int rc = reserve(ctx, ctx->st_prof, 16 * sizeof(uint64_t));
        if (rc)
            return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->st_prof.p, 0, 16 * sizeof(uint64_t),
                                    ctx->stream));
        a.prof = (unsigned long long *)ctx->st_prof.p;
    }
    )
    hipStream_t s = side ? side : ctx->stream;
    if (!side) {
        ctx->timing_valid = false;
        HIP_TRY(ctx, hipEventRecord(ctx->ev[0], s));
    }
    if (n > kPlanOneWg) {
        const uint32_t parts = (uint32_t)((n + 1023) / 1024);
        HIP_TRY(ctx, hipMemsetAsync(a.bucket_pos, 0, 64 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_plan_decompress_a, dim3(parts), dim3(1024), 0, s,
                           a);
        hipLaunchKernelGGL(k_plan_decompress_b, dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_plan_decompress_c, dim3(parts), dim3(1024), 0, s,
                           a);
    } else {
        hipLaunchKernelGGL(k_plan_decompress, dim3(1), dim3(1024), 0, s, a);
    }
    if (!side)
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], s));
    if (ctx->decode_kernel == 0)
        hipLaunchKernelGGL(k_decompress_sequential, dim3((uint32_t)n),
                           dim3(64), 0, s, a);
#ifdef SNAPMI_TESTING
    else if (ctx->decode_kernel == 2)
        hipLaunchKernelGGL(k_decompress_streams2, dim3((uint32_t)n), dim3(64),
                           0, s, a);
#endif
    else {
        if (n > ctx->decode_many_min)
            hipLaunchKernelGGL(
                k_decompress_streams3_many,
                dim3((uint32_t)((n + kManyStreams - 1) / kManyStreams)),
                dim3(64), 0, s, a);
        else
            hipLaunchKernelGGL(k_decompress_streams3, dim3((uint32_t)n),
                               dim3(64), 0, s, a);
        // the streams of fewer than 256 compressed bytes, one per lane (how
        // many there are only the device knows: workgroups without any leave
        // at once, in both launches)
        hipLaunchKernelGGL(k_decompress_tiny,
                           dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, a);
        // ... and those of under 512 bytes in and out, 32 per wavefront
        hipLaunchKernelGGL(k_decompress_small,
                           dim3((uint32_t)((n + 31) / 32)), dim3(64), 0, s, a);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (side)
        return SNAPMI_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], s));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], s));
    ctx->timing_valid = true;
    ctx->timing_is_compress = false;
    ctx->last_kernel = ctx->decode_kernel == 0   ? "k_decompress_sequential"
                       : ctx->decode_kernel == 2 ? "k_decompress_streams2"
                                                 : "k_decompress_streams3";
    ctx->dominant_split = false;
    ctx->codec_launches = 1;
    return SNAPMI_OK;
}

// ---------------------------------------------------------------------
// A batch of few streams waits for its longest one: a stream is decoded by
// one wavefront, 0.14 GiB/s, so the 702 KB of urls.10K are 4.9 ms whatever
// else the batch holds (extras.sweep: 64 MiB of the corpus round decoded at
// 12.8 GiB/s; now 34).  For batches of at most kBatchLongMaxN streams the long ones
// (long_stream_rule, k_long_plan; at
// most kBatchLongMaxL of them) go the way of snapmi_decompress_stream instead - scan, cuts, pieces,
// all long streams of the batch in the same launches (k_bstream_*) - and the
// batch's own launch skips them (mode 3).  The price is one look at the
// lengths on the host (k_long_plan, a copy of what it found, a stream
// synchronisation: ~30 us), which is why large batches, whose long streams
// hide behind each other, do not take it.
// ---------------------------------------------------------------------
// compressed bytes from which a stream can be worth its pieces
// (long_stream_rule; the test build reads SNAPMI_LONG_STREAM: the scalar
// entry points and snapmi_decompress_batch both use it).  Measured per bench
// input at 16 / 64 / 256 KiB (profiles/r4_scalar_latency.txt): the ten small
// launches of the scan cost ~0.5 ms, a wavefront decodes 100-250 MB/s of
// text: html (23 KB) 0.62 -> 0.73 ms through pieces, kppkn.gtb's 69 KB
// 2.12 -> 1.43, urls.10K 4.7 -> 1.3, fireworks.jpeg (literals) 0.17 -> 0.40
static size_t long_stream_min()
{
    static const size_t v = [] {
#ifdef SNAPMI_TESTING
        if (const char *e = getenv("SNAPMI_LONG_STREAM"))
            return (size_t)atoll(e);
#endif
        return (size_t)(32 << 10);
    }();
    return v;
}

constexpr size_t kBatchLongMaxN = 16384;
constexpr uint32_t kBatchLongMaxL = 4096;

#define BL_CHECK(name)                                                        \
    do {                                                                      \
        hipError_t _e = hipGetLastError();                                    \
        if (_e != hipSuccess)                                                 \
            return fail_ctx(ctx, SNAPMI_E_DEVICE, "launch of " #name ": %s",  \
                            hipGetErrorString(_e));                           \
    } while (0)

// Segment size of the scan of long streams (k_stream_scan and the levels
// above it): 4 KiB when there is enough of them to fill the chip with walks
// (one 2 GiB stream: 2.0 ms of scan at 280 GiB/s), 1 KiB below that - the
// scan of a few hundred KiB is then a wait for the longest walk of ONE
// wavefront, ~1 200 hops of ~630 cycles in a 4 KiB segment (0.5 ms whatever
// the size), and a quarter of that with four times the lanes.
static uint32_t stream_seg_log2(const snapmi_ctx *ctx, uint64_t long_bytes)
{
    if (ctx->stream_seg_log2)
        return ctx->stream_seg_log2;
    return long_bytes < ((uint64_t)256 << 20) ? 10u : 12u;
}

// Segments per wavefront of k_stream_scan (StreamArgs::scan_segs) for a call
// whose long streams hold `nseg` segments together: the kernel's full group of
// 64 when that still makes kScanFill wavefronts, else halved until it does
// (not below 8).  A scan wavefront of 64 segments of 1 KiB hands its 64 lanes
// eight rounds of entry walks and then trunks of which the longest is three
// to four times the average - ~1 100 hops of ~630 cycles, 280 us, whoever
// else is on the chip - and 32 MiB of long streams are 506 such wavefronts,
// two per CU.  Measured (profiles/r5_scan_groups.txt): 64 MiB of the corpus
// round 1.359 -> 1.301 ms per call with 16 (8: 1.357 - the groups are a
// second wave of workgroups then), one 126 MB stream as a batch of one 1.690
// -> 1.585 with 32, Decoder::decompress of lcet10.txt 1.076 -> 0.993 with 8,
// 256 MiB 1.622 with 64 and 1.702 with 32: hence 1 024.  (The cuts kernel's
// 512 segments per wavefront were measured the same way: 64 .. 512 are equal,
// its time is the one walk every lane has.)
static uint32_t stream_scan_segs(const snapmi_ctx *ctx, uint64_t nseg)
{
    if (ctx->stream_scan_segs)
        return ctx->stream_scan_segs;
    uint32_t segs = kScanSegs;
    while (segs > 8 && nseg / segs < kScanFill)
        segs /= 2;
    return segs;
}

// pinned host staging of a context (grow-only): pageable copies go through
// the runtime's own staging buffer one at a time, process-wide - eight
// threads calling snappy_compress would queue there
static int pin_reserve(snapmi_ctx *ctx, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap)
        return SNAPMI_OK;
    if (*p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipHostFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return SNAPMI_OK;
}

static int decompress_batch_long(snapmi_ctx *ctx,
                                 const void *const *d_in_ptrs,
                                 const uint64_t *d_in_lens,
                                 void *const *d_out_ptrs,
                                 const uint64_t *d_out_caps,
                                 uint64_t *d_out_lens, snapmi_error *d_errs,
                                 size_t n, bool *done)
{
    *done = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc;
    const size_t list_bytes = 16 + (size_t)kBatchLongMaxL * sizeof(LongItem);
    if ((rc = reserve(ctx, ctx->bl_modes, 2 * n + 64)) ||
        (rc = reserve(ctx, ctx->bl_list, list_bytes)))
        return rc;
    if (ctx->pin_bl_cap < list_bytes) {
        if (ctx->pin_bl) {
            HIP_TRY(ctx, hipStreamSynchronize(s));
            HIP_TRY(ctx, hipHostFree(ctx->pin_bl));
            ctx->pin_bl = nullptr;
            ctx->pin_bl_cap = 0;
        }
        HIP_TRY(ctx, hipHostMalloc(&ctx->pin_bl, list_bytes,
                                   hipHostMallocDefault));
        ctx->pin_bl_cap = list_bytes;
    }
    uint8_t *modes = (uint8_t *)ctx->bl_modes.p, *modes2 = modes + n;
    uint32_t *d_count = (uint32_t *)ctx->bl_list.p;
    LongItem *d_list = (LongItem *)((uint8_t *)ctx->bl_list.p + 16);
    HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 16, s));
    hipLaunchKernelGGL(k_long_plan, dim3(1), dim3(1024), 0, s, d_in_ptrs,
                       d_in_lens, d_out_ptrs, d_out_caps, (uint32_t)n,
                       (uint64_t)long_stream_min(), modes, d_list,
                       kBatchLongMaxL,
                       d_count);
    BL_CHECK(k_long_plan);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pin_bl, ctx->bl_list.p, list_bytes,
                                hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    const uint32_t found = *(const uint32_t *)ctx->pin_bl;
    // (none - or so many that they fill the chip a wavefront each: 3 670
    // long streams in 1 GiB of the corpus round decode in 4.0 ms through
    // pieces and in 5.9 a wavefront each, 3 058 streams of urls.10K at 314
    // and 288 GiB/s, and the gain goes on shrinking: the caller goes on
    // without modes)
    if (found == 0 || found > kBatchLongMaxL)
        return SNAPMI_OK;
    const uint32_t L = found;
    const LongItem *items = (const LongItem *)((const uint8_t *)ctx->pin_bl + 16);

    // ---- geometry, scratch, descriptors ----------------------------------
    // first workgroup of every stream, per kernel: scan, super, super3,
    // spread3, spread2, cuts, pieces
    enum { kPScan, kPSuper, kPSuper3, kPSpread3, kPSpread2, kPCuts, kPPieces, kPre };
    // (descriptors and prefixes are written into pinned memory of the
    // context: their copy to the device needs no wait - the next call's
    // synchronisation behind k_long_plan comes before they are written again)
    const size_t desc_bytes0 = (size_t)L * sizeof(StreamArgs);
    const size_t pre_count = (size_t)kPre * (L + 1);
    if ((rc = pin_reserve(ctx, &ctx->pin_bl2, &ctx->pin_bl2_cap,
                          desc_bytes0 + pre_count * sizeof(uint32_t) + 64)))
        return rc;
    StreamArgs *const descs = (StreamArgs *)ctx->pin_bl2;
    uint32_t *const pre = (uint32_t *)((uint8_t *)ctx->pin_bl2 + desc_bytes0);
    size_t rows = 0, blocks = 0, cuts = 0, pieces = 0;
    uint64_t long_bytes = 0;
    for (uint32_t j = 0; j < L; j++)
        long_bytes += items[j].in_len;
    const uint32_t seg_log2 = stream_seg_log2(ctx, long_bytes);
    const uint64_t seg = 1ull << seg_log2;
    const uint32_t scan_segs = stream_scan_segs(ctx, long_bytes / seg + L);
    for (uint32_t j = 0; j < L; j++) {
        StreamArgs &a = descs[j];
        memset(&a, 0, sizeof a);
        a.seg_log2 = seg_log2;
        a.scan_segs = scan_segs;
        a.in = (const uint8_t *)items[j].in;
        a.in_len = items[j].in_len;
        a.out = (uint8_t *)items[j].out;
        a.out_cap = items[j].out_cap;
        a.out_len = (unsigned long long *)(d_out_lens + items[j].idx);
        a.err = d_errs ? d_errs + items[j].idx : nullptr;
        a.fb_mode = modes2 + items[j].idx;
        a.nseg = (uint32_t)((a.in_len + seg - 1) / seg + 1);
        a.nsuper = (a.nseg + kSegPerSuper - 1) / kSegPerSuper;
        a.nsuper3 = (a.nsuper + kSegPerSuper - 1) / kSegPerSuper;
        a.kmax = (uint32_t)(items[j].dlen / kStreamChunk + 2);
        uint32_t *pj = &pre[j];
        const size_t st = L + 1;
        pj[kPScan * st] = (a.nseg + scan_segs - 1) / scan_segs;
        pj[kPSuper * st] = a.nsuper;
        pj[kPSuper3 * st] = a.nsuper3;
        pj[kPSpread3 * st] = (a.nsuper3 + 63) / 64;
        pj[kPSpread2 * st] = (a.nsuper + 63) / 64;
        pj[kPCuts * st] = (a.nseg + kCutSegs - 1) / kCutSegs;
        pj[kPPieces * st] = (a.kmax + 255) / 256;
        rows += (size_t)a.nseg + ((size_t)a.nsuper + a.nsuper3) * kSegPerSuper;
        blocks += (size_t)a.nseg + a.nsuper + a.nsuper3;
        cuts += (size_t)a.kmax + 1;
        pieces += a.kmax;
    }
    uint32_t grid[kPre];
    for (int k = 0; k < kPre; k++) { // counts -> exclusive prefix, total last
        uint32_t *pk = &pre[(size_t)k * (L + 1)];
        uint32_t acc = 0;
        for (uint32_t j = 0; j < L; j++) {
            const uint32_t c = pk[j];
            pk[j] = acc;
            acc += c;
        }
        pk[L] = acc;
        grid[k] = acc;
    }
    const size_t t_bytes = 64 + (size_t)L * 64 + rows * kEntry * 16 +
                           blocks * 16 + cuts * 16;
    const size_t d_stride = 8 + 8 + 8 + 8 + 8 + sizeof(snapmi_error) + 1;
    const size_t d_bytes = pieces * d_stride + 64;
    const size_t desc_bytes = desc_bytes0;
    const size_t pre_bytes = pre_count * sizeof(uint32_t);
    if ((rc = reserve(ctx, ctx->sd_tables, t_bytes)) ||
        (rc = reserve(ctx, ctx->sd_desc, d_bytes)) ||
        (rc = reserve(ctx, ctx->bl_descs, desc_bytes + pre_bytes + 64)))
        return rc;
    {
        unsigned long long *t = (unsigned long long *)ctx->sd_tables.p;
        unsigned long long *meta = t;
        t += (size_t)L * 8;
        unsigned long long *e_all = t; // the e-tables of all streams: one fill
        unsigned long long *e = e_all;
        t += blocks * 2;
        uint8_t *q = (uint8_t *)ctx->sd_desc.p;
        const void **c_in = (const void **)q;
        q += pieces * 8;
        unsigned long long *c_inlen = (unsigned long long *)q;
        q += pieces * 8;
        void **c_out = (void **)q;
        q += pieces * 8;
        unsigned long long *c_cap = (unsigned long long *)q;
        q += pieces * 8;
        unsigned long long *c_outlen = (unsigned long long *)q;
        q += pieces * 8;
        snapmi_error *c_err = (snapmi_error *)q;
        q += pieces * sizeof(snapmi_error);
        uint8_t *c_mode = q;
        size_t k0 = 0;
        for (uint32_t j = 0; j < L; j++) {
            StreamArgs &a = descs[j];
            a.meta = meta + (size_t)j * 8;
            a.e1 = e;
            e += (size_t)a.nseg * 2;
            a.e2 = e;
            e += (size_t)a.nsuper * 2;
            a.e3 = e;
            e += (size_t)a.nsuper3 * 2;
            a.s1 = t;
            t += (size_t)a.nseg * kEntry * 2;
            a.s2 = t;
            t += (size_t)a.nsuper * kSegPerSuper * kEntry * 2;
            a.s3 = t;
            t += (size_t)a.nsuper3 * kSegPerSuper * kEntry * 2;
            a.cuts = t;
            t += ((size_t)a.kmax + 1) * 2;
            a.c_in = c_in + k0;
            a.c_inlen = c_inlen + k0;
            a.c_out = c_out + k0;
            a.c_cap = c_cap + k0;
            a.c_outlen = c_outlen + k0;
            a.c_err = c_err + k0;
            a.c_mode = c_mode + k0;
            k0 += a.kmax;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->bl_descs.p, descs,
                                    desc_bytes + pre_bytes,
                                    hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemsetAsync(e_all, 0xFF, blocks * 16, s));
        HIP_TRY(ctx, hipMemsetAsync(modes2, 3, n, s));
        const StreamArgs *dd = (const StreamArgs *)ctx->bl_descs.p;
        const uint32_t *dp =
            (const uint32_t *)((uint8_t *)ctx->bl_descs.p + desc_bytes);
        auto B = [&](int k) {
            BatchStreams b;
            b.descs = dd;
            b.pre = k < 0 ? nullptr : dp + (size_t)k * (L + 1);
            b.n = L;
            return b;
        };
        // the batch's other streams beside all this, on the second stream
        // (their longest is 0.6-0.8 ms of one wavefront on the corpus)
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        // from here on every way out joins the side stream again: whatever
        // it was given still writes the caller's arrays and reads bl_modes /
        // bl_order, which the next call reuses
        struct SideJoin {
            snapmi_ctx *c;
            hipStream_t s;
            bool joined = false;
            void join()
            {
                if (joined)
                    return;
                joined = true;
                if (hipEventRecord(c->ev_join, c->stream2) != hipSuccess ||
                    hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    (void)hipStreamSynchronize(c->stream2);
                }
            }
            ~SideJoin() { join(); }
        } side{ctx, s};
        if ((rc = launch_decompress(ctx, d_in_ptrs, d_in_lens, d_out_ptrs,
                                    d_out_caps, d_out_lens, d_errs, modes, n,
                                    nullptr, 0, ctx->stream2,
                                    &ctx->bl_order)))
            return rc;
        hipLaunchKernelGGL(k_bstream_head, dim3(L), dim3(1), 0, s, B(-1));
        BL_CHECK(k_bstream_head);
        hipLaunchKernelGGL(k_bstream_scan, dim3(grid[kPScan]), dim3(64), 0, s,
                           B(kPScan));
        BL_CHECK(k_bstream_scan);
        hipLaunchKernelGGL(k_bstream_super, dim3(grid[kPSuper]), dim3(kEntry),
                           0, s, B(kPSuper));
        BL_CHECK(k_bstream_super);
        hipLaunchKernelGGL(k_bstream_super3, dim3(grid[kPSuper3]),
                           dim3(kEntry), 0, s, B(kPSuper3));
        BL_CHECK(k_bstream_super3);
        hipLaunchKernelGGL(k_bstream_chain, dim3(L), dim3(1), 0, s, B(-1));
        BL_CHECK(k_bstream_chain);
        hipLaunchKernelGGL(k_bstream_spread3, dim3(grid[kPSpread3]), dim3(64),
                           0, s, B(kPSpread3));
        BL_CHECK(k_bstream_spread3);
        hipLaunchKernelGGL(k_bstream_spread2, dim3(grid[kPSpread2]), dim3(64),
                           0, s, B(kPSpread2));
        BL_CHECK(k_bstream_spread2);
        hipLaunchKernelGGL(k_bstream_cuts, dim3(grid[kPCuts]), dim3(64), 0, s,
                           B(kPCuts));
        BL_CHECK(k_bstream_cuts);
        hipLaunchKernelGGL(k_bstream_pieces, dim3(grid[kPPieces]), dim3(256),
                           0, s, B(kPPieces));
        BL_CHECK(k_bstream_pieces);
        // the pieces of the long streams; which of the long ones were
        // irregular; those, by the wavefront decoder
        if ((rc = launch_decompress(ctx, c_in, (const uint64_t *)c_inlen,
                                    c_out, (const uint64_t *)c_cap,
                                    (uint64_t *)c_outlen, c_err, c_mode,
                                    pieces)))
            return rc;
        hipLaunchKernelGGL(k_bstream_finish, dim3(L), dim3(1024), 0, s, B(-1));
        BL_CHECK(k_bstream_finish);
        side.join();
        // (without timing events: snapmi_last_timing reports the pieces)
        if ((rc = launch_decompress(ctx, d_in_ptrs, d_in_lens, d_out_ptrs,
                                    d_out_caps, d_out_lens, d_errs, modes2,
                                    n, nullptr, 0, s, nullptr)))
            return rc;
    }
    *done = true;
    return SNAPMI_OK;
}

} // namespace snapmi

extern "C" {

int snapmi_decompress_batch(snapmi_ctx *ctx, const void *const *d_in_ptrs,
                            const uint64_t *d_in_lens,
                            void *const *d_out_ptrs,
                            const uint64_t *d_out_caps, uint64_t *d_out_lens,
                            snapmi_error *d_errs, size_t n)
{
    if (!ctx)
        return SNAPMI_E_ARGUMENT;
    if (n == 0)
        return SNAPMI_OK;
    if (!d_in_ptrs || !d_in_lens || !d_out_ptrs || !d_out_caps ||
        !d_out_lens || n > 0x7FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "decompress_batch: bad args");
    // (the look at the batch waits for the device once: never while the
    // caller's stream is being captured into a graph - such a caller gets the
    // enqueue-only path, a wavefront per stream)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (ctx->batch_long_streams && n <= kBatchLongMaxN &&
        cap == hipStreamCaptureStatusNone) {
        bool done = false;
        const int rc = decompress_batch_long(ctx, d_in_ptrs, d_in_lens,
                                             d_out_ptrs, d_out_caps,
                                             d_out_lens, d_errs, n, &done);
        if (rc || done)
            return rc;
    }
    return launch_decompress(ctx, d_in_ptrs, d_in_lens, d_out_ptrs, d_out_caps,
                             d_out_lens, d_errs, nullptr, n);
}

#define STREAM_CHECK(name)                                                    \
    do {                                                                      \
        hipError_t _e = hipGetLastError();                                    \
        if (_e != hipSuccess)                                                 \
            return fail_ctx(ctx, SNAPMI_E_DEVICE, "launch of " #name ": %s",  \
                            hipGetErrorString(_e));                           \
    } while (0)

int snapmi_decompress_stream(snapmi_ctx *ctx, const void *d_in,
                             uint64_t in_len, void *d_out, uint64_t out_cap,
                             uint64_t *d_out_len, snapmi_error *d_err)
{
    if (!ctx || !d_out_len || !d_err || (in_len && !d_in) ||
        (out_cap && !d_out))
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "decompress_stream: bad args");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // pieces: one per 64 KiB of output; the output cannot exceed out_cap nor
    // ~21.4x the input (a 3-byte copy element yields at most 64 bytes)
    uint64_t bound = out_cap;
    if (in_len < (1ull << 40) && in_len * 22 < bound)
        bound = in_len * 22;
    const uint64_t kmax64 = bound / kStreamChunk + 2;
    const uint32_t seg_log2 = stream_seg_log2(ctx, in_len);
    const uint64_t seg = 1ull << seg_log2;
    const uint64_t nseg64 = (in_len + seg - 1) / seg + 1;
    if (kmax64 > 0x3FFFFFFFu || nseg64 > 0x3FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "decompress_stream: too long");
    StreamArgs a;
    a.in = (const uint8_t *)d_in;
    a.in_len = in_len;
    a.out = (uint8_t *)d_out;
    a.out_cap = out_cap;
    a.out_len = (unsigned long long *)d_out_len;
    a.err = d_err;
    a.nseg = (uint32_t)nseg64;
    a.seg_log2 = seg_log2;
    a.scan_segs = stream_scan_segs(ctx, a.nseg);
    a.nsuper = (a.nseg + kSegPerSuper - 1) / kSegPerSuper;
    a.nsuper3 = (a.nsuper + kSegPerSuper - 1) / kSegPerSuper;
    a.kmax = (uint32_t)kmax64;
    a.fb_mode = nullptr;
    const size_t blocks = (size_t)a.nseg + a.nsuper + a.nsuper3;
    const size_t rows = (size_t)a.nseg +
                        ((size_t)a.nsuper + a.nsuper3) * kSegPerSuper;
    const size_t t_bytes = 64 + rows * kEntry * 16 + blocks * 16 +
                           ((size_t)a.kmax + 1) * 16;
    // descriptors: the pieces, then the whole stream as batch entry [kmax]
    const size_t d_stride = 8 + 8 + 8 + 8 + 8 + sizeof(snapmi_error);
    const size_t d_bytes = ((size_t)a.kmax + 1) * (d_stride + 1) + 64;
    int rc;
    if ((rc = reserve(ctx, ctx->sd_tables, t_bytes)) ||
        (rc = reserve(ctx, ctx->sd_desc, d_bytes)))
        return rc;
    unsigned long long *t = (unsigned long long *)ctx->sd_tables.p;
    a.meta = t;
    t += 8;
    a.s1 = t;
    t += (size_t)a.nseg * kEntry * 2;
    a.s2 = t;
    t += (size_t)a.nsuper * kSegPerSuper * kEntry * 2;
    a.s3 = t;
    t += (size_t)a.nsuper3 * kSegPerSuper * kEntry * 2;
    a.e1 = t; // e1, e2, e3 contiguous: one memset
    t += (size_t)a.nseg * 2;
    a.e2 = t;
    t += (size_t)a.nsuper * 2;
    a.e3 = t;
    t += (size_t)a.nsuper3 * 2;
    a.cuts = t;
    const size_t m = (size_t)a.kmax + 1;
    uint8_t *q = (uint8_t *)ctx->sd_desc.p;
    a.c_in = (const void **)q;
    q += m * 8;
    a.c_inlen = (unsigned long long *)q;
    q += m * 8;
    a.c_out = (void **)q;
    q += m * 8;
    a.c_cap = (unsigned long long *)q;
    q += m * 8;
    a.c_outlen = (unsigned long long *)q;
    q += m * 8;
    a.c_err = (snapmi_error *)q;
    q += m * sizeof(snapmi_error);
    a.c_mode = q;

    // entry [kmax]: the stream itself, for the sequential decoder
    struct {
        const void *in;
        uint64_t in_len;
        void *out;
        uint64_t cap;
    } whole = {d_in, in_len, d_out, out_cap};
    HIP_TRY(ctx, hipMemcpyAsync((void *)&a.c_in[a.kmax], &whole.in, 8,
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(&a.c_inlen[a.kmax], &whole.in_len, 8,
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(&a.c_out[a.kmax], &whole.out, 8,
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(&a.c_cap[a.kmax], &whole.cap, 8,
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(&a.c_mode[a.kmax], 0, 1, s));
    HIP_TRY(ctx, hipMemsetAsync(a.e1, 0xFF, blocks * 16, s));

    hipLaunchKernelGGL(k_stream_head, dim3(1), dim3(1), 0, s, a);
    STREAM_CHECK(k_stream_head);
    hipLaunchKernelGGL(k_stream_scan,
                       dim3((a.nseg + a.scan_segs - 1) / a.scan_segs),
                       dim3(64), 0, s, a);
    STREAM_CHECK(k_stream_scan);
    hipLaunchKernelGGL(k_stream_super, dim3(a.nsuper), dim3(kEntry), 0, s, a);
    STREAM_CHECK(k_stream_super);
    hipLaunchKernelGGL(k_stream_super3, dim3(a.nsuper3), dim3(kEntry), 0, s,
                       a);
    STREAM_CHECK(k_stream_super3);
    hipLaunchKernelGGL(k_stream_chain, dim3(1), dim3(1), 0, s, a);
    STREAM_CHECK(k_stream_chain);
    hipLaunchKernelGGL(k_stream_spread3, dim3((a.nsuper3 + 63) / 64),
                       dim3(64), 0, s, a);
    STREAM_CHECK(k_stream_spread3);
    hipLaunchKernelGGL(k_stream_spread2, dim3((a.nsuper + 63) / 64), dim3(64),
                       0, s, a);
    STREAM_CHECK(k_stream_spread2);
    hipLaunchKernelGGL(k_stream_cuts, dim3((a.nseg + kCutSegs - 1) / kCutSegs),
                       dim3(64), 0, s, a);
    STREAM_CHECK(k_stream_cuts);
    hipLaunchKernelGGL(k_stream_pieces, dim3((a.kmax + 255) / 256), dim3(256),
                       0, s, a);
    STREAM_CHECK(k_stream_pieces);
    // the pieces, unless the scan gave up (meta[2] == 1) ...
    rc = launch_decompress(ctx, a.c_in, (const uint64_t *)a.c_inlen, a.c_out,
                           (const uint64_t *)a.c_cap, (uint64_t *)a.c_outlen,
                           a.c_err, a.c_mode, a.kmax, a.meta + 2, 0);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_stream_finish, dim3(1), dim3(1024), 0, s, a);
    // ... and the sequential decoder over the whole stream if anything was
    // irregular: it owns the error report
    return launch_decompress(ctx, a.c_in + a.kmax,
                             (const uint64_t *)a.c_inlen + a.kmax,
                             a.c_out + a.kmax,
                             (const uint64_t *)a.c_cap + a.kmax, d_out_len,
                             d_err, a.c_mode + a.kmax, 1, a.meta + 2, 1);
}

int snapmi_stream_decode_path(snapmi_ctx *ctx)
{
    if (!ctx || !ctx->sd_tables.p)
        return -1;
    unsigned long long meta[4];
    if (hipSetDevice(ctx->device) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(meta, ctx->sd_tables.p, sizeof meta,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return meta[2] ? 1 : 0;
}

int snapmi_decompress_len_batch(snapmi_ctx *ctx,
                                const void *const *d_in_ptrs,
                                const uint64_t *d_in_lens,
                                uint64_t *d_out_lens, snapmi_error *d_errs,
                                size_t n)
{
    if (!ctx)
        return SNAPMI_E_ARGUMENT;
    if (n == 0)
        return SNAPMI_OK;
    if (!d_in_ptrs || !d_in_lens || !d_out_lens || n > 0x7FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                        "decompress_len_batch: bad args");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DecompressArgs a;
    a.in_ptrs = d_in_ptrs;
    a.in_lens = d_in_lens;
    a.out_ptrs = nullptr;
    a.out_caps = nullptr;
    a.out_lens = d_out_lens;
    a.errs = d_errs;
    a.modes = nullptr;
    a.n_streams = (uint32_t)n;
    a.order = nullptr;
    a.bucket_pos = nullptr;
    a.prof = nullptr;
    a.gate = nullptr;
    a.gate_value = 0;
    hipLaunchKernelGGL(k_decompress_len, dim3((uint32_t)((n + 255) / 256)),
                       dim3(256), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return SNAPMI_OK;
}

PROF(
// experiment builds only: copy out the 16 cycle counters of the last
// compress batch (not part of include/snapmi.h)
SNAPMI_API int snapmi_debug_profile(snapmi_ctx *ctx, uint64_t *out16)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out16, ctx->st_prof.p, 16 * sizeof(uint64_t),
                           hipMemcpyDeviceToHost));
    return SNAPMI_OK;
}
)

int snapmi_last_timing(snapmi_ctx *ctx, snapmi_timing *out)
{
    if (!ctx || !out)
        return SNAPMI_E_ARGUMENT;
    memset(out, 0, sizeof *out);
    if (!ctx->timing_valid)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "no batch has been timed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
    HIP_TRY(ctx, hipEventElapsedTime(&out->plan_ms, ctx->ev[0], ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&out->codec_ms, ctx->ev[1], ctx->ev[2]));
    HIP_TRY(ctx,
            hipEventElapsedTime(&out->compact_ms, ctx->ev[2], ctx->ev[3]));
    HIP_TRY(ctx, hipEventElapsedTime(&out->total_ms, ctx->ev[0], ctx->ev[3]));
    out->codec_launches = ctx->codec_launches;
    out->dominant_ms = out->codec_ms;
    if (ctx->dominant_split)
        HIP_TRY(ctx, hipEventElapsedTime(&out->dominant_ms, ctx->ev[4],
                                         ctx->ev[5]));
    return SNAPMI_OK;
}

// ----------------------------------------------------------------------
// scalar mirrors (host buffers): stage through device memory, batch of 1
// ----------------------------------------------------------------------
namespace {

struct OneDesc {
    const void *in_ptr;
    uint64_t in_len;
    void *out_ptr;
    uint64_t out_cap;
    uint64_t out_len;
    uint64_t pad;
    snapmi_error err;
};

// host buffers of up to this many bytes go through the context's pinned
// staging (one host memcpy each way, no pageable device copy)
constexpr size_t kPinStage = 8u << 20;


int run_one(snapmi_ctx *ctx, bool compress, const uint8_t *input,
            size_t input_len, uint8_t *output, size_t output_cap,
            size_t *written, snapmi_error *err)
{
    if (!ctx || !written || (!input && input_len) || (!output && output_cap))
        return SNAPMI_E_ARGUMENT;
    *written = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    // Device output buffer: what the kernels may write.  For compression the
    // reference demands max_compress_len (checked on the device as well).
    size_t dev_out = output_cap;
    size_t dl = 0; // decompress: the length the header announces
    if (!compress) {
        snapmi_error he;
        if (input_len && snapmi_decompress_len(input, input_len, &dl, &he) ==
                             SNAPMI_OK &&
            dl < dev_out)
            dev_out = dl;
    } else {
        size_t need = snapmi_max_compress_len(input_len);
        if (need && need < dev_out)
            dev_out = need;
    }
    // (inputs up to a few MiB are staged through pinned memory of the
    // context; larger ones are copied from where they lie)
    const bool staged = input_len <= kPinStage && dev_out <= kPinStage;
    if ((rc = reserve(ctx, ctx->st_in, input_len + 16)) ||
        (rc = reserve(ctx, ctx->st_out, dev_out + 64)) ||
        (rc = reserve(ctx, ctx->st_desc, sizeof(OneDesc))) ||
        (rc = pin_reserve(ctx, &ctx->pin_desc, &ctx->pin_desc_cap,
                          2 * sizeof(OneDesc))) ||
        (staged && ((rc = pin_reserve(ctx, &ctx->pin_in, &ctx->pin_in_cap,
                                      input_len + 16)) ||
                    (rc = pin_reserve(ctx, &ctx->pin_out, &ctx->pin_out_cap,
                                      dev_out + 64)))))
        return rc;
    OneDesc *hd = (OneDesc *)ctx->pin_desc; // [0] in, [1] back
    memset(&hd[0], 0, sizeof hd[0]);
    hd[0].in_ptr = ctx->st_in.p;
    hd[0].in_len = input_len;
    hd[0].out_ptr = ctx->st_out.p;
    hd[0].out_cap = output_cap; // the caller's capacity is what is validated
    hipStream_t s = ctx->stream;
    if (input_len) {
        const void *from = input;
        if (staged) {
            memcpy(ctx->pin_in, input, input_len);
            from = ctx->pin_in;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->st_in.p, from, input_len,
                                    hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st_desc.p, &hd[0], sizeof hd[0],
                                hipMemcpyHostToDevice, s));
    OneDesc *d = (OneDesc *)ctx->st_desc.p;
    if (compress) {
        uint64_t hl = input_len;
        rc = snapmi_compress_batch(ctx, &d->in_ptr, &d->in_len, &hl,
                                   &d->out_ptr, &d->out_cap, &d->out_len,
                                   &d->err, 1);
    } else if (long_stream_rule(input_len, dl, long_stream_min())) {
        // one wavefront decodes ~140 MB/s: long streams go through the
        // parallel single-stream path
        rc = snapmi_decompress_stream(ctx, ctx->st_in.p, input_len,
                                      ctx->st_out.p, output_cap, &d->out_len,
                                      &d->err);
    } else {
        // (straight to the wavefront decoder: the host has just made the
        // decision snapmi_decompress_batch would wait for the device to make)
        rc = launch_decompress(ctx, &d->in_ptr, &d->in_len, &d->out_ptr,
                               &d->out_cap, &d->out_len, &d->err, nullptr, 1);
    }
    if (rc)
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(&hd[1], ctx->st_desc.p, sizeof hd[1],
                                hipMemcpyDeviceToHost, s));
    if (staged) {
        // the result comes along in the same round trip: as much as the
        // kernels can have written (the length is not known to the host yet)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pin_out, ctx->st_out.p,
                                    dev_out, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (compress && (rc = release_batch_scratch(ctx)))
        return rc;
    const OneDesc &h = hd[1];
    if (err)
        *err = h.err;
    if (h.err.kind != SNAPMI_OK)
        return h.err.kind;
    if (h.out_len > output_cap)
        return fail_ctx(ctx, SNAPMI_E_DEVICE, "device wrote %llu > cap %zu",
                        (unsigned long long)h.out_len, output_cap);
    if (h.out_len) {
        if (staged)
            memcpy(output, ctx->pin_out, h.out_len);
        else
            HIP_TRY(ctx, hipMemcpy(output, ctx->st_out.p, h.out_len,
                                   hipMemcpyDeviceToHost));
    }
    *written = (size_t)h.out_len;
    return SNAPMI_OK;
}

} // namespace

int snapmi_raw_compress(snapmi_ctx *ctx, const uint8_t *input,
                        size_t input_len, uint8_t *output, size_t output_cap,
                        size_t *written, snapmi_error *err)
{
    return run_one(ctx, true, input, input_len, output, output_cap, written,
                   err);
}

int snapmi_raw_decompress(snapmi_ctx *ctx, const uint8_t *input,
                          size_t input_len, uint8_t *output,
                          size_t output_cap, size_t *written,
                          snapmi_error *err)
{
    return run_one(ctx, false, input, input_len, output, output_cap, written,
                   err);
}

// ----------------------------------------------------------------------
// libsnappy C API (snappy-c.h), as bound by the reference's snappy-cpp
// crate.  The reference's wrappers are stateless and may be called from any
// number of threads at once (snappy-cpp/src/lib.rs:13-64), so these calls
// share a small process-wide pool of contexts on device SNAPMI_DEVICE
// (default 0; created on demand, up to SNAPMI_SEAM_CONTEXTS, default 2).
//
// Round 5: concurrent calls are COMBINED.  One call is ~1.5 ms of a lone
// wavefront per block whatever else the GPU does, so eight callers on eight
// streams got 3.3-3.8x the rate of one (round 4) - but a batch of sixteen such
// streams takes about as long as one.  A caller stages its input in pinned
// memory of its own thread, queues a request and either finds it done by
// another caller or becomes a leader: it takes a context, waits a few
// microseconds for requests that are just arriving, takes every queued
// request of its kind and runs them as ONE snapmi_compress_batch /
// snapmi_decompress_batch - per-request copies in, one launch, per-request
// copies out, one wait.  Every caller then moves its own bytes from its
// pinned buffer to the buffer it was given.  Results and errors are per
// stream, exactly those of the batch call.  Inputs of more than kPinStage
// bytes go one by one, as before.
// ----------------------------------------------------------------------
namespace {
struct SeamPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<snapmi_ctx *> idle;
    size_t created = 0, cap = 0;
    bool broken = false; // a context could not be created: do not retry

    // (mu held) a context if one is idle or may still be created; *wait =
    // whether one will come back
    snapmi_ctx *checkout_locked(std::unique_lock<std::mutex> &lock,
                                bool block, bool *none)
    {
        *none = false;
        if (cap == 0) {
            // (two: one batch runs while the next one gathers - with more
            // contexts the callers spread over more, smaller batches: 16
            // callers on alice29.txt 1 240 / 1 060 / 740 MB/s with 2 / 4 / 8,
            // profiles/r5_seam_sweep.txt)
            cap = 2;
            if (const char *e = getenv("SNAPMI_SEAM_CONTEXTS"))
                cap = (size_t)(atoi(e) < 1 ? 1 : atoi(e));
        }
        for (;;) {
            if (!idle.empty()) {
                snapmi_ctx *c = idle.back();
                idle.pop_back();
                return c;
            }
            if (created < cap && !broken) {
                created++; // reserved: created outside the lock
                lock.unlock();
                int dev = 0;
                if (const char *e = getenv("SNAPMI_DEVICE"))
                    dev = atoi(e);
                snapmi_ctx *c = nullptr;
                const bool ok = snapmi_ctx_create(dev, nullptr, &c) == SNAPMI_OK;
                lock.lock();
                if (ok)
                    return c;
                created--;
                broken = true; // (snapmi_ctx_create has printed why)
                cv.notify_all();
            }
            if (created == 0) {
                *none = true; // no context and none can be made
                return nullptr;
            }
            if (!block)
                return nullptr;
            cv.wait(lock);
        }
    }
    snapmi_ctx *checkout()
    {
        std::unique_lock<std::mutex> lock(mu);
        bool none;
        return checkout_locked(lock, true, &none);
    }
    void give_back(snapmi_ctx *c)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            idle.push_back(c);
        }
        cv.notify_all();
    }
};
SeamPool g_pool;

struct SeamLease {
    snapmi_ctx *ctx;
    SeamLease() : ctx(g_pool.checkout()) {}
    ~SeamLease()
    {
        if (ctx)
            g_pool.give_back(ctx);
    }
};

// set by an atexit handler: the process is leaving, the HIP runtime with it
std::atomic<bool> g_seam_exiting{false};
struct SeamExitHook {
    SeamExitHook()
    {
        atexit([] { g_seam_exiting.store(true, std::memory_order_release); });
    }
} g_seam_exit_hook;

// pinned staging of the calling thread (at most 2 MiB kept between calls,
// freed with the thread)
struct ThreadPin {
    void *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap)
            return true;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return false;
        }
        cap = want;
        return true;
    }
    // a call that needed a large buffer does not leave it with the thread
    // for good: above kKeep the buffer goes back once the call is over
    static constexpr size_t kKeep = (size_t)2 << 20;
    void trim()
    {
        if (cap > kKeep) {
            (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
        }
    }
    ~ThreadPin()
    {
        // (a thread that ends while the process is leaving may find the HIP
        // runtime gone: its own teardown frees pinned memory then, and a
        // call into it would not come back)
        if (p && !g_seam_exiting.load(std::memory_order_acquire))
            (void)hipHostFree(p);
    }
};
thread_local ThreadPin tl_pin_in, tl_pin_out;

struct SeamReq {
    bool compress;
    size_t in_len, out_cap; // the caller's
    size_t dev_out;         // bytes the device may write = bytes coming back
    const uint8_t *pin_in;  // the caller's pinned copy of its input
    uint8_t *pin_out;       // ... and room for dev_out bytes of result
    int state;              // 0 queued, 1 taken by a leader, 2 done
    int rc;
    size_t written;
    snapmi_error err;
};

// The single-launch path of a lone small call (round 6): a request of under
// 256 bytes (compress: input; uncompress: compressed bytes, at most 256 of
// output) that has no company runs as ONE kernel whose descriptor, input and
// output lie in pinned host memory - the caller's staging buffers, which the
// device reaches over the link - and whose end the host sees by polling a
// word of that memory: no copy commands, no plan kernel, no stream
// synchronisation (round 5: two copies each way, two to three kernels and a
// hipStreamSynchronize, ~80 us for 200 bytes).  The reference's seam is a
// plain function call (snappy-cpp/src/lib.rs:13-64); this is as close as a
// device gets.  Returns false when it cannot run (the batch path takes over).
struct SeamTinyBlock {  // in ctx->pin_desc
    uint64_t in_ptr, in_len, out_ptr, out_cap, out_len;
    snapmi_error err;
    uint32_t order0;         // DecompressArgs::order: stream 0
    uint32_t bucket_pos[66]; // (unused by the kernel; room the args point at)
    uint32_t done;
};

bool seam_tiny(snapmi_ctx *ctx, SeamReq *r)
{
    if (r->in_len == 0 || r->in_len >= kTinyCompress)
        return false;
    if (r->compress ? !ctx->tiny_stream_kernel : r->dev_out > 256)
        return false;
    if (pin_reserve(ctx, &ctx->pin_desc, &ctx->pin_desc_cap,
                    sizeof(SeamTinyBlock) + 64) != SNAPMI_OK)
        return false;
    SeamTinyBlock *h = (SeamTinyBlock *)ctx->pin_desc;
    void *d_blk = nullptr, *d_in = nullptr, *d_out = nullptr;
    if (hipHostGetDevicePointer(&d_blk, h, 0) != hipSuccess ||
        hipHostGetDevicePointer(&d_in, (void *)r->pin_in, 0) != hipSuccess ||
        hipHostGetDevicePointer(&d_out, r->pin_out, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    SeamTinyBlock *d = (SeamTinyBlock *)d_blk;
    const uint32_t seq = ++ctx->seam_seq ? ctx->seam_seq : ++ctx->seam_seq;
    h->in_ptr = (uint64_t)(uintptr_t)d_in;
    h->in_len = r->in_len;
    h->out_ptr = (uint64_t)(uintptr_t)d_out;
    h->out_cap = r->out_cap < r->dev_out ? r->out_cap : r->dev_out;
    h->out_len = 0;
    memset(&h->err, 0, sizeof h->err);
    h->order0 = 0;
    h->done = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (r->compress) {
        hipLaunchKernelGGL(k_seam_compress_tiny, dim3(1), dim3(64), 0,
                           ctx->stream, (const uint8_t *)d_in,
                           (uint32_t)r->in_len, (uint8_t *)d_out,
                           (unsigned long long *)&d->out_len, &d->done, seq);
    } else {
        DecompressArgs a;
        memset(&a, 0, sizeof a);
        a.in_ptrs = (const void *const *)&d->in_ptr;
        a.in_lens = &d->in_len;
        a.out_ptrs = (void *const *)&d->out_ptr;
        a.out_caps = &d->out_cap;
        a.out_lens = &d->out_len;
        a.errs = &d->err;
        a.n_streams = 1;
        a.order = &d->order0;
        a.bucket_pos = d->bucket_pos;
        hipLaunchKernelGGL(k_seam_decompress_tiny, dim3(1), dim3(64), 0,
                           ctx->stream, a, &d->done, seq);
    }
    if (hipGetLastError() != hipSuccess)
        return false;
    // the end: the device's release store of `seq` (spin; after 2 ms of it,
    // the stream's own wait - a queue behind somebody else's work)
    volatile uint32_t *done = &h->done;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t spins = 0;; spins++) {
        if (*done == seq) {
            seen = true;
            break;
        }
        if ((spins & 255) == 255 &&
            std::chrono::steady_clock::now() - t0 >
                std::chrono::milliseconds(2))
            break;
    }
    if (!seen && (hipStreamSynchronize(ctx->stream) != hipSuccess ||
                  *done != seq)) {
        (void)hipGetLastError();
        r->rc = SNAPMI_E_DEVICE;
        r->written = 0;
        memset(&r->err, 0, sizeof r->err);
        r->err.kind = SNAPMI_E_DEVICE;
        return true;
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    r->err = h->err;
    r->rc = r->compress ? SNAPMI_OK : h->err.kind;
    r->written = r->rc == SNAPMI_OK ? (size_t)h->out_len : 0;
    if (r->written > r->dev_out) {
        r->rc = SNAPMI_E_DEVICE;
        r->written = 0;
    }
    return true;
}

// one batch of requests of one kind on `ctx` (no lock held)
void seam_execute(snapmi_ctx *ctx, const std::vector<SeamReq *> &batch)
{
    const size_t n = batch.size();
    const bool compress = batch[0]->compress;
    if (n == 1 && hipSetDevice(ctx->device) == hipSuccess &&
        seam_tiny(ctx, batch[0]))
        return;
    auto fail_all = [&](int rc) {
        for (SeamReq *r : batch) {
            r->rc = rc;
            r->written = 0;
            memset(&r->err, 0, sizeof r->err);
            r->err.kind = rc;
        }
    };
    if (hipSetDevice(ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        return fail_all(SNAPMI_E_DEVICE);
    }
    // device slabs: inputs and outputs back to back, 16-byte aligned
    std::vector<size_t> in_off(n), out_off(n);
    size_t in_total = 0, out_total = 0;
    for (size_t i = 0; i < n; i++) {
        in_off[i] = in_total;
        in_total += (batch[i]->in_len + 16 + 15) & ~(size_t)15;
        out_off[i] = out_total;
        out_total += (batch[i]->dev_out + 64 + 15) & ~(size_t)15;
    }
    // descriptors, structure of arrays: in_ptrs, in_lens, out_ptrs, out_caps,
    // out_lens (8 bytes each), errs (32)
    const size_t desc_bytes = n * (5 * 8 + sizeof(snapmi_error));
    int rc;
    if ((rc = reserve(ctx, ctx->st_in, in_total + 16)) ||
        (rc = reserve(ctx, ctx->st_out, out_total + 64)) ||
        (rc = reserve(ctx, ctx->st_desc, desc_bytes)) ||
        (rc = pin_reserve(ctx, &ctx->pin_desc, &ctx->pin_desc_cap,
                          2 * desc_bytes)))
        return fail_all(rc);
    uint8_t *hd = (uint8_t *)ctx->pin_desc, *hback = hd + desc_bytes;
    uint8_t *dd = (uint8_t *)ctx->st_desc.p;
    uint64_t *h_in_ptrs = (uint64_t *)hd, *h_in_lens = h_in_ptrs + n,
             *h_out_ptrs = h_in_lens + n, *h_out_caps = h_out_ptrs + n;
    memset(hd, 0, desc_bytes);
    for (size_t i = 0; i < n; i++) {
        h_in_ptrs[i] = (uint64_t)(uintptr_t)((uint8_t *)ctx->st_in.p + in_off[i]);
        h_in_lens[i] = batch[i]->in_len;
        h_out_ptrs[i] =
            (uint64_t)(uintptr_t)((uint8_t *)ctx->st_out.p + out_off[i]);
        // (the caller's capacity was checked against what the call needs -
        // max_compress_len / the header's length - before it got here; the
        // kernels get what the request's slab holds, so that their own cap
        // checks keep them inside it)
        h_out_caps[i] = batch[i]->out_cap < batch[i]->dev_out
                            ? batch[i]->out_cap
                            : batch[i]->dev_out;
    }
    hipStream_t s = ctx->stream;
    bool ok = true;
    for (size_t i = 0; i < n && ok; i++)
        if (batch[i]->in_len)
            ok = hipMemcpyAsync((uint8_t *)ctx->st_in.p + in_off[i],
                                batch[i]->pin_in, batch[i]->in_len,
                                hipMemcpyHostToDevice, s) == hipSuccess;
    ok = ok && hipMemcpyAsync(dd, hd, desc_bytes, hipMemcpyHostToDevice, s) ==
                   hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        return fail_all(SNAPMI_E_DEVICE);
    }
    const void *const *d_in_ptrs = (const void *const *)dd;
    const uint64_t *d_in_lens = (const uint64_t *)(dd + 8 * n);
    void *const *d_out_ptrs = (void *const *)(dd + 16 * n);
    const uint64_t *d_out_caps = (const uint64_t *)(dd + 24 * n);
    uint64_t *d_out_lens = (uint64_t *)(dd + 32 * n);
    snapmi_error *d_errs = (snapmi_error *)(dd + 40 * n);
    if (compress)
        rc = snapmi_compress_batch(ctx, d_in_ptrs, d_in_lens, h_in_lens,
                                   d_out_ptrs, d_out_caps, d_out_lens, d_errs,
                                   n);
    else
        rc = snapmi_decompress_batch(ctx, d_in_ptrs, d_in_lens, d_out_ptrs,
                                     d_out_caps, d_out_lens, d_errs, n);
    if (rc) {
        (void)hipStreamSynchronize(s);
        return fail_all(rc);
    }
    ok = hipMemcpyAsync(hback, dd, desc_bytes, hipMemcpyDeviceToHost, s) ==
         hipSuccess;
    // the results come along in the same round trip: as much as the kernels
    // can have written (the lengths are not known to the host yet)
    for (size_t i = 0; i < n && ok; i++)
        if (batch[i]->dev_out)
            ok = hipMemcpyAsync(batch[i]->pin_out,
                                (uint8_t *)ctx->st_out.p + out_off[i],
                                batch[i]->dev_out, hipMemcpyDeviceToHost,
                                s) == hipSuccess;
    if (hipStreamSynchronize(s) != hipSuccess || !ok) {
        (void)hipGetLastError();
        return fail_all(SNAPMI_E_DEVICE);
    }
    (void)release_batch_scratch(ctx);
    const uint64_t *b_out_lens = (const uint64_t *)(hback + 32 * n);
    const snapmi_error *b_errs = (const snapmi_error *)(hback + 40 * n);
    for (size_t i = 0; i < n; i++) {
        SeamReq *r = batch[i];
        r->err = b_errs[i];
        r->rc = b_errs[i].kind;
        r->written = b_errs[i].kind == SNAPMI_OK ? (size_t)b_out_lens[i] : 0;
        if (r->written > r->dev_out) { // cannot be: the device checks the caps
            r->rc = SNAPMI_E_DEVICE;
            r->written = 0;
        }
    }
}

struct SeamCombiner {
    std::deque<SeamReq *> q; // under g_pool.mu
    static constexpr size_t kMaxBatch = 1024;
    static constexpr size_t kMaxBytes = (size_t)1 << 30;
    // (under g_pool.mu) batches running now; when a request last had company
    size_t in_flight = 0;
    std::chrono::steady_clock::time_point last_company{};

    void run(SeamReq *r)
    {
        std::unique_lock<std::mutex> lock(g_pool.mu);
        r->state = 0;
        q.push_back(r);
        for (;;) {
            if (r->state == 2)
                return;
            if (r->state == 0) {
                bool none = false;
                snapmi_ctx *ctx = g_pool.checkout_locked(lock, false, &none);
                if (r->state != 0) { // (the lock was dropped meanwhile)
                    if (ctx) {
                        g_pool.idle.push_back(ctx);
                        g_pool.cv.notify_all();
                    }
                    continue;
                }
                if (none) { // no device: only this request fails here
                    for (auto it = q.begin(); it != q.end(); ++it)
                        if (*it == r) {
                            q.erase(it);
                            break;
                        }
                    r->rc = SNAPMI_E_DEVICE;
                    r->written = 0;
                    r->state = 2;
                    return;
                }
                if (ctx) {
                    lead(lock, ctx, r);
                    continue;
                }
            }
            g_pool.cv.wait(lock);
        }
    }
    // (mu held on entry and exit) gather, run, hand out
    void lead(std::unique_lock<std::mutex> &lock, snapmi_ctx *ctx, SeamReq *r)
    {
        // requests that are arriving right now join: until the queue has not
        // grown for ~8 us, 50 us at most (a call is ~1 500 us of GPU time) -
        // unless this caller has been alone for a while (no second request
        // queued or in flight during the last 2 ms): a single-threaded
        // caller pays no window at all
        const auto t_enter = std::chrono::steady_clock::now();
        if (q.size() > 1 || in_flight > 0)
            last_company = t_enter;
        if (t_enter - last_company <= std::chrono::milliseconds(2)) {
            size_t seen = q.size();
            lock.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            auto t_grow = t0;
            for (;;) {
                std::this_thread::yield();
                const auto now = std::chrono::steady_clock::now();
                lock.lock();
                const size_t have = q.size();
                lock.unlock();
                if (have != seen) {
                    seen = have;
                    t_grow = now;
                }
                if (now - t_grow > std::chrono::microseconds(8) ||
                    now - t0 > std::chrono::microseconds(50))
                    break;
            }
            lock.lock();
        }
        if (r->state != 0) { // another leader took it during the window
            g_pool.idle.push_back(ctx);
            g_pool.cv.notify_all();
            return;
        }
        std::vector<SeamReq *> batch;
        size_t bytes = 0;
        for (auto it = q.begin(); it != q.end();) {
            SeamReq *x = *it;
            const size_t cost = x->in_len + x->dev_out;
            if (x->compress == r->compress &&
                (x == r || (batch.size() < kMaxBatch - 1 &&
                            bytes + cost <= kMaxBytes))) {
                x->state = 1;
                bytes += cost;
                batch.push_back(x);
                it = q.erase(it);
            } else {
                ++it;
            }
        }
        if (batch.size() > 1)
            last_company = std::chrono::steady_clock::now();
        in_flight++;
        lock.unlock();
        seam_execute(ctx, batch);
        lock.lock();
        in_flight--;
        for (SeamReq *x : batch)
            x->state = 2;
        g_pool.idle.push_back(ctx);
        g_pool.cv.notify_all();
    }
};
SeamCombiner g_seam;

// one seam call: through the combiner, or alone when the input is large
int seam_call(bool compress, const uint8_t *input, size_t input_len,
              uint8_t *output, size_t output_cap, size_t dev_out,
              size_t *written, const char *what)
{
    *written = 0;
    if (input_len > kPinStage || dev_out > kPinStage ||
        !tl_pin_in.reserve(input_len + 16) ||
        !tl_pin_out.reserve(dev_out + 64)) {
        SeamLease lease;
        snapmi_ctx *ctx = lease.ctx;
        if (!ctx) // (snapmi_ctx_create has printed why)
            return SNAPMI_E_DEVICE;
        snapmi_error err;
        const int rc = run_one(ctx, compress, input, input_len, output,
                               output_cap, written, &err);
        if (rc >= SNAPMI_E_DEVICE)
            fprintf(stderr, "snapmi: %s: %s\n", what, snapmi_last_error(ctx));
        return rc;
    }
    if (input_len)
        memcpy(tl_pin_in.p, input, input_len);
    SeamReq r;
    r.compress = compress;
    r.in_len = input_len;
    r.out_cap = output_cap;
    r.dev_out = dev_out;
    r.pin_in = (const uint8_t *)tl_pin_in.p;
    r.pin_out = (uint8_t *)tl_pin_out.p;
    r.rc = SNAPMI_E_DEVICE;
    r.written = 0;
    g_seam.run(&r);
    if (r.rc == SNAPMI_OK && r.written) {
        memcpy(output, r.pin_out, r.written);
        *written = r.written;
    }
    tl_pin_in.trim();
    tl_pin_out.trim();
    if (r.rc >= SNAPMI_E_DEVICE)
        fprintf(stderr, "snapmi: %s: device failure (no CPU fallback)\n",
                what);
    return r.rc;
}
} // namespace

size_t snappy_max_compressed_length(size_t source_length)
{
    return 32 + source_length + source_length / 6;
}

snappy_status snappy_uncompressed_length(const char *compressed,
                                         size_t compressed_length,
                                         size_t *result)
{
    // libsnappy reads a varint32: at most 5 bytes, value < 2^32.
    uint64_t v = 0;
    size_t n = compressed_length < 5 ? compressed_length : 5;
    size_t h = host_varint((const uint8_t *)compressed, n, &v);
    if (h == 0 || v > kMaxInput)
        return SNAPPY_INVALID_INPUT;
    *result = (size_t)v;
    return SNAPPY_OK;
}

snappy_status snappy_compress(const char *input, size_t input_length,
                              char *compressed, size_t *compressed_length)
{
    if (!compressed_length)
        return SNAPPY_INVALID_INPUT;
    if (*compressed_length < snappy_max_compressed_length(input_length))
        return SNAPPY_BUFFER_TOO_SMALL;
    size_t written = 0;
    size_t dev_out = snapmi_max_compress_len(input_length);
    if (dev_out == 0 || dev_out > *compressed_length)
        dev_out = *compressed_length;
    const int rc = seam_call(true, (const uint8_t *)input, input_length,
                             (uint8_t *)compressed, *compressed_length,
                             dev_out, &written, "snappy_compress");
    if (rc == SNAPMI_BUFFER_TOO_SMALL)
        return SNAPPY_BUFFER_TOO_SMALL;
    // (snappy_status has no "device" value: such a failure is printed and
    // reported as the one status a caller cannot mistake for success)
    if (rc != SNAPMI_OK)
        return SNAPPY_INVALID_INPUT;
    *compressed_length = written;
    return SNAPPY_OK;
}

snappy_status snappy_uncompress(const char *compressed,
                                size_t compressed_length, char *uncompressed,
                                size_t *uncompressed_length)
{
    if (!uncompressed_length)
        return SNAPPY_INVALID_INPUT;
    size_t need = 0;
    if (snappy_uncompressed_length(compressed, compressed_length, &need) !=
        SNAPPY_OK)
        return SNAPPY_INVALID_INPUT;
    if (*uncompressed_length < need)
        return SNAPPY_BUFFER_TOO_SMALL;
    size_t written = 0;
    const int rc = seam_call(false, (const uint8_t *)compressed,
                             compressed_length, (uint8_t *)uncompressed,
                             *uncompressed_length, need, &written,
                             "snappy_uncompress");
    if (rc != SNAPMI_OK)
        return SNAPPY_INVALID_INPUT;
    *uncompressed_length = written;
    return SNAPPY_OK;
}

snappy_status snappy_validate_compressed_buffer(const char *compressed,
                                                size_t compressed_length)
{
    size_t need = 0;
    if (snappy_uncompressed_length(compressed, compressed_length, &need) !=
        SNAPPY_OK)
        return SNAPPY_INVALID_INPUT;
    std::vector<char> tmp(need ? need : 1);
    size_t n = need;
    return snappy_uncompress(compressed, compressed_length, tmp.data(), &n);
}

} // extern "C"
