// snapmi_decompress.hip -- Snappy raw stream decompressor for gfx950 (CDNA4).
//
// Semantics (element order, every bounds check, which error wins and with
// which field values) follow the reference src/decompress.rs exactly; the
// execution model is CDNA4's:
//
//   * one wavefront per raw stream (a raw stream has no block index --
//     reference src/compress.rs:128-153 -- so the stream is the parallel
//     unit; the batch supplies thousands of them and the dispatcher balances
//     them dynamically: one 64-thread workgroup per stream);
//   * the tag stream is parsed out of a 256-byte register window of the
//     compressed bytes (v_readlane + scalar shifts), so walking from one
//     element to the next costs no memory round trip;
//   * the decoder state (s, d, lengths, offsets) is wave-uniform in SGPRs;
//     the 64 lanes move the bytes: literals 256 B per instruction, copies
//     (len <= 64) one byte per lane, overlapping copies (offset < len) by
//     replicating the pattern with a per-lane modulo;
//   * the output stays in HBM/L2 (no LDS), which keeps 32 waves per CU
//     resident; a back-reference that may read bytes this wave stored since
//     its last drain first waits for those stores (workgroup-scope fence =
//     s_waitcnt vmcnt(0) on gfx950), tracked with one watermark.
#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"

namespace snapmi {

namespace {

// reference bytes::read_varu64, src/bytes.rs:73-90 (returns header length,
// 0 = invalid).  Executed redundantly by every lane on uniform data.
__device__ __forceinline__ uint32_t read_varint(const uint8_t *p, uint64_t n,
                                                uint64_t *value)
{
    uint64_t acc = 0;
    uint32_t shift = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t b = p[i];
        if (shift >= 64)
            return 0;
        if (b < 0x80) {
            *value = acc | ((uint64_t)b << shift);
            return (uint32_t)i + 1;
        }
        acc |= (uint64_t)(b & 0x7F) << shift;
        shift += 7;
    }
    return 0;
}

// Header::read + the checks of Decoder::decompress, reference
// src/decompress.rs:75-95,362-374.  Returns kind; on success fills hdr/dlen.
__device__ __forceinline__ int read_header(const uint8_t *in, uint64_t in_len,
                                           uint32_t *hdr, uint64_t *dlen,
                                           snapmi_error *errs, uint64_t i)
{
    uint64_t v = 0;
    const uint32_t h = read_varint(in, in_len, &v);
    if (h == 0) {
        set_error(errs, i, SNAPMI_HEADER, 0, 0, 0);
        return SNAPMI_HEADER;
    }
    if (v > kMaxInput) {
        set_error(errs, i, SNAPMI_TOO_BIG, v, kMaxInput, 0);
        return SNAPMI_TOO_BIG;
    }
    *hdr = h;
    *dlen = v;
    return SNAPMI_OK;
}

} // namespace

// decompress_len, reference src/decompress.rs:30-35: one thread per stream.
__global__ __launch_bounds__(256) void k_decompress_len(DecompressArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_streams)
        return;
    const uint64_t in_len = a.in_lens[i];
    a.out_lens[i] = 0;
    if (in_len == 0) {
        set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
        return;
    }
    uint32_t hdr;
    uint64_t dlen;
    if (read_header((const uint8_t *)a.in_ptrs[i], in_len, &hdr, &dlen,
                    a.errs, i) != SNAPMI_OK)
        return;
    a.out_lens[i] = dlen;
    set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
}

#define SNAPMI_FAIL(kind, fa, fb, fc)                                         \
    do {                                                                      \
        if (lane == 0) {                                                      \
            set_error(a.errs, st, (kind), (fa), (fb), (fc));                  \
            a.out_lens[st] = 0;                                               \
        }                                                                     \
        return;                                                               \
    } while (0)

// ---------------------------------------------------------------------
// K2: one wavefront per raw stream.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_decompress_streams(DecompressArgs a)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t st = blockIdx.x;
    const uint8_t *in = (const uint8_t *)a.in_ptrs[st];
    const uint64_t in_len = a.in_lens[st];

    // reference Decoder::decompress, src/decompress.rs:75-95
    if (in_len == 0)
        SNAPMI_FAIL(SNAPMI_EMPTY, 0, 0, 0);
    uint32_t hdr = 0;
    uint64_t dst_len = 0;
    {
        snapmi_error *e = lane == 0 ? a.errs : nullptr;
        if (read_header(in, in_len, &hdr, &dst_len, e, st) != SNAPMI_OK) {
            if (lane == 0)
                a.out_lens[st] = 0;
            return;
        }
    }
    const uint64_t cap = a.out_caps[st];
    if (dst_len > cap)
        SNAPMI_FAIL(SNAPMI_BUFFER_TOO_SMALL, cap, dst_len, 0);

    const uint8_t *src = in + hdr;
    const uint64_t src_len = in_len - hdr;
    uint8_t *dst = (uint8_t *)a.out_ptrs[st];

    ByteWindow win;
    win.init(src, src_len);

    uint64_t s = 0;      // position in src
    uint64_t d = 0;      // position in dst
    uint64_t pend = 0;   // dst[pend..d) may still be in flight (stores)

    // reference Decompress::decompress, src/decompress.rs:130-148
    while (s < src_len) {
        const uint32_t w = win.get32(s); // tag + the 3 bytes after it
        const uint32_t tag = w & 0xFF;
        s += 1;
        if ((tag & 3) == 0) {
            // reference read_literal, src/decompress.rs:161-228
            uint64_t len = (tag >> 2) + 1;
            if (len >= 61) {
                if (s + 4 > src_len)
                    SNAPMI_FAIL(SNAPMI_LITERAL, 4, src_len - s, dst_len - d);
                const uint32_t nb = (uint32_t)len - 60;
                const uint32_t raw = win.get32(s);
                len = (uint64_t)(nb == 4 ? raw
                                         : raw & ((1u << (8 * nb)) - 1)) +
                      1;
                s += nb;
            }
            if (src_len - s < len || dst_len - d < len)
                SNAPMI_FAIL(SNAPMI_LITERAL, len, src_len - s, dst_len - d);
            const uint8_t *from = src + s;
            uint8_t *to = dst + d;
            for (uint64_t i = 4 * lane; i + 4 <= len; i += 4 * kWave)
                st32u(to + i, ld32u(from + i));
            const uint64_t t = len & ~3ull;
            if (lane < (len & 3))
                to[t + lane] = from[t + lane];
            s += len;
            d += len;
        } else {
            // reference read_copy + TagEntry::offset,
            // src/decompress.rs:233-343,433-474
            const uint32_t kind = tag & 3;
            const uint32_t nb = kind == 1 ? 1 : (kind == 2 ? 2 : 4);
            const uint32_t len =
                kind == 1 ? 4 + ((tag >> 2) & 7) : 1 + (tag >> 2);
            uint64_t offset = kind == 1 ? (uint64_t)(tag >> 5) << 8 : 0;
            if (s + 4 <= src_len) {
                if (nb == 1)
                    offset |= (w >> 8) & 0xFF;
                else if (nb == 2)
                    offset |= (w >> 8) & 0xFFFF;
                else
                    offset |= win.get32(s);
            } else if (nb == 1) {
                if (s >= src_len)
                    SNAPMI_FAIL(SNAPMI_COPY_READ, 1, src_len - s, 0);
                offset |= (w >> 8) & 0xFF;
            } else if (nb == 2) {
                if (s + 1 >= src_len)
                    SNAPMI_FAIL(SNAPMI_COPY_READ, 2, src_len - s, 0);
                offset |= (w >> 8) & 0xFFFF;
            } else {
                SNAPMI_FAIL(SNAPMI_COPY_READ, 4, src_len - s, 0);
            }
            s += nb;
            if (d <= offset - 1) // wrapping, also catches offset == 0
                SNAPMI_FAIL(SNAPMI_OFFSET, offset, d, 0);
            const uint64_t end = d + len;
            if (end > dst_len)
                SNAPMI_FAIL(SNAPMI_COPY_WRITE, len, dst_len - d, 0);

            // Source bytes are dst[d-offset .. min(d, d-offset+len)).  If
            // any of them may still be an in-flight store of this wave,
            // drain the stores first.
            const uint64_t from0 = d - offset;
            const uint64_t src_end = offset < len ? d : from0 + len;
            if (src_end > pend) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                pend = d;
            }
            if (lane < len) {
                uint32_t i = lane;
                if (offset < len) {
                    // pattern replication: i mod offset, exact for i,offset<64
                    const uint32_t o = (uint32_t)offset;
                    const uint32_t q =
                        (uint32_t)(((float)lane + 0.5f) *
                                   __builtin_amdgcn_rcpf((float)o));
                    i = lane - q * o;
                }
                dst[d + lane] = dst[from0 + i];
            }
            d = end;
        }
    }
    if (d != dst_len)
        SNAPMI_FAIL(SNAPMI_HEADER_MISMATCH, dst_len, d, 0);
    if (lane == 0) {
        set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
        a.out_lens[st] = dst_len;
    }
}

} // namespace snapmi
