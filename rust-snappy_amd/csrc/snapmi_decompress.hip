// snapmi_decompress.hip -- Snappy raw stream decompressor for gfx950 (CDNA4).
//
// Semantics (element order, every bounds check, which error wins and with
// which field values) follow the reference src/decompress.rs exactly.  The
// reference walks one element at a time (tag dispatch loop, :130-148).  A raw
// stream has no block index (src/compress.rs:128-153), so the stream is the
// parallel unit; a batch supplies thousands of them, sorted longest first
// (k_plan_decompress*), and every size class has its kernel:
//
//   k_decompress_streams3 (+ _many)   one wavefront per stream, the default.
//       decode_windows3 (third generation): 256 compressed bytes per window;
//       the lanes look at tag bytes only to find the element starts (four
//       rounds of pointer jumping per 64-byte group), the starts are
//       compacted so that lane t holds the t-th ELEMENT of the window, which
//       is then decoded, checked and placed (DPP scan) per lane; elements
//       whose source is complete are copied in one lane-parallel step of
//       whole 16-byte pieces (lane order of one DS instruction resolves the
//       overlaps), far sources requested for the whole window at once; the
//       elements that read the window's own output follow one by one, by
//       the whole wave.  A 4 KiB ring of recent output lives in LDS, the
//       ring goes to HBM 256 bytes at a time.  The last 337 bytes of a
//       stream are decode_windows2's (second generation: 64-byte windows, an
//       element per lane that sits on its first byte), which is also
//       k_decompress_streams2, the cross-check of the test build.
//   k_decompress_tiny / _small   streams of under 256 / 512 compressed bytes
//       whose output is no larger: one per LANE (64 / 32 per wavefront),
//       input and output staged in LDS, the reference's loop per lane.
//   k_decompress_sequential   the reference's loop, one element at a time:
//       what names every error (any failed check of the wide paths hands the
//       stream over), and the fallback of a device that fails the LDS
//       store-order self-check.
//   k_stream_* / k_long_plan + k_bstream_*   ONE long stream on many
//       wavefronts: (exit, produced) per segment and entry offset by walks
//       that hop through LDS (scan), two levels above, one short sequential
//       pass, the element boundaries at every 64 KiB of output (cuts), and
//       the pieces between them through k_decompress_streams3; the same for
//       the long streams of a small batch.
//
// DESIGN.md section 4.2 has the history (byte-per-lane kernel of round 1,
// gone; 354 -> 16 ms at cfg2) and what bounds each of them.
#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"

namespace snapmi {

namespace {


// reference bytes::read_varu64, src/bytes.rs:73-90 (returns header length,
// 0 = invalid).  Executed redundantly by every lane on uniform data.
__device__ __forceinline__ uint32_t read_varint(gcptr p, uint64_t n,
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
__device__ __forceinline__ int read_header(gcptr in, uint64_t in_len,
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

#define SNAPMI_FAIL(kind, fa, fb, fc)                                         \
    do {                                                                      \
        if (lane == 0) {                                                      \
            set_error(a.errs, st, (kind), (fa), (fb), (fc));                  \
            a.out_lens[st] = 0;                                               \
        }                                                                     \
        return;                                                               \
    } while (0)

// The reference's element-at-a-time loop (src/decompress.rs:130-343), state
// in SGPRs, bytes moved by the lanes.  Used to finish a stream once the wide
// path has met something irregular; resumes at (s, d).
__device__ __forceinline__ void decode_sequential(const DecompressArgs &a,
                                               uint64_t st, uint32_t lane,
                                               gcptr src,
                                               uint64_t src_len, gptr dst,
                                               uint64_t dst_len, uint64_t s,
                                               uint64_t d)
{
    ByteWindow win;
    win.init(src, src_len);
    uint64_t pend = 0; // dst[pend..d) may still be in flight (stores)

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
            gcptr from = src + s;
            gptr to = dst + d;
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

// wave64 DPP helpers (row_shr inside rows of 16, row_bcast across rows)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_get(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK,
                                                 0xF, false);
}
// Inclusive prefix sum over the wave: six v_add_u32_dpp.  (Through the
// update_dpp builtin every step is a v_mov_dpp plus an add - the compiler
// does not fold them - and the decoders' window loop is bound by VALU issue.)
// A DPP operand written by the instruction in front needs two wait states;
// the compiler cannot see into the asm, so the first pad also covers the
// worst case in front of it (a VALU write of EXEC: five wait states).
__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t v)
{
    asm volatile(
        "s_nop 4\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// OR of v over all lanes (uniform result)
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= dpp_get<0x111, 0xF>(v);
    v |= dpp_get<0x112, 0xF>(v);
    v |= dpp_get<0x114, 0xF>(v);
    v |= dpp_get<0x118, 0xF>(v);
    v |= dpp_get<0x142, 0xA>(v);
    v |= dpp_get<0x143, 0xC>(v);
    return rdlane(v, 63);
}

__device__ __forceinline__ uint32_t popc_below(uint64_t mask)
{
    // number of set bits of `mask` strictly below this lane
    return __builtin_amdgcn_mbcnt_hi(
        (uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// 8 bytes at src[pos..] with bytes at or past `avail` read as zero
__device__ __forceinline__ uint64_t ld64g(gcptr src, uint64_t pos,
                                          uint64_t avail)
{
    if (pos + 8 <= avail) {
        uint64_t v;
        __builtin_memcpy(&v, src + pos, 8);
        return v;
    }
    uint64_t v = 0;
    for (uint32_t k = 0; k < 7; k++)
        if (pos + k < avail)
            v |= (uint64_t)src[pos + k] << (8 * k);
    return v;
}

// 16 bytes as two qwords
struct B16x {
    uint64_t lo, hi;
};


// 8 bytes at src[pos..] for a stream of avail >= 8 readable bytes; bytes at
// or past `avail` read as zero.  Branch-free: the load address is clamped
// into the stream and the value shifted back into place.
__device__ __forceinline__ uint64_t ld64c(gcptr src, uint64_t pos,
                                          uint64_t avail)
{
    const uint64_t pc = pos < avail - 8 ? pos : avail - 8;
    uint64_t v;
    __builtin_memcpy(&v, src + pc, 8);
    const uint64_t sh = pos - pc; // 0 inside the stream
    return sh < 8 ? v >> (8 * sh) : 0;
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
    if (read_header((gcptr)a.in_ptrs[i], in_len, &hdr, &dlen,
                    a.errs, i) != SNAPMI_OK)
        return;
    a.out_lens[i] = dlen;
    set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
}

// Streams of fewer than kTiny = 2^kTinyLog2 compressed bytes are decoded one
// per LANE (k_decompress_tiny below), streams of under kSmallDec bytes whose
// output is no larger on half the lanes of a wavefront (k_decompress_small,
// round 5); the sort puts both classes last, and the plan leaves the number
// of streams in front of them in bucket_pos[64] and [65].
constexpr uint32_t kTinyLog2 = 8;
constexpr uint32_t kSmallDecLog2 = 9;
constexpr uint32_t kSmallDec = 1u << kSmallDecLog2;
// sort classes, descending in the dispatch order: kFirstWide + floor(log2
// (compressed length)) - 8 for the wavefront decoder, kSmallClass, then
// floor(log2(length)) of the tiny ones
constexpr uint32_t kSmallClass = kTinyLog2;
constexpr uint32_t kFirstWide = kTinyLog2 + 1;

// The size class a stream is sorted by: floor(log2(compressed length)) -
// except that a raw stream of under kTiny bytes whose header promises MORE
// than kTiny bytes of output (a run of zeros: 200 bytes of copies are 4 KiB)
// (or a piece of a long stream with as much) is not the lane-per-stream
// kernel's: it keeps its output in LDS, kTiny bytes per lane; what does not
// fit would be left to one lane moving bytes through global memory.  Such a
// stream, and any of 256 .. 511 compressed bytes, whose output is at most
// kSmallDec bytes is k_decompress_small's; the rest goes with the first class
// of the wavefront decoder.
__device__ __forceinline__ uint32_t plan_class(const DecompressArgs &a,
                                               uint32_t i)
{
    const uint64_t len = a.in_lens[i];
    if (a.modes && a.modes[i] == 3) // not this launch's: behind everything
        return 0;
    const uint32_t bk = len ? 63 - (uint32_t)__builtin_clzll(len) : 0;
    if (len == 0 || bk > kSmallDecLog2 - 1)
        return len ? (bk < 62 ? bk + 1 : 63) : 0;
    const uint32_t mode = a.modes ? a.modes[i] : 0;
    uint64_t dl = 0;
    bool header = true;
    if (mode == 2) // a piece of a long stream: its output is out_caps
        dl = a.out_caps[i];
    else if (mode == 0 && read_varint((gcptr)a.in_ptrs[i], len, &dl) == 0)
        header = false; // the lane-per-stream kernel reports it
    if (bk < kTinyLog2 && (!header || dl <= (1u << kTinyLog2)))
        return bk;
    if (mode == 0 && header && dl <= kSmallDec)
        return kSmallClass;
    return kFirstWide;
}

// ---------------------------------------------------------------------
// Plan: dispatch order.  Streams differ in size by orders of magnitude and a
// stream is decoded by one wavefront, so the longest streams must start
// first or they become the tail of the launch.  One workgroup sorts the
// stream indices by floor(log2(compressed length)), largest bucket first
// (counting sort: histogram, bucket offsets, scatter).
// ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_plan_decompress(DecompressArgs a)
{
    __shared__ uint32_t hist[64];
    if (a.gate && *a.gate != a.gate_value)
        return;
    if (threadIdx.x < 64)
        hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_streams; i += blockDim.x) {
        const uint32_t bk = plan_class(a, i);
        atomicAdd(&hist[63 - bk], 1u); // reversed: big buckets first
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 64; k++) {
            if (k == 64 - kFirstWide) // the wavefront decoder's streams
                a.bucket_pos[64] = run;
            if (k == 64 - kSmallClass) // ... and k_decompress_small's
                a.bucket_pos[65] = run;
            const uint32_t c = hist[k];
            hist[k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_streams; i += blockDim.x) {
        const uint32_t bk = plan_class(a, i);
        a.order[atomicAdd(&hist[63 - bk], 1u)] = i;
    }
}

// The same sort on many workgroups (10.7 M streams: 19 ms of a 33 ms pass in
// the one-workgroup kernel): histogram, offsets, scatter.  bucket_pos[0, 64)
// = counts, then cursors.
__global__ __launch_bounds__(1024) void k_plan_decompress_a(DecompressArgs a)
{
    __shared__ uint32_t hist[64];
    if (a.gate && *a.gate != a.gate_value)
        return;
    if (threadIdx.x < 64)
        hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_streams) {
        const uint32_t bk = plan_class(a, i);
        atomicAdd(&hist[63 - bk], 1u); // reversed: big buckets first
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x])
        atomicAdd(&a.bucket_pos[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_plan_decompress_b(DecompressArgs a)
{
    if (a.gate && *a.gate != a.gate_value)
        return;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 64; k++) {
            if (k == 64 - kFirstWide)
                a.bucket_pos[64] = run;
            if (k == 64 - kSmallClass)
                a.bucket_pos[65] = run;
            const uint32_t c = a.bucket_pos[k];
            a.bucket_pos[k] = run;
            run += c;
        }
    }
}

__global__ __launch_bounds__(1024) void k_plan_decompress_c(DecompressArgs a)
{
    __shared__ uint32_t hist[64], base[64];
    if (a.gate && *a.gate != a.gate_value)
        return;
    if (threadIdx.x < 64)
        hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bk = 0, mine = 0;
    if (i < a.n_streams) {
        bk = 63 - plan_class(a, i);
        mine = atomicAdd(&hist[bk], 1u); // my place among this workgroup's
    }
    __syncthreads();
    // one range per bucket and workgroup
    if (threadIdx.x < 64 && hist[threadIdx.x])
        base[threadIdx.x] = atomicAdd(&a.bucket_pos[threadIdx.x],
                                      hist[threadIdx.x]);
    __syncthreads();
    if (i < a.n_streams)
        a.order[base[bk] + mine] = i;
}

// ---------------------------------------------------------------------
// One wavefront per raw stream: the wide path.
//
// Two window loops share the stream prologue, a 4 KiB ring of recent output
// in LDS and the hand-over to the sequential decoder:
//
// decode_windows2 (second generation; k_decompress_streams2 and the last
// bytes of every stream in k_decompress_streams3) - 64 compressed bytes per
// window, an element is copied by the lane that sits on its first byte:
//
//   1. every lane decodes the element that would start at its byte of the
//      64-byte window, and loads 16 literal bytes speculatively;
//   2. the real element starts are the orbit of lane 0 under "next element"
//      (one ds_bpermute per round, see the loop) - no compaction, the
//      records stay where they were decoded;
//   3. a DPP scan of the output lengths over the start lanes places every
//      element; the window is cut after 2048 output bytes (kWinMax), so a
//      4 KiB ring of recent output in LDS always keeps 2 KiB of history that
//      the window's own writes cannot touch;
//   4. ONE lane-parallel copy step: every element whose source is complete
//      before the window (literals; copies from in front of it) is copied by
//      its lane, 16 bytes per trip - source: the speculative literal bytes,
//      the ring, or HBM for what the ring no longer holds - and stored to
//      the ring as WHOLE 16-byte pieces, last piece first: overlapping lanes
//      of one DS store are applied in ascending lane order, so the excess
//      bytes of a short element are overwritten by the elements that follow;
//   5. the few elements that read the window's own output or repeat a short
//      period are swept in stream order, each by the whole wave (lane k =
//      byte k, k mod offset for overlaps);
//   6. the ring goes to HBM 256 bytes at a time (one dword per lane), so
//      global stores are whole aligned lines instead of 64 byte stores.
//
// decode_windows3 (third generation, the default) - 256 compressed bytes per
// window and one ELEMENT per lane; described at the function.
//
// Both loops are bound by instruction issue - VALU (a wave64 integer
// instruction holds its SIMD for four cycles) and the CU's one scalar unit
// about equally, the LDS pipeline third (tests/hw/lds_cost.hip) - so they are
// written for few instructions: predicates are 64-bit lane masks in SGPRs,
// combined with scalar ALU operations and handed back to the vector side as
// they are (inverse ballot); positions are 32-bit (no scalar 64-bit compare
// exists); tests that only matter near the end of the input sit behind one
// uniform branch; addresses are a uniform base plus a 32-bit lane offset.
//
// Everything irregular - a failed check, an element cut off by the end of
// the input - stops the wide path at a window boundary: the ring is stored
// and the sequential decoder above finishes the stream from (s, d) with the
// reference's exact error.  tests/model_decoder.py and tests/model_decoder3.py
// restate the two loops lane by lane on the CPU (test infrastructure, not
// used here).
// ---------------------------------------------------------------------
namespace {
#define SNAPMI_RING 4096 // experiment builds: 8192 / 16384 (make ring_variants)
constexpr uint32_t kRing2 = SNAPMI_RING;
constexpr uint32_t kWinMax = 2048;
typedef __attribute__((address_space(3))) uint8_t l_u8;
typedef __attribute__((address_space(3))) uint16_t l_u16x;
typedef __attribute__((address_space(3))) uint32_t l_u32;
typedef __attribute__((address_space(3))) uint64_t l_u64;

__device__ __forceinline__ uint32_t lds_ld32(const l_u8 *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t lds_ld64(const l_u8 *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void lds_st64(l_u8 *p, uint64_t v)
{
    __builtin_memcpy(p, &v, 8);
}
__device__ __forceinline__ void lds_st32(l_u8 *p, uint32_t v)
{
    __builtin_memcpy(p, &v, 4);
}
__device__ __forceinline__ void lds_st16(l_u8 *p, uint16_t v)
{
    __builtin_memcpy(p, &v, 2);
}

struct Ring2 {
    l_u8 *rg;       // kRing2 bytes + a 16-byte mirror of its first 16
    gptr dst;
    uint32_t lane;
    uint32_t gflush; // dst[0, gflush) has been stored
    uint32_t fenced; // ... and those stores are known to be complete

    // ring[0,16) again behind the end: unaligned reads may run past it
    __device__ __forceinline__ void mirror() const
    {
        if (lane < 4)
            lds_st32(rg + kRing2 + 4 * lane, lds_ld32(rg + 4 * lane));
    }
    // whole 256-byte pieces of dst[gflush, d), one dword per lane
    __device__ __forceinline__ void flush_chunks(uint32_t d)
    {
        while (gflush + 256 <= d) {
            const uint32_t p = gflush + 4 * lane;
            st32u(dst + p, lds_ld32(rg + (p & (kRing2 - 1))));
            gflush += 256;
        }
    }
    // everything up to `upto`, bytewise (before a long literal, at the end,
    // when the sequential decoder takes over)
    __device__ __forceinline__ void flush_partial(uint32_t upto)
    {
        for (uint32_t p = gflush + lane; p < upto; p += kWave)
            dst[p] = rg[p & (kRing2 - 1)];
        gflush = upto;
    }
    // a far source must be a completed store
    __device__ __forceinline__ void fence_for(uint32_t limit)
    {
        if (limit > fenced) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            fenced = gflush;
        }
    }
};
} // namespace

namespace {
#define SNAPMI_FAIL_V(ret, kind, fa, fb, fc)                                  \
    do {                                                                      \
        if (lane == 0) {                                                      \
            set_error(a.errs, st, (kind), (fa), (fb), (fc));                  \
            a.out_lens[st] = 0;                                               \
        }                                                                     \
        return ret;                                                           \
    } while (0)

// The wide path's state: everything uniform (SGPRs).
struct Wide {
    gcptr src;        // elements (behind the varint header)
    gptr dst;
    uint64_t src_len, dst_len;
    uint64_t st;      // stream index
    // Positions are 32-bit: there is no scalar 64-bit compare, each one in
    // a window loop would be two VALU instructions.  dst_len < 2^32 by the
    // format; a compressed stream of 4 GiB or more (legal, if every element
    // is tiny) is left to the sequential decoder as a whole, and so is an
    // output within 4 KiB of 2^32 (window arithmetic like d + W + 16 must
    // not wrap).
    uint32_t slen, dlen;
    uint32_t s, d;    // positions in src / dst
    uint32_t ring_lo; // the ring holds dst[max(ring_lo, d - 4096), d)
    Ring2 R;
    PROF(
    uint64_t n_win = 0, n_elem = 0, n_dep = 0, n_fence = 0, n_far = 0,
             n_trip = 0, n_run = 0, n_win3 = 0;
    )
    __device__ __forceinline__ bool too_big() const
    {
        return src_len > 0xFFE00000ull || dst_len > 0xFFFFF000ull;
    }
};

// Stored chunk, header, capacity (reference Decoder::decompress,
// src/decompress.rs:75-95).  False: the stream is finished (copied, or its
// error is written).
__device__ __forceinline__ bool open_stream(const DecompressArgs &a,
                                            const uint32_t lane, Wide &x,
                                            l_u8 *ring_mem,
                                            const uint32_t slot)
{
    const uint64_t st = a.order[slot]; // slot: position in the sorted order
    gcptr in = (gcptr)a.in_ptrs[st];
    const uint64_t in_len = a.in_lens[st];
    const bool piece = a.modes && a.modes[st] == 2;

    if (a.modes && a.modes[st] == 3)
        return false; // another launch decodes this stream
    if (a.modes && a.modes[st] == 1) {
        // stored frame chunk (reference src/read.rs:173-199): the payload is
        // the data (wave_copy)
        const uint64_t cap0 = a.out_caps[st];
        if (in_len > cap0)
            SNAPMI_FAIL_V(false, SNAPMI_BUFFER_TOO_SMALL, cap0, in_len, 0);
        wave_copy<false>((gptr)a.out_ptrs[st], in, in_len, lane);
        if (lane == 0) {
            set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
            a.out_lens[st] = in_len;
        }
        return false;
    }
    uint32_t hdr = 0;
    uint64_t dst_len = 0;
    if (piece) { // elements [in, in + in_len) produce exactly out_caps bytes
        dst_len = a.out_caps[st];
    } else {
        if (in_len == 0)
            SNAPMI_FAIL_V(false, SNAPMI_EMPTY, 0, 0, 0);
        snapmi_error *e = lane == 0 ? a.errs : nullptr;
        if (read_header(in, in_len, &hdr, &dst_len, e, st) != SNAPMI_OK) {
            if (lane == 0)
                a.out_lens[st] = 0;
            return false;
        }
    }
    // the header came through vector loads: pin it (and everything derived
    // from it) to SGPRs so the decoder state is scalar
    hdr = uni(hdr);
    dst_len = uni64(dst_len);
    const uint64_t cap = a.out_caps[st];
    if (dst_len > cap)
        SNAPMI_FAIL_V(false, SNAPMI_BUFFER_TOO_SMALL, cap, dst_len, 0);
    x.st = st;
    x.src = in + hdr;
    x.src_len = in_len - hdr;
    x.dst = (gptr)a.out_ptrs[st];
    x.dst_len = dst_len;
    x.slen = (uint32_t)x.src_len;
    x.dlen = (uint32_t)dst_len;
    x.s = x.d = x.ring_lo = 0;
    x.R.rg = ring_mem;
    x.R.dst = x.dst;
    x.R.lane = lane;
    x.R.gflush = 0;
    x.R.fenced = 0;
    return true;
}

// The end of a stream: what the wide path has in the ring is stored; an
// irregular stream is finished (and its error named) by the sequential
// decoder; HeaderMismatch, reference src/decompress.rs:149-157.
__device__ __forceinline__ void close_stream(const DecompressArgs &a,
                                             const uint32_t lane, Wide &x,
                                             const bool irregular)
{
    const uint64_t st = x.st;
    x.R.flush_partial(x.d);
    if (irregular) {
        decode_sequential(a, st, lane, x.src, x.src_len, x.dst, x.dst_len,
                          x.s, x.d);
        return;
    }
    PROF(
    if (lane == 0 && a.prof) {
        atomicAdd(&a.prof[7], (unsigned long long)x.n_win3);
        atomicAdd(&a.prof[8], (unsigned long long)x.n_run);
        atomicAdd(&a.prof[9], (unsigned long long)x.n_far);
        atomicAdd(&a.prof[10], (unsigned long long)x.n_win);
        atomicAdd(&a.prof[11], (unsigned long long)x.n_trip);
        atomicAdd(&a.prof[12], (unsigned long long)x.n_elem);
        atomicAdd(&a.prof[13], (unsigned long long)x.n_fence);
        atomicAdd(&a.prof[14], (unsigned long long)x.n_dep);
        atomicAdd(&a.prof[15], 1ull);
    }
    )
    if (x.d != x.dst_len)
        SNAPMI_FAIL(SNAPMI_HEADER_MISMATCH, x.dst_len, x.d, 0);
    if (lane == 0) {
        set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
        a.out_lens[st] = x.dst_len;
    }
}

// The second-generation window loop from (x.s, x.d) to the end of the
// input.  True: something irregular, the sequential decoder takes over.
__device__ __forceinline__ bool decode_windows2(Wide &x, const uint32_t lane)
{
    gcptr src = x.src;
    gptr dst = x.dst;
    const uint64_t src_len = x.src_len;
    const uint32_t slen = x.slen, dlen = x.dlen;
    uint32_t s = x.s, d = x.d, ring_lo = x.ring_lo;
    Ring2 &R = x.R;
    l_u8 *const rg = R.rg;
    // streams too short for the 8-byte window loads go to the sequential
    // decoder at once; so does the first failed check
    bool irregular = src_len < 8;
    uint64_t w = irregular || s >= slen
                     ? 0
                     : ld64c(src, (uint64_t)s + lane, src_len); // src[s+lane..]
    while (!irregular && s < slen) {
        COUNT(x.n_win);
        // bytes of input left, as far as this window can see (<= 2^20)
        const uint32_t rem = slen - s < (1u << 20) ? slen - s : 1u << 20;
        // ---- 1. the element that would start at src[s + lane] ------------
        const uint32_t tag = (uint32_t)w & 0xFF;
        const uint32_t b14 = (uint32_t)(w >> 8); // the 4 bytes after the tag
        const uint32_t type = tag & 3, n6 = tag >> 2;
        const bool is_lit = type == 0;
        // The bytes behind the tag hold a little-endian number of nb bytes:
        // the rest of a literal's length (nb = 0..4, reference read_literal,
        // src/decompress.rs:161-228) or a copy's offset (nb = 1, 2, 4,
        // TagEntry::offset / read_copy, :233-250,433-474) - one extraction
        // for both (this loop is bound by VALU issue: every instruction that
        // goes is 0.4 % of the kernel)
        const uint32_t lnb = (n6 > 59 ? n6 : 59) - 59; // extra length bytes
        const uint32_t cnb = type + (type == 3);        // 1, 2, 4
        const uint32_t sh = 32 - 8 * (is_lit ? lnb : cnb);
        const uint32_t ext = (b14 << (sh & 31)) >> (sh & 31); // (nb = 0: unused)
        const uint32_t lraw = lnb ? ext : n6; // length - 1
        const uint32_t hd = 1 + lnb;
        // Predicates live as 64-bit lane masks in SGPRs and are combined
        // there: a __ballot of a compound per-lane bool costs two VALU
        // instructions on top of its compares (the bool is materialised and
        // compared again), a scalar AND of two simple ballots costs none.
        const uint64_t M_lit = __ballot(is_lit);
        // literals of more than 64 bytes: not in a window
        const uint64_t M_lng = M_lit & __ballot(lraw >= 64);
        const uint32_t clen = type == 1 ? 4 + (n6 & 7) : 1 + n6;
        const uint32_t off = ext | (type == 1 ? (tag & 0xE0u) << 3 : 0);
        const uint32_t olen = is_lit ? lraw + 1 : clen;      // if !lng
        const uint32_t enc = is_lit ? hd + lraw + 1 : 1 + cnb; // if !lng
        // the whole element lies inside the input (:189-217 src side, CopyRead)
        // (an extended literal length is read as 4 bytes: :189-198)
        // (a window with 160 bytes of input in front of it - all but the
        // last two or three of a stream - needs none of the per-lane tests:
        // lane + 5 + 64 + 16 <= rem whatever the element)
        const bool deep = rem >= 160;
        const uint64_t M_fits =
            deep ? ~M_lng
                 : ~M_lng & __ballot(lane < rem && enc <= rem - lane &&
                                     !(is_lit && lnb && lane + 5 > rem));
        // 16 literal bytes, speculatively (only windows well inside the input)
        const bool inner = rem >= 64 + 5 + 16;
        B16x lit16;
        // (uniform base + 32-bit lane offset: the load takes the base from
        // SGPRs and no 64-bit vector adds are spent on the address)
        const gcptr win_src = src + s;
        if (inner) {
            __builtin_memcpy(&lit16, win_src + (lane + hd), 16);
        } else {
            lit16.lo = lit16.hi = 0;
        }
        // ---- 2. element starts: the orbit of lane 0 under "next" ----------
        // A ds_bpermute costs the CU's one LDS pipeline about as much as
        // seven VALU instructions cost its four SIMDs (tests/hw/lds_cost.hip:
        // 6.4 cycles), so the discovery moves ONE dword per round: the wave
        // is two halves of 32 positions, and a lane's set R (32 bits, its own
        // half) holds the first nodes of its chain INCLUDING the frontier -
        // the node not yet expanded, which is the set's highest bit because
        // chains run forward.  One round: R |= R[frontier].  After k rounds a
        // set holds 2^k + 1 nodes; a half has at most 16 (every element has
        // two bytes or more), so four rounds finish it.  A lane whose element
        // ends its chain - long literal, behind the input, or reaching into
        // the other half / out of the window - has no frontier beyond itself
        // and fetches its own set.  The two halves are strung together by
        // the scalar unit afterwards.
        const uint32_t nx = lane + enc;
        const uint64_t T = M_lng | (deep ? 0 : __ballot(lane >= rem));
        const uint64_t M_stay = ~T & __ballot((nx ^ lane) < 32);
        uint32_t Rr = (1u << (lane & 31)) |
                      (__builtin_amdgcn_inverse_ballot_w64(M_stay)
                           ? 1u << (nx & 31)
                           : 0);
        const int c4 = (int)((lane | 31) << 2);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const int sel = c4 - 4 * (int)__builtin_clz(Rr);
            Rr |= (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)Rr);
        }
        uint64_t S = rdlane(Rr, 0);
        {
            // the last start of the lower half: does its element lead into
            // the upper half?
            const uint32_t t0 = 31 - (uint32_t)__builtin_clz((uint32_t)S);
            const uint32_t nx0 = rdlane(nx, t0);
            if (!((T >> t0) & 1) && nx0 < kWave)
                S |= (uint64_t)rdlane(Rr, nx0) << 32;
        }
        // ---- 3. placement, window cut, checks ------------------------------
        const uint64_t M_elem = S & M_fits;
        const uint32_t o =
            __builtin_amdgcn_inverse_ballot_w64(M_elem) ? olen : 0;
        const uint32_t incl = wave_inclusive_add(o);
        // the window ends in front of the first start that is a long literal
        // or does not fit, and after kWinMax output bytes
        const uint64_t stop = S & ~M_fits;
        const uint64_t below =
            stop ? ((1ull << __builtin_ctzll(stop)) - 1) : ~0ull;
        const uint64_t K = M_elem & __ballot(incl <= kWinMax) & below;
        if (K == 0) {
            // lane 0 is a literal of more than 64 bytes: wave_copy
            const uint32_t lng0 = (uint32_t)M_lng & 1u;
            const uint64_t Lq = (uint64_t)rdlane(lraw, 0) + 1;
            const uint32_t h0 = rdlane(hd, 0);
            if (!lng0 || rem < h0 || slen - (s + h0) < Lq || dlen - d < Lq) {
                irregular = true; // the sequential decoder names the error
                break;
            }
            R.flush_partial(d);
            wave_copy<false>(dst + d, src + s + h0, Lq, lane);
            s += h0 + (uint32_t)Lq;
            d += (uint32_t)Lq;
            R.gflush = d;
            ring_lo = d; // these bytes are not in the ring
            if (s < slen)
                w = ld64c(src, (uint64_t)s + lane, src_len);
            continue;
        }
        const uint32_t last = 63 - (uint32_t)__builtin_clzll(K);
        const uint32_t W = rdlane(incl, last);         // output of the window
        const uint32_t cur = last + rdlane(enc, last); // input consumed
        const uint32_t dstp = d + (incl - o);          // element's position
        // reference checks :209-217 (dst side), :245-250, :327-332
        const uint64_t M_cpy = K & ~M_lit;
        if (W > dlen - d ||
            (M_cpy & (__ballot(off == 0) | __ballot(off > dstp))) != 0) {
            irregular = true;
            break;
        }
        PROF(
        x.n_elem += __builtin_popcountll(K);
        )
        // next window's bytes: issued now, consumed after the expand
        // (plain unaligned loads while the next window lies inside the
        // input - a uniform test; the clamped form only at the stream's end)
        uint64_t w_next;
        if (slen - s >= cur + kWave + 8)
            __builtin_memcpy(&w_next, src + (s + cur) + lane, 8);
        else
            w_next = ld64c(src, (uint64_t)s + cur + lane, src_len);

        // ---- 4. the lane-parallel copy step --------------------------------
        const uint32_t q = dstp - off;               // copy source (if cpy)
        // (a copy that a lane moves by itself does not overlap its source:
        // olen <= off, it reads olen bytes)
        const uint32_t qe = q + olen;
        // (whole 16-byte pieces are stored: up to 15 bytes behind the window's
        // output may be clobbered too)
        const uint32_t dW = d + W + 16;
        uint32_t safe_lo = dW > kRing2 ? dW - kRing2 : 0;
        safe_lo = safe_lo > ring_lo ? safe_lo : ring_lo;
        const uint64_t M_ring = __ballot(q >= safe_lo);
        // 16-byte loads from HBM must stay inside the buffers: an element
        // that ends within 15 bytes of the input's / the output's end is
        // left to the sweep, which moves exactly its bytes
        // (for a literal near the end of the input the exact figure, for a
        // copy from HBM 64: elements of a window have at most 64 bytes)
        // lanes that copy their element themselves: literals (with their 16
        // speculative bytes inside the input), copies whose whole source lies
        // in front of the window, in the ring's safe part or stored already
        const uint64_t M_litok =
            deep ? M_lit
                 : (inner ? M_lit & __ballot(lane + hd + ((olen + 15) & ~15u) <=
                                             rem)
                          : 0);
        const uint64_t M_src = __ballot(qe <= d) & __ballot(olen <= off);
        const uint64_t M_farok =
            dlen >= 64 ? __ballot(qe <= R.gflush) & __ballot(q <= dlen - 64)
                       : 0;
        const uint64_t M_lw =
            K &
            (M_litok | (~M_lit & M_src & (M_ring | M_farok)));
        const uint64_t M_far = M_lw & ~M_lit & ~M_ring; // source in HBM
        const uint64_t M_rng = M_lw & ~M_lit & M_ring;  // source in the ring
        {
            if (M_far) {
                PROF(
                x.n_far += __builtin_popcountll(M_far);
                )
                if ((M_far & __ballot(qe > R.fenced)) != 0) {
                    R.fence_for(0xFFFFFFFFu);
                    COUNT(x.n_fence);
                }
            }
            // Every lane stores WHOLE 16-byte pieces with one ds_write_b128:
            // the bytes past an element's end land on the following elements,
            // whose lanes are higher and whose stores of the same instruction
            // therefore win (overlapping lanes of one DS store are applied in
            // ascending lane order: tests/hw/lds_write_order.hip, checked at
            // context creation).  The pieces go last to first, so the excess
            // of an element's last piece is repaired by the first pieces of
            // its successors, which are all written by the last trip.
            const uint64_t M_g16 = M_lw & __ballot(olen > 16);
            const uint32_t top =
                M_g16 == 0 ? 0
                           : ((M_g16 & __ballot(olen > 48))
                                  ? 48
                                  : ((M_g16 & __ballot(olen > 32)) ? 32 : 16));
            // an element's own bytes wrap around the ring's end only in a
            // window whose output does (uniform)
            const bool wraps = (d & (kRing2 - 1)) + W > kRing2;
            for (uint32_t c = top;; c -= 16) {
                const uint64_t M_act = c == 0 ? M_lw : M_lw & __ballot(c < olen);
                if (M_act != 0) {
                    COUNT(x.n_trip);
                    // source: 16 bytes from the literal, the ring, or HBM
                    // (a lane that loads nothing below stores nothing either:
                    // whatever v starts with - the literal bytes of trip 0 -
                    // is never seen)
                    B16x v = lit16;
                    if (c != 0 && __builtin_amdgcn_inverse_ballot_w64(
                                      M_act & M_lit)) {
                        __builtin_memcpy(&v, win_src + (lane + hd + c), 16);
                    }
                    if ((M_act & M_far) != 0 &&
                        __builtin_amdgcn_inverse_ballot_w64(M_act & M_far))
                        __builtin_memcpy(&v, dst + (q + c), 16);
                    if (__builtin_amdgcn_inverse_ballot_w64(M_act & M_rng)) {
                        // (only the lanes that need it: the LDS serves a wave's
                        // scattered unaligned reads a few lanes per cycle)
                        // (one unaligned 16-byte read costs the LDS one
                        // cycle per lane, two 8-byte reads two:
                        // tests/hw/lds_cost.hip; it may run into the mirror)
                        __builtin_memcpy(&v, rg + ((q + c) & (kRing2 - 1)), 16);
                    }
                    const uint32_t wa = (dstp + c) & (kRing2 - 1);
                    // (rare) the element's own bytes wrap around the ring's
                    // end: those lanes store bytewise, after the others
                    uint64_t M_strad = 0;
                    uint32_t m = 16;
                    if (wraps) {
                        m = olen - c < 16 ? olen - c : 16;
                        M_strad = M_act & __ballot(wa + m > kRing2);
                    }
                    if (__builtin_amdgcn_inverse_ballot_w64(M_act & ~M_strad))
                        __builtin_memcpy(rg + wa, &v, 16); // may reach the mirror
                    if (M_strad != 0 &&
                        __builtin_amdgcn_inverse_ballot_w64(M_strad)) {
                        for (uint32_t j = 0; j < m; j++) {
                            const uint64_t part = j < 8 ? v.lo : v.hi;
                            rg[(wa + j) & (kRing2 - 1)] =
                                (uint8_t)(part >> (8 * (j & 7)));
                        }
                    }
                }
                if (c == 0)
                    break;
            }
        }
        // ---- 5. the sweep: elements that depend on this window -------------
        uint64_t dep = K & ~M_lw;
        while (dep) {
            COUNT(x.n_dep);
            const uint32_t i = (uint32_t)__builtin_ctzll(dep);
            dep &= dep - 1;
            const uint32_t ni = rdlane(olen, i), di = rdlane(dstp, i);
            uint32_t val = 0;
            if ((M_lit >> i) & 1) {
                // (only windows at the very end of the input, `inner` false)
                if (lane < ni)
                    val = src[s + i + rdlane(hd, i) + lane];
            } else {
                const uint32_t qi = rdlane(q, i), oi = rdlane(off, i);
                const uint32_t nsrc = ni < oi ? ni : oi;
                uint32_t kk = lane;
                if (oi < ni) { // overlapping: byte k repeats byte k mod oi
                    const uint32_t quo = (uint32_t)(
                        ((float)lane + 0.5f) *
                        __builtin_amdgcn_rcpf((float)oi));
                    kk = lane - quo * oi;
                }
                const bool in_ring = qi >= safe_lo;
                if (!in_ring) {
                    // bytes the ring has lost that are not stored yet (only
                    // right after a long literal): store them first
                    if (qi + nsrc > R.gflush)
                        R.flush_partial(di);
                    if (qi + nsrc > R.fenced) {
                        R.fence_for(0xFFFFFFFFu);
                        COUNT(x.n_fence);
                    }
                }
                if (lane < ni)
                    val = in_ring ? (uint32_t)rg[(qi + kk) & (kRing2 - 1)]
                                  : (uint32_t)dst[qi + kk];
            }
            if (lane < ni)
                rg[(di + lane) & (kRing2 - 1)] = (uint8_t)val;
        }
        // The mirror of ring[0,16) behind the ring's end, once per window (a
        // uniform test): the window's stores - whole 16-byte pieces, so up to
        // d + W + 16 - touched ring[0,16), or spilled over the end into the
        // mirror.  Within the window nobody reads through it what these
        // stores changed: such a source lies below safe_lo; the sweep and the
        // flush address the ring bytewise / dword-aligned.
        {
            const uint32_t a0 = d & (kRing2 - 1);
            if (a0 < 16 || a0 + W + 16 > kRing2)
                R.mirror();
        }
        // ---- 6. advance; whole 256-byte pieces go to HBM -------------------
        d += W;
        s += cur;
        w = w_next;
        R.flush_chunks(d);
    }
    x.s = s;
    x.d = d;
    x.ring_lo = ring_lo;
    return irregular;
}

// ---------------------------------------------------------------------
// The third-generation window loop: kG3 x 64 compressed bytes per window,
// one ELEMENT per lane.
//
// The second generation spends its instructions on lanes that hold no
// element: every lane decodes "the element that would start at my byte" in
// full (offset, length, checks, source classes), and one lane in four or five
// is a real start.  Instruction issue - vector and scalar - is what bounds
// the loop, so here the expensive part runs on real elements only:
//
//   1. LENGTHS.  A window is kG3 groups of 64 input bytes; lane l looks at
//      the TAG bytes at s + 64 g + l, just far enough to know how many
//      compressed bytes the element there would take (tag only: a literal
//      with length bytes - 61 bytes or more - always ends a window).
//   2. STARTS.  Per group one ds_bpermute per round, four rounds (a lane's
//      32-bit set of its 32-byte sub-window, frontier as the highest bit:
//      R |= R[frontier]); the 2 kG3 sub-windows are strung together by the
//      scalar unit (two v_readlane each).
//   3. COMPACTION.  The t-th start of the window goes to lane t: rank by
//      population count, position through a 64 kG3-byte table in LDS.
//   4. DECODE, PLACEMENT, CHECKS as in the second generation, but every lane
//      below the element count holds a real element (40 of 64 lanes on the
//      corpus instead of 14).
//   5. COPY.  Every element whose source is complete before the window -
//      literals, copies from in front of it - is copied by its lane, as in
//      the second generation: 16 bytes per trip, whole-piece stores resolved
//      by lane order, the far sources of the whole window requested at once.
//      Then the LATE elements - copies that read the window's own output
//      (2.3 per window on text, 8 on html) or overlap themselves - are moved
//      one by one, in stream order, by the whole wave (lane k = byte k).
//
// A window needs kTail3 bytes of input in front of it, so that none of its
// loads can leave the input; the last bytes of a stream (and streams shorter
// than that) are decode_windows2's.  True: something irregular.
// ---------------------------------------------------------------------
#define SNAPMI_G3 4
constexpr uint32_t kG3 = SNAPMI_G3;
static_assert(kG3 == 2 || kG3 == 4,
              "positions in a window are masked with 64 kG3 - 1; 8 groups do "
              "not fit 64 VGPRs");
// the last position (64 kG3 - 1), a tag, 60 literal bytes read as whole
// 16-byte pieces
constexpr uint32_t kTail3 = 64 * kG3 + 1 + 64 + 16;

__device__ __forceinline__ bool decode_windows3(Wide &x, const uint32_t lane,
                                                l_u8 *const postab)
{
    gcptr src = x.src;
    gptr dst = x.dst;
    const uint32_t slen = x.slen, dlen = x.dlen;
    uint32_t s = x.s, d = x.d, ring_lo = x.ring_lo;
    Ring2 &R = x.R;
    l_u8 *const rg = R.rg;
    bool irregular = false;
    uint32_t tg[kG3]; // tag bytes of the window at s, one per group
    bool have = false;
    const int c4 = (int)((lane | 31) << 2);
    const uint32_t self = 1u << (lane & 31);

    while (slen - s >= kTail3) {
        COUNT(x.n_win3);
        const gcptr win_src = src + s;
        if (!have) {
#pragma unroll
            for (uint32_t g = 0; g < kG3; g++)
                tg[g] = win_src[64 * g + lane];
        }
        // ---- 1. lengths, 2. starts -------------------------------------
        uint32_t nx[kG3], Rr[kG3];
#pragma unroll
        for (uint32_t g = 0; g < kG3; g++) {
            const uint32_t t = tg[g], type = t & 3;
            const uint32_t pos = 64 * g + lane;
            // copies take 2, 3, 5 bytes; a literal its tag and n6 + 1 bytes;
            // a literal with length bytes (n6 >= 60) ends every chain: its
            // "next" lies behind the window
            uint32_t enc =
                type ? (0x05030200u >> (8 * type)) & 0xFF : (t >> 2) + 2;
            enc = (t & 0xF3) == 0xF0 ? 255 : enc;
            nx[g] = pos + enc;
            // (the chain stays in this lane's 32-byte sub-window)
            Rr[g] = self | ((nx[g] ^ pos) < 32 ? 1u << (nx[g] & 31) : 0);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
#pragma unroll
            for (uint32_t g = 0; g < kG3; g++) {
                const int sel = c4 - 4 * (int)__builtin_clz(Rr[g]);
                Rr[g] |= (uint32_t)__builtin_amdgcn_ds_bpermute(sel,
                                                                (int)Rr[g]);
            }
        }
        // the sub-windows, strung together: sub-window k is entered at its
        // byte e; its last start's element leads into a later one (or out
        // of the window)
        uint32_t S32[2 * kG3];
#pragma unroll
        for (uint32_t kk = 0; kk < 2 * kG3; kk++)
            S32[kk] = 0;
        {
            uint32_t k = 0, e = 0;
#pragma unroll
            for (uint32_t kk = 0; kk < 2 * kG3; kk++) {
                if (k == kk) {
                    const uint32_t g = kk >> 1, half = 32 * (kk & 1);
                    const uint32_t Sk = rdlane(Rr[g], half + e);
                    S32[kk] = Sk;
                    const uint32_t lt = 31 - (uint32_t)__builtin_clz(Sk);
                    const uint32_t nxa = rdlane(nx[g], half + lt);
                    k = nxa >> 5;
                    e = nxa & 31;
                }
            }
        }
        uint64_t S[kG3];
#pragma unroll
        for (uint32_t g = 0; g < kG3; g++)
            S[g] = ((uint64_t)S32[2 * g + 1] << 32) | S32[2 * g];
        // ---- 3. compaction ---------------------------------------------
        uint32_t base = 0;
#pragma unroll
        for (uint32_t g = 0; g < kG3; g++) {
            // (a lane that is no start writes behind the table: a select is
            // cheaper than an exec region - the scalar unit is what is short)
            const uint32_t r = __builtin_amdgcn_inverse_ballot_w64(S[g])
                                   ? base + popc_below(S[g])
                                   : 64 * kG3 + lane;
            postab[r] = (uint8_t)(64 * g + lane);
            base += (uint32_t)__builtin_popcountll(S[g]);
        }
        const uint32_t n_el = base < kWave ? base : kWave;
        const uint64_t M_act = n_el == kWave ? ~0ull : (1ull << n_el) - 1;
        // (lanes behind the last element read what an earlier window left:
        // any position in the window will do, they are masked by M_act)
        const uint32_t pos = postab[lane] & (64 * kG3 - 1);
        // ---- 4. the element, in full -----------------------------------
        uint64_t w;
        __builtin_memcpy(&w, win_src + pos, 8);
        B16x lit16;
        __builtin_memcpy(&lit16, win_src + (pos + 1), 16);
        const uint32_t tag = (uint32_t)w & 0xFF;
        const uint32_t b14 = (uint32_t)(w >> 8); // the 4 bytes after the tag
        const uint32_t type = tag & 3, n6 = tag >> 2;
        const uint64_t M_lit = __ballot(type == 0);
        const uint64_t M_lng = M_act & M_lit & __ballot(n6 >= 60);
        // a copy's offset: 1, 2 or 4 bytes behind the tag (TagEntry::offset /
        // read_copy, reference src/decompress.rs:233-250,433-474)
        const uint32_t cnb = type + (type == 3);
        const uint32_t sh = 32 - 8 * cnb;
        const uint32_t ext = (b14 << (sh & 31)) >> (sh & 31);
        const uint32_t off = ext | (type == 1 ? (tag & 0xE0u) << 3 : 0);
        // output bytes: a literal's and a long copy's n6 + 1, copy-1's 4..11
        const uint32_t olen = type == 1 ? 4 + (n6 & 7) : 1 + n6;
        const uint32_t enc = type == 0 ? n6 + 2 : 1 + cnb;
        const uint64_t M_el = M_act & ~M_lng;
        const uint32_t o =
            __builtin_amdgcn_inverse_ballot_w64(M_el) ? olen : 0;
        const uint32_t incl = wave_inclusive_add(o);
        // the window ends in front of a literal with length bytes, and after
        // kWinMax output bytes
        const uint64_t below =
            M_lng ? ((1ull << __builtin_ctzll(M_lng)) - 1) : ~0ull;
        const uint64_t K = M_el & __ballot(incl <= kWinMax) & below;
        if (K == 0) {
            // lane 0 is a literal with length bytes (reference read_literal,
            // src/decompress.rs:161-228): moved by the whole wave
            const uint32_t lnb = rdlane(n6, 0) - 59; // 1..4 length bytes
            const uint32_t b0 = rdlane(b14, 0);
            const uint64_t Lq =
                (uint64_t)(lnb == 4 ? b0 : b0 & ((1u << (8 * lnb)) - 1)) + 1;
            const uint32_t h0 = 1 + lnb;
            if (slen - (s + h0) < Lq || dlen - d < Lq) {
                irregular = true; // the sequential decoder names the error
                break;
            }
            R.flush_partial(d);
            wave_copy<false>(dst + d, src + s + h0, Lq, lane);
            s += h0 + (uint32_t)Lq;
            d += (uint32_t)Lq;
            R.gflush = d;
            ring_lo = d; // these bytes are not in the ring
            have = false;
            continue;
        }
        // (K is a prefix of the lanes)
        const uint32_t nK = (uint32_t)__builtin_popcountll(K);
        const uint32_t W = rdlane(incl, nK - 1);       // output of the window
        const uint32_t cur = rdlane(pos, nK - 1) + rdlane(enc, nK - 1);
        const uint32_t dstp = d + (incl - o);          // element's position
        // reference checks :209-217 (dst side), :245-250, :327-332
        const uint64_t M_cpy = K & ~M_lit;
        if (W > dlen - d ||
            (M_cpy & (__ballot(off == 0) | __ballot(off > dstp))) != 0) {
            irregular = true;
            break;
        }
        PROF(
        x.n_elem += nK;
        )
        // next window's tags: issued now, consumed after the copy step
        uint32_t tgn[kG3];
        const bool have_next = slen - (s + cur) >= kTail3;
        if (have_next) {
#pragma unroll
            for (uint32_t g = 0; g < kG3; g++)
                tgn[g] = win_src[cur + 64 * g + lane];
        } else {
#pragma unroll
            for (uint32_t g = 0; g < kG3; g++)
                tgn[g] = 0;
        }
        // ---- 5. the copy step, in in-order runs ------------------------
        const uint32_t q = dstp - off; // copy source (if cpy)
        const uint32_t qe = q + olen;  // (its end, for a copy that does not
                                       // overlap itself)
        // (whole 16-byte pieces are stored: up to 15 bytes behind the window's
        // output may be clobbered too)
        const uint32_t dW = d + W + 16;
        uint32_t safe_lo = dW > kRing2 ? dW - kRing2 : 0;
        safe_lo = safe_lo > ring_lo ? safe_lo : ring_lo;
        const uint64_t M_ring = __ballot(q >= safe_lo);
        // A source in HBM must have been stored, and its 16-byte loads must
        // stay inside the buffer.  Both hold by construction while the ring
        // is whole (a far source ends 1968 bytes or more in front of d, the
        // stores lag d by less than 256); only behind a long literal, whose
        // bytes are not in the ring, must the lanes be asked.
        const bool whole = ring_lo + kRing2 <= dW;
        const uint64_t M_farok =
            whole ? ~0ull
                  : (dlen >= 64 ? __ballot(qe <= R.gflush) &
                                      __ballot(q <= dlen - 64)
                                : 0);
        // LATE elements are moved one by one, in stream order, by the whole
        // wave, after everything else: copies that read this window's own
        // output (qe > d: that includes a copy that overlaps itself) and
        // sources neither in the ring nor stored.  (2.3 per window on text,
        // 8 on html: a loop of in-order RUNS of lanes - whole-piece stores
        // per run - was measured first and cost the scalar unit five times
        // as much per dependent element.)
        const uint64_t M_late =
            M_cpy & (__ballot(qe > d) | ~(M_ring | M_farok));
        const uint64_t M_lw = K & ~M_late; // copied by their own lanes, now
        // an element's own bytes wrap around the ring's end only in a
        // window whose output does (uniform)
        const uint32_t r0 = d & (kRing2 - 1);
        const bool wraps = r0 + W > kRing2;
        const uint64_t M_far = M_lw & M_cpy & ~M_ring; // source in HBM
        const uint64_t M_rng = M_lw & M_cpy & M_ring;  // ... in the ring
        // The first piece of every element whose source is not in the ring:
        // literal bytes are here already (lit16); far sources are an L2 miss
        // each (64 bytes from HBM or the MALL, a microsecond or two), and
        // that latency is what a wave waits for most - all of them at once.
        B16x v0 = lit16;
        if (M_far) {
            PROF(
            x.n_far += __builtin_popcountll(M_far);
            )
            // sources in HBM must be completed stores
            if ((M_far & __ballot(qe > R.fenced)) != 0) {
                R.fence_for(0xFFFFFFFFu);
                COUNT(x.n_fence);
            }
            if (__builtin_amdgcn_inverse_ballot_w64(M_far))
                __builtin_memcpy(&v0, dst + q, 16);
        }
        // Every lane stores WHOLE 16-byte pieces with one ds_write_b128 per
        // trip, last piece first; the excess of a short element lands on
        // the elements behind it - higher lanes of the same instruction,
        // which win (see decode_windows2), or late elements, which are
        // written afterwards.
        const uint64_t M_g16 = M_lw & __ballot(olen > 16);
        if (!wraps && M_g16 == 0) {
            // the usual window: one piece per element, no wrap
            COUNT(x.n_trip);
            B16x v = v0;
            if (__builtin_amdgcn_inverse_ballot_w64(M_rng))
                __builtin_memcpy(&v, rg + (q & (kRing2 - 1)), 16);
            if (__builtin_amdgcn_inverse_ballot_w64(M_lw))
                __builtin_memcpy(rg + (dstp & (kRing2 - 1)), &v, 16);
        } else {
            const uint32_t top =
                M_g16 == 0 ? 0
                           : ((M_g16 & __ballot(olen > 48))
                                  ? 48
                                  : ((M_g16 & __ballot(olen > 32)) ? 32 : 16));
            for (uint32_t c = top;; c -= 16) {
                const uint64_t M_actc =
                    c == 0 ? M_lw : M_lw & __ballot(c < olen);
                if (M_actc != 0) {
                    COUNT(x.n_trip);
                    B16x v = v0;
                    if (c != 0) {
                        if (__builtin_amdgcn_inverse_ballot_w64(M_actc & M_lit))
                            __builtin_memcpy(&v, win_src + (pos + 1 + c), 16);
                        if ((M_actc & M_far) != 0 &&
                            __builtin_amdgcn_inverse_ballot_w64(M_actc & M_far))
                            __builtin_memcpy(&v, dst + (q + c), 16);
                    }
                    if (__builtin_amdgcn_inverse_ballot_w64(M_actc & M_rng))
                        __builtin_memcpy(&v, rg + ((q + c) & (kRing2 - 1)),
                                         16);
                    const uint32_t wa = (dstp + c) & (kRing2 - 1);
                    // (rare) the element's own bytes wrap around the ring's
                    // end: those lanes store bytewise, after the others
                    uint64_t M_strad = 0;
                    uint32_t m = 16;
                    if (wraps) {
                        m = olen - c < 16 ? olen - c : 16;
                        M_strad = M_actc & __ballot(wa + m > kRing2);
                    }
                    if (__builtin_amdgcn_inverse_ballot_w64(M_actc & ~M_strad))
                        __builtin_memcpy(rg + wa, &v, 16); // may reach the mirror
                    if (M_strad != 0 &&
                        __builtin_amdgcn_inverse_ballot_w64(M_strad)) {
                        for (uint32_t j = 0; j < m; j++) {
                            const uint64_t part = j < 8 ? v.lo : v.hi;
                            rg[(wa + j) & (kRing2 - 1)] =
                                (uint8_t)(part >> (8 * (j & 7)));
                        }
                    }
                }
                if (c == 0)
                    break;
            }
        }
        // the late elements: lane k = byte k of the element (exactly its
        // bytes, addressed bytewise: no mirror)
        uint64_t late = M_late;
        if ((M_late & ~M_ring) == 0) {
            // (the usual case: every late source is in the ring)
            while (late) {
                COUNT(x.n_dep);
                const uint32_t i = (uint32_t)__builtin_ctzll(late);
                late &= late - 1;
                const uint32_t F = rdlane(dstp, i), ni = rdlane(olen, i);
                const uint32_t qi = rdlane(q, i), oi = F - qi;
                uint32_t kk = lane;
                if (oi < ni) { // overlapping: byte k repeats byte k mod oi
                    const uint32_t quo = (uint32_t)(
                        ((float)lane + 0.5f) *
                        __builtin_amdgcn_rcpf((float)oi));
                    kk = lane - quo * oi;
                }
                if (lane < ni)
                    rg[(F + lane) & (kRing2 - 1)] =
                        rg[(qi + kk) & (kRing2 - 1)];
            }
        }
        while (late) {
            COUNT(x.n_dep);
            const uint32_t i = (uint32_t)__builtin_ctzll(late);
            late &= late - 1;
            const uint32_t F = rdlane(dstp, i), ni = rdlane(olen, i);
            const uint32_t qi = rdlane(q, i), oi = F - qi;
            uint32_t kk = lane;
            if (oi < ni) { // overlapping: byte k repeats byte k mod oi
                const uint32_t quo = (uint32_t)(
                    ((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)oi));
                kk = lane - quo * oi;
            }
            const bool in_ring = qi >= safe_lo;
            if (!in_ring) {
                const uint32_t nsrc = ni < oi ? ni : oi;
                // bytes the ring has lost that are not stored yet (only
                // right after a long literal): store them first
                if (qi + nsrc > R.gflush)
                    R.flush_partial(F);
                if (qi + nsrc > R.fenced) {
                    R.fence_for(0xFFFFFFFFu);
                    COUNT(x.n_fence);
                }
            }
            if (lane < ni) {
                const uint32_t val =
                    in_ring ? (uint32_t)rg[(qi + kk) & (kRing2 - 1)]
                            : (uint32_t)dst[qi + kk];
                rg[(F + lane) & (kRing2 - 1)] = (uint8_t)val;
            }
        }
        // the mirror for the next window (see decode_windows2)
        if (r0 < 16 || r0 + W + 16 > kRing2)
            R.mirror();
        // ---- 6. advance; whole 256-byte pieces go to HBM -----------------
        d += W;
        s += cur;
#pragma unroll
        for (uint32_t g = 0; g < kG3; g++)
            tg[g] = tgn[g];
        have = have_next;
        R.flush_chunks(d);
    }
    x.s = s;
    x.d = d;
    x.ring_lo = ring_lo;
    return irregular;
}
} // namespace

// ---------------------------------------------------------------------
// Tiny streams, one per LANE: the reference's loop (src/decompress.rs:75-95,
// 130-343) as it stands, every lane on a stream of its own.  A wavefront per
// stream spends ten microseconds of dependent round trips (descriptors,
// header, first window, flush) on a stream of 150 bytes; here 64 streams
// share them.  Literals move 16 bytes at a time where both buffers allow it,
// copies bytewise (they may overlap).  Errors and their fields are the
// reference's, lane by lane.
// ---------------------------------------------------------------------
namespace {
#define SNAPMI_TINY_FAIL(kind, fa, fb, fc)                                    \
    do {                                                                      \
        set_error(a.errs, st, (kind), (fa), (fb), (fc));                      \
        a.out_lens[st] = 0;                                                   \
        return;                                                               \
    } while (0)

__device__ __forceinline__ void decode_tiny(const DecompressArgs &a,
                                            const uint64_t st)
{
    gcptr in = (gcptr)a.in_ptrs[st];
    const uint64_t in_len = a.in_lens[st];
    gptr dst = (gptr)a.out_ptrs[st];
    const uint32_t mode = a.modes ? a.modes[st] : 0;
    const uint64_t cap = a.out_caps[st];
    if (mode == 3)
        return; // another launch decodes this stream
    if (mode == 1) { // stored frame chunk (reference src/read.rs:173-199)
        if (in_len > cap)
            SNAPMI_TINY_FAIL(SNAPMI_BUFFER_TOO_SMALL, cap, in_len, 0);
        for (uint64_t i = 0; i < in_len; i++)
            dst[i] = in[i];
        set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
        a.out_lens[st] = in_len;
        return;
    }
    uint32_t hdr = 0;
    uint64_t dst_len = 0;
    if (mode == 2) { // headerless piece: exactly out_caps bytes
        dst_len = cap;
    } else {
        if (in_len == 0)
            SNAPMI_TINY_FAIL(SNAPMI_EMPTY, 0, 0, 0);
        if (read_header(in, in_len, &hdr, &dst_len, a.errs, st) != SNAPMI_OK) {
            a.out_lens[st] = 0;
            return;
        }
        if (dst_len > cap)
            SNAPMI_TINY_FAIL(SNAPMI_BUFFER_TOO_SMALL, cap, dst_len, 0);
    }
    gcptr src = in + hdr;
    const uint64_t src_len = in_len - hdr;
    uint64_t s = 0, d = 0;
    while (s < src_len) {
        const uint32_t tag = src[s];
        s += 1;
        if ((tag & 3) == 0) {
            // read_literal, src/decompress.rs:161-228
            uint64_t len = (tag >> 2) + 1;
            if (len >= 61) {
                if (s + 4 > src_len)
                    SNAPMI_TINY_FAIL(SNAPMI_LITERAL, 4, src_len - s,
                                     dst_len - d);
                const uint32_t nb = (uint32_t)len - 60;
                const uint32_t raw = ld32u(src + s);
                len = (uint64_t)(nb == 4 ? raw
                                         : raw & ((1u << (8 * nb)) - 1)) +
                      1;
                s += nb;
            }
            if (src_len - s < len || dst_len - d < len)
                SNAPMI_TINY_FAIL(SNAPMI_LITERAL, len, src_len - s,
                                 dst_len - d);
            uint64_t i = 0;
            for (; i + 16 <= len; i += 16) {
                const u32x4 t = ld128g(src + s + i);
                __builtin_memcpy(dst + d + i, &t, 16);
            }
            for (; i < len; i++)
                dst[d + i] = src[s + i];
            s += len;
            d += len;
        } else {
            // read_copy + TagEntry::offset, src/decompress.rs:233-343,433-474
            const uint32_t kind = tag & 3;
            const uint32_t nb = kind == 1 ? 1 : (kind == 2 ? 2 : 4);
            const uint32_t len =
                kind == 1 ? 4 + ((tag >> 2) & 7) : 1 + (tag >> 2);
            uint64_t offset = kind == 1 ? (uint64_t)(tag >> 5) << 8 : 0;
            if (s + nb > src_len)
                SNAPMI_TINY_FAIL(SNAPMI_COPY_READ, nb, src_len - s, 0);
            uint32_t v = 0;
            for (uint32_t k = 0; k < nb; k++)
                v |= (uint32_t)src[s + k] << (8 * k);
            offset |= v;
            s += nb;
            if (d <= offset - 1) // wrapping, also catches offset == 0
                SNAPMI_TINY_FAIL(SNAPMI_OFFSET, offset, d, 0);
            const uint64_t end = d + len;
            if (end > dst_len)
                SNAPMI_TINY_FAIL(SNAPMI_COPY_WRITE, len, dst_len - d, 0);
            // (bytewise and in order: the source may overlap the
            // destination; a lane's own stores are visible to its loads)
            for (uint32_t k = 0; k < len; k++)
                dst[d + k] = dst[d - offset + k];
            d = end;
        }
    }
    if (d != dst_len)
        SNAPMI_TINY_FAIL(SNAPMI_HEADER_MISMATCH, dst_len, d, 0);
    set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
    a.out_lens[st] = dst_len;
}
} // namespace

// 8 waves per SIMD (64 VGPRs, 5 KiB of LDS each): a window is a chain of
// LDS / HBM round trips, and two more waves to switch to are worth more than
// a few spilled dwords (36.0 -> 32.2 ms at cfg2 for the second generation).
#define SNAPMI_DEC2_WAVES 8
__attribute__((amdgpu_waves_per_eu(SNAPMI_DEC2_WAVES, SNAPMI_DEC2_WAVES)))
__global__ __launch_bounds__(64) void k_decompress_streams3(DecompressArgs a)
{
    // the ring, its 16-byte mirror, the compaction table (+ 64 bytes that
    // take the writes of lanes without an element)
    __shared__ __attribute__((aligned(16)))
    uint8_t ring_mem[kRing2 + 16 + 64 * kG3 + 64];
    const uint32_t lane = threadIdx.x;
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    // workgroups [0, n_big): one stream each; the tiny streams behind them
    // (the sort put them last) are k_decompress_tiny's, one per LANE
    if (blockIdx.x >= uni(a.bucket_pos[64]))
        return;
    Wide x;
    if (!open_stream(a, lane, x, (l_u8 *)ring_mem, blockIdx.x))
        return;
    bool irregular = x.too_big();
    if (!irregular)
        irregular = decode_windows3(x, lane, (l_u8 *)ring_mem + kRing2 + 16);
    if (!irregular)
        irregular = decode_windows2(x, lane);
    close_stream(a, lane, x, irregular);
}

// The same decoder for batches of millions of streams: a workgroup takes
// kManyStreams positions of the sorted order, so the dispatcher hands out a
// sixteenth of the workgroups - at 10.7 M streams of 200 bytes, all of them
// k_decompress_tiny's, the 10.7 M empty workgroups of the launch above were
// 1.1 ms of a 4.4 ms pass.  The positions are a whole grid apart (workgroup w:
// w, w + G, w + 2G ...): every workgroup starts with one of the G longest
// streams and gets one of every size stratum, so a few giant streams in such
// a batch do not end up behind each other in one wavefront.
// (No occupancy attribute: the loop around the body costs registers, and held
// to 64 VGPRs it spills; four wavefronts per SIMD are plenty for streams that
// small.)
__global__ __launch_bounds__(64) void k_decompress_streams3_many(
    DecompressArgs a)
{
    __shared__ __attribute__((aligned(16)))
    uint8_t ring_mem[kRing2 + 16 + 64 * kG3 + 64];
    const uint32_t lane = threadIdx.x;
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    const uint32_t n_big = uni(a.bucket_pos[64]);
    for (uint32_t j = 0; j < kManyStreams; j++) {
        const uint32_t slot = blockIdx.x + j * gridDim.x;
        if (slot >= n_big)
            return;
        Wide x;
        if (!open_stream(a, lane, x, (l_u8 *)ring_mem, slot))
            continue;
        bool irregular = x.too_big();
        if (!irregular)
            irregular =
                decode_windows3(x, lane, (l_u8 *)ring_mem + kRing2 + 16);
        if (!irregular)
            irregular = decode_windows2(x, lane);
        close_stream(a, lane, x, irregular);
    }
}

#ifdef SNAPMI_TESTING // (libsnapmi_test.so only)
// The second generation alone (option decode_kernel = 2): kept as the
// cross-check every decoder parity test also runs through.
__attribute__((amdgpu_waves_per_eu(SNAPMI_DEC2_WAVES, SNAPMI_DEC2_WAVES)))
__global__ __launch_bounds__(64) void k_decompress_streams2(DecompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring_mem[kRing2 + 16];
    const uint32_t lane = threadIdx.x;
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    Wide x;
    if (!open_stream(a, lane, x, (l_u8 *)ring_mem, blockIdx.x))
        return;
    const bool irregular = x.too_big() || decode_windows2(x, lane);
    close_stream(a, lane, x, irregular);
}
#endif

// The reference's loop alone, one element at a time (option decode_kernel =
// 0): what a context falls back to when the self-check of the LDS store
// order fails, and a third opinion for the tests.
__global__ __launch_bounds__(64) void k_decompress_sequential(DecompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring_mem[16];
    const uint32_t lane = threadIdx.x;
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    Wide x;
    if (!open_stream(a, lane, x, (l_u8 *)ring_mem, blockIdx.x))
        return;
    decode_sequential(a, x.st, lane, x.src, x.src_len, x.dst, x.dst_len, 0, 0);
}

// Tiny streams (fewer than kTiny compressed bytes), one per LANE.  A stream
// whose output is tiny as well (the usual case) is staged in LDS - input and
// output dword-interleaved across the lanes, so that lane l's byte k is at
// ((k >> 2) * kLanes + l) * 4 + (k & 3) and the lanes of an access fall on
// different banks - decoded there with the reference's loop, and stored with
// 16-byte accesses: four LDS latencies per byte moved instead of two HBM
// round trips.  The others run the same loop on global memory.
// k_decompress_small (round 5) is the same body with 512 bytes of input and of
// output per lane on 32 lanes of a wavefront, for the streams between the
// tiny ones and the wavefront decoder's (400-byte pages decoded at 106 GiB/s
// a wavefront each, between 290 and 170 for their neighbours).
template <uint32_t kLog2, uint32_t kLanes>
__device__ __forceinline__ void decompress_lane_streams(
    const DecompressArgs &a, const uint64_t first, const uint64_t end)
{
    constexpr uint32_t kT = 1u << kLog2;
    __shared__ __attribute__((aligned(16))) uint32_t tin[kT / 4 * kLanes];
    __shared__ __attribute__((aligned(16))) uint32_t tout[kT / 4 * kLanes];
    const uint32_t lane = threadIdx.x;
    if (first + (uint64_t)blockIdx.x * kLanes >= end)
        return;
    const uint64_t i = first + (uint64_t)blockIdx.x * kLanes + lane;
    if (lane >= kLanes || i >= end)
        return;
    const uint64_t st = a.order[i];
    gcptr in = (gcptr)a.in_ptrs[st];
    const uint32_t in_len = (uint32_t)a.in_lens[st]; // < kT
    const uint32_t mode = a.modes ? a.modes[st] : 0;
    // the header decides: a stream with a tiny output goes through LDS
    uint64_t dl = 0;
    uint32_t hdr = 0;
    bool small = mode == 0 && in_len > 0;
    if (small) {
        hdr = read_varint(in, in_len, &dl);
        small = hdr != 0 && dl <= kT && dl <= a.out_caps[st];
    }
    if (!small) {
        decode_tiny(a, st);
        return;
    }
    typedef __attribute__((address_space(3))) uint32_t l_u32t;
    l_u8 *const bin = (l_u8 *)(l_u32t *)tin;
    l_u8 *const bout = (l_u8 *)(l_u32t *)tout;
    auto at = [lane](uint32_t k) {
        return ((k >> 2) * kLanes + lane) * 4 + (k & 3);
    };
    // stage the elements: whole dwords (reads stay inside the stream: the
    // last partial dword bytewise)
    const uint32_t slen = in_len - hdr;
    gcptr src = in + hdr;
    {
        uint32_t k = 0;
        for (; k + 4 <= slen; k += 4)
            *(l_u32t *)(bin + at(k)) = ld32u(src + k);
        for (; k < slen; k++)
            bin[at(k)] = src[k];
    }
    const uint32_t dlen = (uint32_t)dl;
    uint32_t s = 0, d = 0;
    while (s < slen) {
        const uint32_t tag = bin[at(s)];
        s += 1;
        if ((tag & 3) == 0) {
            uint32_t len = (tag >> 2) + 1;
            if (len >= 61) {
                if (s + 4 > slen)
                    SNAPMI_TINY_FAIL(SNAPMI_LITERAL, 4, slen - s, dlen - d);
                const uint32_t nb = len - 60;
                uint32_t raw = 0;
                for (uint32_t k = 0; k < 4; k++)
                    raw |= (uint32_t)bin[at(s + k)] << (8 * k);
                // (u64: a length field of 0xFFFFFFFF + 1 must not wrap)
                const uint64_t l64 =
                    (uint64_t)(nb == 4 ? raw : raw & ((1u << (8 * nb)) - 1)) +
                    1;
                s += nb;
                if (slen - s < l64 || dlen - d < l64)
                    SNAPMI_TINY_FAIL(SNAPMI_LITERAL, l64, slen - s, dlen - d);
                len = (uint32_t)l64;
            } else if (slen - s < len || dlen - d < len) {
                SNAPMI_TINY_FAIL(SNAPMI_LITERAL, len, slen - s, dlen - d);
            }
            for (uint32_t k = 0; k < len; k++)
                bout[at(d + k)] = bin[at(s + k)];
            s += len;
            d += len;
        } else {
            const uint32_t kind = tag & 3;
            const uint32_t nb = kind == 1 ? 1 : (kind == 2 ? 2 : 4);
            const uint32_t len =
                kind == 1 ? 4 + ((tag >> 2) & 7) : 1 + (tag >> 2);
            uint64_t offset = kind == 1 ? (uint64_t)(tag >> 5) << 8 : 0;
            if (s + nb > slen)
                SNAPMI_TINY_FAIL(SNAPMI_COPY_READ, nb, slen - s, 0);
            uint32_t v = 0;
            for (uint32_t k = 0; k < nb; k++)
                v |= (uint32_t)bin[at(s + k)] << (8 * k);
            offset |= v;
            s += nb;
            if (d <= offset - 1) // wrapping, also catches offset == 0
                SNAPMI_TINY_FAIL(SNAPMI_OFFSET, offset, d, 0);
            if (d + len > dlen)
                SNAPMI_TINY_FAIL(SNAPMI_COPY_WRITE, len, dlen - d, 0);
            const uint32_t from = d - (uint32_t)offset;
            for (uint32_t k = 0; k < len; k++)
                bout[at(d + k)] = bout[at(from + k)];
            d += len;
        }
    }
    if (d != dlen)
        SNAPMI_TINY_FAIL(SNAPMI_HEADER_MISMATCH, dlen, d, 0);
    gptr dst = (gptr)a.out_ptrs[st];
    {
        uint32_t k = 0;
        for (; k + 4 <= dlen; k += 4)
            st32u(dst + k, *(const l_u32t *)(bout + at(k)));
        for (; k < dlen; k++)
            dst[k] = bout[at(k)];
    }
    set_error(a.errs, st, SNAPMI_OK, 0, 0, 0);
    a.out_lens[st] = dlen;
}

__global__ __launch_bounds__(64) void k_decompress_tiny(DecompressArgs a)
{
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    decompress_lane_streams<kTinyLog2, kWave>(a, uni(a.bucket_pos[65]),
                                               a.n_streams);
}

// The libsnappy seam's single-launch path (snapmi_api.hip, seam_tiny): ONE
// stream of under 256 compressed bytes; descriptor, input and output in
// pinned host memory, *done polled by the host (k_seam_compress_tiny).
__global__ __launch_bounds__(64) void k_seam_decompress_tiny(
    DecompressArgs a, uint32_t *done, uint32_t seq)
{
    decompress_lane_streams<kTinyLog2, kWave>(a, 0, 1);
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(done, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(64) void k_decompress_small(DecompressArgs a)
{
    if (a.gate && uni64(*a.gate) != a.gate_value)
        return;
    decompress_lane_streams<kSmallDecLog2, 32>(a, uni(a.bucket_pos[64]),
                                               uni(a.bucket_pos[65]));
}

// ---------------------------------------------------------------------
// One long raw stream on many wavefronts (snapmi_decompress_stream).
//
// A raw stream has no index and its elements form a chain, but where the
// chain goes is cheap to tabulate: a lane that starts at byte o of a 4 KiB
// segment and hops from element to element leaves the segment at some
// position, having produced some number of bytes.  Chains from neighbouring
// offsets merge within a few elements, so the 64 lanes of a wave mostly read
// the same bytes.  Segments -> super-segments (64 segments) -> one short
// sequential pass over the super-segments -> the exact (src, dst) positions
// of the element boundaries at every 64 KiB of output -> those pieces are
// decoded by k_decompress_streams like independent streams.  Anything
// irregular (an element the scan cannot follow, a piece that fails one of
// the reference's checks - which includes a copy reaching back into another
// piece) sets meta[2] and the whole stream is decoded by the sequential
// path instead, which also produces the reference's exact error.
// ---------------------------------------------------------------------
namespace {
constexpr unsigned long long kNone = ~0ull;
// k_stream_scan follows a chain at most this many elements behind its
// segment's end.  Compressible data lands on a tabulated offset (within
// kEntry = 8 bytes of a 4 KiB boundary) at the next boundary unless an
// element of 9+ encoded bytes jumps over it, which costs a segment of small
// elements (~1 300 hops of 3 bytes) until the next chance; incompressible
// data (64 KiB literals) lands with probability 1/512 per element.  8 192
// hops cover several misses in a row and all but (511/512)^8192 = 1e-7 of
// the literal walks, and bound what a crafted stream can cost to 16 hops
// per input byte (it was quadratic in the input length without a bound).
constexpr uint32_t kScanOverrun = 8192;

typedef unsigned long long su64x2 __attribute__((ext_vector_type(2)));

// hop over the element at p; false if it does not fit in the stream
__device__ __forceinline__ bool elem_step(gcptr in, uint64_t in_len,
                                          uint64_t &p, uint64_t &out)
{
    const uint32_t tag = in[p];
    const uint32_t type = tag & 3;
    if (type == 0) {
        const uint32_t n6 = tag >> 2;
        uint64_t len = n6 + 1, hd = 1;
        if (n6 >= 60) {
            const uint32_t nb = n6 - 59;
            if (p + 1 + nb > in_len)
                return false;
            uint32_t v = 0;
            for (uint32_t k = 0; k < nb; k++)
                v |= (uint32_t)in[p + 1 + k] << (8 * k);
            len = (uint64_t)v + 1;
            hd = 1 + nb;
        }
        if (in_len - (p + hd) < len)
            return false;
        p += hd + len;
        out += len;
    } else {
        const uint32_t cnb = type == 1 ? 1 : (type == 2 ? 2 : 4);
        if (p + 1 + cnb > in_len)
            return false;
        p += 1 + cnb;
        out += type == 1 ? 4 + ((tag >> 2) & 7) : 1 + (tag >> 2);
    }
    return true;
}

// level 1 = segments (4 KiB; 1 KiB for short streams: StreamArgs::seg_log2),
// 2 = 64 of them, 3 = 64 of those
static_assert(kSegPerSuper == 64, "level_bytes shifts by 6 per level");
template <int L>
__device__ __forceinline__ uint64_t level_bytes(const StreamArgs &a)
{
    return 1ull << (a.seg_log2 + 6 * (L - 1));
}
template <int L>
__device__ __forceinline__ su64x2 *level_table(const StreamArgs &a)
{
    return (su64x2 *)(L == 1 ? a.s1 : (L == 2 ? a.s2 : a.s3));
}
template <int L>
__device__ __forceinline__ su64x2 *level_entry(const StreamArgs &a)
{
    return (su64x2 *)(L == 1 ? a.e1 : (L == 2 ? a.e2 : a.e3));
}

// Follow the chain from p to the end of the level-L block that contains p
// (or of the stream).  Tables: level 1 [segment][o]: enter the segment at
// offset o < kEntry; levels 2, 3 [block][child][o]: enter the block at offset
// o < kEntry of its child (a block of the level below).  A position deeper than
// kEntry bytes inside a child - the chain landed there after a long element - is
// first taken to that child's end one level down.
template <int L>
__device__ __forceinline__ bool reach_end(const StreamArgs &a, uint64_t &p,
                                          uint64_t &out);
template <>
__device__ __forceinline__ bool reach_end<1>(const StreamArgs &a, uint64_t &p,
                                             uint64_t &out)
{
    const uint64_t seg = p >> a.seg_log2;
    const uint64_t o = p - (seg << a.seg_log2);
    if (o < kEntry) {
        const su64x2 e = level_table<1>(a)[seg * kEntry + o];
        if (e.x == kNone)
            return false;
        p = e.x;
        out += e.y;
        return true;
    }
    uint64_t end = (seg + 1) << a.seg_log2;
    if (end > a.in_len)
        end = a.in_len;
    while (p < end)
        if (!elem_step((gcptr)a.in, a.in_len, p, out))
            return false;
    return true;
}
template <int L>
__device__ __forceinline__ bool reach_end(const StreamArgs &a, uint64_t &p,
                                          uint64_t &out)
{
    const uint64_t B = level_bytes<L>(a), Bc = level_bytes<L - 1>(a);
    const uint64_t blk = p / B;
    uint64_t end = (blk + 1) * B;
    if (end > a.in_len)
        end = a.in_len;
    while (p < end) {
        const uint64_t rel = p - blk * B;
        const uint64_t c = rel / Bc, o = rel - c * Bc;
        if (o < kEntry) {
            const su64x2 e =
                level_table<L>(a)[(blk * kSegPerSuper + c) * kEntry + o];
            if (e.x == kNone)
                return false;
            p = e.x;
            out += e.y;
            return true;
        }
        if (!reach_end<L - 1>(a, p, out))
            return false;
    }
    return true;
}

// table of one level-L block (L = 2, 3), children right to left: entering at
// child c continues, after that child, with an entry already tabulated
template <int L>
__device__ __forceinline__ void build_level(const StreamArgs &a, uint32_t wg)
{
    __shared__ su64x2 tab[kSegPerSuper * kEntry]; // 16 KiB
    if (a.meta[2])
        return;
    const uint64_t B = level_bytes<L>(a), Bc = level_bytes<L - 1>(a);
    const uint64_t blk = wg;
    uint64_t end = (blk + 1) * B;
    if (end > a.in_len)
        end = a.in_len;
    for (int c = kSegPerSuper - 1; c >= 0; c--) {
        uint64_t p = blk * B + (uint64_t)c * Bc + threadIdx.x, out = 0;
        bool ok = p < a.in_len && reach_end<L - 1>(a, p, out);
        while (ok && p < end) {
            const uint64_t rel = p - blk * B;
            const uint64_t c2 = rel / Bc, o2 = rel - c2 * Bc;
            if (o2 < kEntry) {
                const su64x2 e = tab[c2 * kEntry + o2]; // c2 > c: done before
                ok = e.x != kNone;
                p = e.x;
                out += e.y;
                break;
            }
            ok = reach_end<L - 1>(a, p, out);
        }
        tab[c * kEntry + threadIdx.x] = (su64x2){ok ? p : kNone, out};
        __syncthreads();
    }
    su64x2 *dst = level_table<L>(a) + blk * kSegPerSuper * kEntry;
    for (uint32_t i = threadIdx.x; i < kSegPerSuper * kEntry; i += kEntry)
        dst[i] = tab[i];
}

// entries of level L-1 from the entries of level L: one lane per block of
// level L walks its children
template <int L>
__device__ __forceinline__ void spread_level(const StreamArgs &a, uint32_t wg)
{
    if (a.meta[2])
        return;
    const uint64_t blk = (uint64_t)wg * blockDim.x + threadIdx.x;
    const uint64_t B = level_bytes<L>(a), Bc = level_bytes<L - 1>(a);
    if (blk * B >= a.in_len)
        return;
    const su64x2 e = level_entry<L>(a)[blk];
    if (e.x == kNone)
        return; // a long element spans this block
    uint64_t p = e.x, out = e.y;
    uint64_t end = (blk + 1) * B;
    if (end > a.in_len)
        end = a.in_len;
    while (p < end) {
        level_entry<L - 1>(a)[p / Bc] = (su64x2){p, out};
        if (!reach_end<L - 1>(a, p, out)) {
            a.meta[2] = 1;
            return;
        }
    }
}
} // namespace

__device__ __forceinline__ void stream_head(const StreamArgs &a, const uint32_t wg)
{
    // header checks of Decoder::decompress (src/decompress.rs:75-95); any
    // failure is left to the sequential decoder, which reports it
    uint32_t hdr = 0;
    uint64_t dlen = 0;
    a.meta[0] = 0;
    a.meta[1] = 0;
    a.meta[2] = 1;
    a.meta[3] = 0;
    if (a.in_len == 0)
        return;
    if (read_header((gcptr)a.in, a.in_len, &hdr, &dlen, nullptr, 0) !=
        SNAPMI_OK)
        return;
    if (dlen > a.out_cap || (dlen + kStreamChunk - 1) / kStreamChunk > a.kmax)
        return;
    a.meta[0] = hdr;
    a.meta[1] = dlen;
    a.meta[2] = 0;
    a.meta[3] = (dlen + kStreamChunk - 1) / kStreamChunk;
}

// ---------------------------------------------------------------------
// k_stream_scan: the level-1 table, (exit, produced) for the kEntry entry
// offsets of every 4 KiB segment.
//
// Rounds 1-3 gave every (segment, entry) its own lane and let it hop through
// HBM: ~45 instructions and one dependent load per element, and the eight
// chains of a segment are ONE chain after a few elements - seven eighths of
// the hops were duplicates (8.1 ms of the 14.2 ms of a 2 GiB stream).  Now:
//   * the hop is two LDS reads and ~42 straight-line instructions: the tag
//     from the lane's window of the input (a ring of kHopLines lines; up to
//     kHopFetch lines are fetched WHILE the lanes hop through the ones that
//     are there - the round structure of k_match_blocks), then (encoded
//     bytes, produced bytes) from a 256-entry table of tags.  Literals with
//     length bytes - one per 64 KiB of incompressible data - are done between
//     the rounds;
//   * a wavefront owns 64 segments.  Phase A walks all 8 x 64 entries for
//     kHopMid bytes only; phase B walks ONE trunk per segment from where
//     entry 0 stood after phase A - entries that stood at the same place are
//     the trunk plus what they had produced - and the chains that had not
//     joined (chains through the bytes of a long literal never meet);
//   * a walk that leaves its segment off the landing zone goes on through
//     the next one - but kHopMid bytes into it, it stands as a rule where
//     that segment's own trunk started, and is that trunk from there on;
//   * walks are handed out lane by lane from a pool, so a lane that is done
//     takes the next walk instead of waiting for the longest.
// Positions are 32-bit offsets from the wavefront's first segment; a chain
// that a literal of a GiB takes out of that range finishes in the 64-bit
// loop of the old kernel (elem_step, through HBM).
// Measured (one 2 GiB stream, profiles/r4_stream_scan_steps.txt): 8.07 ms ->
// 2.0 ms; a wavefront is bound by the latency of its own dependent chain
// (alone on the chip or not), the lanes of a round are busy to ~55 %.
// ---------------------------------------------------------------------
#define SNAPMI_HOP_LINE 32
#define SNAPMI_HOP_ITERS 12
constexpr uint32_t kHopLine = SNAPMI_HOP_LINE; // bytes of a line
constexpr uint32_t kHopLines = 4;              // lines of a lane's ring
constexpr uint32_t kHopFetch = 2;              // lines fetched per round
// hops per round: a lane whose elements are larger than kHopFetch * kHopLine
// / kHopIters bytes on average runs out of window before the round is over
constexpr uint32_t kHopIters = SNAPMI_HOP_ITERS;
constexpr uint32_t kHopMid = 128;               // phase A walks this far
constexpr uint32_t kHopFar = 1u << 30;
constexpr uint32_t kHopNone = 0xFFFFFFFFu;

struct HopShared {
    l_u32 *win;  // [kHopLines * kHopLine / 4][64]: dword r of lane l at win[r * 64 + l]
    l_u16x *lut; // [256] encoded bytes | produced << 8; 0 = length bytes follow
    gcptr in;    // the stream
    uint64_t in_len;
    uint64_t wbase;  // stream offset of the wavefront's first segment
    uint32_t seg;    // segment size of this call (a power of two)
    uint32_t mis;    // (in + wbase) mod kHopLine: the window's lines are aligned
    uint32_t lastR;  // end of the stream from wbase, at most 2^31
    uint64_t lastP;  // the same + mis, exact
};

// one lane's walk
struct Hopper {
    uint32_t L0, nl;   // lines L0 .. L0 + nl - 1 of the aligned view are in the window
    uint32_t R;        // position from wbase
    uint32_t out;      // produced
    uint32_t over;     // elements hopped behind the segment's end
    uint32_t endR;     // the segment's end (or the stream's)
    uint32_t stopR;    // phase A pauses at the first element start here
    uint32_t stopOut;  // k_stream_cuts: ... at the first one that has produced this
    bool run;          // hopping
    bool slow;         // stands in front of a literal with length bytes
    bool fail;         // the chain cannot be followed
    bool punt;         // left the 32-bit range: p64 / out64 are its end
    uint64_t p64, out64;

    // (the loop condition of rounds 1-3: the exit is the first element start
    // at or behind the segment's end that lies within kEntry bytes of a
    // segment boundary, or the end of the stream)
    __device__ __forceinline__ bool more(const HopShared &h) const
    {
        return R < h.lastR && (R < endR || (R & (h.seg - 1)) >= kEntry);
    }
    __device__ __forceinline__ void start(const HopShared &h, uint32_t r,
                                          uint32_t o, uint32_t e, uint32_t st,
                                          bool valid)
    {
        L0 = 0; nl = 0; R = r; out = o; over = 0; endR = e; stopR = st;
        stopOut = kHopNone;
        slow = false; punt = false; p64 = 0; out64 = 0;
        fail = !valid;
        run = valid && more(h) && R < stopR;
    }
    // between rounds: the literal with length bytes the lane stands at
    // (src/decompress.rs:213-232), its bytes through HBM
    __device__ __forceinline__ void long_literal(const HopShared &h)
    {
        slow = false;
        const uint64_t p = h.wbase + R;
        const uint32_t tag = h.in[p];
        const uint32_t nb = (tag >> 2) - 59;
        if (p + 1 + nb > h.in_len) {
            fail = true;
            return;
        }
        uint32_t v = 0;
        for (uint32_t k = 0; k < nb; k++)
            v |= (uint32_t)h.in[p + 1 + k] << (8 * k);
        const uint64_t len = (uint64_t)v + 1;
        if (h.in_len - (p + 1 + nb) < len) {
            fail = true;
            return;
        }
        if (len + R >= kHopFar) {
            // out of the 32-bit range: the rest of the walk right here
            uint64_t q = p + 1 + nb + len, qo = (uint64_t)out + len;
            const uint64_t end = h.wbase + endR;
            bool ok = true;
            while (ok && q < h.in_len &&
                   (q < end || (q & (h.seg - 1)) >= kEntry)) {
                if (q >= end && ++over > kScanOverrun) {
                    ok = false;
                    break;
                }
                ok = elem_step(h.in, h.in_len, q, qo);
            }
            punt = true;
            fail = !ok;
            p64 = q;
            out64 = qo;
            return;
        }
        R += 1 + nb + (uint32_t)len;
        out += (uint32_t)len;
        run = more(h) && R < stopR && out < stopOut;
    }
    __device__ __forceinline__ void result(const HopShared &h, uint64_t &p,
                                           uint64_t &o) const
    {
        p = punt ? p64 : h.wbase + R;
        o = punt ? out64 : (uint64_t)out;
    }
};

// Rounds over a pool of `npool` walks: a lane without a walk takes the next
// one (init(item, w)), hops while its window reaches, and hands the walk to
// done(item, w) when it stands; done returns false to send it on (with a new
// stopR).
template <class INIT, class DONE>
__device__ __forceinline__ void hop_pool(const HopShared &h, uint32_t npool,
                                         INIT init, DONE done)
{
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    const uint32_t lane = threadIdx.x;
    const l_u8 *const winb = (const l_u8 *)(h.win + lane);
    uint32_t next = 0; // uniform: first walk not handed out
    bool busy = false;
    uint32_t item = 0;
    Hopper w;
    w.run = false;
    w.slow = false;
    w.nl = 0;
    w.L0 = 0;
    for (;;) {
        if (busy && !w.run && !w.slow) { // this lane's walk stands
            if (done(item, w)) // ... and is over
                busy = false;
        }
        {
            const uint64_t M = __ballot(!busy);
            const uint32_t mine =
                next + __builtin_amdgcn_mbcnt_hi(
                           (uint32_t)(M >> 32),
                           __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0));
            if (!busy && mine < npool) {
                item = mine;
                init(item, w);
                busy = true;
            }
            next += (uint32_t)__builtin_popcountll(M);
            if (next > npool)
                next = npool;
        }
        if (__ballot(w.slow)) {
            if (w.slow)
                w.long_literal(h);
        }
        if (!__ballot(busy))
            break;
        // ---- one round: fetch up to kHopFetch lines behind the window ------
        uint32_t want = 0, cnt = 0;
        if (w.run) {
            const uint32_t lp = (w.R + h.mis) / kHopLine;
            if (w.nl != 0 && lp >= w.L0 && lp < w.L0 + w.nl) {
                w.nl -= lp - w.L0; // the lines below are used up
                w.L0 = lp;
            } else {
                w.L0 = lp; // nothing of use: the lane sits this round out
                w.nl = 0;
            }
            want = w.L0 + w.nl;
            cnt = kHopLines - w.nl < kHopFetch ? kHopLines - w.nl : kHopFetch;
            // (a line that starts behind the stream's end is never fetched:
            // the stream's last line may be the last of the allocation)
            while (cnt && (uint64_t)(want + cnt - 1) * kHopLine >= h.lastP)
                cnt--;
        }
        u32x4 f[kHopFetch * kHopLine / 16];
#pragma unroll
        for (uint32_t i = 0; i < kHopFetch * kHopLine / 16; i++)
            f[i] = (u32x4){0, 0, 0, 0};
        {
            const g_u32x4 *g = (const g_u32x4 *)(h.in + h.wbase - h.mis +
                                                 (uint64_t)want * kHopLine);
#pragma unroll
            for (uint32_t c = 0; c < kHopFetch; c++)
                if (c < cnt) {
#pragma unroll
                    for (uint32_t i = 0; i < kHopLine / 16; i++)
                        f[c * (kHopLine / 16) + i] = g[c * (kHopLine / 16) + i];
                }
        }
        // ---- at most kHopIters hops through the lines that are there (LDS
        // only: the fetch is in flight).  Straight-line code: a divergent
        // region in this loop costs the scalar unit three instructions per
        // flag it carries.
        {
            uint32_t R = w.R, out = w.out, over = w.over;
            bool run = w.run, slow = false, fail = w.fail;
            const uint32_t lo = w.L0 * kHopLine - h.mis, span = w.nl * kHopLine;
            const uint32_t endR = w.endR, stopOut = w.stopOut;
            const uint32_t capR = w.stopR < h.lastR ? w.stopR : h.lastR;
            const uint32_t segm = h.seg - kEntry; // landing zone: none set
#define SNAPMI_HOP_UNROLL SNAPMI_HOP_ITERS
            for (uint32_t it = 0; it < kHopIters; it += SNAPMI_HOP_UNROLL) {
                // (one look at "can anyone hop" per SNAPMI_HOP_UNROLL hops - by
                // default per round: a lane that cannot, idles through them.
                // With a test and a branch per hop the compiler keeps the
                // loop's flags in step with three scalar instructions each,
                // 65 instructions per hop; unrolled it is 42, and the next
                // tag is on its way while the flags of this one are computed:
                // 3.39 -> 2.03 ms)
                if (!__builtin_amdgcn_ballot_w64(run && R - lo < span))
                    break;
#pragma unroll
                for (uint32_t u = 0; u < SNAPMI_HOP_UNROLL; u++) {
                    const bool can = run && R - lo < span;
                    const uint32_t x =
                        (R + h.mis) & (kHopLines * kHopLine - 1);
                    const uint32_t tag = winb[((x & ~3u) << 6) + (x & 3)];
                    const uint32_t lv = h.lut[tag];
                    const uint32_t e = can ? lv : 0; // waiting: no step
                    over += (can && R >= endR) ? 1u : 0u;
                    R += e & 0xFF;
                    out += e >> 8;
                    slow = slow || (can && lv == 0);
                    // (the element behind the limit is hopped, then the walk
                    // fails)
                    fail = fail || over > kScanOverrun || R > h.lastR;
                    run = run && !slow && !fail && R < capR &&
                          out < stopOut &&
                          (R < endR || (R & segm) != 0);
                }
            }
            w.R = R;
            w.out = out;
            w.over = over;
            w.run = run;
            w.slow = slow;
            w.fail = fail;
        }
        // ---- the fetched lines into the ring ------------------------------
#pragma unroll
        for (uint32_t c = 0; c < kHopFetch; c++)
            if (c < cnt) {
                l_u32 *dst = h.win +
                             ((want + c) & (kHopLines - 1)) * (kHopLine / 4) * 64 +
                             lane;
#pragma unroll
                for (uint32_t i = 0; i < kHopLine / 16; i++) {
                    const u32x4 v = f[c * (kHopLine / 16) + i];
                    dst[(4 * i + 0) * 64] = v.x;
                    dst[(4 * i + 1) * 64] = v.y;
                    dst[(4 * i + 2) * 64] = v.z;
                    dst[(4 * i + 3) * 64] = v.w;
                }
            }
        w.nl += cnt;
    }
}

// the table of tags: encoded bytes | produced bytes << 8, 0 for a literal
// with length bytes
__device__ __forceinline__ void hop_lut(uint16_t *lutbuf)
{
    for (uint32_t t = threadIdx.x; t < 256; t += 64) {
        const uint32_t type = t & 3, n6 = t >> 2;
        uint32_t enc, prod;
        if (type == 0) {
            enc = n6 < 60 ? n6 + 2 : 0;
            prod = n6 < 60 ? n6 + 1 : 0;
        } else {
            enc = type == 1 ? 2 : (type == 2 ? 3 : 5);
            prod = type == 1 ? 4 + (n6 & 7) : 1 + n6;
        }
        lutbuf[t] = (uint16_t)(enc | (prod << 8));
    }
    __syncthreads();
}

__device__ __forceinline__ void stream_scan(const StreamArgs &a, const uint32_t wg)
{
    __shared__ uint32_t winbuf[kHopLines * (kHopLine / 4) * 64];
    __shared__ uint16_t lutbuf[256];
    __shared__ uint32_t midR[64 * kEntry], midO[64 * kEntry];
    __shared__ uint16_t todo[64 * kEntry];
    __shared__ unsigned long long trunkP[64], trunkO[64];
    __shared__ uint32_t todoO[64 * (kEntry - 1)];
    __shared__ uint8_t trunkL[64], todoL[64 * (kEntry - 1)];
    if (a.meta[2])
        return;
    const uint32_t lane = threadIdx.x;
    HopShared h;
    h.win = (l_u32 *)winbuf;
    h.lut = (l_u16x *)lutbuf;
    h.in = (gcptr)a.in;
    h.in_len = a.in_len;
    const uint32_t S = a.scan_segs; // segments of this wavefront: 8 .. 64
    const uint64_t seg0 = (uint64_t)wg * S;
    const uint32_t sl2 = a.seg_log2; // (the wavefront's S segments: < 2^31)
    h.seg = 1u << sl2;
    h.wbase = seg0 << sl2;
    h.mis = (uint32_t)((uintptr_t)a.in + h.wbase) & (kHopLine - 1);
    // the grid covers nseg + 1 sentinel: its last workgroup can begin behind
    // the input (lastR = 0 then: every walk is invalid and reads nothing)
    const uint64_t left = a.in_len > h.wbase ? a.in_len - h.wbase : 0;
    h.lastR = left < (1ull << 31) ? (uint32_t)left : 1u << 31;
    h.lastP = left + h.mis;
    const uint32_t nloc =
        a.nseg - seg0 < S ? (uint32_t)(a.nseg - seg0) : S; // segments here
    hop_lut(lutbuf);
    su64x2 *const table = level_table<1>(a) + seg0 * kEntry;
    const uint32_t lastR = h.lastR;
    auto seg_end = [lastR, sl2](uint32_t sl) {
        return ((sl + 1) << sl2) < lastR ? ((sl + 1) << sl2) : lastR;
    };

    // ---- phase A: every entry, kHopMid bytes far --------------------------
    hop_pool(
        h, nloc * kEntry,
        [&](uint32_t it, Hopper &w) {
            const uint32_t sl = it / kEntry,
                           r0 = (sl << sl2) + it % kEntry;
            w.start(h, r0, 0, seg_end(sl), (sl << sl2) + kHopMid, r0 < lastR);
        },
        [&](uint32_t it, Hopper &w) {
            const bool paused = !w.fail && !w.punt && w.more(h);
            midR[it] = paused ? w.R : kHopNone;
            midO[it] = w.out;
            if (!paused) { // the chain is over, or cannot be followed
                uint64_t p, o;
                w.result(h, p, o);
                table[it] = (su64x2){w.fail ? kNone : p, o};
            }
            return true;
        });
    __syncthreads();

    // ---- the chains that stand elsewhere than their segment's entry 0 -----
    uint32_t ntodo = 0;
    for (uint32_t o = 1; o < kEntry; o++) {
        const uint32_t it = lane * kEntry + o;
        const uint32_t r = lane < nloc ? midR[it] : kHopNone;
        const bool own = r != kHopNone && r != midR[lane * kEntry];
        const uint64_t M = __ballot(own);
        if (own)
            todo[ntodo + __builtin_amdgcn_mbcnt_hi(
                             (uint32_t)(M >> 32),
                             __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0))] =
                (uint16_t)it;
        ntodo += (uint32_t)__builtin_popcountll(M);
    }
    __syncthreads();

    // ---- phase B: one trunk per segment, and those ------------------------
    // A walk that leaves its segment off the landing zone goes on through the
    // next one (30 % of the trunks on text, and again with the same odds: the
    // longest of 64 walks was 4 600 hops where the average is 1 190) - but
    // kHopMid bytes into that segment it stands, as a rule, where that
    // segment's own trunk started: then it IS that trunk from there on, and
    // its result is a sum.  Links go to higher segments only; they are
    // resolved from the last segment down.
    for (uint32_t i = lane; i < 64; i += 64)
        trunkL[i] = 0xFF;
    auto next_stop = [nloc, sl2](uint32_t sl) {
        return sl + 1 < nloc ? ((sl + 1) << sl2) + kHopMid : kHopNone;
    };
    hop_pool(
        h, nloc + ntodo,
        [&](uint32_t it, Hopper &w) {
            // (the last segment's trunk first: it has nobody to join)
            const uint32_t e =
                it < nloc ? (nloc - 1 - it) * kEntry : todo[it - nloc];
            const uint32_t r = midR[e];
            // (a trunk counts from 0: its entries add what they had)
            w.start(h, r, it < nloc ? 0 : midO[e], seg_end(e / kEntry),
                    next_stop(e / kEntry), r != kHopNone);
        },
        [&](uint32_t it, Hopper &w) {
            uint32_t link = 0xFF;
            if (!w.fail && !w.punt && w.more(h)) { // stands at a stop
                const uint32_t t = w.R >> sl2;
                if (t < nloc && midR[t * kEntry] == w.R) {
                    link = t;
                } else {
                    w.stopR = next_stop(t);
                    w.run = true;
                    return false;
                }
            }
            uint64_t p, o;
            w.result(h, p, o);
            if (it < nloc) {
                const uint32_t sl = nloc - 1 - it;
                trunkP[sl] = w.fail ? kNone : p;
                trunkO[sl] = o;
                trunkL[sl] = (uint8_t)link;
            } else if (link == 0xFF) {
                table[todo[it - nloc]] = (su64x2){w.fail ? kNone : p, o};
            } else {
                todoO[it - nloc] = w.out;
            }
            if (it >= nloc)
                todoL[it - nloc] = (uint8_t)link;
            return true;
        });
    __syncthreads();
    if (lane == 0) {
        for (uint32_t sl = nloc; sl-- > 0;) {
            const uint32_t t = trunkL[sl];
            if (t != 0xFF) {
                trunkP[sl] = trunkP[t];
                trunkO[sl] += trunkO[t];
            }
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < ntodo; i += 64) {
        const uint32_t t = todoL[i];
        if (t != 0xFF)
            table[todo[i]] = (su64x2){trunkP[t], todoO[i] + trunkO[t]};
    }
    if (lane < nloc) {
        const uint32_t t0 = midR[lane * kEntry];
        for (uint32_t o = 0; o < kEntry; o++) {
            const uint32_t it = lane * kEntry + o;
            if (t0 != kHopNone && midR[it] == t0)
                table[it] = (su64x2){trunkP[lane], trunkO[lane] + midO[it]};
        }
    }
}

// the one sequential pass: a table lookup per 16 MiB of input
__device__ __forceinline__ void stream_chain(const StreamArgs &a, const uint32_t wg)
{
    if (a.meta[2])
        return;
    uint64_t p = a.meta[0], out = 0;
    bool ok = true;
    while (ok && p < a.in_len) {
        level_entry<3>(a)[p / level_bytes<3>(a)] = (su64x2){p, out};
        ok = reach_end<3>(a, p, out);
    }
    // the elements must end with the stream and produce the announced
    // length (src/decompress.rs:141-147)
    if (!ok || p != a.in_len || out != a.meta[1])
        a.meta[2] = 1;
}

// The element boundary at (or first behind) every 64 KiB of output: the
// segments whose stretch of the chain holds such a boundary are collected -
// kCutSegs segments per wavefront, one in eight on text - and walked from
// where the chain enters them by the pool of k_stream_scan (windows in LDS),
// a lane per segment, standing at every boundary on the way.  (Rounds 1-3:
// one lane per segment hopping through HBM, the wavefront as slow as its
// slowest lane: 1.7 ms of a 2 GiB stream.)
__device__ __forceinline__ void stream_cuts(const StreamArgs &a, const uint32_t wg)
{
    __shared__ uint32_t winbuf[kHopLines * (kHopLine / 4) * 64];
    __shared__ uint16_t lutbuf[256];
    __shared__ uint16_t todo[kCutSegs];
    if (a.meta[2])
        return;
    const uint32_t lane = threadIdx.x;
    const uint64_t seg0 = (uint64_t)wg * kCutSegs;
    if (seg0 == 0 && lane == 0) {
        const uint64_t K = a.meta[3];
        a.cuts[0] = a.meta[0];
        a.cuts[1] = 0;
        a.cuts[K * 2] = a.in_len;
        a.cuts[K * 2 + 1] = a.meta[1];
    }
    HopShared h;
    h.win = (l_u32 *)winbuf;
    h.lut = (l_u16x *)lutbuf;
    h.in = (gcptr)a.in;
    h.in_len = a.in_len;
    const uint32_t sl2 = a.seg_log2;
    h.seg = 1u << sl2;
    h.wbase = seg0 << sl2;
    h.mis = (uint32_t)((uintptr_t)a.in + h.wbase) & (kHopLine - 1);
    const uint64_t left = a.in_len > h.wbase ? a.in_len - h.wbase : 0;
    h.lastR = left < (1ull << 31) ? (uint32_t)left : 1u << 31;
    h.lastP = left + h.mis;
    hop_lut(lutbuf);
    const uint64_t dlen = a.meta[1];
    // ---- the segments with a boundary in their stretch --------------------
    uint32_t ntodo = 0;
    for (uint32_t j = 0; j < kCutSegs; j += 64) {
        const uint64_t seg = seg0 + j + lane;
        bool has = false;
        if ((seg << sl2) < a.in_len) {
            const su64x2 e = level_entry<1>(a)[seg];
            if (e.x != kNone) {
                uint64_t np = e.x, nout = e.y;
                if (!reach_end<1>(a, np, nout)) {
                    a.meta[2] = 1;
                } else {
                    const uint64_t t = (e.y / kStreamChunk + 1) * kStreamChunk;
                    has = t <= nout && t < dlen;
                }
            }
        }
        const uint64_t M = __ballot(has);
        if (has)
            todo[ntodo + __builtin_amdgcn_mbcnt_hi(
                             (uint32_t)(M >> 32),
                             __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0))] =
                (uint16_t)(j + lane);
        ntodo += (uint32_t)__builtin_popcountll(M);
    }
    __syncthreads();
    // ---- walk them: a stop at every boundary ------------------------------
    uint64_t out0 = 0, nout = 0, k = 0; // of this lane's walk
    const uint32_t lastR = h.lastR;
    hop_pool(
        h, ntodo,
        [&](uint32_t it, Hopper &w) {
            const uint32_t sl = todo[it];
            const su64x2 e = level_entry<1>(a)[seg0 + sl];
            uint64_t np = e.x;
            out0 = e.y;
            nout = e.y;
            reach_end<1>(a, np, nout); // (true: it was, above)
            k = out0 / kStreamChunk + 1;
            const uint32_t endR =
                ((sl + 1) << sl2) < lastR ? ((sl + 1) << sl2) : lastR;
            w.start(h, (uint32_t)(e.x - h.wbase), 0, endR, kHopNone, true);
            w.stopOut = (uint32_t)(k * kStreamChunk - out0);
            w.run = w.run && w.out < w.stopOut;
        },
        [&](uint32_t it, Hopper &w) {
            const uint64_t qo = out0 + w.out;
            if (w.fail || w.punt || qo < k * kStreamChunk) {
                // cannot be followed here (or a literal of a GiB): the
                // sequential decoder owns the stream
                a.meta[2] = 1;
                return true;
            }
            // (a long element can cover several boundaries)
            while (k * kStreamChunk <= qo && k * kStreamChunk < dlen) {
                a.cuts[k * 2] = h.wbase + w.R;
                a.cuts[k * 2 + 1] = qo;
                k++;
            }
            if (k * kStreamChunk <= nout && k * kStreamChunk < dlen) {
                w.stopOut = (uint32_t)(k * kStreamChunk - out0);
                w.run = w.more(h);
                if (w.run)
                    return false;
                a.meta[2] = 1; // (the stretch ends before its last boundary)
            }
            return true;
        });
}

__device__ __forceinline__ void stream_pieces(const StreamArgs &a, const uint32_t wg)
{
    const uint64_t k = (uint64_t)wg * blockDim.x + threadIdx.x;
    if (k >= a.kmax)
        return;
    const bool live = a.meta[2] == 0 && k < a.meta[3];
    const uint64_t s0 = live ? a.cuts[k * 2] : 0,
                   d0 = live ? a.cuts[k * 2 + 1] : 0;
    const uint64_t s1 = live ? a.cuts[k * 2 + 2] : 0,
                   d1 = live ? a.cuts[k * 2 + 3] : 0;
    a.c_in[k] = a.in + s0;
    a.c_inlen[k] = s1 - s0;
    a.c_out[k] = a.out + d0;
    a.c_cap[k] = d1 - d0;
    a.c_mode[k] = 2;
    a.c_outlen[k] = 0;
    a.c_err[k].kind = SNAPMI_OK;
}

__device__ __forceinline__ void stream_finish(const StreamArgs &a, const uint32_t wg)
{
    __shared__ uint32_t bad;
    if (threadIdx.x == 0)
        bad = a.meta[2] != 0;
    __syncthreads();
    const uint64_t K = a.meta[3];
    if (!bad)
        for (uint64_t k = threadIdx.x; k < K; k += blockDim.x)
            if (a.c_err[k].kind != SNAPMI_OK || a.c_outlen[k] != a.c_cap[k])
                atomicOr(&bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.meta[2] = bad; // 1: the sequential decoder runs next
        if (a.fb_mode) // in a batch: 0 = decode it in the launch behind, 3 = not
            *a.fb_mode = bad ? 0 : 3;
        if (!bad) {
            a.out_len[0] = a.meta[1];
            set_error(a.err, 0, SNAPMI_OK, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------
// The kernels: for ONE stream (snapmi_decompress_stream), and for the long
// streams of a batch (snapmi_decompress_batch, few streams: every long stream
// gets its pieces instead of one wavefront) - the same bodies, a workgroup of
// the batched launch first finds its stream.
// ---------------------------------------------------------------------
// b.pre[s] = first workgroup of stream s in this launch, pre[L] = all of them;
// pre == nullptr: one workgroup per stream
__device__ __forceinline__ bool batch_find(const BatchStreams &b, uint32_t &s,
                                           uint32_t &wg)
{
    const uint32_t g = blockIdx.x;
    if (!b.pre) {
        s = g;
        wg = 0;
        return g < b.n;
    }
    if (g >= b.pre[b.n])
        return false;
    uint32_t lo = 0, hi = b.n; // pre[lo] <= g < pre[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (b.pre[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    s = lo;
    wg = g - b.pre[lo];
    return true;
}
#define SNAPMI_STREAM_KERNEL(name, bounds, call)                              \
    __global__ __launch_bounds__(bounds) void k_stream_##name(StreamArgs a)   \
    {                                                                         \
        const uint32_t wg = blockIdx.x;                                       \
        call;                                                                 \
    }                                                                         \
    __global__ __launch_bounds__(bounds) void k_bstream_##name(BatchStreams b) \
    {                                                                         \
        uint32_t s_, wg;                                                      \
        if (!batch_find(b, s_, wg))                                           \
            return;                                                           \
        const StreamArgs a = b.descs[s_];                                     \
        call;                                                                 \
    }
SNAPMI_STREAM_KERNEL(head, 64, stream_head(a, wg))
SNAPMI_STREAM_KERNEL(scan, 64, stream_scan(a, wg))
SNAPMI_STREAM_KERNEL(super, kEntry, build_level<2>(a, wg))
SNAPMI_STREAM_KERNEL(super3, kEntry, build_level<3>(a, wg))
SNAPMI_STREAM_KERNEL(chain, 64, stream_chain(a, wg))
SNAPMI_STREAM_KERNEL(spread3, 64, spread_level<3>(a, wg))
SNAPMI_STREAM_KERNEL(spread2, 64, spread_level<2>(a, wg))
SNAPMI_STREAM_KERNEL(cuts, 64, stream_cuts(a, wg))
SNAPMI_STREAM_KERNEL(pieces, 256, stream_pieces(a, wg))
SNAPMI_STREAM_KERNEL(finish, 1024, stream_finish(a, wg))

// The long streams of a batch (long_stream_rule) whose header announces no
// more than the caller's buffer holds (anything else is left to the wavefront
// decoder, which names the error).  modes[i] = 3 for them (the
// batch's own launch skips them), 0 for the others.
__global__ __launch_bounds__(1024) void k_long_plan(
    const void *const *in_ptrs, const uint64_t *in_lens, void *const *out_ptrs,
    const uint64_t *out_caps, uint32_t n, uint64_t min_len, uint8_t *modes,
    LongItem *list, uint32_t cap, uint32_t *count)
{
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t len = in_lens[i];
        uint8_t mode = 0;
        if (len >= min_len) {
            uint64_t dl = 0;
            const uint32_t hdr = read_varint((gcptr)in_ptrs[i], len, &dl);
            if (hdr && dl <= out_caps[i] && long_stream_rule(len, dl, min_len)) {
                const uint32_t slot = atomicAdd(count, 1u);
                if (slot < cap) {
                    LongItem it;
                    it.idx = i;
                    it.pad = 0;
                    it.in_len = len;
                    it.dlen = dl;
                    it.in = in_ptrs[i];
                    it.out = out_ptrs[i];
                    it.out_cap = out_caps[i];
                    list[slot] = it;
                    mode = 3;
                }
            }
        }
        modes[i] = mode;
    }
}

} // namespace snapmi
