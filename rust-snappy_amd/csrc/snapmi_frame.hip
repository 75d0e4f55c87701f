// snapmi_frame.hip -- Snappy frame format on the device (gfx950).
//
// Reference: src/frame.rs (chunk layout, compress_frame), src/crc32.rs
// (masked CRC32C), src/write.rs (chunking of write::FrameEncoder),
// src/read.rs (read::FrameDecoder state machine).  A framed stream is a
// 10-byte stream identifier followed by chunks
//     type(1) | len24 LE (= 4 + payload) | masked crc32c(uncompressed) LE | payload
// and every data chunk (<= 64 KiB of input) is an independent raw stream, so
// the chunk is the parallel unit: the raw codec kernels run unchanged on a
// batch whose "streams" are the chunks.
//
// Kernels here:
//   k_crc32c          masked CRC32C, one wavefront per buffer, 64 interleaved
//                     partial CRCs advanced 256 bytes per step with 4 LDS
//                     table lookups, combined by GF(2) multiplication
//   k_frame_chunks    compress side: chunk descriptors for the raw compressor
//   k_frame_sizes     compressed-vs-stored decision (src/frame.rs:85) + sizes
//   k_scan_u64        exclusive scan (one workgroup)
//   k_frame_emit      headers + payloads into the framed stream
//   k_frame_walk      decode side: sequential walk over the chunk headers
//                     (the format has no index) with the reference's checks
//   k_frame_index     decode side with a side index: one thread per chunk
//   k_frame_lens      decompressed length of every data chunk
//   k_frame_desc      descriptors for the raw decompressor
//   k_frame_verify    CRC comparison, first error in stream order
#include <hip/hip_runtime.h>

#include <string.h>
#include <vector>

#include <chrono>

#include "snapmi.h"
#include "snapmi_ctx.hpp"
#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"

using namespace snapmi;

namespace snapmi {

constexpr uint32_t kCrcPoly = 0x82F63B78u;      // reference build.rs:6
constexpr uint32_t kMaxChunk = 76490;           // reference src/frame.rs:12
constexpr uint32_t kFrameSlot = 76496;          // kMaxChunk rounded to 16

// Tables for k_crc32c, generated once per context on the host (the
// reference generates its tables at build time, build.rs:97-124).
struct CrcTables {
    uint32_t z256[4][256]; // state advanced by 256 zero bytes, per input byte
    uint32_t lane_mul[64]; // x^(8*4*(64-j)) mod P: final advance of lane j
    uint32_t init_adv[65537]; // 0xFFFFFFFF advanced by n zero bytes
};

struct FrameChunk { // one data chunk of a framed stream (decode side)
    uint64_t payload_off; // offset of the payload in the stream
    uint32_t payload_len;
    uint32_t crc;   // stored masked crc
    uint32_t type;  // 0 compressed, 1 stored
    uint32_t pad;
};

// GF(2) polynomial product mod P, reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t gf_mul(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0)
                break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// ---------------------------------------------------------------------
// K4: masked CRC32C of n buffers (<= 65536 bytes each), one wavefront each.
// The message is zero-padded at the FRONT to a multiple of 256 bytes (leading
// zeros do not move a zero state); lane j owns dwords j, j+64, ... and keeps
// a_j = Z256(a_j) ^ dword, where Z256 = "advance 256 zero bytes" through four
// 256-entry tables in LDS.  The 64 partial states are advanced to the end
// (one GF(2) product each), XOR-reduced, and the contribution of the initial
// state 0xFFFFFFFF is added from a table.  Equals crc32c_slice16 /
// the SSE4.2 instruction of reference src/crc32.rs:59-111; mask :35-38.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_crc32c(const void *const *ptrs,
                                               const uint64_t *lens,
                                               uint32_t *out, uint32_t n,
                                               const CrcTables *tab)
{
    __shared__ uint32_t z[4 * 256];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x;
    if (i >= n)
        return;
    for (uint32_t k = lane; k < 1024; k += 64)
        z[k] = tab->z256[k >> 8][k & 255];
    __syncthreads();
    gcptr m = (gcptr)ptrs[i];
    const uint32_t len = (uint32_t)lens[i];
    const uint32_t pad = (256 - (len & 255)) & 255;
    const uint32_t steps = (len + pad) >> 8;
    uint32_t a = 0;
    for (uint32_t t = 0; t < steps; t++) {
        const uint32_t pos = 256 * t + 4 * lane; // in the padded message
        uint32_t w;
        if (pos >= pad) {
            w = ld32u(m + (pos - pad));
        } else if (pos + 4 <= pad) {
            w = 0;
        } else { // the dword that straddles the start of the data
            w = 0;
            for (uint32_t b = pad - pos; b < 4; b++)
                w |= (uint32_t)m[pos + b - pad] << (8 * b);
        }
        a = z[a & 255] ^ z[256 + ((a >> 8) & 255)] ^
            z[512 + ((a >> 16) & 255)] ^ z[768 + (a >> 24)] ^ w;
    }
    uint32_t f = gf_mul(tab->lane_mul[lane], a);
    for (uint32_t o = 32; o; o >>= 1)
        f ^= __shfl_xor(f, o);
    if (lane == 0) {
        const uint32_t c = ~(f ^ tab->init_adv[len]);
        out[i] = ((c >> 15) | (c << 17)) + 0xA282EAD8u; // src/crc32.rs:37
    }
}

// ---------------------------------------------------------------------
// compress side
// ---------------------------------------------------------------------
struct FrameCompressArgs {
    const uint8_t *in;
    uint64_t in_len;
    uint8_t *out;
    uint64_t out_cap;
    uint64_t *out_len;       // [1]
    uint64_t *chunk_offsets; // optional [n+1]
    uint32_t n;              // chunks of the whole stream
    // The stream is processed in segments of `cnt` chunks starting at chunk
    // `lo`, so the per-chunk scratch (one 76 KiB slot each) stays bounded;
    // the arrays below are per segment, `base` carries the output offset.
    uint32_t lo, cnt;
    uint64_t *base; // [1] payload bytes emitted by earlier segments
    // optional [n+1]: input offset of every chunk (snapmi_frame_compress_chunks:
    // the caller decides where chunks end); nullptr = 65536-byte multiples
    const uint64_t *chunk_in_off;
    uint32_t ident; // 10 when the stream identifier is written, 0 when not
    // scratch
    const void **in_ptrs;
    uint64_t *in_lens;
    void **slot_ptrs;
    uint64_t *clens;   // raw-compressed length of every chunk
    uint32_t *crcs;
    uint64_t *sizes;   // 8 + payload
    uint64_t *offs;    // exclusive scan of sizes, [n+1]
    uint8_t *slots;
};

__global__ void k_frame_chunks(FrameCompressArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.cnt)
        return;
    uint64_t off, len;
    if (a.chunk_in_off) {
        off = a.chunk_in_off[a.lo + i];
        len = a.chunk_in_off[a.lo + i + 1] - off;
    } else {
        off = (uint64_t)(a.lo + i) * kMaxBlock;
        len = a.in_len - off < kMaxBlock ? a.in_len - off : kMaxBlock;
    }
    a.in_ptrs[i] = a.in + off;
    a.in_lens[i] = len;
    a.slot_ptrs[i] = a.slots + (uint64_t)i * kFrameSlot;
}

// reference compress_frame, src/frame.rs:83-89
__global__ void k_frame_sizes(FrameCompressArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.cnt)
        return;
    const uint64_t len = a.in_lens[i];
    const uint64_t clen = a.clens[i];
    const bool stored = clen >= len - len / 8;
    a.sizes[i] = 8 + (stored ? len : clen);
}

// exclusive scan of n u64 values, out[n] = total; one workgroup
__global__ __launch_bounds__(1024) void k_scan_u64(const uint64_t *in,
                                                   uint64_t *out, uint32_t n)
{
    __shared__ uint64_t wave_tot[16];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t x = i < n ? in[i] : 0;
        uint64_t sx = x;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(sx, o);
            if (lane >= o)
                sx += t;
        }
        if (lane == 63)
            wave_tot[w] = sx;
        __syncthreads();
        uint64_t before = 0, all = 0;
        for (uint32_t j = 0; j < (blockDim.x >> 6); j++) {
            const uint64_t t = wave_tot[j];
            if (j < w)
                before += t;
            all += t;
        }
        __syncthreads();
        if (i < n)
            out[i] = carry + before + sx - x;
        carry += all;
    }
    if (threadIdx.x == 0)
        out[n] = carry;
}

// one workgroup per chunk: header (src/frame.rs:91-93) + payload
__global__ __launch_bounds__(256) void k_frame_emit(FrameCompressArgs a)
{
    const uint32_t i = blockIdx.x;   // chunk of this segment
    const uint32_t gi = a.lo + i;    // chunk of the stream
    const uint64_t len = a.in_lens[i];
    const uint64_t clen = a.clens[i];
    const bool stored = clen >= len - len / 8;
    const uint64_t payload = stored ? len : clen;
    const uint64_t at = a.ident + a.base[0] + a.offs[i];
    gptr o = (gptr)a.out + at;
    if (gi == 0 && a.ident && threadIdx.x < 10) {
        const uint8_t ident[10] = {0xFF, 0x06, 0x00, 0x00, 's',
                                   'N',  'a',  'P',  'p',  'Y'};
        ((gptr)a.out)[threadIdx.x] = ident[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        const uint32_t cl = 4 + (uint32_t)payload;
        const uint32_t crc = a.crcs[i];
        o[0] = stored ? 0x01 : 0x00;
        o[1] = (uint8_t)cl;
        o[2] = (uint8_t)(cl >> 8);
        o[3] = (uint8_t)(cl >> 16);
        o[4] = (uint8_t)crc;
        o[5] = (uint8_t)(crc >> 8);
        o[6] = (uint8_t)(crc >> 16);
        o[7] = (uint8_t)(crc >> 24);
        const uint64_t end = a.ident + a.base[0] + a.offs[a.cnt];
        if (a.chunk_offsets) {
            a.chunk_offsets[gi] = at;
            if (gi + 1 == a.n)
                a.chunk_offsets[a.n] = end;
        }
        if (gi + 1 == a.n)
            a.out_len[0] = end;
    }
    gcptr from = stored ? (gcptr)a.in_ptrs[i]
                        : (gcptr)(a.slots + (uint64_t)i * kFrameSlot);
    gptr to = o + 8;
    for (uint64_t k = 4 * threadIdx.x; k + 4 <= payload; k += 4 * blockDim.x)
        st32u(to + k, ld32u(from + k));
    const uint64_t t = payload & ~3ull;
    if (threadIdx.x < (payload & 3))
        to[t + threadIdx.x] = from[t + threadIdx.x];
}

// after a segment's emit: carry its bytes into the next segment's offsets
__global__ void k_frame_advance(FrameCompressArgs a)
{
    a.base[0] += a.offs[a.cnt];
}

// ---------------------------------------------------------------------
// decode side
// ---------------------------------------------------------------------
struct FrameDecodeArgs {
    const uint8_t *in;
    uint64_t in_len;
    uint8_t *out; // may be nullptr (lengths only)
    uint64_t out_cap;
    uint64_t *out_len; // [1]
    snapmi_error *err; // [1]
    const uint64_t *index; // optional chunk header offsets
    uint32_t n_index;
    uint32_t cap_chunks; // capacity of chunks[]
    uint32_t flags;      // SNAPMI_FRAME_CONTINUATION: identifier already seen
    // first 10 bytes of the reference decoder's `src` scratch buffer when
    // this call starts (all zero for a fresh decoder): see frame_short_varint
    uint8_t stale[10];
    // scratch
    FrameChunk *chunks;
    uint32_t *meta;      // [0] data chunks found, [1] overflow flag,
                         // [2] index of the data chunk a structural error
                         //     precedes (0xFFFFFFFF = none), [3] the side
                         //     index met a chunk only the walk can judge
    snapmi_error *serr;  // [1] structural error of the walk
    uint64_t *dlens;     // [n] decompressed length per data chunk
    uint64_t *offs;      // [n+1]
    snapmi_error *cerrs; // [n] per-chunk errors (length stage, decode stage)
    const void **in_ptrs;
    uint64_t *in_lens;
    void **out_ptrs;
    uint64_t *out_caps;
    uint64_t *out_lens;
    uint8_t *modes;
    uint32_t *crcs; // computed
    // parallel header walk (k_fw_*): per 32 MiB segment of the stream, the
    // chains that start at a plausible header inside its first 76 494 bytes
    // and reach the segment's end: {entry, exit, data chunks on the way}
    unsigned long long *fw_cand; // [nseg * kFwCands * 3]
    uint32_t *fw_ncand;          // [nseg]
    unsigned long long *fw_entry; // [nseg + 1] the real chain's entry per segment
    uint32_t *fw_base;           // [nseg + 1] data chunks in front of the segment
    uint32_t nseg;
    unsigned long long fw_seg; // segment bytes (32 MiB; smaller in tests)
};

__device__ inline void walk_fail(const FrameDecodeArgs &a, uint32_t n_data,
                                 int kind, uint64_t fa, uint64_t fb)
{
    a.serr[0].kind = kind;
    a.serr[0].reserved = 0;
    a.serr[0].a = fa;
    a.serr[0].b = fb;
    a.serr[0].c = 0;
    a.meta[0] = n_data;
    a.meta[2] = n_data;
}

// A compressed chunk whose payload is shorter than 10 bytes and holds no
// varint terminator (every byte >= 0x80; an empty payload too).  The
// reference calls decompress_len on its WHOLE 76 490-byte scratch buffer
// `src` (src/read.rs:216), so the varint continues into whatever earlier
// reads left there: this chunk's own 4 header bytes at src[0..4) (read.rs:118),
// and the bodies of earlier stream-identifier / skippable / padding /
// compressed chunks (read.rs:151,157,168,214; stored chunks go to `dst`).
// The outcome then is TooBig, UnsupportedChunkLength, or - when the phantom
// length is acceptable - whatever Decoder::decompress says about the real
// payload: Empty or Header (src/decompress.rs:80-83).  Rare and always an
// error, so the model of src[0..10) is rebuilt here by walking the stream
// again from its start up to `stop` (the offset of this chunk's header).
__device__ inline void frame_short_varint(const FrameDecodeArgs &a,
                                          uint64_t stop, uint32_t n_data)
{
    gcptr in = (gcptr)a.in;
    uint8_t m[10];
    for (int k = 0; k < 10; k++)
        m[k] = a.stale[k];
    uint64_t r = 0;
    for (;;) { // every chunk before `stop` was accepted by the walk
        const uint32_t hd = ld32u(in + r);
        for (int k = 0; k < 4; k++)
            m[k] = (uint8_t)(hd >> (8 * k));
        const uint32_t ty = hd & 0xFF;
        const uint64_t len = hd >> 8;
        uint64_t body = r + 4, blen = len; // bytes read into src[0..blen)
        if (ty == 0x00) {
            body = r + 8;
            blen = len - 4;
        } else if (ty == 0x01) {
            blen = 0;
        }
        for (uint64_t k = 0; k < blen && k < 10; k++)
            m[k] = in[body + k];
        if (r == stop)
            break;
        r += 4 + len;
    }
    // decompress_len(&src): read_varu64 (src/bytes.rs:73-90) over m[0..10)
    uint64_t v = 0;
    uint32_t shift = 0;
    bool ok = false;
    for (int k = 0; k < 10; k++) {
        const uint64_t b = m[k];
        if (b < 0x80) {
            v |= b << shift;
            ok = true;
            break;
        }
        v |= (b & 0x7F) << shift;
        shift += 7;
    }
    const uint64_t sn = (ld32u(in + stop) >> 8) - 4;
    if (!ok)
        walk_fail(a, n_data, SNAPMI_HEADER, 0, 0);
    else if (v > kMaxInput)
        walk_fail(a, n_data, SNAPMI_TOO_BIG, v, kMaxInput);
    else if (v > kMaxBlock) // read.rs:217-222
        walk_fail(a, n_data, SNAPMI_UNSUPPORTED_CHUNK_LENGTH, v, 0);
    else // Decoder::decompress(&src[0..sn]), src/decompress.rs:80-83
        walk_fail(a, n_data, sn == 0 ? SNAPMI_EMPTY : SNAPMI_HEADER, 0, 0);
}

// true when the payload [p, p + pl) needs frame_short_varint
__device__ inline bool short_varint(gcptr p, uint64_t pl)
{
    if (pl >= 10)
        return false;
    for (uint64_t k = 0; k < pl; k++)
        if (p[k] < 0x80)
            return false;
    return true;
}

// ---------------------------------------------------------------------
// Parallel header walk.  The frame format is a linked list without an index:
// k_frame_walk below hops from header to header, 0.7 us per hop (1.1 s for the
// 1 048 576 chunks of a 64 GiB stream).  But a header is easy to recognise
// (type byte, 24-bit length within the format's limits, chunk inside the
// stream: 0.2 % of random positions pass) and a chain that starts at a wrong
// position dies within a hop or two.  So, per 32 MiB segment of the stream:
//   k_fw_candidates  every position of the segment's first 76 494 bytes (the
//                    real chain must touch down there) that looks like a
//                    header is followed to the segment's end; the survivors
//                    are recorded as {entry, exit, data chunks};
//   k_fw_resolve     one thread strings the segments together: the real
//                    chain enters segment k where it left segment k-1;
//   k_fw_emit        one thread per segment walks its real chain again and
//                    writes the chunk records at their final index.
// Only well-formed stretches are decided here: the first header that the
// reference's reader would reject (or a chunk the stale-buffer rule applies
// to) makes the chain "die", the resolve step then finds no continuation and
// sets meta[3] = 2, and the sequential walk runs and reports the error.
// ---------------------------------------------------------------------
constexpr uint32_t kFwCands = 32;
constexpr uint32_t kFwScan = 4 + kMaxChunk; // longest chunk, header included

// One hop of a well-formed chain: the chunk at r (header checks of
// src/read.rs:119-187 that need no history).  Returns false if the reader
// would stop here; *data = 1 for a compressed / stored chunk.
__device__ inline bool fw_hop(const FrameDecodeArgs &a, uint64_t &r,
                              uint32_t &data)
{
    gcptr in = (gcptr)a.in;
    if (a.in_len - r < 4)
        return false;
    const uint32_t hd = ld32u(in + r);
    const uint32_t ty = hd & 0xFF;
    const uint64_t len = hd >> 8;
    data = 0;
    if (len > kMaxChunk || (ty >= 0x02 && ty <= 0x7F) ||
        a.in_len - r - 4 < len)
        return false;
    if (ty == 0xFF) {
        if (len != 6)
            return false;
        const uint8_t body[6] = {'s', 'N', 'a', 'P', 'p', 'Y'};
        for (int k = 0; k < 6; k++)
            if (in[r + 4 + k] != body[k])
                return false;
    } else if (ty <= 0x01) {
        if (len < 4 || (ty == 0x01 && len - 4 > kMaxBlock))
            return false;
        if (ty == 0x00 && short_varint(in + r + 8, len - 4))
            return false; // read.rs:216: the sequential walk's business
        data = 1;
    }
    r += 4 + len;
    return true;
}

__global__ __launch_bounds__(256) void k_fw_candidates(FrameDecodeArgs a)
{
    const uint64_t seg = blockIdx.y;
    const uint64_t lo = seg * a.fw_seg;
    uint64_t end = lo + a.fw_seg; // the chain leaves the segment at or past it
    if (end > a.in_len)
        end = a.in_len;
    const uint64_t off = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= kFwScan || lo + off >= a.in_len)
        return;
    if (seg == 0 && off != 0)
        return; // the stream starts at its first byte
    uint64_t r = lo + off;
    uint32_t n = 0, data = 0;
    while (r < end) {
        if (!fw_hop(a, r, data))
            return; // not a chain (or the real one's first bad chunk)
        n += data;
    }
    const uint32_t k = atomicAdd(&a.fw_ncand[seg], 1u);
    if (k < kFwCands) {
        unsigned long long *c = a.fw_cand + (seg * kFwCands + k) * 3;
        c[0] = lo + off;
        c[1] = r;
        c[2] = n;
    }
}

__global__ void k_fw_resolve(FrameDecodeArgs a)
{
    gcptr in = (gcptr)a.in;
    a.meta[0] = 0;
    a.meta[1] = 0;
    a.meta[2] = 0xFFFFFFFFu;
    a.meta[3] = 2; // until proven well-formed: the sequential walk decides
    a.serr[0].kind = SNAPMI_OK;
    // the first chunk must be the stream identifier (src/read.rs:123-128)
    if (!(a.flags & SNAPMI_FRAME_CONTINUATION) &&
        (a.in_len < 4 || in[0] != 0xFF))
        return;
    uint64_t cur = 0;
    uint32_t total = 0;
    for (uint32_t seg = 0; seg < a.nseg; seg++) {
        a.fw_entry[seg] = ~0ull;
        a.fw_base[seg] = total;
        if (cur >= ((uint64_t)seg + 1) * a.fw_seg)
            continue; // (a chunk of <= 76 KiB spans a whole small test segment)
        const uint32_t nc =
            a.fw_ncand[seg] < kFwCands ? a.fw_ncand[seg] : kFwCands;
        bool found = false;
        for (uint32_t k = 0; k < nc && !found; k++) {
            const unsigned long long *c =
                a.fw_cand + ((uint64_t)seg * kFwCands + k) * 3;
            if (c[0] == cur) {
                a.fw_entry[seg] = cur;
                cur = c[1];
                total += (uint32_t)c[2];
                found = true;
            }
        }
        if (!found)
            return; // the real chain does not survive this segment
    }
    if (cur != a.in_len)
        return;
    a.fw_base[a.nseg] = total;
    a.meta[0] = total;
    a.meta[1] = total > a.cap_chunks ? 1 : 0;
    a.meta[3] = 0;
}

__global__ void k_fw_emit(FrameDecodeArgs a)
{
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= a.nseg || a.meta[3] != 0 || a.meta[1] != 0)
        return;
    uint64_t r = a.fw_entry[seg];
    if (r == ~0ull)
        return;
    gcptr in = (gcptr)a.in;
    uint64_t end = ((uint64_t)seg + 1) * a.fw_seg;
    if (end > a.in_len)
        end = a.in_len;
    uint32_t idx = a.fw_base[seg];
    while (r < end) {
        const uint32_t hd = ld32u(in + r);
        const uint32_t ty = hd & 0xFF;
        const uint32_t len = hd >> 8;
        if (ty <= 0x01) {
            FrameChunk c;
            c.payload_off = r + 8;
            c.payload_len = len - 4;
            c.crc = ld32u(in + r + 4);
            c.type = ty;
            c.pad = 0;
            a.chunks[idx++] = c;
        }
        r += 4 + (uint64_t)len;
    }
}

// Sequential walk over the chunk headers: reference FrameDecoder::read,
// src/read.rs:111-236 (checks in the reference's order).  One thread: every
// hop depends on the previous header.
__global__ void k_frame_walk(FrameDecodeArgs a)
{
    if (blockIdx.x || threadIdx.x)
        return;
    gcptr in = (gcptr)a.in;
    uint64_t r = 0;
    uint32_t nd = 0;
    bool seen_ident = (a.flags & SNAPMI_FRAME_CONTINUATION) != 0;
    a.meta[1] = 0;
    a.meta[2] = 0xFFFFFFFFu;
    a.meta[3] = 0;
    a.serr[0].kind = SNAPMI_OK;
    for (;;) {
        if (r == a.in_len)
            break; // clean EOF, :119-121
        if (a.in_len - r < 4) {
            walk_fail(a, nd, SNAPMI_E_UNEXPECTED_EOF, 0, 0);
            return;
        }
        const uint32_t hd = ld32u(in + r);
        r += 4;
        const uint32_t ty = hd & 0xFF;
        const uint64_t len = hd >> 8;
        if (!seen_ident) { // :123-128
            if (ty != 0xFF) {
                walk_fail(a, nd, SNAPMI_STREAM_HEADER, ty, 0);
                return;
            }
            seen_ident = true;
        }
        if (len > kMaxChunk) { // :129-135
            walk_fail(a, nd, SNAPMI_UNSUPPORTED_CHUNK_LENGTH, len, 0);
            return;
        }
        if (ty >= 0x02 && ty <= 0x7F) { // :138-142
            walk_fail(a, nd, SNAPMI_UNSUPPORTED_CHUNK_TYPE, ty, 0);
            return;
        }
        if ((ty >= 0x80 && ty <= 0xFD) || ty == 0xFE) { // skippable, padding
            if (a.in_len - r < len) {
                walk_fail(a, nd, SNAPMI_E_UNEXPECTED_EOF, 0, 0);
                return;
            }
            r += len;
        } else if (ty == 0xFF) { // :159-172
            if (len != 6) {
                walk_fail(a, nd, SNAPMI_UNSUPPORTED_CHUNK_LENGTH, len, 1);
                return;
            }
            if (a.in_len - r < 6) {
                walk_fail(a, nd, SNAPMI_E_UNEXPECTED_EOF, 0, 0);
                return;
            }
            const uint8_t body[6] = {'s', 'N', 'a', 'P', 'p', 'Y'};
            uint64_t got = 0;
            bool same = true;
            for (int k = 0; k < 6; k++) {
                const uint8_t b = in[r + k];
                got |= (uint64_t)b << (8 * k);
                same = same && b == body[k];
            }
            if (!same) {
                walk_fail(a, nd, SNAPMI_STREAM_HEADER_MISMATCH, got, 0);
                return;
            }
            r += 6;
        } else { // 0x00 compressed / 0x01 stored: :173-235
            if (len < 4) {
                walk_fail(a, nd, SNAPMI_UNSUPPORTED_CHUNK_LENGTH, len, 0);
                return;
            }
            if (a.in_len - r < 4) {
                walk_fail(a, nd, SNAPMI_E_UNEXPECTED_EOF, 0, 0);
                return;
            }
            const uint32_t crc = ld32u(in + r);
            r += 4;
            const uint64_t pl = len - 4;
            if (ty == 0x01 && pl > kMaxBlock) { // :182-187
                walk_fail(a, nd, SNAPMI_UNSUPPORTED_CHUNK_LENGTH, pl, 0);
                return;
            }
            if (a.in_len - r < pl) {
                walk_fail(a, nd, SNAPMI_E_UNEXPECTED_EOF, 0, 0);
                return;
            }
            if (ty == 0x00 && short_varint(in + r, pl)) { // read.rs:216
                frame_short_varint(a, r - 8, nd);
                return;
            }
            if (nd < a.cap_chunks) {
                FrameChunk c;
                c.payload_off = r;
                c.payload_len = (uint32_t)pl;
                c.crc = crc;
                c.type = ty;
                c.pad = 0;
                a.chunks[nd] = c;
            } else {
                a.meta[1] = 1; // overflow: caller reruns with more room
            }
            nd++;
            r += pl;
        }
    }
    a.meta[0] = nd;
}

// With a side index: chunk i's header is at index[i]; all chunks must be
// data chunks written by snapmi_frame_compress (the stream identifier is
// checked here as well).
__global__ void k_frame_index(FrameDecodeArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    gcptr in = (gcptr)a.in;
    // (meta[] and serr are cleared by the host before this launch)
    if (i == 0 && !(a.flags & SNAPMI_FRAME_CONTINUATION)) {
        const uint8_t ident[10] = {0xFF, 0x06, 0x00, 0x00, 's',
                                   'N',  'a',  'P',  'p',  'Y'};
        bool ok = a.in_len >= 10;
        for (int k = 0; ok && k < 10; k++)
            ok = in[k] == ident[k];
        if (!ok)
            a.meta[3] = 1; // the walk reports what is wrong with the start
    }
    // an index without entries tiles nothing: only a stream that ends right
    // behind its identifier (or an empty continuation) has no chunks
    if (i == 0 && a.n_index == 0 &&
        a.in_len != ((a.flags & SNAPMI_FRAME_CONTINUATION) ? 0u : 10u))
        a.meta[3] = 1;
    if (i >= a.n_index)
        return;
    const uint64_t r = a.index[i];
    FrameChunk c;
    c.type = 0xFF;
    c.payload_len = 0;
    c.crc = 0;
    c.payload_off = 0;
    c.pad = 0;
    // The index is the caller's: it is taken as a hint, never trusted.  A
    // header that is not an in-bounds data chunk ending exactly at the next
    // index entry (the entries must tile the stream from the identifier to
    // its end: other chunk types in between are the walk's business), or a
    // chunk only the walk can judge (frame_short_varint), sends the whole
    // stream to the walk, which owns every error report.
    bool good = a.in_len >= 8 && r <= a.in_len - 8;
    if (good) {
        const uint32_t hd = ld32u(in + r);
        const uint32_t ty = hd & 0xFF, len = hd >> 8;
        good = ty <= 1 && len >= 4 && len <= kMaxChunk &&
               len <= a.in_len - r - 4 && !(ty == 1 && len - 4 > kMaxBlock) &&
               a.index[i + 1] == r + 4 + len && // nothing unseen in between
               (i + 1 < a.n_index || a.index[i + 1] == a.in_len) &&
               (i > 0 ||
                r == ((a.flags & SNAPMI_FRAME_CONTINUATION) ? 0u : 10u)) &&
               !(ty == 0 && short_varint(in + r + 8, len - 4));
        if (good) {
            c.type = ty;
            c.payload_len = len - 4;
            c.crc = ld32u(in + r + 4);
            c.payload_off = r + 8;
        }
    }
    if (!good)
        a.meta[3] = 1;
    a.chunks[i] = c;
}

// decompressed length of every data chunk: reference src/read.rs:181-187
// (stored) and :215-222 (compressed: decompress_len, then dn <= 65536)
__global__ void k_frame_lens(FrameDecodeArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = a.meta[0];
    if (i >= n || i >= a.cap_chunks)
        return;
    const FrameChunk c = a.chunks[i];
    snapmi_error e;
    e.kind = SNAPMI_OK;
    e.reserved = 0;
    e.a = e.b = e.c = 0;
    uint64_t dn = 0;
    if (c.type > 1) { // only reachable through a bad side index
        e.kind = SNAPMI_UNSUPPORTED_CHUNK_TYPE;
        e.a = c.pad;
    } else if (c.type == 1) {
        dn = c.payload_len;
    } else {
        gcptr p = (gcptr)a.in + c.payload_off;
        uint64_t acc = 0;
        uint32_t shift = 0, used = 0;
        bool ok = false;
        for (uint32_t k = 0; k < c.payload_len; k++) {
            const uint32_t b = p[k];
            if (shift >= 64)
                break;
            if (b < 0x80) {
                acc |= (uint64_t)b << shift;
                used = k + 1;
                ok = true;
                break;
            }
            acc |= (uint64_t)(b & 0x7F) << shift;
            shift += 7;
        }
        (void)used;
        if (c.payload_len == 0) {
            // reference: decompress_len of the scratch reads a stale byte;
            // the decode of the empty payload then fails with Empty
            e.kind = SNAPMI_EMPTY;
        } else if (!ok) {
            e.kind = SNAPMI_HEADER;
        } else if (acc > kMaxInput) {
            e.kind = SNAPMI_TOO_BIG;
            e.a = acc;
            e.b = kMaxInput;
        } else if (acc > kMaxBlock) {
            e.kind = SNAPMI_UNSUPPORTED_CHUNK_LENGTH;
            e.a = acc;
        } else {
            dn = acc;
        }
    }
    a.dlens[i] = e.kind == SNAPMI_OK ? dn : 0;
    a.cerrs[i] = e;
}

// descriptors for the raw decompressor (one "stream" per data chunk)
__global__ void k_frame_desc(FrameDecodeArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = a.meta[0];
    if (i >= n || i >= a.cap_chunks)
        return;
    const FrameChunk c = a.chunks[i];
    const bool bad = a.cerrs[i].kind != SNAPMI_OK;
    const bool fits = a.offs[n] <= a.out_cap;
    a.in_ptrs[i] = a.in + c.payload_off;
    // a chunk that already failed (or an output that does not fit) is
    // decoded as an empty stored chunk: nothing is written
    a.in_lens[i] = (bad || !fits) ? 0 : c.payload_len;
    a.modes[i] = (bad || !fits) ? 1 : (uint8_t)c.type;
    a.out_ptrs[i] = a.out + a.offs[i];
    a.out_caps[i] = a.dlens[i];
    a.out_lens[i] = 0;
}

// Final verdict: first error in stream order (reference processes chunk by
// chunk: length checks, raw decode, checksum :225-232 / :189-196).
__global__ __launch_bounds__(1024) void k_frame_verify(FrameDecodeArgs a,
                                                       const snapmi_error *derrs)
{
    __shared__ uint32_t first_bad;
    const uint32_t n = a.meta[0] < a.cap_chunks ? a.meta[0] : a.cap_chunks;
    if (threadIdx.x == 0)
        first_bad = 0xFFFFFFFFu;
    __syncthreads();
    const bool decoded = a.out != nullptr && a.offs[n] <= a.out_cap;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        bool bad = a.cerrs[i].kind != SNAPMI_OK;
        if (!bad && decoded) {
            if (derrs[i].kind != SNAPMI_OK) {
                a.cerrs[i] = derrs[i];
                bad = true;
            } else if (a.crcs[i] != a.chunks[i].crc) {
                snapmi_error e;
                e.kind = SNAPMI_CHECKSUM;
                e.reserved = 0;
                e.a = a.chunks[i].crc;
                e.b = a.crcs[i];
                e.c = 0;
                a.cerrs[i] = e;
                bad = true;
            }
        }
        if (bad)
            atomicMin(&first_bad, i);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t sidx = a.meta[2]; // structural error before chunk sidx
        snapmi_error e;
        e.kind = SNAPMI_OK;
        e.reserved = 0;
        e.a = e.b = e.c = 0;
        if (first_bad != 0xFFFFFFFFu && first_bad < sidx)
            e = a.cerrs[first_bad];
        else if (a.serr[0].kind != SNAPMI_OK)
            e = a.serr[0];
        else if (a.out != nullptr && a.offs[n] > a.out_cap) {
            e.kind = SNAPMI_BUFFER_TOO_SMALL;
            e.a = a.out_cap;
            e.b = a.offs[n];
        }
        a.err[0] = e;
        // On an error the chunks in front of the failing one are decoded and
        // checked: the reference's reader has handed them out by then
        // (src/read.rs:112-118), so their byte count is reported.
        uint32_t upto = n;
        if (e.kind != SNAPMI_OK) {
            upto = first_bad < sidx ? first_bad : (sidx < n ? sidx : n);
            if (!decoded)
                upto = 0;
        }
        a.out_len[0] = a.offs[upto];
    }
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------
static int ensure_tables(snapmi_ctx *ctx)
{
    if (ctx->fr_tables_ready)
        return SNAPMI_OK;
    int rc = reserve(ctx, ctx->fr_tables, sizeof(CrcTables));
    if (rc)
        return rc;
    std::vector<uint8_t> buf(sizeof(CrcTables));
    CrcTables *t = (CrcTables *)buf.data();
    uint32_t byte_tab[256];
    for (uint32_t i = 0; i < 256; i++) { // reference build.rs:110-124
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (c >> 1) ^ kCrcPoly : c >> 1;
        byte_tab[i] = c;
    }
    auto adv1 = [&](uint32_t s) { return byte_tab[s & 255] ^ (s >> 8); };
    for (int k = 0; k < 4; k++)
        for (uint32_t b = 0; b < 256; b++) {
            uint32_t s = b << (8 * k);
            for (int z = 0; z < 256; z++)
                s = adv1(s);
            t->z256[k][b] = s;
        }
    // x^(8*n) mod P = the state 0x80000000 (x^0) advanced by n zero bytes
    for (uint32_t j = 0; j < 64; j++) {
        uint32_t s = 1u << 31;
        for (uint32_t z = 0; z < 4 * (64 - j); z++)
            s = adv1(s);
        t->lane_mul[j] = s;
    }
    uint32_t s = 0xFFFFFFFFu;
    for (uint32_t n = 0; n <= 65536; n++) {
        t->init_adv[n] = s;
        s = adv1(s);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fr_tables.p, buf.data(),
                                sizeof(CrcTables), hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->fr_tables_ready = true;
    return SNAPMI_OK;
}

template <class T> static T *carve(uint8_t *&p, size_t count)
{
    uintptr_t a = ((uintptr_t)p + 15) & ~(uintptr_t)15;
    T *r = (T *)a;
    p = (uint8_t *)(a + count * sizeof(T));
    return r;
}

} // namespace snapmi

extern "C" {

size_t snapmi_frame_max_len(size_t n)
{
    const size_t chunks = (n + kMaxBlock - 1) / kMaxBlock;
    return 10 + n + 8 * chunks;
}

int snapmi_crc32c_masked_batch(snapmi_ctx *ctx, const void *const *d_ptrs,
                               const uint64_t *d_lens, uint32_t *d_out,
                               size_t n)
{
    if (!ctx || (n && (!d_ptrs || !d_lens || !d_out)) || n > 0x7FFFFFFFu)
        return SNAPMI_E_ARGUMENT;
    if (n == 0)
        return SNAPMI_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_tables(ctx);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_crc32c, dim3((uint32_t)n), dim3(64), 0, ctx->stream,
                       d_ptrs, d_lens, d_out, (uint32_t)n,
                       (const CrcTables *)ctx->fr_tables.p);
    HIP_TRY(ctx, hipGetLastError());
    return SNAPMI_OK;
}

} // extern "C"

namespace snapmi {
// n chunks of d_in: at 65536-byte multiples (d_chunk_in_off == nullptr) or at
// the given input offsets ([n+1], device); shared by the two entry points
static int frame_compress_impl(snapmi_ctx *ctx, const void *d_in,
                               uint64_t in_len, uint32_t n,
                               const uint64_t *d_chunk_in_off, bool ident,
                               void *d_out, uint64_t *d_out_len,
                               uint64_t *d_chunk_offsets,
                               const uint32_t *h_chunk_lens = nullptr)
{
    hipStream_t s = ctx->stream;
    int rc = ensure_tables(ctx);
    if (rc)
        return rc;
    // segments of at most lane_segment_blocks chunks (16 GiB of input by
    // default): 76 KiB of slot + 72 KiB of tokens per chunk of a segment
    const uint32_t seg = n < ctx->lane_segment_blocks
                             ? n
                             : ctx->lane_segment_blocks;
    const size_t desc_bytes =
        (size_t)seg * (8 + 8 + 8 + 8 + 4 + 8 + 8) + 8 + 16 * 8;
    if ((rc = reserve(ctx, ctx->fr_desc, desc_bytes)) ||
        (rc = reserve(ctx, ctx->fr_slots, (size_t)seg * kFrameSlot)))
        return rc;
    FrameCompressArgs a;
    uint8_t *p = (uint8_t *)ctx->fr_desc.p;
    a.in = (const uint8_t *)d_in;
    a.in_len = in_len;
    a.out = (uint8_t *)d_out;
    a.out_cap = 0;
    a.out_len = d_out_len;
    a.chunk_offsets = d_chunk_offsets;
    a.n = n;
    a.chunk_in_off = d_chunk_in_off;
    a.ident = ident ? 10 : 0;
    a.base = carve<uint64_t>(p, 1);
    a.in_ptrs = carve<const void *>(p, seg);
    a.in_lens = carve<uint64_t>(p, seg);
    a.slot_ptrs = carve<void *>(p, seg);
    a.clens = carve<uint64_t>(p, seg);
    a.sizes = carve<uint64_t>(p, seg);
    a.offs = carve<uint64_t>(p, seg + 1);
    a.crcs = carve<uint32_t>(p, seg);
    a.slots = (uint8_t *)ctx->fr_slots.p;

    HIP_TRY(ctx, hipMemsetAsync(a.base, 0, sizeof(uint64_t), s));
    for (uint32_t lo = 0; lo < n; lo += seg) {
        const uint32_t cnt = n - lo < seg ? n - lo : seg;
        a.lo = lo;
        a.cnt = cnt;
        const uint32_t tb = 256, gb = (cnt + tb - 1) / tb;
        hipLaunchKernelGGL(k_frame_chunks, dim3(gb), dim3(tb), 0, s, a);
        // The checksums only need the input: they run on the side stream,
        // under the match finder (which waits on HBM round trips and leaves
        // 16 KiB of LDS per CU free - room for three CRC workgroups).  Fusing
        // the CRC into the match finder's own block loads would add ~380
        // vector instructions per 128-byte line to a loop of ~150 per round
        // to save a kernel of 1 % of the pass: not done.
        const bool side = ctx->frame_crc_side_stream;
        if (side) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_crc[0], s));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_crc[0], 0));
        }
        hipLaunchKernelGGL(k_crc32c, dim3(cnt), dim3(64), 0,
                           side ? ctx->stream2 : s, a.in_ptrs, a.in_lens,
                           a.crcs, cnt, (const CrcTables *)ctx->fr_tables.p);
        if (side)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_crc[1], ctx->stream2));
        // every chunk is a one-block raw stream: cnt blocks, no scratch slots
        // (chunks cut short by the caller - flushes, short reads - are
        // counted for the small-block kernels)
        uint64_t c8 = 0;
        if (h_chunk_lens)
            for (uint32_t i = lo; i < lo + cnt; i++)
                c8 += h_chunk_lens[i] <= 8192;
        rc = launch_compress(ctx, a.in_ptrs, a.in_lens, a.slot_ptrs, nullptr,
                             a.clens, nullptr, cnt, cnt, 0, 0xF, c8);
        if (rc)
            return rc;
        if (side)
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_crc[1], 0));
        hipLaunchKernelGGL(k_frame_sizes, dim3(gb), dim3(tb), 0, s, a);
        hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, a.sizes,
                           a.offs, cnt);
        hipLaunchKernelGGL(k_frame_emit, dim3(cnt), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_frame_advance, dim3(1), dim3(1), 0, s, a);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SNAPMI_OK;
}
} // namespace snapmi

extern "C" {

int snapmi_frame_compress(snapmi_ctx *ctx, const void *d_in, uint64_t in_len,
                          void *d_out, uint64_t out_cap, uint64_t *d_out_len,
                          uint64_t *d_chunk_offsets)
{
    if (!ctx || !d_out_len || (in_len && (!d_in || !d_out)))
        return SNAPMI_E_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (in_len == 0) { // the identifier is written lazily: src/write.rs:154-170
        HIP_TRY(ctx, hipMemsetAsync(d_out_len, 0, sizeof(uint64_t), s));
        return SNAPMI_OK;
    }
    if (out_cap < snapmi_frame_max_len(in_len))
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                        "frame_compress: out_cap %llu < frame_max_len %zu",
                        (unsigned long long)out_cap,
                        snapmi_frame_max_len(in_len));
    const uint64_t n64 = (in_len + kMaxBlock - 1) / kMaxBlock;
    if (n64 > 0x7FFFFFFFu)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "frame_compress: too long");
    return frame_compress_impl(ctx, d_in, in_len, (uint32_t)n64, nullptr, true,
                               d_out, d_out_len, d_chunk_offsets);
}

int snapmi_frame_compress_chunks(snapmi_ctx *ctx, const void *d_in,
                                 const uint32_t *h_chunk_lens, size_t n,
                                 uint32_t flags, void *d_out, uint64_t out_cap,
                                 uint64_t *d_out_len,
                                 uint64_t *d_chunk_offsets)
{
    if (!ctx || !d_out_len || (n && (!d_in || !d_out || !h_chunk_lens)) ||
        n > 0x7FFFFFFFu || (flags & ~(uint32_t)SNAPMI_FRAME_NO_IDENT))
        return SNAPMI_E_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out_len, 0, sizeof(uint64_t), s));
        return SNAPMI_OK;
    }
    // input offsets of the chunks: chunk i = d_in[off[i], off[i+1])
    std::vector<uint64_t> off(n + 1);
    off[0] = 0;
    for (size_t i = 0; i < n; i++) {
        if (h_chunk_lens[i] == 0 || h_chunk_lens[i] > kMaxBlock)
            return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                            "frame_compress_chunks: chunk %zu has %u bytes "
                            "(1..65536)", i, h_chunk_lens[i]);
        off[i + 1] = off[i] + h_chunk_lens[i];
    }
    const bool ident = !(flags & SNAPMI_FRAME_NO_IDENT);
    const uint64_t need = (ident ? 10 : 0) + off[n] + 8 * (uint64_t)n;
    if (out_cap < need)
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                        "frame_compress_chunks: out_cap %llu < %llu",
                        (unsigned long long)out_cap, (unsigned long long)need);
    int rc = reserve(ctx, ctx->fr_chunk_off, (n + 1) * sizeof(uint64_t));
    if (rc)
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->fr_chunk_off.p, off.data(),
                                (n + 1) * sizeof(uint64_t),
                                hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s)); // `off` is pageable host memory
    return frame_compress_impl(ctx, d_in, off[n], (uint32_t)n,
                               (const uint64_t *)ctx->fr_chunk_off.p, ident,
                               d_out, d_out_len, d_chunk_offsets,
                               h_chunk_lens);
}

int snapmi_frame_index_host(const void *h_in, uint64_t in_len,
                            uint64_t *h_offsets, uint64_t cap,
                            uint64_t *n_chunks)
{
    // the regular cases of k_frame_walk (reference src/read.rs:111-236);
    // anything else is left to the device walk, which owns the error report
    if (!n_chunks || (in_len && !h_in))
        return 1;
    const uint8_t *in = (const uint8_t *)h_in;
    uint64_t r = 0, nd = 0;
    bool seen_ident = false;
    while (r != in_len) {
        if (in_len - r < 4)
            return 1;
        const uint32_t ty = in[r];
        const uint64_t len = (uint64_t)in[r + 1] | ((uint64_t)in[r + 2] << 8) |
                             ((uint64_t)in[r + 3] << 16);
        const uint64_t at = r;
        r += 4;
        if (!seen_ident && ty != 0xFF)
            return 1;
        seen_ident = true;
        if (len > kMaxChunk || (ty >= 0x02 && ty <= 0x7F) ||
            in_len - r < len)
            return 1;
        if (ty == 0xFF) {
            if (len != 6 || memcmp(in + r, "sNaPpY", 6) != 0)
                return 1;
        } else if (ty <= 0x01) {
            if (len < 4 || (ty == 0x01 && len - 4 > kMaxBlock))
                return 1;
            if (h_offsets) {
                if (nd + 1 >= cap)
                    return 1;
                h_offsets[nd] = at;
            }
            nd++;
        }
        r += len;
    }
    if (h_offsets) {
        if (nd + 1 > cap)
            return 1;
        h_offsets[nd] = in_len;
    }
    *n_chunks = nd;
    return 0;
}

// meta[] and the structural error slot before the index kernel (whose
// threads only ever raise meta[3])
__global__ void k_frame_meta_init(FrameDecodeArgs a)
{
    a.meta[0] = a.n_index;
    a.meta[1] = 0;
    a.meta[2] = 0xFFFFFFFFu;
    a.meta[3] = 0;
    a.serr[0].kind = SNAPMI_OK;
}

// Device memory -> pinned (device-mapped) host memory by a kernel: 16-byte
// stores over the host link.  On this platform a hipMemcpyAsync to the host
// and one from the host, on two streams of one process, ran one after the
// other (SNAPMI_PIPE_TRACE: a 0.55 GB copy in took 26 ms behind a 1 GiB copy
// out) although the link is full duplex (tests/hw/pcie_duplex.py: 2 x 48
// GB/s); a copy kernel beside a copy-engine transfer the other way does
// overlap.  The destination is brought to a 16-byte boundary first.
__global__ __launch_bounds__(256) void k_to_host(uint8_t *host,
                                                 const uint8_t *dev,
                                                 uint64_t n)
{
    gptr to = (gptr)host;
    gcptr from = (gcptr)dev;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthr = (uint64_t)gridDim.x * blockDim.x;
    uint64_t head = (16 - ((uintptr_t)host & 15)) & 15;
    if (head > n)
        head = n;
    if (tid < head)
        to[tid] = from[tid];
    const uint64_t body = (n - head) & ~15ull;
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    for (uint64_t i = 16 * tid; i < body; i += 16 * nthr)
        *(g_u32x4 *)(to + head + i) = ld128g(from + head + i);
    const uint64_t tail = n - head - body;
    if (tid < tail)
        to[head + body + tid] = from[head + body + tid];
}

// is [p, p + n) host memory a kernel may write (pinned and mapped)?
static bool device_visible_at(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory: not an error here
        return false;
    }
    // (the copy kernel stores through the HOST address: memory that is
    // mapped at another device address - hipHostRegister without unified
    // addressing - goes home by hipMemcpyAsync instead)
    return at.type == hipMemoryTypeHost && at.devicePointer == p;
}
// k_to_host may store to [p, p + n): both ends pinned and mapped at their
// host address (a buffer that starts in a pinned region and leaves it - a
// caller's offset into a smaller registration - must not reach the kernel)
static bool device_visible(const void *p, size_t n)
{
    if (!p || !n)
        return false;
    return device_visible_at(p) &&
           device_visible_at((const uint8_t *)p + (n - 1));
}

static int copy_home(snapmi_ctx *ctx, hipStream_t st, uint8_t *h_dst,
                     const void *d_src, uint64_t n, bool by_kernel)
{
    if (by_kernel) {
        // enough workgroups to keep the link busy, few enough to leave the
        // codec its CUs
        hipLaunchKernelGGL(k_to_host, dim3(128), dim3(256), 0, st, h_dst,
                           (const uint8_t *)d_src, n);
        HIP_TRY(ctx, hipGetLastError());
        return SNAPMI_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, st));
    return SNAPMI_OK;
}

static int mailbox(snapmi_ctx *ctx)
{
    if (!ctx->h_mail)
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_mail, 256,
                                   hipHostMallocDefault));
    return SNAPMI_OK;
}

// A few words of a result, from device memory into pinned (device-mapped)
// host memory.  A hipMemcpyAsync of 16 bytes would do the same - through the
// copy engine, where it waits behind whatever bulk copy is under way in that
// direction (20 ms behind a 1 GiB result): a kernel store does not queue.
__global__ void k_post_words(uint32_t *host_mapped, const uint32_t *dev,
                             uint32_t n)
{
    if (threadIdx.x < n)
        host_mapped[threadIdx.x] = dev[threadIdx.x];
    __threadfence_system();
}

// snapmi_frame_scan_host, stopping in front of data chunk number `max_data`
// (status 3: the caller's output buffer is full)
static int frame_scan(const void *h_in, uint64_t in_len, uint32_t flags,
                      uint8_t *stale10, uint64_t *h_offsets, uint64_t cap,
                      uint64_t max_data, uint64_t *n_chunks,
                      uint64_t *consumed)
{
    if (!n_chunks || !consumed || (in_len && !h_in))
        return SNAPMI_E_ARGUMENT;
    const uint8_t *in = (const uint8_t *)h_in;
    uint64_t r = 0, nd = 0;
    bool seen_ident = (flags & SNAPMI_FRAME_CONTINUATION) != 0;
    int status = 0;
    while (r != in_len) {
        if (in_len - r < 4) {
            status = 2;
            break;
        }
        const uint32_t ty = in[r];
        const uint64_t len = (uint64_t)in[r + 1] | ((uint64_t)in[r + 2] << 8) |
                             ((uint64_t)in[r + 3] << 16);
        if ((!seen_ident && ty != 0xFF) || len > kMaxChunk ||
            (ty >= 0x02 && ty <= 0x7F) || (ty == 0xFF && len != 6) ||
            (ty <= 0x01 && (len < 4 || (ty == 0x01 && len - 4 > kMaxBlock)))) {
            status = 1;
            break;
        }
        if (in_len - r - 4 < len) {
            status = 2;
            break;
        }
        if (ty == 0xFF && memcmp(in + r + 4, "sNaPpY", 6) != 0) {
            status = 1;
            break;
        }
        seen_ident = true;
        if (ty <= 0x01) {
            if (nd == max_data) {
                status = 3;
                break;
            }
            if (h_offsets) {
                if (nd + 1 >= cap)
                    return SNAPMI_E_ARGUMENT;
                h_offsets[nd] = r;
            }
            nd++;
        }
        if (stale10) { // the reference reader's src[0..10), see frame_short_varint
            memcpy(stale10, in + r, 4);
            const uint64_t body = ty == 0x00 ? r + 8 : r + 4;
            const uint64_t blen = ty == 0x00 ? len - 4 : (ty == 0x01 ? 0 : len);
            memcpy(stale10, in + body, (size_t)(blen < 10 ? blen : 10));
        }
        r += 4 + len;
    }
    if (h_offsets) {
        if (nd + 1 > cap)
            return SNAPMI_E_ARGUMENT;
        h_offsets[nd] = r;
    }
    *n_chunks = nd;
    *consumed = r;
    return status;
}

int snapmi_frame_scan_host(const void *h_in, uint64_t in_len, uint32_t flags,
                           uint8_t *stale10, uint64_t *h_offsets,
                           uint64_t cap, uint64_t *n_chunks,
                           uint64_t *consumed)
{
    return frame_scan(h_in, in_len, flags, stale10, h_offsets, cap,
                      ~0ull, n_chunks, consumed);
}

int snapmi_frame_decompress_ex(snapmi_ctx *ctx, const void *d_in,
                               uint64_t in_len, void *d_out, uint64_t out_cap,
                               uint64_t *d_out_len, snapmi_error *d_err,
                               const uint64_t *d_chunk_offsets,
                               uint64_t n_chunks, uint32_t flags,
                               const uint8_t *stale10)
{
    if (!ctx || !d_out_len || !d_err || (in_len && !d_in) ||
        (flags & ~(uint32_t)SNAPMI_FRAME_CONTINUATION))
        return SNAPMI_E_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = ensure_tables(ctx);
    if (rc)
        return rc;
    // capacity for the chunk table: exact with an index, else an estimate
    // that is retried once with the exact count
    bool use_index = d_chunk_offsets != nullptr;
    // long streams without an index: find the headers in parallel (k_fw_*)
    bool parallel_walk = in_len >= ctx->frame_parallel_walk_min;
    uint64_t cap = use_index ? n_chunks : in_len / 2048 + 64;
    const uint64_t fw_seg = ctx->frame_walk_segment;
    const uint64_t nseg64 = (in_len + fw_seg - 1) / fw_seg;
    if (nseg64 == 0 || nseg64 > 65535)
        parallel_walk = false; // (empty; grid limit: 2 TiB at the default)
    const uint32_t nseg = parallel_walk ? (uint32_t)nseg64 : 0;
    for (int attempt = 0; attempt < 4; attempt++) {
        if (cap > 0x7FFFFFFFu)
            return fail_ctx(ctx, SNAPMI_E_ARGUMENT, "frame: too many chunks");
        const size_t n = (size_t)cap;
        const size_t meta_bytes = 64 + sizeof(snapmi_error) +
                                  n * (sizeof(FrameChunk) + 8 + 8 +
                                       sizeof(snapmi_error) * 2 + 8 * 5 + 1 +
                                       4) + 32 * 16 +
                                  (size_t)(nseg + 2) *
                                      (kFwCands * 24 + 4 + 8 + 4);
        if ((rc = reserve(ctx, ctx->fr_meta, meta_bytes)))
            return rc;
        FrameDecodeArgs a;
        uint8_t *p = (uint8_t *)ctx->fr_meta.p;
        a.in = (const uint8_t *)d_in;
        a.in_len = in_len;
        a.out = (uint8_t *)d_out;
        a.out_cap = out_cap;
        a.out_len = d_out_len;
        a.err = d_err;
        a.index = use_index ? d_chunk_offsets : nullptr;
        a.n_index = use_index ? (uint32_t)n_chunks : 0;
        a.cap_chunks = (uint32_t)n;
        a.flags = flags;
        for (int k = 0; k < 10; k++)
            a.stale[k] = stale10 ? stale10[k] : 0;
        a.meta = carve<uint32_t>(p, 4);
        a.serr = carve<snapmi_error>(p, 1);
        a.chunks = carve<FrameChunk>(p, n);
        a.dlens = carve<uint64_t>(p, n);
        a.offs = carve<uint64_t>(p, n + 1);
        a.cerrs = carve<snapmi_error>(p, n);
        snapmi_error *derrs = carve<snapmi_error>(p, n);
        a.in_ptrs = carve<const void *>(p, n);
        a.in_lens = carve<uint64_t>(p, n);
        a.out_ptrs = carve<void *>(p, n);
        a.out_caps = carve<uint64_t>(p, n);
        a.out_lens = carve<uint64_t>(p, n);
        a.crcs = carve<uint32_t>(p, n);
        a.modes = carve<uint8_t>(p, n);
        a.nseg = nseg;
        a.fw_seg = fw_seg;
        a.fw_cand = carve<unsigned long long>(p, (size_t)nseg * kFwCands * 3);
        a.fw_entry = carve<unsigned long long>(p, (size_t)nseg + 1);
        a.fw_ncand = carve<uint32_t>(p, nseg);
        a.fw_base = carve<uint32_t>(p, (size_t)nseg + 1);

        const uint32_t tb = 256;
        const uint32_t gb = n ? (uint32_t)((n + tb - 1) / tb) : 1;
        if (use_index) {
            hipLaunchKernelGGL(k_frame_meta_init, dim3(1), dim3(1), 0, s, a);
            hipLaunchKernelGGL(k_frame_index, dim3(gb), dim3(tb), 0, s, a);
        } else if (parallel_walk) {
            HIP_TRY(ctx, hipMemsetAsync(a.fw_ncand, 0, a.nseg * 4, s));
            hipLaunchKernelGGL(k_fw_candidates,
                               dim3((kFwScan + 255) / 256, a.nseg), dim3(256),
                               0, s, a);
            hipLaunchKernelGGL(k_fw_resolve, dim3(1), dim3(1), 0, s, a);
            hipLaunchKernelGGL(k_fw_emit, dim3((a.nseg + 63) / 64), dim3(64),
                               0, s, a);
        } else {
            hipLaunchKernelGGL(k_frame_walk, dim3(1), dim3(64), 0, s, a);
        }
        // the number of data chunks decides the launch sizes below
        uint32_t meta[4];
        if ((rc = mailbox(ctx)))
            return rc;
        hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, s,
                           (uint32_t *)ctx->h_mail, a.meta, 4u);
        HIP_TRY(ctx, hipStreamSynchronize(s));
        memcpy(meta, (const void *)ctx->h_mail, sizeof meta);
        if (use_index && meta[3]) { // the index does not tile the stream with
            use_index = false;      // plain data chunks: the walk decides
            cap = n_chunks + in_len / 65536 + 64;
            continue;
        }
        if (parallel_walk && meta[3] == 2) { // something in the stream that
            parallel_walk = false;           // only the sequential walk judges
            continue;
        }
        if (meta[1] && !use_index && cap < meta[0]) { // table too small
            cap = meta[0];
            continue;
        }
        const uint32_t nd = meta[0] < n ? meta[0] : (uint32_t)n;
        const uint32_t gd = nd ? (nd + tb - 1) / tb : 1;
        hipLaunchKernelGGL(k_frame_lens, dim3(gd), dim3(tb), 0, s, a);
        hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, s, a.dlens,
                           a.offs, nd);
        if (d_out && nd) {
            hipLaunchKernelGGL(k_frame_desc, dim3(gd), dim3(tb), 0, s, a);
            rc = launch_decompress(ctx, a.in_ptrs, a.in_lens, a.out_ptrs,
                                   a.out_caps, a.out_lens, derrs, a.modes,
                                   nd);
            if (rc)
                return rc;
            // CRC of what was produced (chunks that were skipped have
            // out_lens 0: their CRC is never compared)
            hipLaunchKernelGGL(k_crc32c, dim3(nd), dim3(64), 0, s,
                               (const void *const *)a.out_ptrs, a.out_lens,
                               a.crcs, nd,
                               (const CrcTables *)ctx->fr_tables.p);
        }
        hipLaunchKernelGGL(k_frame_verify, dim3(1), dim3(1024), 0, s, a,
                           (const snapmi_error *)derrs);
        HIP_TRY(ctx, hipGetLastError());
        return SNAPMI_OK;
    }
    return fail_ctx(ctx, SNAPMI_E_DEVICE, "frame: chunk table retry failed");
}

int snapmi_frame_decompress(snapmi_ctx *ctx, const void *d_in,
                            uint64_t in_len, void *d_out, uint64_t out_cap,
                            uint64_t *d_out_len, snapmi_error *d_err,
                            const uint64_t *d_chunk_offsets,
                            uint64_t n_chunks)
{
    return snapmi_frame_decompress_ex(ctx, d_in, in_len, d_out, out_cap,
                                      d_out_len, d_err, d_chunk_offsets,
                                      n_chunks, 0, nullptr);
}

// ----------------------------------------------------------------------
// Host-buffer forms (H2D + kernels + D2H, blocking): what a host-language
// FrameEncoder / FrameDecoder (the Rust shim, the Python mirror, tools/szip)
// calls per batch of chunks.
// ----------------------------------------------------------------------
size_t snapmi_frame_encode_bound(size_t total_bytes, size_t n_chunks)
{
    return 10 + total_bytes + 8 * n_chunks;
}

} // extern "C"

// ---- the pipeline behind the two host-buffer calls -----------------------
// A batch is cut into slices; slice i+1 is on its way to the device (copy
// stream 1) while the kernels of slice i run (the context's stream) and the
// result of slice i-1 goes back to the host (copy stream 2): PCIe is full
// duplex, and the three legs of a batch cost about the same (a 4 GiB corpus
// batch: 78 ms in, 60 ms of kernels, 39 ms out - 177 ms one after the other).
// Three slots of device staging, so that none of the three legs waits for a
// buffer of the other two.  Host memory from snapmi_host_alloc (pinned) is
// what makes the copies asynchronous; pageable memory works, one leg at a
// time.
namespace {
constexpr int kSlots = 3;
struct PipeSlot {
    snapmi::DevBuf in, out, desc; // desc: u64 len | snapmi_error | index...
    hipEvent_t ev_h2d = nullptr, ev_k = nullptr, ev_d2h = nullptr;
    struct Result {
        uint64_t len;
        snapmi_error e;
    } *h_res = nullptr;            // pinned
    uint64_t *h_off = nullptr;     // pinned: chunk offsets of the slice
    size_t h_off_cap = 0;
};
} // namespace

struct snapmi_host_pipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    PipeSlot slot[kSlots];
};

namespace snapmi {
void host_pipe_destroy(snapmi_ctx *ctx)
{
    snapmi_host_pipe *p = ctx->pipe;
    if (!p)
        return;
    if (p->s_in) {
        (void)hipStreamSynchronize(p->s_in);
        (void)hipStreamDestroy(p->s_in);
    }
    if (p->s_out) {
        (void)hipStreamSynchronize(p->s_out);
        (void)hipStreamDestroy(p->s_out);
    }
    for (PipeSlot &sl : p->slot) {
        for (DevBuf *b : {&sl.in, &sl.out, &sl.desc})
            if (b->p)
                (void)hipFree(b->p);
        for (hipEvent_t e : {sl.ev_h2d, sl.ev_k, sl.ev_d2h})
            if (e)
                (void)hipEventDestroy(e);
        if (sl.h_res)
            (void)hipHostFree(sl.h_res);
        if (sl.h_off)
            (void)hipHostFree(sl.h_off);
    }
    delete p;
    ctx->pipe = nullptr;
}
} // namespace snapmi

static int host_pipe(snapmi_ctx *ctx, snapmi_host_pipe **out)
{
    if (!ctx->pipe) {
        snapmi_host_pipe *p = new snapmi_host_pipe;
        ctx->pipe = p; // (freed with the context whatever fails below)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
        for (PipeSlot &sl : p->slot) {
            HIP_TRY(ctx, hipEventCreate(&sl.ev_h2d));
            HIP_TRY(ctx, hipEventCreate(&sl.ev_k));
            HIP_TRY(ctx, hipEventCreate(&sl.ev_d2h));
            HIP_TRY(ctx, hipHostMalloc((void **)&sl.h_res, sizeof *sl.h_res,
                                       hipHostMallocDefault));
        }
    }
    *out = ctx->pipe;
    return SNAPMI_OK;
}

// Every exit of a host-buffer call that comes after its first asynchronous
// operation goes through this: an early return (a failed allocation, a
// capacity check, an error of the codec call) must not hand the caller's
// buffers back while copies of earlier slices still read or write them, and
// the next call relies on "the previous call ended with its streams idle".
namespace {
struct PipeDrain {
    snapmi_ctx *ctx;
    snapmi_host_pipe *pipe;
    bool armed = true;
    ~PipeDrain()
    {
        if (!armed)
            return;
        (void)hipStreamSynchronize(pipe->s_in);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(pipe->s_out);
    }
};
} // namespace

static int slot_offsets(snapmi_ctx *ctx, PipeSlot &sl, size_t n)
{
    if (n <= sl.h_off_cap)
        return SNAPMI_OK;
    if (sl.h_off)
        HIP_TRY(ctx, hipHostFree(sl.h_off));
    sl.h_off = nullptr;
    sl.h_off_cap = 0;
    const size_t want = n + n / 4 + 64;
    HIP_TRY(ctx, hipHostMalloc((void **)&sl.h_off, want * sizeof(uint64_t),
                               hipHostMallocDefault));
    sl.h_off_cap = want;
    return SNAPMI_OK;
}

// a slot's device buffer grows only when nothing of the slot is in flight
static int slot_reserve(snapmi_ctx *ctx, snapmi::DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap)
        return SNAPMI_OK;
    if (b.p) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return SNAPMI_OK;
}

extern "C" {

int snapmi_frame_encode_host(snapmi_ctx *ctx, const uint8_t *h_in,
                             const uint32_t *h_chunk_lens, size_t n,
                             uint32_t flags, uint8_t *h_out, size_t out_cap,
                             size_t *written)
{
    if (!ctx || !written || (n && (!h_in || !h_chunk_lens || !h_out)) ||
        n > 0x7FFFFFFFu || (flags & ~(uint32_t)SNAPMI_FRAME_NO_IDENT))
        return SNAPMI_E_ARGUMENT;
    *written = 0;
    if (n == 0)
        return SNAPMI_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
        if (h_chunk_lens[i] == 0 || h_chunk_lens[i] > kMaxBlock)
            return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                            "frame_encode_host: chunk %zu has %u bytes "
                            "(1..65536)", i, h_chunk_lens[i]);
        total += h_chunk_lens[i];
    }
    const size_t need = snapmi_frame_encode_bound(total, n);
    if (out_cap < need - ((flags & SNAPMI_FRAME_NO_IDENT) ? 10 : 0))
        return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                        "frame_encode_host: out_cap %zu < %zu", out_cap, need);
    snapmi_host_pipe *P;
    int rc = host_pipe(ctx, &P);
    if (rc)
        return rc;
    hipStream_t sK = ctx->stream;
    PipeDrain drain{ctx, P};
    // Slices: the match finder wants large launches (its latency floor is
    // ~35 ms whatever the size: DESIGN 4.1), the pipeline wants several
    // slices; by default a batch of 1 GiB or more is cut in two to three.
    const uint64_t slice_bytes = ctx->host_encode_slice;
    struct Slice {
        size_t c0, c1;      // chunks [c0, c1)
        uint64_t b0, bytes; // input bytes
    };
    std::vector<Slice> sl;
    {
        const uint64_t k = total <= slice_bytes
                               ? 1
                               : (total + slice_bytes - 1) / slice_bytes;
        const uint64_t per = (total + k - 1) / k;
        size_t c = 0;
        uint64_t b = 0;
        while (c < n) {
            Slice x{c, c, b, 0};
            while (x.c1 < n && (x.bytes < per || x.c1 == x.c0)) {
                x.bytes += h_chunk_lens[x.c1];
                x.c1++;
            }
            c = x.c1;
            b += x.bytes;
            sl.push_back(x);
        }
    }
    const size_t ns = sl.size();
    uint64_t out_off = 0;
    const bool out_by_kernel =
        (ctx->host_copy_kernel & 2) && device_visible(h_out, out_cap);
    // step t: copy slice t in, start the kernels of slice t-1, send the
    // result of slice t-2 home
    for (size_t t = 0; t < ns + 2; t++) {
        if (t < ns) {
            PipeSlot &s = P->slot[t % kSlots];
            const Slice &x = sl[t];
            const size_t cn = x.c1 - x.c0;
            if ((rc = slot_reserve(ctx, s.in, x.bytes + 16)) ||
                (rc = slot_reserve(ctx, s.out,
                                   snapmi_frame_encode_bound(x.bytes, cn) +
                                       64)) ||
                (rc = slot_reserve(ctx, s.desc, 64 + (cn + 1) * 8)) ||
                (rc = slot_offsets(ctx, s, cn + 1)))
                return rc;
            HIP_TRY(ctx, hipMemcpyAsync(s.in.p, h_in + x.b0, x.bytes,
                                        hipMemcpyHostToDevice, P->s_in));
            HIP_TRY(ctx, hipEventRecord(s.ev_h2d, P->s_in));
        }
        if (t >= 1 && t - 1 < ns) {
            PipeSlot &s = P->slot[(t - 1) % kSlots];
            const Slice &x = sl[t - 1];
            const size_t cn = x.c1 - x.c0;
            s.h_off[0] = 0;
            for (size_t i = 0; i < cn; i++)
                s.h_off[i + 1] = s.h_off[i] + h_chunk_lens[x.c0 + i];
            uint64_t *d_len = (uint64_t *)s.desc.p;
            HIP_TRY(ctx, hipStreamWaitEvent(sK, s.ev_h2d, 0));
            if (t - 1 >= kSlots) // the slot's last result has left
                HIP_TRY(ctx, hipStreamWaitEvent(sK, s.ev_d2h, 0));
            const bool ident =
                t - 1 == 0 && !(flags & SNAPMI_FRAME_NO_IDENT);
            // (the chunk offsets are read where they lie, in pinned host
            // memory: a copy would wait behind the bulk copy of the next
            // slice; so would one of the result)
            rc = snapmi::frame_compress_impl(
                ctx, s.in.p, x.bytes, (uint32_t)cn, s.h_off, ident, s.out.p,
                d_len, nullptr, h_chunk_lens + x.c0);
            if (rc)
                return rc;
            hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, sK,
                               (uint32_t *)&s.h_res->len,
                               (const uint32_t *)d_len, 2u);
            HIP_TRY(ctx, hipEventRecord(s.ev_k, sK));
        }
        if (t >= 2) {
            PipeSlot &s = P->slot[(t - 2) % kSlots];
            HIP_TRY(ctx, hipEventSynchronize(s.ev_k));
            const uint64_t flen = s.h_res->len;
            if (out_off + flen > out_cap)
                return fail_ctx(ctx, SNAPMI_E_DEVICE,
                                "frame_encode_host: %llu > cap",
                                (unsigned long long)(out_off + flen));
            if ((rc = copy_home(ctx, P->s_out, h_out + out_off, s.out.p,
                                flen, out_by_kernel)))
                return rc;
            HIP_TRY(ctx, hipEventRecord(s.ev_d2h, P->s_out));
            out_off += flen;
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(P->s_out));
    drain.armed = false; // (every copy in and every kernel lies in front)
    *written = (size_t)out_off;
    return SNAPMI_OK;
}

int snapmi_frame_decode_host(snapmi_ctx *ctx, const uint8_t *h_in,
                             size_t in_len, uint32_t flags, uint8_t *stale10,
                             uint8_t *h_out, size_t out_cap, size_t *written,
                             size_t *consumed, snapmi_error *err)
{
    if (!ctx || !written || !consumed || (in_len && !h_in) ||
        (out_cap && !h_out) ||
        (flags & ~(uint32_t)(SNAPMI_FRAME_CONTINUATION | SNAPMI_FRAME_FINAL)))
        return SNAPMI_E_ARGUMENT;
    *written = 0;
    *consumed = 0;
    if (err)
        memset(err, 0, sizeof *err);
    if (in_len == 0)
        return SNAPMI_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    snapmi_host_pipe *P;
    int rc = host_pipe(ctx, &P);
    if (rc)
        return rc;
    hipStream_t sK = ctx->stream;
    PipeDrain drain{ctx, P};
    const bool final = (flags & SNAPMI_FRAME_FINAL) != 0;
    uint32_t cflag = flags & SNAPMI_FRAME_CONTINUATION;
    uint8_t stale[10] = {0}; // the reader's src[0..10) in front of `pos`
    if (stale10)
        memcpy(stale, stale10, 10);
    const uint64_t slice_chunks = ctx->host_decode_slice_chunks;
    const bool out_by_kernel =
        (ctx->host_copy_kernel & 1) && device_visible(h_out, out_cap);

    static const bool trace = getenv("SNAPMI_PIPE_TRACE") != nullptr;
    static hipEvent_t ev_base = nullptr;
    if (trace) {
        if (!ev_base)
            (void)hipEventCreate(&ev_base);
        (void)hipEventRecord(ev_base, sK);
    }
    const auto t_start = std::chrono::steady_clock::now();
    auto now_us = [&] {
        return (long)std::chrono::duration_cast<std::chrono::microseconds>(
                   std::chrono::steady_clock::now() - t_start)
            .count();
    };
    // what a slice in flight is
    struct Flight {
        bool live = false, clean = false;
        uint64_t used = 0;
    } fl[kSlots];
    uint64_t pos = 0;      // input consumed by the slices issued so far
    uint64_t wrote = 0;    // output bytes of the slices retired so far
    uint64_t budget = out_cap / kMaxBlock; // chunks the output still holds
    bool more = true;      // another slice may follow
    int result = SNAPMI_OK;
    snapmi_error first_err;
    memset(&first_err, 0, sizeof first_err);
    bool disagree = false;

    // the result of the slice in slot k goes home
    auto retire = [&](int k) -> int {
        PipeSlot &s = P->slot[k];
        Flight &f = fl[k];
        if (!f.live)
            return SNAPMI_OK;
        f.live = false;
        HIP_TRY(ctx, hipEventSynchronize(s.ev_k));
        if (trace) {
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, ev_base, s.ev_h2d);
            (void)hipEventElapsedTime(&b, ev_base, s.ev_k);
            fprintf(stderr, "[pipe] slot %d: h2d done %.1f ms, kernels done "
                            "%.1f ms (device clock)\n", k, a, b);
        }
        const uint64_t len = s.h_res->len;
        if (wrote + len > out_cap)
            return fail_ctx(ctx, SNAPMI_E_DEVICE,
                            "frame_decode_host: %llu > cap",
                            (unsigned long long)(wrote + len));
        if (result == SNAPMI_OK && !disagree) {
            // (nothing behind a failed slice is delivered)
            if (len) {
                if (int e = copy_home(ctx, P->s_out, h_out + wrote, s.out.p,
                                      len, out_by_kernel))
                    return e;
                HIP_TRY(ctx, hipEventRecord(s.ev_d2h, P->s_out));
            }
            wrote += len;
            if (s.h_res->e.kind != SNAPMI_OK) {
                result = s.h_res->e.kind;
                first_err = s.h_res->e;
            } else if (!f.clean) {
                disagree = true; // the host scan saw a bad or cut-off chunk
            }
        }
        return SNAPMI_OK;
    };

    // step t: slice t is issued (copy in on one stream, kernels behind it on
    // the context's); then slice t-1 is retired - its kernels lie in front of
    // slice t's on the same stream and snapmi_frame_decompress_ex waits for
    // its own header pass, so they are done - and its output goes home on the
    // third stream while slice t decodes
    for (uint64_t t = 0;; t++) {
        const int k = (int)(t % kSlots);
        // the slot's previous slice (t - 3) is retired by now (see below)
        if (more && result == SNAPMI_OK && !disagree) {
            PipeSlot &s = P->slot[k];
            const uint64_t maxc = budget < slice_chunks ? budget : slice_chunks;
            uint8_t stale_in[10], stale_work[10];
            memcpy(stale_in, stale, 10);
            memcpy(stale_work, stale, 10);
            uint64_t nd = 0, used = 0;
            const uint8_t *in = h_in + pos;
            const uint64_t left = in_len - pos;
            // the scan stops at the first cut-off (2) or rejected (1) chunk,
            // or in front of the first data chunk beyond the slice (3)
            int status = frame_scan(in, left, cflag, nullptr, nullptr, 0, maxc,
                                    &nd, &used);
            if (status > 3)
                return status;
            if ((rc = slot_offsets(ctx, s, nd + 1)))
                return rc;
            status = frame_scan(in, left, cflag, stale_work, s.h_off, nd + 1,
                                maxc, &nd, &used);
            if (status > 3)
                return status;
            if (status == 3 && used == 0 && budget == 0) {
                if (pos == 0 && out_cap < kMaxBlock)
                    return fail_ctx(ctx, SNAPMI_E_ARGUMENT,
                                    "frame_decode_host: out_cap below 65536");
                more = false; // the output buffer is full: call again
            } else if (used == 0 && status == 2 && !final) {
                more = false; // not one whole chunk here: supply more input
            } else if (left == 0) {
                more = false;
            } else {
                const bool clean =
                    status == 3 || status == 0 || (status == 2 && !final);
                const uint64_t dec_len = clean ? used : left;
                if ((rc = slot_reserve(ctx, s.in, dec_len + 16)) ||
                    (rc = slot_reserve(ctx, s.out, nd * kMaxBlock + 64)) ||
                    (rc = slot_reserve(ctx, s.desc, 128 + (nd + 1) * 8)))
                    return rc;
                if (trace)
                    fprintf(stderr, "[pipe] t=%lu scanned %ld us\n",
                            (unsigned long)t, now_us());
                HIP_TRY(ctx, hipMemcpyAsync(s.in.p, in, dec_len,
                                            hipMemcpyHostToDevice, P->s_in));
                HIP_TRY(ctx, hipEventRecord(s.ev_h2d, P->s_in));
                if (trace)
                    fprintf(stderr, "[pipe] t=%lu h2d issued %ld us (%llu B)\n",
                            (unsigned long)t, now_us(),
                            (unsigned long long)dec_len);
                uint64_t *d_len = (uint64_t *)s.desc.p;
                snapmi_error *d_err =
                    (snapmi_error *)((uint8_t *)s.desc.p + 64);
                HIP_TRY(ctx, hipStreamWaitEvent(sK, s.ev_h2d, 0));
                if (t >= kSlots) // the slot's last result has left
                    HIP_TRY(ctx, hipStreamWaitEvent(sK, s.ev_d2h, 0));
                const bool with_index = clean && nd > 0;
                // (the side index is read where it lies, in pinned host
                // memory, and the result is posted by a kernel: copies on
                // this stream would wait behind the bulk copies)
                rc = snapmi_frame_decompress_ex(
                    ctx, s.in.p, dec_len, s.out.p, nd * kMaxBlock, d_len,
                    d_err, with_index ? s.h_off : nullptr, nd, cflag,
                    stale_in);
                if (rc)
                    return rc;
                static_assert(sizeof(PipeSlot::Result) == 40, "len + error");
                hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, sK,
                                   (uint32_t *)&s.h_res->len,
                                   (const uint32_t *)d_len, 2u);
                hipLaunchKernelGGL(k_post_words, dim3(1), dim3(64), 0, sK,
                                   (uint32_t *)&s.h_res->e,
                                   (const uint32_t *)d_err, 8u);
                HIP_TRY(ctx, hipEventRecord(s.ev_k, sK));
                if (trace)
                    fprintf(stderr, "[pipe] t=%lu kernels issued %ld us\n",
                            (unsigned long)t, now_us());
                fl[k].live = true;
                fl[k].clean = clean;
                fl[k].used = used;
                if (clean) {
                    pos += used;
                    budget -= nd;
                    cflag = SNAPMI_FRAME_CONTINUATION;
                    memcpy(stale, stale_work, 10);
                }
                // 3: the slice was full, the next one follows; anything else
                // ends the batch (all consumed / a cut-off tail / an error
                // the device is about to name)
                more = status == 3;
            }
        } else {
            more = false;
        }
        if (t >= 1 && (rc = retire((int)((t - 1) % kSlots))))
            return rc;
        if (trace)
            fprintf(stderr, "[pipe] t=%lu retired %ld us\n", (unsigned long)t,
                    now_us());
        if (!more)
            break;
    }
    for (int k = 0; k < kSlots; k++)
        if ((rc = retire(k)))
            return rc;
    HIP_TRY(ctx, hipStreamSynchronize(P->s_out));
    drain.armed = false; // all slices retired, their copies home are done
    if (trace) {
        for (int k = 0; k < kSlots; k++) {
            float a = 0;
            if (hipEventElapsedTime(&a, ev_base, P->slot[k].ev_d2h) ==
                hipSuccess)
                fprintf(stderr, "[pipe] slot %d: last d2h done %.1f ms\n", k,
                        a);
        }
        fprintf(stderr, "[pipe] done %ld us\n", now_us());
    }
    *written = (size_t)wrote;
    if (result != SNAPMI_OK) {
        if (err)
            *err = first_err;
        return result;
    }
    if (disagree)
        return fail_ctx(ctx, SNAPMI_E_DEVICE,
                        "frame_decode_host: host scan and device walk "
                        "disagree");
    *consumed = (size_t)pos;
    if (stale10)
        memcpy(stale10, stale, 10);
    return SNAPMI_OK;
}

} // extern "C"
