// snapmi_compress.hip -- Snappy raw block compressor for gfx950 (CDNA4).
//
// What is computed is fixed by the reference (src/compress.rs: one greedy
// LZ77 parse per <=64 KiB block with a 16-bit hash table, the skip heuristic
// and the literal / copy-1 / copy-2 encoders) and must be bit-exact.  How it
// is computed is CDNA4-native:
//
//   * one wavefront (a 64-thread workgroup) owns one 64 KiB block; blocks are
//     independent (fresh zeroed table per block, offsets never leave the
//     block: reference src/compress.rs:148,514-516), so the grid is simply
//     "all blocks of all streams of the batch";
//   * the u16 hash table (<=32 KiB) lives in LDS, zeroed with ds_write_b128;
//     5 blocks are resident per CU (5 x 32 KiB = 160 KiB);
//   * the sequential state (s, next_emit, d, skip, hashes) is wave-uniform
//     and sits in SGPRs; hashes are taken from a 256-byte register window of
//     the input (v_readlane), not from memory;
//   * a probe is verified AND extended in one step: the 64 lanes load
//     4 bytes each at candidate+4i and s+4i, one ballot gives the match
//     length (reference extend_match, src/compress.rs:378-412, is a serial
//     8-byte loop);
//   * literals are copied 256 bytes per wave instruction.
//
// Blocks 0 of every stream are written straight into the caller's output
// (after the varint); later blocks go to scratch slots and are moved into
// place by k_compact once the sizes are known (reference Encoder::compress
// concatenates them serially, src/compress.rs:128-153).
#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"

namespace snapmi {

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x, uint32_t shift)
{
    return (x * 0x1E35A7BDu) >> shift; // reference src/compress.rs:523-525
}

struct BlockEnc {
    const uint8_t *src; // block start
    uint64_t avail;     // readable bytes from src (>= n)
    uint32_t n;         // block length
    uint8_t *dst;
    uint32_t d;
    uint32_t lane;

    // reference emit_literal, src/compress.rs:433-474
    __device__ __forceinline__ void emit_literal(uint32_t from, uint32_t to)
    {
        const uint32_t len = to - from;
        const uint32_t n1 = len - 1;
        uint32_t hdr;
        if (n1 < 60) {
            if (lane == 0)
                dst[d] = (uint8_t)(n1 << 2);
            hdr = 1;
        } else if (n1 < 256) {
            if (lane == 0) {
                dst[d] = 60 << 2;
                dst[d + 1] = (uint8_t)n1;
            }
            hdr = 2;
        } else {
            if (lane == 0) {
                dst[d] = 61 << 2;
                dst[d + 1] = (uint8_t)n1;
                dst[d + 2] = (uint8_t)(n1 >> 8);
            }
            hdr = 3;
        }
        uint8_t *o = dst + d + hdr;
        const uint8_t *in = src + from;
        for (uint32_t i = 4 * lane; i + 4 <= len; i += 4 * kWave)
            st32u(o + i, ld32u(in + i));
        const uint32_t t = len & ~3u;
        if (lane < (len & 3u))
            o[t + lane] = in[t + lane];
        d += hdr + len;
    }

    // reference emit_copy / emit_copy2, src/compress.rs:323-369
    __device__ __forceinline__ void emit_copy(uint32_t offset, uint32_t len)
    {
        while (len >= 68) {
            if (lane == 0) {
                dst[d] = (uint8_t)((63u << 2) | 2u);
                dst[d + 1] = (uint8_t)offset;
                dst[d + 2] = (uint8_t)(offset >> 8);
            }
            d += 3;
            len -= 64;
        }
        if (len > 64) {
            if (lane == 0) {
                dst[d] = (uint8_t)((59u << 2) | 2u);
                dst[d + 1] = (uint8_t)offset;
                dst[d + 2] = (uint8_t)(offset >> 8);
            }
            d += 3;
            len -= 60;
        }
        if (len <= 11 && offset <= 2047) {
            if (lane == 0) {
                dst[d] =
                    (uint8_t)(((offset >> 8) << 5) | ((len - 4) << 2) | 1u);
                dst[d + 1] = (uint8_t)offset;
            }
            d += 2;
        } else {
            if (lane == 0) {
                dst[d] = (uint8_t)(((len - 1) << 2) | 2u);
                dst[d + 1] = (uint8_t)offset;
                dst[d + 2] = (uint8_t)(offset >> 8);
            }
            d += 3;
        }
    }

    // Length of the common prefix of src[cand..] and src[s..], bounded by the
    // block end (reference: 4-byte verify at src/compress.rs:239-243 /
    // :305-306 plus extend_match :378-412, fused).  >= 4 means "hit".
    __device__ __forceinline__ uint32_t match_len(uint32_t cand, uint32_t s)
    {
        uint32_t len = 0;
        const uint32_t room = n - s;
        for (uint32_t pos = 0;; pos += 4 * kWave) {
            const uint32_t off = pos + 4 * lane;
            uint32_t eq = 0;
            if (off < room) {
                const uint32_t a = ld32g(src, cand + off, avail);
                const uint32_t b = ld32g(src, s + off, avail);
                const uint32_t x = a ^ b;
                eq = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
                const uint32_t lim = room - off;
                eq = eq < lim ? eq : lim;
            }
            const uint64_t stop = __ballot(eq < 4);
            if (stop == 0) {
                len += 4 * kWave;
                continue;
            }
            const uint32_t f = (uint32_t)__builtin_ctzll(stop);
            len += 4 * f + (uint32_t)__builtin_amdgcn_readlane(eq, f);
            return len;
        }
    }
};

} // namespace

// ---------------------------------------------------------------------
// K1: one wavefront per block.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_compress_blocks(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t table[kMaxTable];

    const uint32_t lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t nblocks = a.blk_first[a.n_streams];
    if (b >= nblocks)
        return;

    // stream lookup: blk_first[st] <= b < blk_first[st + 1]
    uint32_t lo = 0, hi = a.n_streams;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.blk_first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t st = lo;
    const uint32_t k = b - a.blk_first[st];
    const uint64_t total = a.in_lens[st];
    const uint64_t boff = (uint64_t)k * kMaxBlock;

    BlockEnc e;
    e.lane = lane;
    e.src = (const uint8_t *)a.in_ptrs[st] + boff;
    e.avail = total - boff;
    e.n = e.avail < kMaxBlock ? (uint32_t)e.avail : kMaxBlock;
    e.d = 0;
    if (k == 0) {
        // varint(total) then block 0, in place: reference
        // src/compress.rs:128 and src/bytes.rs:61-70
        e.dst = (uint8_t *)a.out_ptrs[st];
        if (lane == 0) {
            uint64_t v = total;
            uint32_t i = 0;
            while (v >= 0x80) {
                e.dst[i++] = (uint8_t)v | 0x80;
                v >>= 7;
            }
            e.dst[i] = (uint8_t)v;
        }
        e.dst += varint_len(total);
    } else {
        const uint32_t slot = a.slot_first[st] + k - 1;
        if (slot >= a.host_slots)
            return; // stream rejected by k_plan_compress (E_ARGUMENT)
        e.dst = a.scratch + (uint64_t)slot * kSlotBytes;
    }
    const uint32_t n = e.n;

    if (n < kMinNonLiteral) { // reference src/compress.rs:140-146
        e.emit_literal(0, n);
        if (lane == 0)
            a.blk_size[b] = e.d;
        return;
    }

    // table sizing + zero fill: reference src/compress.rs:491-518
    uint32_t shift = 32 - 8, tsize = 256;
    while (tsize < kMaxTable && tsize < n) {
        shift--;
        tsize *= 2;
    }
    for (uint32_t i = 8 * lane; i < tsize; i += 8 * kWave)
        *(uint4 *)&table[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    ByteWindow win;
    win.init(e.src, e.avail);

    // reference Block::compress, src/compress.rs:195-317
    uint32_t s = 1, next_emit = 0;
    const uint32_t s_limit = n - kInputMargin;
    uint32_t next_hash = hash32(win.get32(1), shift);
    for (;;) {
        uint32_t skip = 32, s_next = s, cand, mlen;
        for (;;) { // probe loop, :207-245
            s = s_next;
            const uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            if (s_next > s_limit)
                goto done;
            cand = uni(table[next_hash]);
            if (lane == 0)
                table[next_hash] = (uint16_t)s;
            next_hash = hash32(win.get32(s_next), shift);
            mlen = e.match_len(cand, s);
            if (mlen >= 4)
                break;
        }
        e.emit_literal(next_emit, s);
        for (;;) { // copy chain, :258-315
            const uint32_t base = s;
            s += mlen;
            e.emit_copy(base - cand, mlen);
            next_emit = s;
            if (s >= s_limit)
                goto done;
            const uint64_t x = win.get64(s - 1);
            if (lane == 0)
                table[hash32((uint32_t)x, shift)] = (uint16_t)(s - 1);
            const uint32_t h = hash32((uint32_t)(x >> 8), shift);
            cand = uni(table[h]);
            if (lane == 0)
                table[h] = (uint16_t)s;
            mlen = e.match_len(cand, s);
            if (mlen < 4) {
                next_hash = hash32((uint32_t)(x >> 16), shift);
                s += 1;
                break;
            }
        }
    }
done:
    if (next_emit < n) // reference done(), src/compress.rs:417-426
        e.emit_literal(next_emit, n);
    if (lane == 0)
        a.blk_size[b] = e.d;
}

// ---------------------------------------------------------------------
// Plan: per-stream validation + block table.  One workgroup of 1024 threads
// walks the streams 1024 at a time with a workgroup-wide exclusive scan.
// Reference checks: src/compress.rs:104-125.
// ---------------------------------------------------------------------
__device__ __forceinline__ uint2 wg_scan2(uint32_t x, uint32_t y,
                                          uint2 *wave_tot, uint2 *total)
{
    // inclusive scan inside the wave
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t sx = x, sy = y;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t tx = __shfl_up(sx, o), ty = __shfl_up(sy, o);
        if (lane >= o) {
            sx += tx;
            sy += ty;
        }
    }
    if (lane == 63)
        wave_tot[w] = make_uint2(sx, sy);
    __syncthreads();
    uint32_t bx = 0, by = 0, allx = 0, ally = 0;
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t i = 0; i < nw; i++) {
        const uint2 t = wave_tot[i];
        if (i < w) {
            bx += t.x;
            by += t.y;
        }
        allx += t.x;
        ally += t.y;
    }
    __syncthreads();
    *total = make_uint2(allx, ally);
    return make_uint2(bx + sx - x, by + sy - y); // exclusive
}

__global__ __launch_bounds__(1024) void k_plan_compress(CompressArgs a)
{
    __shared__ uint2 wave_tot[16];
    uint32_t carry_b = 0, carry_s = 0;
    for (uint32_t base = 0; base < a.n_streams; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        uint32_t nb = 0;
        if (i < a.n_streams) {
            const uint64_t len = a.in_lens[i];
            const uint64_t need = max_compress_len_u64(len);
            a.out_lens[i] = 0;
            if (need == 0) {
                set_error(a.errs, i, SNAPMI_TOO_BIG, len, kMaxInput, 0);
            } else if (a.out_caps && a.out_caps[i] < need) {
                set_error(a.errs, i, SNAPMI_BUFFER_TOO_SMALL, a.out_caps[i],
                          need, 0);
            } else if (len == 0) { // src/compress.rs:120-125
                ((uint8_t *)a.out_ptrs[i])[0] = 0;
                a.out_lens[i] = 1;
                set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
            } else {
                nb = (uint32_t)((len + kMaxBlock - 1) / kMaxBlock);
                set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
            }
        }
        uint2 tot;
        const uint2 ex = wg_scan2(nb, nb ? nb - 1 : 0, wave_tot, &tot);
        uint32_t fb = carry_b + ex.x, fs = carry_s + ex.y;
        if (i < a.n_streams) {
            // The launch was sized from the host's copy of the lengths; a
            // stream that does not fit in it is rejected, never overrun.
            if (nb && ((uint64_t)fb + nb > a.host_blocks ||
                       (uint64_t)fs + (nb - 1) > a.host_slots))
                set_error(a.errs, i, SNAPMI_E_ARGUMENT, a.in_lens[i], 0, 0);
            a.blk_first[i] = fb;
            a.slot_first[i] = fs;
        }
        carry_b += tot.x;
        carry_s += tot.y;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.blk_first[a.n_streams] = carry_b;
        a.slot_first[a.n_streams] = carry_s;
    }
}

// Exclusive scan of blk_size (u32) into blk_off (u64), one workgroup.
__global__ __launch_bounds__(1024) void k_scan_sizes(CompressArgs a)
{
    __shared__ uint64_t wave_tot[16];
    const uint32_t *blk_size = a.blk_size;
    uint64_t *blk_off = a.blk_off;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t x = i < nblocks ? blk_size[i] : 0;
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
        if (i < nblocks)
            blk_off[i] = carry + before + sx - x;
        carry += all;
    }
    if (threadIdx.x == 0)
        blk_off[nblocks] = carry;
}

// ---------------------------------------------------------------------
// K3: move blocks 1.. of every stream from their scratch slot to
// out_ptrs[st] + varint + sum(size of earlier blocks); block 0's workgroup
// publishes the stream's compressed length.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(CompressArgs a)
{
    const uint32_t b = blockIdx.x;
    const uint32_t nblocks = a.blk_first[a.n_streams];
    if (b >= nblocks)
        return;
    uint32_t lo = 0, hi = a.n_streams;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.blk_first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t st = lo;
    const uint32_t first = a.blk_first[st];
    const uint32_t k = b - first;
    const uint64_t total = a.in_lens[st];
    const uint32_t vl = varint_len(total);
    const uint32_t nb = (uint32_t)((total + kMaxBlock - 1) / kMaxBlock);
    if ((uint64_t)first + nb > a.host_blocks ||
        (uint64_t)a.slot_first[st] + (nb - 1) > a.host_slots)
        return; // rejected by k_plan_compress (E_ARGUMENT), out_lens stays 0
    if (k == 0) {
        if (threadIdx.x == 0)
            a.out_lens[st] = vl + (a.blk_off[first + nb] - a.blk_off[first]);
        return;
    }
    const uint8_t *from =
        a.scratch + (uint64_t)(a.slot_first[st] + k - 1) * kSlotBytes;
    uint8_t *to =
        (uint8_t *)a.out_ptrs[st] + vl + (a.blk_off[b] - a.blk_off[first]);
    const uint32_t size = a.blk_size[b];
    // align the destination to 16 bytes, then 16-byte stores fed by
    // unaligned 16-byte loads (the slot is 16-aligned, `to` is arbitrary).
    uint32_t head = (uint32_t)((16 - ((uintptr_t)to & 15)) & 15);
    if (head > size)
        head = size;
    if (threadIdx.x < head)
        to[threadIdx.x] = from[threadIdx.x];
    const uint32_t body = (size - head) & ~15u;
    for (uint32_t i = 16 * threadIdx.x; i < body; i += 16 * blockDim.x) {
        uint4 v;
        __builtin_memcpy(&v, from + head + i, 16);
        *(uint4 *)(to + head + i) = v;
    }
    const uint32_t tail = size - head - body;
    if (threadIdx.x < tail)
        to[head + body + threadIdx.x] = from[head + body + threadIdx.x];
}

} // namespace snapmi
