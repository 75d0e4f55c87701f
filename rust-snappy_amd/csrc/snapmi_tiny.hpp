// snapmi: one small raw stream compressed by ONE thread.
//
// The batch compressor gives a 64 KiB block to a lane (tables in HBM) or to a
// wavefront (table in LDS).  A stream of a few hundred bytes is neither: its
// whole state - input, the reference's table for that length (256 entries
// for inputs up to 256 bytes, the next power of two above the length beyond),
// output - is a few KiB at most, so many of them fit a CU's LDS and a probe
// costs an LDS round trip instead of an HBM one (k_compress_tiny: streams
// under 256 bytes, 64 per wavefront; k_compress_small: streams under 2 KiB, a
// few per wavefront; snapmi_compress.hip).
//
// The algorithm is the reference's, statement for statement
// (src/compress.rs:99-154 compress, :195-317 compress_block, :378-412
// extend_match, :323-369 emit_copy, :433-474 emit_literal, :491-518 table
// size), for a stream of ONE block (n <= 65 536), written over a memory
// policy M so that the very same text is run on the host by
// tests/test_tiny_lane_cpu.py (byte arrays) and on the device (LDS):
//
//   uint32_t M::in8(k)         input byte k
//   uint32_t M::in32(k)        input bytes k .. k+3, little endian (k + 4 <= n)
//   uint32_t M::tab(h)         table entry h, zero at the start
//   void     M::tab_set(h, v)  v < n
//   void     M::out8(k, v)     output byte k
//   void     M::out32(k, v)    output bytes k .. k+3, k a multiple of 4
#ifndef SNAPMI_TINY_HPP
#define SNAPMI_TINY_HPP

#include <stdint.h>

#if defined(__HIPCC__)
#define SNAPMI_LANE_FN __host__ __device__ __forceinline__
#else
#define SNAPMI_LANE_FN inline
#endif

namespace snapmi {

// streams of 1 .. kTinyCompress - 1 bytes are k_compress_tiny's, streams of
// kTinyCompress .. kSmallCompress - 1 bytes k_compress_small's
constexpr uint32_t kTinyCompress = 256;
constexpr uint32_t kSmallCompress = 2048;
// What such a stream can grow to.  A literal of L bytes costs L + 1 (+ 1 over
// 60 bytes, + 2 over 256), and while every offset is under 2048 a copy costs
// 2 bytes for 4 .. 11 bytes of input or 3 for 12 and more: a literal + copy
// pair expands only when the literal is longer than 256 bytes, and then by
// one byte.  Under 256 bytes of input: header (2) + the last literal's tag
// (2).  Under 2048: header (2) + at most 7 such pairs + the last tag (3).
constexpr uint32_t kTinyOutMax = kTinyCompress - 1 + 2 + 2;
constexpr uint32_t kSmallOutSlack = 2 + 7 + 3; // output <= input + this

template <class M>
SNAPMI_LANE_FN uint32_t tiny_put_literal(M &m, uint32_t d, uint32_t from,
                                         uint32_t len)
{
    const uint32_t n1 = len - 1; // src/compress.rs:433-474
    if (n1 <= 59) {
        m.out8(d++, n1 << 2);
    } else if (n1 < 256) {
        m.out8(d++, 60u << 2);
        m.out8(d++, n1);
    } else {
        m.out8(d++, 61u << 2);
        m.out8(d++, n1 & 255);
        m.out8(d++, n1 >> 8);
    }
    // bytes up to the output's next dword, whole dwords, the rest
    uint32_t k = 0;
    for (; k < len && ((d + k) & 3); k++)
        m.out8(d + k, m.in8(from + k));
    for (; k + 4 <= len; k += 4)
        m.out32(d + k, m.in32(from + k));
    for (; k < len; k++)
        m.out8(d + k, m.in8(from + k));
    return d + len;
}

template <class M>
SNAPMI_LANE_FN uint32_t tiny_put_copy(M &m, uint32_t d, uint32_t offset,
                                      uint32_t len)
{
    // src/compress.rs:323-357
    while (len >= 68) {
        m.out8(d++, (63u << 2) | 2);
        m.out8(d++, offset & 255);
        m.out8(d++, offset >> 8);
        len -= 64;
    }
    if (len > 64) {
        m.out8(d++, (59u << 2) | 2);
        m.out8(d++, offset & 255);
        m.out8(d++, offset >> 8);
        len -= 60;
    }
    if (len <= 11 && offset <= 2047) {
        m.out8(d++, ((offset >> 8) << 5) | ((len - 4) << 2) | 1);
        m.out8(d++, offset & 255);
    } else {
        m.out8(d++, ((len - 1) << 2) | 2);
        m.out8(d++, offset & 255);
        m.out8(d++, offset >> 8);
    }
    return d;
}

SNAPMI_LANE_FN uint32_t tiny_hash(uint32_t x, uint32_t shift)
{
    return (x * 0x1E35A7BDu) >> shift; // src/compress.rs:523-525
}

// entries of the table of a block of n bytes: src/compress.rs:491-518
SNAPMI_LANE_FN uint32_t tiny_table_size(uint32_t n)
{
    uint32_t size = 256;
    while (size < 16384 && size < n)
        size *= 2;
    return size;
}

// n = 1 .. 65 536 input bytes (one block); returns the stream's length
template <class M> SNAPMI_LANE_FN uint32_t tiny_compress(M &m, uint32_t n)
{
    uint32_t d = 0;
    for (uint32_t v = n;;) { // the header: src/compress.rs:127
        if (v < 128) {
            m.out8(d++, v);
            break;
        }
        m.out8(d++, (v & 127) | 128);
        v >>= 7;
    }
    if (n < 17) // src/compress.rs:140-146
        return tiny_put_literal(m, d, 0, n);

    uint32_t shift = 24;
    for (uint32_t size = 256; size < 16384 && size < n; size *= 2)
        shift--;
    const uint32_t s_limit = n - 15;
    uint32_t s = 1, next_emit = 0;
    uint32_t next_hash = tiny_hash(m.in32(1), shift);
    bool done = false;
    while (!done) {
        // the skip loop, src/compress.rs:204-245
        uint32_t skip = 32, s_next = s, cand = 0;
        for (;;) {
            s = s_next;
            const uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            if (s_next > s_limit) {
                done = true;
                break;
            }
            cand = m.tab(next_hash);
            m.tab_set(next_hash, s);
            next_hash = tiny_hash(m.in32(s_next), shift);
            if (m.in32(s) == m.in32(cand))
                break;
        }
        if (done)
            break;
        d = tiny_put_literal(m, d, next_emit, s - next_emit);
        // the copy chain, src/compress.rs:258-315
        for (;;) {
            const uint32_t base = s;
            uint32_t c = cand + 4;
            s += 4;
            // extend_match: the longest common run, whatever the stride
            bool open = true;
            while (s + 4 <= n) {
                const uint32_t z = m.in32(s) ^ m.in32(c);
                if (z) {
                    s += (uint32_t)__builtin_ctz(z) >> 3;
                    open = false;
                    break;
                }
                s += 4;
                c += 4;
            }
            if (open)
                while (s < n && m.in8(s) == m.in8(c)) {
                    s++;
                    c++;
                }
            d = tiny_put_copy(m, d, base - cand, s - base);
            next_emit = s;
            if (s >= s_limit) {
                done = true;
                break;
            }
            const uint32_t x0 = m.in32(s - 1), x1 = m.in32(s);
            m.tab_set(tiny_hash(x0, shift), s - 1);
            const uint32_t h = tiny_hash(x1, shift);
            cand = m.tab(h);
            m.tab_set(h, s);
            if (x1 != m.in32(cand)) {
                next_hash = tiny_hash(m.in32(s + 1), shift);
                s++;
                break;
            }
        }
    }
    if (next_emit < n) // src/compress.rs:417-426
        d = tiny_put_literal(m, d, next_emit, n - next_emit);
    return d;
}

} // namespace snapmi
#endif
