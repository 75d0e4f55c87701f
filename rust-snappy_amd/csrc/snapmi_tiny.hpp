// snapmi: one raw stream of fewer than 256 bytes compressed by ONE thread.
//
// The batch compressor gives a 64 KiB block to a lane (tables in HBM) or to a
// wavefront (table in LDS).  A stream of a couple of hundred bytes is neither:
// its whole state - input, the reference's smallest hash table (256 entries,
// and every position fits a byte), output - is under 800 bytes, so 64 of them
// fit a wavefront's share of LDS and a round costs LDS latency instead of an
// HBM round trip (k_compress_tiny, snapmi_compress.hip).
//
// The algorithm is the reference's, statement for statement
// (src/compress.rs:99-154 compress, :195-317 compress_block, :378-412
// extend_match, :323-369 emit_copy, :433-474 emit_literal, :491-518 table
// size for inputs under 256 bytes: 256 entries, shift 24), written over a
// memory policy M so that the very same text is run on the host by
// tests/test_tiny_lane_cpu.py (byte arrays) and on the device (one lane's
// dword-interleaved LDS columns):
//
//   uint32_t M::in8(k)         input byte k
//   uint32_t M::in32(k)        input bytes k .. k+3, little endian (k + 4 <= n)
//   uint32_t M::tab(h)         table entry h (0 .. 255), zero at the start
//   void     M::tab_set(h, v)
//   void     M::out8(k, v)     output byte k (k < kTinyOutMax)
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

// streams of 1 .. kTinyCompress - 1 bytes are k_compress_tiny's
constexpr uint32_t kTinyCompress = 256;
// What such a stream can grow to: 2 bytes of header + the elements, and the
// elements of a block never exceed the input by more than the tag of its last
// literal (1 or 2 bytes) - a literal of L bytes in front of a copy costs
// L + 1 (+ 1 over 60 bytes) and the copy 2 or 3 bytes for at least 4 (12)
// bytes of input, so no literal + copy pair expands.
constexpr uint32_t kTinyOutMax = kTinyCompress - 1 + 2 + 2;

template <class M>
SNAPMI_LANE_FN uint32_t tiny_put_literal(M &m, uint32_t d, uint32_t from,
                                         uint32_t len)
{
    const uint32_t n1 = len - 1; // src/compress.rs:433-474
    if (n1 <= 59) {
        m.out8(d++, n1 << 2);
    } else {
        m.out8(d++, 60u << 2);
        m.out8(d++, n1);
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
    // src/compress.rs:323-357 (offsets are under 256 here)
    while (len >= 68) {
        m.out8(d++, (63u << 2) | 2);
        m.out8(d++, offset);
        m.out8(d++, 0);
        len -= 64;
    }
    if (len > 64) {
        m.out8(d++, (59u << 2) | 2);
        m.out8(d++, offset);
        m.out8(d++, 0);
        len -= 60;
    }
    if (len <= 11) {
        m.out8(d++, ((len - 4) << 2) | 1);
        m.out8(d++, offset);
    } else {
        m.out8(d++, ((len - 1) << 2) | 2);
        m.out8(d++, offset);
        m.out8(d++, 0);
    }
    return d;
}

SNAPMI_LANE_FN uint32_t tiny_hash(uint32_t x)
{
    return (x * 0x1E35A7BDu) >> 24; // src/compress.rs:523-525, 256 entries
}

// n = 1 .. kTinyCompress - 1 input bytes; returns the stream's length
template <class M> SNAPMI_LANE_FN uint32_t tiny_compress(M &m, uint32_t n)
{
    uint32_t d = 0;
    if (n < 128) { // the header: src/compress.rs:127
        m.out8(d++, n);
    } else {
        m.out8(d++, (n & 127) | 128);
        m.out8(d++, n >> 7);
    }
    if (n < 17) // src/compress.rs:140-146
        return tiny_put_literal(m, d, 0, n);

    const uint32_t s_limit = n - 15;
    uint32_t s = 1, next_emit = 0;
    uint32_t next_hash = tiny_hash(m.in32(1));
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
            next_hash = tiny_hash(m.in32(s_next));
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
            m.tab_set(tiny_hash(x0), s - 1);
            const uint32_t h = tiny_hash(x1);
            cand = m.tab(h);
            m.tab_set(h, s);
            if (x1 != m.in32(cand)) {
                next_hash = tiny_hash(m.in32(s + 1));
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
