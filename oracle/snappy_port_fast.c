/*
 * snappy_port_fast.c -- TIMING-ONLY restatement of the reference's raw codec
 * WITH its fast paths.  Test/bench infrastructure, never shipped, never the
 * parity oracle: parity is pinned on snappy_oracle.c, which restates the
 * reference's RESULTS (bytes, and which error wins) with plain loops.  What
 * that file leaves out on purpose - the reference's 16-byte blind copies, its
 * tag lookup table, its three copy strategies - is what makes the reference
 * as fast as C++ snappy (README.md:135-158), so a "reference CPU path timed
 * beside the GPU" (bench.py's cpu_baseline) that omits them sells the CPU
 * short (round 5: the plain port decoded at 0.65 GiB/s per thread, libsnappy
 * 1.1.8 at 1.84).  This file restates them, line for line in behaviour:
 *
 *   compress    src/compress.rs:99-154 (driver), :195-317 (match finder),
 *               :378-412 (extend_match, 8 bytes at a time), :433-474
 *               (emit_literal with the 16-byte fast path), :323-369
 *               (emit_copy), :491-518 (table: 1 024-entry small table for
 *               short inputs, 16 384 entries otherwise)
 *   decompress  src/decompress.rs:130-148 (dispatch), :161-228 (read_literal
 *               with the 16-byte fast path), :233-343 (read_copy: two 8-byte
 *               moves / 16 bytes at a time with the overlap pre-roll / byte
 *               by byte), :398-474 + build.rs:40-67 (the tag lookup table,
 *               one masked u32 load for the offset)
 *
 * Entry points carry the snappy-c.h signatures so that oracle/snappy_oracle.c
 * ::snapo_bench_ext times them like it times libsnappy.  tests/
 * test_oracle_cpu.py checks their bytes and statuses against the oracle on
 * the corpus, random inputs and the error KATs - a timing whose results are
 * wrong would be worthless.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define MAX_INPUT_SIZE 0xFFFFFFFFull
#define MAX_BLOCK_SIZE 65536u
#define MAX_TABLE_SIZE 16384u
#define SMALL_TABLE_SIZE 1024u
#define INPUT_MARGIN 15u
#define MIN_NON_LITERAL_BLOCK 17u

#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)

static inline uint32_t ld32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint64_t ld64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
static inline void cp8(uint8_t *d, const uint8_t *s) { memcpy(d, s, 8); }
static inline void cp16(uint8_t *d, const uint8_t *s) { memcpy(d, s, 16); }

size_t snapf_max_compressed_length(size_t n)
{
    if ((uint64_t)n > MAX_INPUT_SIZE)
        return 0;
    uint64_t m = 32 + (uint64_t)n + (uint64_t)n / 6;
    return m > MAX_INPUT_SIZE ? 0 : (size_t)m;
}

/* ---- compress -------------------------------------------------------- */

typedef struct {
    const uint8_t *src;
    size_t n;
    uint8_t *dst;
    size_t d;
} fblock;

/* src/compress.rs:433-474 */
static inline void emit_literal(fblock *b, size_t lit_start, size_t lit_end)
{
    const size_t len = lit_end - lit_start;
    const size_t n = len - 1;
    uint8_t *dst = b->dst;
    if (LIKELY(n <= 59)) {
        dst[b->d++] = (uint8_t)(n << 2);
        if (LIKELY(len <= 16 && lit_start + 16 <= b->n)) {
            cp16(dst + b->d, b->src + lit_start); /* :440-453 */
            b->d += len;
            return;
        }
    } else if (n < 256) {
        dst[b->d] = 60 << 2;
        dst[b->d + 1] = (uint8_t)n;
        b->d += 2;
    } else {
        dst[b->d] = 61 << 2;
        dst[b->d + 1] = (uint8_t)n;
        dst[b->d + 2] = (uint8_t)(n >> 8);
        b->d += 3;
    }
    memcpy(dst + b->d, b->src + lit_start, len);
    b->d += len;
}

/* src/compress.rs:363-369 */
static inline void emit_copy2(fblock *b, size_t offset, size_t len)
{
    uint8_t *o = b->dst + b->d;
    o[0] = (uint8_t)(((len - 1) << 2) | 2);
    o[1] = (uint8_t)offset;
    o[2] = (uint8_t)(offset >> 8);
    b->d += 3;
}

/* src/compress.rs:323-357 */
static inline void emit_copy(fblock *b, size_t offset, size_t len)
{
    while (UNLIKELY(len >= 68)) {
        emit_copy2(b, offset, 64);
        len -= 64;
    }
    if (UNLIKELY(len > 64)) {
        emit_copy2(b, offset, 60);
        len -= 60;
    }
    if (len <= 11 && offset <= 2047) {
        uint8_t *o = b->dst + b->d;
        o[0] = (uint8_t)(((offset >> 8) << 5) | ((len - 4) << 2) | 1);
        o[1] = (uint8_t)offset;
        b->d += 2;
    } else {
        emit_copy2(b, offset, len);
    }
}

/* src/compress.rs:195-317 */
static void compress_block(fblock *b, uint16_t *table, unsigned shift)
{
    const uint8_t *src = b->src;
    const size_t n = b->n;
    size_t s = 1, next_emit = 0;
    const size_t s_limit = n - INPUT_MARGIN;
    uint32_t next_hash = (ld32(src + s) * 0x1E35A7BDu) >> shift;

    for (;;) {
        uint32_t skip = 32;
        size_t s_next = s, cand;
        for (;;) { /* :204-245 */
            s = s_next;
            const uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            if (UNLIKELY(s_next > s_limit))
                goto done;
            cand = table[next_hash];
            table[next_hash] = (uint16_t)s;
            next_hash = (ld32(src + s_next) * 0x1E35A7BDu) >> shift;
            if (ld32(src + s) == ld32(src + cand))
                break;
        }
        emit_literal(b, next_emit, s);
        for (;;) { /* :258-315 */
            const size_t base = s;
            size_t c = cand + 4;
            s += 4;
            /* extend_match :378-412 */
            while (s + 8 <= n) {
                const uint64_t z = ld64(src + s) ^ ld64(src + c);
                if (z) {
                    s += (size_t)__builtin_ctzll(z) >> 3;
                    goto extended;
                }
                s += 8;
                c += 8;
            }
            while (s < n && src[s] == src[c]) {
                s++;
                c++;
            }
        extended:
            emit_copy(b, base - cand, s - base);
            next_emit = s;
            if (UNLIKELY(s >= s_limit))
                goto done;
            const uint64_t x = ld64(src + s - 1);
            table[((uint32_t)x * 0x1E35A7BDu) >> shift] = (uint16_t)(s - 1);
            const uint32_t h = ((uint32_t)(x >> 8) * 0x1E35A7BDu) >> shift;
            cand = table[h];
            table[h] = (uint16_t)s;
            if ((uint32_t)(x >> 8) != ld32(src + cand)) {
                next_hash = ((uint32_t)(x >> 16) * 0x1E35A7BDu) >> shift;
                s++;
                break;
            }
        }
    }
done:
    if (next_emit < n)
        emit_literal(b, next_emit, n); /* :417-426 */
}

/* snappy-c.h snappy_compress; src/compress.rs:99-154 */
int snapf_compress(const char *input, size_t input_len, char *compressed,
                   size_t *compressed_length)
{
    /* the reference's Encoder owns both tables (:67-70); a thread's own */
    static __thread uint16_t big[MAX_TABLE_SIZE];
    static __thread uint16_t small[SMALL_TABLE_SIZE];
    const size_t min = snapf_max_compressed_length(input_len);
    if (min == 0)
        return 1;
    if (*compressed_length < min)
        return 2;
    uint8_t *out = (uint8_t *)compressed;
    const uint8_t *in = (const uint8_t *)input;
    if (input_len == 0) {
        out[0] = 0;
        *compressed_length = 1;
        return 0;
    }
    size_t d = 0;
    for (uint64_t v = input_len;;) { /* src/bytes.rs:61-70 */
        if (v < 0x80) {
            out[d++] = (uint8_t)v;
            break;
        }
        out[d++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    fblock b;
    b.dst = out;
    b.d = d;
    size_t pos = 0;
    while (pos < input_len) {
        size_t n = input_len - pos;
        if (n > MAX_BLOCK_SIZE)
            n = MAX_BLOCK_SIZE;
        b.src = in + pos;
        b.n = n;
        if (n < MIN_NON_LITERAL_BLOCK) {
            emit_literal(&b, 0, n);
        } else {
            /* block_table :491-518 */
            unsigned shift = 32 - 8;
            size_t table_size = 256;
            while (table_size < MAX_TABLE_SIZE && table_size < n) {
                shift--;
                table_size *= 2;
            }
            uint16_t *table = table_size <= SMALL_TABLE_SIZE ? small : big;
            memset(table, 0, table_size * sizeof(uint16_t));
            compress_block(&b, table, shift);
        }
        pos += n;
    }
    *compressed_length = b.d;
    return 0;
}

/* ---- decompress ------------------------------------------------------ */

static const uint32_t WORD_MASK[5] = {0, 0xFF, 0xFFFF, 0xFFFFFF, 0xFFFFFFFF};

/* build.rs:40-67: extra bytes << 11 | copy-1 offset high bits << 8 | len */
static uint16_t TAG[256];
static int tag_ready;
static void tag_init(void)
{
    for (unsigned b = 0; b < 256; b++) {
        unsigned e;
        switch (b & 3) {
        case 0: {
            const unsigned l = (b >> 2) + 1;
            e = l <= 60 ? l : (l - 60) << 11;
            break;
        }
        case 1:
            e = (1u << 11) | (((b >> 5) & 7) << 8) | (4 + ((b >> 2) & 7));
            break;
        case 2:
            e = (2u << 11) | (1 + (b >> 2));
            break;
        default:
            e = (4u << 11) | (1 + (b >> 2));
        }
        TAG[b] = (uint16_t)e;
    }
    __atomic_store_n(&tag_ready, 1, __ATOMIC_RELEASE);
}

int snapf_uncompressed_length(const char *compressed, size_t n,
                              size_t *result)
{
    const uint8_t *p = (const uint8_t *)compressed;
    uint64_t v = 0;
    unsigned shift = 0;
    for (size_t i = 0; i < n && i < 10; i++) { /* src/bytes.rs:73-90 */
        const uint8_t b = p[i];
        if (b < 0x80) {
            if (i == 9 && b > 1)
                return 1;
            v |= (uint64_t)b << shift;
            if (v > MAX_INPUT_SIZE)
                return 1;
            *result = (size_t)v;
            return 0;
        }
        v |= (uint64_t)(b & 0x7F) << shift;
        shift += 7;
    }
    return 1;
}

/* snappy-c.h snappy_uncompress; src/decompress.rs:75-95, :130-343 */
int snapf_uncompress(const char *compressed, size_t compressed_length,
                     char *uncompressed, size_t *uncompressed_length)
{
    if (!__atomic_load_n(&tag_ready, __ATOMIC_ACQUIRE))
        tag_init();
    if (compressed_length == 0)
        return 1;
    const uint8_t *in = (const uint8_t *)compressed;
    /* header */
    uint64_t dlen = 0;
    size_t hdr = 0;
    {
        unsigned shift = 0;
        int ok = 0;
        for (size_t i = 0; i < compressed_length && i < 10; i++) {
            const uint8_t b = in[i];
            if (b < 0x80) {
                if (i == 9 && b > 1)
                    return 1;
                dlen |= (uint64_t)b << shift;
                hdr = i + 1;
                ok = 1;
                break;
            }
            dlen |= (uint64_t)(b & 0x7F) << shift;
            shift += 7;
        }
        if (!ok || dlen > MAX_INPUT_SIZE)
            return 1;
    }
    if (dlen > *uncompressed_length)
        return 2;
    const uint8_t *src = in + hdr;
    const size_t src_len = compressed_length - hdr;
    uint8_t *dst = (uint8_t *)uncompressed;
    const size_t dst_len = (size_t)dlen;
    size_t s = 0, d = 0;

    while (s < src_len) {
        const uint8_t byte = src[s++];
        if ((byte & 3) == 0) {
            /* read_literal :161-228 */
            uint64_t len = (uint64_t)(byte >> 2) + 1;
            if (LIKELY(len <= 16 && s + 16 <= src_len && d + 16 <= dst_len)) {
                cp16(dst + d, src + s);
                d += (size_t)len;
                s += (size_t)len;
                continue;
            }
            if (len >= 61) {
                if ((uint64_t)s + 4 > (uint64_t)src_len)
                    return 1;
                const unsigned nb = (unsigned)len - 60;
                len = (uint64_t)(ld32(src + s) & WORD_MASK[nb]) + 1;
                s += nb;
            }
            if ((uint64_t)(src_len - s) < len || (uint64_t)(dst_len - d) < len)
                return 1;
            memcpy(dst + d, src + s, (size_t)len);
            s += (size_t)len;
            d += (size_t)len;
        } else {
            /* read_copy :233-343 */
            const unsigned e = TAG[byte];
            const unsigned nb = e >> 11;
            size_t trailer;
            if (LIKELY(s + 4 <= src_len)) {
                trailer = ld32(src + s) & WORD_MASK[nb];
            } else if (nb == 1) {
                if (s >= src_len)
                    return 1;
                trailer = src[s];
            } else if (nb == 2) {
                if (s + 1 >= src_len)
                    return 1;
                trailer = (size_t)src[s] | (size_t)src[s + 1] << 8;
            } else {
                return 1;
            }
            const size_t offset = (e & 0x700) | trailer;
            const size_t len = e & 0xFF;
            s += nb;
            if (UNLIKELY(d <= offset - 1)) /* wrapping: offset == 0 too */
                return 1;
            const size_t end = d + len;
            if (LIKELY(offset >= 8 && len <= 16 && d + 16 <= dst_len)) {
                uint8_t *dp = dst + d;
                const uint8_t *sp = dp - offset;
                cp8(dp, sp); /* the second move may read what the first wrote */
                cp8(dp + 8, sp + 8);
            } else if (end + 24 <= dst_len) {
                uint8_t *dp = dst + d;
                const uint8_t *sp = dp - offset;
                for (;;) { /* pre-roll until source and target are 16 apart */
                    const size_t diff = (size_t)(dp - sp);
                    if (diff >= 16)
                        break;
                    memmove(dp, sp, 16);
                    d += diff;
                    dp += diff;
                }
                while (d < end) {
                    cp16(dp, sp);
                    sp += 16;
                    dp += 16;
                    d += 16;
                }
            } else {
                if (end > dst_len)
                    return 1;
                while (d != end) {
                    dst[d] = dst[d - offset];
                    d++;
                }
            }
            d = end;
        }
    }
    if (d != dst_len)
        return 1;
    *uncompressed_length = dst_len;
    return 0;
}
