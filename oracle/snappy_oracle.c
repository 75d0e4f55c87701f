/*
 * snappy_oracle.c -- CPU restatement (plain C) of rust-snappy 1.1.1's raw
 * block codec, CRC32C and frame layer.
 *
 * TEST INFRASTRUCTURE ONLY (see snappy_oracle.h).  Written from the behaviour
 * of the reference, not copied from it; every function cites the reference
 * lines it restates.  Paths are relative to /root/reference.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_getaffinity, pthread_setaffinity_np (bench driver) */
#endif
#include "snappy_oracle.h"

#include <string.h>

#define MAX_INPUT_SIZE 0xFFFFFFFFull /* src/lib.rs:93  */
#define MAX_BLOCK_SIZE 65536u        /* src/lib.rs:97  */
#define MAX_TABLE_SIZE 16384u        /* src/compress.rs:11 */
#define INPUT_MARGIN 15u             /* src/compress.rs:20 */
#define MIN_NON_LITERAL_BLOCK 17u    /* src/compress.rs:24 */
#define MAX_COMPRESS_BLOCK_SIZE 76490u /* src/frame.rs:12 */

static __thread snapo_stats g_stats;

void snapo_stats_reset(void) { memset(&g_stats, 0, sizeof g_stats); }
void snapo_stats_get(snapo_stats *out) { *out = g_stats; }

static int fail(snapo_error *e, int kind, uint64_t a, uint64_t b, uint64_t c)
{
    if (e) {
        e->kind = kind;
        e->_pad = 0;
        e->a = a;
        e->b = b;
        e->c = c;
    }
    return kind;
}

/* ---- little-endian helpers: src/bytes.rs:95-118 --------------------- */
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
static inline uint32_t le32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4); /* one unaligned load, as loadu_u32_le */
    return v;
}
static inline uint64_t le64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
#else
static inline uint32_t le32(const uint8_t *p)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 |
           (uint32_t)p[3] << 24;
}
static inline uint64_t le64(const uint8_t *p)
{
    return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32;
}
#endif

/* src/bytes.rs:61-70 */
static size_t put_varint(uint8_t *dst, uint64_t n)
{
    size_t i = 0;
    while (n >= 0x80) {
        dst[i++] = (uint8_t)n | 0x80;
        n >>= 7;
    }
    dst[i++] = (uint8_t)n;
    return i;
}

/*
 * src/bytes.rs:73-90.  Returns header length (0 = invalid).  checked_shl in
 * the reference only rejects shift amounts >= 64; high bits shifted out of a
 * 10th byte are silently dropped, and that is kept here.
 */
static size_t get_varint(const uint8_t *p, size_t n, uint64_t *value)
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

/* ---- compression ----------------------------------------------------- */

/* src/compress.rs:42-53 */
size_t snapo_max_compress_len(size_t input_len)
{
    uint64_t n = input_len;
    if (n > MAX_INPUT_SIZE)
        return 0;
    uint64_t m = 32 + n + n / 6;
    return m > MAX_INPUT_SIZE ? 0 : (size_t)m;
}

/* src/compress.rs:433-474; the 16-byte over-copy at :440-453 only touches
 * bytes past the returned length, so a plain memcpy is byte-equivalent. */
static size_t put_literal(uint8_t *dst, size_t d, const uint8_t *lit,
                          size_t len)
{
    size_t n = len - 1;
    g_stats.literals++;
    g_stats.elements++;
    if (n <= 59) {
        dst[d++] = (uint8_t)(n << 2);
    } else if (n < 256) {
        dst[d++] = 60 << 2;
        dst[d++] = (uint8_t)n;
    } else {
        dst[d++] = 61 << 2;
        dst[d++] = (uint8_t)n;
        dst[d++] = (uint8_t)(n >> 8);
    }
    memcpy(dst + d, lit, len);
    return d + len;
}

/* src/compress.rs:363-369 */
static size_t put_copy2(uint8_t *dst, size_t d, size_t offset, size_t len)
{
    g_stats.elements++;
    dst[d] = (uint8_t)(((len - 1) << 2) | 2);
    dst[d + 1] = (uint8_t)offset;
    dst[d + 2] = (uint8_t)(offset >> 8);
    return d + 3;
}

/* src/compress.rs:323-357 */
static size_t put_copy(uint8_t *dst, size_t d, size_t offset, size_t len)
{
    g_stats.copies++;
    while (len >= 68) {
        d = put_copy2(dst, d, offset, 64);
        len -= 64;
    }
    if (len > 64) {
        d = put_copy2(dst, d, offset, 60);
        len -= 60;
    }
    if (len <= 11 && offset <= 2047) {
        g_stats.elements++;
        dst[d] = (uint8_t)(((offset >> 8) << 5) | ((len - 4) << 2) | 1);
        dst[d + 1] = (uint8_t)offset;
        return d + 2;
    }
    return put_copy2(dst, d, offset, len);
}

/* src/compress.rs:523-525 */
static inline uint32_t hash32(uint32_t x, unsigned shift)
{
    return (x * 0x1E35A7BDu) >> shift;
}

/*
 * One block of at least 17 bytes: src/compress.rs:195-317 (match finder),
 * :378-412 (extend_match), :417-426 (done), :491-518 (table sizing).
 */
static size_t compress_block(const uint8_t *src, size_t n, uint8_t *dst,
                             size_t d, uint16_t *table)
{
    unsigned shift = 32 - 8;
    size_t table_size = 256;
    while (table_size < MAX_TABLE_SIZE && table_size < n) {
        shift--;
        table_size *= 2;
    }
    memset(table, 0, table_size * sizeof(uint16_t));

    size_t s = 1, next_emit = 0;
    const size_t s_limit = n - INPUT_MARGIN;
    uint32_t next_hash = hash32(le32(src + s), shift);

    for (;;) {
        /* probe loop with the skip heuristic, :204-245 */
        uint32_t skip = 32;
        size_t s_next = s, cand;
        for (;;) {
            s = s_next;
            uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            if (s_next > s_limit)
                goto done;
            g_stats.probes++;
            cand = table[next_hash];
            table[next_hash] = (uint16_t)s;
            next_hash = hash32(le32(src + s_next), shift);
            if (le32(src + s) == le32(src + cand))
                break;
        }
        d = put_literal(dst, d, src + next_emit, s - next_emit);
        /* copy chain, :258-315 */
        for (;;) {
            size_t base = s;
            size_t c = cand + 4;
            s += 4;
            /* extend_match, :378-412: 8 bytes at a time, then bytewise */
            while (s + 8 <= n) {
                uint64_t z = le64(src + s) ^ le64(src + c);
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
            d = put_copy(dst, d, base - cand, s - base);
            next_emit = s;
            if (s >= s_limit)
                goto done;
            uint64_t x = le64(src + s - 1);
            table[hash32((uint32_t)x, shift)] = (uint16_t)(s - 1);
            uint32_t h = hash32((uint32_t)(x >> 8), shift);
            cand = table[h];
            table[h] = (uint16_t)s;
            if ((uint32_t)(x >> 8) != le32(src + cand)) {
                next_hash = hash32((uint32_t)(x >> 16), shift);
                s++;
                break;
            }
        }
    }
done:
    if (next_emit < n)
        d = put_literal(dst, d, src + next_emit, n - next_emit);
    return d;
}

/* src/compress.rs:99-154 */
int snapo_compress(const uint8_t *input, size_t input_len, uint8_t *output,
                   size_t output_cap, size_t *written, snapo_error *err)
{
    uint16_t table[MAX_TABLE_SIZE];
    size_t min = snapo_max_compress_len(input_len);
    if (min == 0)
        return fail(err, SNAPO_TOO_BIG, input_len, MAX_INPUT_SIZE, 0);
    if (output_cap < min)
        return fail(err, SNAPO_BUFFER_TOO_SMALL, output_cap, min, 0);
    if (input_len == 0) {
        output[0] = 0;
        *written = 1;
        return fail(err, SNAPO_OK, 0, 0, 0);
    }
    size_t d = put_varint(output, input_len);
    size_t pos = 0;
    while (pos < input_len) {
        size_t n = input_len - pos;
        if (n > MAX_BLOCK_SIZE)
            n = MAX_BLOCK_SIZE;
        if (n < MIN_NON_LITERAL_BLOCK)
            d = put_literal(output, d, input + pos, n);
        else
            d = compress_block(input + pos, n, output, d, table);
        pos += n;
    }
    *written = d;
    return fail(err, SNAPO_OK, 0, 0, 0);
}

/* ---- decompression --------------------------------------------------- */

/* src/decompress.rs:362-374 */
static int read_header(const uint8_t *in, size_t n, size_t *hdr_len,
                       uint64_t *dlen, snapo_error *err)
{
    uint64_t v = 0;
    size_t h = get_varint(in, n, &v);
    if (h == 0)
        return fail(err, SNAPO_HEADER, 0, 0, 0);
    if (v > MAX_INPUT_SIZE)
        return fail(err, SNAPO_TOO_BIG, v, MAX_INPUT_SIZE, 0);
    *hdr_len = h;
    *dlen = v;
    return SNAPO_OK;
}

/* src/decompress.rs:30-35 */
int snapo_decompress_len(const uint8_t *input, size_t input_len,
                         size_t *result, snapo_error *err)
{
    if (input_len == 0) {
        *result = 0;
        return fail(err, SNAPO_OK, 0, 0, 0);
    }
    size_t h;
    uint64_t v;
    int k = read_header(input, input_len, &h, &v, err);
    if (k)
        return k;
    *result = (size_t)v;
    return fail(err, SNAPO_OK, 0, 0, 0);
}

static const uint32_t WORD_MASK[5] = {0, 0xFF, 0xFFFF, 0xFFFFFF, 0xFFFFFFFF};

/* src/decompress.rs:75-95 driver, :130-148 dispatch loop, :161-228 literals,
 * :233-343 copies, :433-474 offset read.  The 16-byte over-copy fast paths
 * only change bytes that are rewritten later, so they are not restated; the
 * order of the checks (and therefore which error wins) is. */
int snapo_decompress(const uint8_t *input, size_t input_len, uint8_t *output,
                     size_t output_cap, size_t *written, snapo_error *err)
{
    if (input_len == 0)
        return fail(err, SNAPO_EMPTY, 0, 0, 0);
    size_t hdr;
    uint64_t dlen64;
    int k = read_header(input, input_len, &hdr, &dlen64, err);
    if (k)
        return k;
    if (dlen64 > output_cap)
        return fail(err, SNAPO_BUFFER_TOO_SMALL, output_cap, dlen64, 0);

    const uint8_t *src = input + hdr;
    const uint64_t src_len = input_len - hdr;
    const uint64_t dst_len = dlen64;
    uint8_t *dst = output;
    uint64_t s = 0, d = 0;

    while (s < src_len) {
        uint8_t tag = src[s++];
        if ((tag & 3) == 0) {
            uint64_t len = (uint64_t)(tag >> 2) + 1;
            if (len >= 61) {
                /* :189-205: needs 4 readable bytes whatever the byte count */
                if (s + 4 > src_len)
                    return fail(err, SNAPO_LITERAL, 4, src_len - s,
                                dst_len - d);
                unsigned nb = (unsigned)(len - 60);
                len = (uint64_t)(le32(src + s) & WORD_MASK[nb]) + 1;
                s += nb;
            }
            if (src_len - s < len || dst_len - d < len)
                return fail(err, SNAPO_LITERAL, len, src_len - s, dst_len - d);
            memcpy(dst + d, src + s, (size_t)len);
            s += len;
            d += len;
        } else {
            unsigned kind = tag & 3;
            unsigned nb = kind == 1 ? 1 : (kind == 2 ? 2 : 4);
            uint64_t len = kind == 1 ? 4 + ((tag >> 2) & 7) : 1 + (tag >> 2);
            uint64_t hi = kind == 1 ? (uint64_t)(tag >> 5) << 8 : 0;
            uint64_t trailer;
            if (s + 4 <= src_len) {
                trailer = le32(src + s) & WORD_MASK[nb];
            } else if (nb == 1) {
                if (s >= src_len)
                    return fail(err, SNAPO_COPY_READ, 1, src_len - s, 0);
                trailer = src[s];
            } else if (nb == 2) {
                if (s + 1 >= src_len)
                    return fail(err, SNAPO_COPY_READ, 2, src_len - s, 0);
                trailer = (uint64_t)src[s] | (uint64_t)src[s + 1] << 8;
            } else {
                return fail(err, SNAPO_COPY_READ, 4, src_len - s, 0);
            }
            uint64_t offset = hi | trailer;
            s += nb;
            if (d <= offset - 1) /* wrapping: also catches offset == 0 */
                return fail(err, SNAPO_OFFSET, offset, d, 0);
            uint64_t end = d + len;
            if (end > dst_len)
                return fail(err, SNAPO_COPY_WRITE, len, dst_len - d, 0);
            if (offset >= len) { /* disjoint: one block move */
                memcpy(dst + d, dst + d - offset, (size_t)len);
                d = end;
            } else {
                for (; d != end; d++) /* overlapping: replicate the pattern */
                    dst[d] = dst[d - offset];
            }
        }
    }
    if (d != dst_len)
        return fail(err, SNAPO_HEADER_MISMATCH, dst_len, d, 0);
    *written = (size_t)dst_len;
    return fail(err, SNAPO_OK, 0, 0, 0);
}

/* ---- CRC32C ---------------------------------------------------------- */

/* build.rs:6,110-124 (table) and src/crc32.rs:85-111: slicing-by-16 and the
 * SSE4.2 instruction both compute the plain reflected CRC-32C, restated
 * here one byte at a time. */
static uint32_t crc_table[256];
static int crc_ready;

static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        crc_table[i] = c;
    }
    crc_ready = 1;
}

uint32_t snapo_crc32c(const uint8_t *buf, size_t n)
{
    if (!crc_ready)
        crc_init();
    uint32_t c = ~0u;
    for (size_t i = 0; i < n; i++)
        c = crc_table[(uint8_t)c ^ buf[i]] ^ (c >> 8);
    return ~c;
}

/* src/crc32.rs:35-38 */
uint32_t snapo_crc32c_masked(const uint8_t *buf, size_t n)
{
    uint32_t c = snapo_crc32c(buf, n);
    return ((c >> 15) | (c << 17)) + 0xA282EAD8u;
}

/* ---- frame layer ----------------------------------------------------- */

static const uint8_t STREAM_IDENT[10] = {0xFF, 0x06, 0x00, 0x00, 's',
                                         'N',  'a',  'P',  'p',  'Y'};

size_t snapo_frame_max_len(size_t n)
{
    size_t chunks = (n + MAX_BLOCK_SIZE - 1) / MAX_BLOCK_SIZE;
    return 10 + chunks * (8 + MAX_COMPRESS_BLOCK_SIZE);
}

/* src/write.rs:123-192 (chunking of write_all + flush) and
 * src/frame.rs:62-104 (compress_frame). */
int snapo_frame_compress(const uint8_t *input, size_t n, uint8_t *out,
                         size_t out_cap, size_t *written, snapo_error *err)
{
    static __thread uint8_t tmp[MAX_COMPRESS_BLOCK_SIZE];
    size_t o = 0;
    if (out_cap < snapo_frame_max_len(n))
        return fail(err, SNAPO_BUFFER_TOO_SMALL, out_cap,
                    snapo_frame_max_len(n), 0);
    if (n == 0) { /* stream identifier is written lazily: write.rs:154-170 */
        *written = 0;
        return fail(err, SNAPO_OK, 0, 0, 0);
    }
    memcpy(out, STREAM_IDENT, 10);
    o = 10;
    for (size_t pos = 0; pos < n;) {
        size_t len = n - pos;
        if (len > MAX_BLOCK_SIZE)
            len = MAX_BLOCK_SIZE;
        const uint8_t *src = input + pos;
        uint32_t sum = snapo_crc32c_masked(src, len);
        size_t clen = 0;
        int k = snapo_compress(src, len, tmp, sizeof tmp, &clen, err);
        if (k)
            return k;
        int raw = clen >= len - len / 8; /* frame.rs:85 */
        size_t payload = raw ? len : clen;
        size_t chunk_len = 4 + payload;
        out[o + 0] = raw ? 0x01 : 0x00;
        out[o + 1] = (uint8_t)chunk_len;
        out[o + 2] = (uint8_t)(chunk_len >> 8);
        out[o + 3] = (uint8_t)(chunk_len >> 16);
        out[o + 4] = (uint8_t)sum;
        out[o + 5] = (uint8_t)(sum >> 8);
        out[o + 6] = (uint8_t)(sum >> 16);
        out[o + 7] = (uint8_t)(sum >> 24);
        memcpy(out + o + 8, raw ? src : tmp, payload);
        o += 8 + payload;
        pos += len;
    }
    *written = o;
    return fail(err, SNAPO_OK, 0, 0, 0);
}

#define SNAPO_UNEXPECTED_EOF (-1)

/* src/read.rs:105-238 driven by read_to_end over an in-memory reader.  The
 * decoder's `src` scratch is modelled explicitly because decompress_len is
 * called on the whole scratch buffer (read.rs:216), stale bytes included. */
int snapo_frame_decompress(const uint8_t *input, size_t n, uint8_t *out,
                           size_t out_cap, size_t *written, snapo_error *err)
{
    static __thread uint8_t src[MAX_COMPRESS_BLOCK_SIZE];
    static __thread uint8_t dst[MAX_BLOCK_SIZE];
    memset(src, 0, sizeof src);
    size_t r = 0, o = 0;
    int seen_ident = 0;
#define NEED(k)                                                               \
    do {                                                                      \
        if (n - r < (size_t)(k))                                              \
            return fail(err, SNAPO_UNEXPECTED_EOF, 0, 0, 0);                  \
    } while (0)
    for (;;) {
        if (r == n)
            break; /* read_exact_eof: clean EOF, read.rs:119-121 */
        NEED(4);
        memcpy(src, input + r, 4);
        r += 4;
        uint8_t ty = src[0];
        if (!seen_ident) {
            if (ty != 0xFF)
                return fail(err, SNAPO_STREAM_HEADER, ty, 0, 0);
            seen_ident = 1;
        }
        uint64_t len = (uint64_t)src[1] | (uint64_t)src[2] << 8 |
                       (uint64_t)src[3] << 16;
        if (len > sizeof src)
            return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, len, 0, 0);
        if (ty >= 0x02 && ty <= 0x7F)
            return fail(err, SNAPO_UNSUPPORTED_CHUNK_TYPE, ty, 0, 0);
        if ((ty >= 0x80 && ty <= 0xFD) || ty == 0xFE) {
            NEED(len);
            memcpy(src, input + r, len);
            r += len;
        } else if (ty == 0xFF) {
            if (len != 6)
                return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, len, 1, 0);
            NEED(len);
            memcpy(src, input + r, len);
            r += len;
            if (memcmp(src, STREAM_IDENT + 4, 6) != 0)
                return fail(err, SNAPO_STREAM_HEADER_MISMATCH, 0, 0, 0);
        } else if (ty == 0x01) {
            if (len < 4)
                return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, len, 0, 0);
            NEED(4);
            uint32_t expected = le32(input + r);
            r += 4;
            size_t m = len - 4;
            if (m > sizeof dst)
                return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, m, 0, 0);
            NEED(m);
            memcpy(dst, input + r, m);
            r += m;
            uint32_t got = snapo_crc32c_masked(dst, m);
            if (expected != got)
                return fail(err, SNAPO_CHECKSUM, expected, got, 0);
            if (out_cap - o < m)
                return fail(err, SNAPO_BUFFER_TOO_SMALL, out_cap, o + m, 0);
            memcpy(out + o, dst, m);
            o += m;
        } else { /* 0x00 compressed */
            if (len < 4)
                return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, len, 0, 0);
            NEED(4);
            uint32_t expected = le32(input + r);
            r += 4;
            size_t sn = len - 4;
            NEED(sn);
            memcpy(src, input + r, sn);
            r += sn;
            size_t dn = 0;
            int k = snapo_decompress_len(src, sizeof src, &dn, err);
            if (k)
                return k;
            if (dn > sizeof dst)
                return fail(err, SNAPO_UNSUPPORTED_CHUNK_LENGTH, dn, 0, 0);
            size_t got_len = 0;
            k = snapo_decompress(src, sn, dst, dn, &got_len, err);
            if (k)
                return k;
            uint32_t got = snapo_crc32c_masked(dst, dn);
            if (expected != got)
                return fail(err, SNAPO_CHECKSUM, expected, got, 0);
            if (out_cap - o < dn)
                return fail(err, SNAPO_BUFFER_TOO_SMALL, out_cap, o + dn, 0);
            memcpy(out + o, dst, dn);
            o += dn;
        }
    }
#undef NEED
    *written = o;
    return fail(err, SNAPO_OK, 0, 0, 0);
}

/* ---- multi-threaded timing driver for bench.py's cpu_baseline leg -------
 * Runs a codec on `threads` pthreads for about `seconds`: every thread loops
 * over the same n streams (compress, or decompress of their compressed form)
 * into private buffers allocated before the clock starts.  Returns
 * uncompressed bytes per second summed over the threads.  The codec is the
 * restatement above, or - snapo_bench_ext - any pair of functions with the
 * snappy-c.h signatures (bench.py passes libsnappy 1.1.8's, the library the
 * reference's own bench compares against, bench/src/bench.rs:117-153).
 * Threads are pinned to the CPUs this process may run on, one each, round
 * robin.  Test/bench infrastructure only. */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>

typedef int (*snappy_c_fn)(const char *, size_t, char *, size_t *);

typedef struct bench_job {
    const uint8_t *const *datas;
    const size_t *lens;
    const uint8_t *const *comps;
    const size_t *clens;
    int n;
    int direction; /* 0 = compress, 1 = decompress */
    double seconds;
    uint64_t rounds; /* out */
    size_t maxlen;
    snappy_c_fn ext_compress, ext_uncompress; /* NULL: the restatement */
    int cpu; /* CPU to pin to, -1 = none */
    pthread_barrier_t *start;
} bench_job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *bench_worker(void *arg)
{
    bench_job *j = (bench_job *)arg;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    size_t cap = snapo_max_compress_len(j->maxlen);
    uint8_t *out = (uint8_t *)malloc(cap ? cap : 64);
    memset(out, 0, cap ? cap : 64); /* pages touched before the clock */
    snapo_error e;
    size_t w;
    pthread_barrier_wait(j->start);
    const double t_end = now_s() + j->seconds;
    uint64_t rounds = 0;
    do {
        for (int i = 0; i < j->n; i++) {
            if (j->ext_compress) {
                w = cap;
                if (j->direction == 0)
                    j->ext_compress((const char *)j->datas[i], j->lens[i],
                                    (char *)out, &w);
                else
                    j->ext_uncompress((const char *)j->comps[i], j->clens[i],
                                      (char *)out, &w);
            } else if (j->direction == 0) {
                snapo_compress(j->datas[i], j->lens[i], out, cap, &w, &e);
            } else {
                snapo_decompress(j->comps[i], j->clens[i], out, cap, &w, &e);
            }
        }
        rounds++;
    } while (now_s() < t_end);
    j->rounds = rounds;
    free(out);
    return NULL;
}

double snapo_bench_ext(const uint8_t *const *datas, const size_t *lens,
                       const uint8_t *const *comps, const size_t *clens, int n,
                       int direction, int threads, double seconds,
                       uint64_t *total_rounds, void *ext_compress,
                       void *ext_uncompress, int pin)
{
    bench_job *jobs = (bench_job *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    size_t maxlen = 0, ubytes = 0;
    for (int i = 0; i < n; i++) {
        if (lens[i] > maxlen)
            maxlen = lens[i];
        ubytes += lens[i];
    }
    /* CPUs this process may use, in order */
    cpu_set_t allowed;
    int cpus[CPU_SETSIZE], ncpu = 0;
    if (pin && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed))
                cpus[ncpu++] = c;
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)threads + 1);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (bench_job){datas, lens, comps, clens, n, direction,
                              seconds, 0, maxlen,
                              (snappy_c_fn)ext_compress,
                              (snappy_c_fn)ext_uncompress,
                              ncpu ? cpus[t % ncpu] : -1, &start};
        pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    pthread_barrier_wait(&start); /* buffers are allocated and touched */
    const double t0 = now_s();
    uint64_t rounds = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        rounds += jobs[t].rounds;
    }
    const double dt = now_s() - t0;
    pthread_barrier_destroy(&start);
    if (total_rounds)
        *total_rounds = rounds;
    free(jobs);
    free(th);
    return (double)rounds * (double)ubytes / dt;
}

double snapo_bench(const uint8_t *const *datas, const size_t *lens,
                   const uint8_t *const *comps, const size_t *clens, int n,
                   int direction, int threads, double seconds,
                   uint64_t *total_rounds)
{
    return snapo_bench_ext(datas, lens, comps, clens, n, direction, threads,
                           seconds, total_rounds, NULL, NULL, 1);
}
