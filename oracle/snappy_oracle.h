/*
 * snappy_oracle.h -- CPU restatement of rust-snappy's raw block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (rust-snappy_amd/, include/snapmi.h) never links or calls it.
 *
 * Parity pin: see oracle/README.md -- pinned against the reference's golden
 * vector (data/Mark.Twain-Tom.Sawyer.txt.rawsnappy, test/tests.rs:200-205),
 * its decoder KATs (test/tests.rs:232-317), its 19 decoder error KATs
 * (test/tests.rs:345-466) and Google libsnappy 1.1.8 (the library the
 * reference cross-tests against, snappy-cpp/src/lib.rs:66-88).
 */
#ifndef SNAPPY_ORACLE_H
#define SNAPPY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error variants of reference src/error.rs:72-180 (same order). */
enum snapo_kind {
    SNAPO_OK = 0,
    SNAPO_TOO_BIG = 1,            /* a=given            b=max                    */
    SNAPO_BUFFER_TOO_SMALL = 2,   /* a=given            b=min                    */
    SNAPO_EMPTY = 3,
    SNAPO_HEADER = 4,
    SNAPO_HEADER_MISMATCH = 5,    /* a=expected_len     b=got_len                */
    SNAPO_LITERAL = 6,            /* a=len              b=src_len   c=dst_len    */
    SNAPO_COPY_READ = 7,          /* a=len              b=src_len                */
    SNAPO_COPY_WRITE = 8,         /* a=len              b=dst_len                */
    SNAPO_OFFSET = 9,             /* a=offset           b=dst_pos                */
    SNAPO_STREAM_HEADER = 10,     /* a=byte                                      */
    SNAPO_STREAM_HEADER_MISMATCH = 11,
    SNAPO_UNSUPPORTED_CHUNK_TYPE = 12,   /* a=byte                               */
    SNAPO_UNSUPPORTED_CHUNK_LENGTH = 13, /* a=len       b=header(0/1)            */
    SNAPO_CHECKSUM = 14           /* a=expected         b=got                    */
};

typedef struct snapo_error {
    int32_t kind;
    uint32_t _pad;
    uint64_t a, b, c;
} snapo_error;

/* src/compress.rs:42-53 */
size_t snapo_max_compress_len(size_t input_len);

/* src/compress.rs:99-154 (Encoder::compress).  Returns kind (0 = ok). */
int snapo_compress(const uint8_t *input, size_t input_len, uint8_t *output,
                   size_t output_cap, size_t *written, snapo_error *err);

/* src/decompress.rs:30-35 */
int snapo_decompress_len(const uint8_t *input, size_t input_len,
                         size_t *result, snapo_error *err);

/* src/decompress.rs:75-95 (Decoder::decompress).  Returns kind (0 = ok). */
int snapo_decompress(const uint8_t *input, size_t input_len, uint8_t *output,
                     size_t output_cap, size_t *written, snapo_error *err);

/* src/crc32.rs:35-38 + :85-111 (slicing-by-16 semantics, computed bytewise) */
uint32_t snapo_crc32c(const uint8_t *buf, size_t n);
uint32_t snapo_crc32c_masked(const uint8_t *buf, size_t n);

/*
 * Frame layer (src/frame.rs:62-104 + src/write.rs:123-192): what
 * write::FrameEncoder::write_all(buf) followed by into_inner() produces.
 * out_cap must be >= snapo_frame_max_len(n).
 */
size_t snapo_frame_max_len(size_t n);
int snapo_frame_compress(const uint8_t *input, size_t n, uint8_t *out,
                         size_t out_cap, size_t *written, snapo_error *err);
/*
 * read::FrameDecoder::read_to_end (src/read.rs:105-238).  Returns kind;
 * an io::ErrorKind::UnexpectedEof is reported as kind = -1.
 */
int snapo_frame_decompress(const uint8_t *input, size_t n, uint8_t *out,
                           size_t out_cap, size_t *written, snapo_error *err);

/* Per-block statistics used by DESIGN.md / bench (not part of parity). */
typedef struct snapo_stats {
    uint64_t probes;    /* iterations of the probe loop, compress.rs:207-245 */
    uint64_t copies;    /* emit_copy calls, compress.rs:273                   */
    uint64_t literals;  /* emit_literal calls                                 */
    uint64_t elements;  /* tag bytes emitted                                  */
} snapo_stats;
void snapo_stats_reset(void);
void snapo_stats_get(snapo_stats *out);

#ifdef __cplusplus
}
#endif
#endif
