/*
 * snapmi_test.h -- knobs of the test suite and of the experiment drivers
 * (tests/, tests/hw/).  Not part of the production ABI of snapmi.h: results
 * never depend on them, and a production host has no use for them.
 *
 * These exist in the TEST build of the library only (libsnapmi_test.so:
 * the same sources compiled with -DSNAPMI_TESTING; rust-snappy_amd/csrc/
 * Makefile).  The product library, libsnapmi.so, does not export
 * snapmi_ctx_set_test_option, ignores the environment, and does not contain
 * the cross-check kernels that only these options reach: snapmi_ctx_set_option
 * values "compress_mode" 2 (both compressors on one ticket), "span_kernel" 0
 * (k_compress_blocks / k_compress_block_lds, one copy per step) and
 * "decode_kernel" 2 (k_decompress_streams2 alone) are SNAPMI_E_ARGUMENT there.
 *
 * Environment: a process that sets SNAPMI_TESTING=1 may also steer a new
 * context with SNAPMI_LANE_WAVES, SNAPMI_LANE_SEGMENT_BLOCKS,
 * SNAPMI_LANE_TABLE_SPREAD, SNAPMI_LANE_MIN_BLOCKS, SNAPMI_FRAME_CRC_SIDE,
 * SNAPMI_LANE_UNCACHED, SNAPMI_LANE_DIRECT, SNAPMI_HOST_COPY_KERNEL,
 * SNAPMI_HOST_ENCODE_SLICE, SNAPMI_HOST_DECODE_CHUNKS, SNAPMI_DECODE_KERNEL,
 * SNAPMI_COMPRESS (the experiment scripts under tests/hw/ do).  Without
 * SNAPMI_TESTING they are ignored.
 */
#ifndef SNAPMI_TEST_H
#define SNAPMI_TEST_H

#include "snapmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 *   "lane_waves_per_cu"    lanes in flight = 64 x this x CUs (default 6)
 *   "lane_max_waves"       cap on the lane kernel's wavefronts (0 = none), so
 *                          a small batch puts several blocks on one lane
 *   "lane_tables_uncached" 1: the lane tables come from an uncached
 *                          allocation (measured: no gain; default 0)
 *   "lane_direct_encode"   1 (default): the lane kernel's encoder writes every
 *                          block at its final position; 0: scratch slot per
 *                          block + a compaction pass
 *   "lane_overlap_encode"  0 (default) never; 1: a lane-kernel segment with
 *                          at least 1.4 blocks per lane is matched in two
 *                          halves, the first half's tokens encoded on a side
 *                          stream meanwhile (measured slower); 2: whenever it
 *                          has two blocks
 *   "lane_table_probe"     1: time the placement even with one try
 *   "lane_tables_renew"    1: free the lane tables now; the next large batch
 *                          allocates (and places) new ones
 *   "lane_epoch_preset"    0..65535: every lane's hash-table epoch is set to
 *                          this before the next lane-kernel launch (reaches
 *                          the 16-bit epoch wrap without 65 535 blocks/lane)
 *   "lds_order_ok"         0: behave as if the LDS atomic order self-check of
 *                          snapmi_ctx_create had failed (lane kernel only)
 *   "frame_crc_side_stream"  0: the frame encoder's CRC kernel runs on the
 *                          main stream
 *   "lane_speculate_max_blocks"  lane-kernel launches of at most this many
 *                          blocks run k_match_blocks_spec (default 24 576)
 *   "decode_many_min"      batches of more streams than this are decoded by
 *                          k_decompress_streams3_many, 16 streams of the
 *                          sorted order per workgroup (default 1 048 576; the
 *                          suite lowers it to reach the kernel with 40 000)
 *   "frame_walk_segment"   segment of the parallel chunk-header walk (>= 128
 *                          KiB; default 32 MiB)
 * Returns SNAPMI_E_ARGUMENT for an unknown name.
 */
SNAPMI_API int snapmi_ctx_set_test_option(snapmi_ctx *ctx, const char *name,
                               int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SNAPMI_TEST_H */
