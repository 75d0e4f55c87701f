// snapmi_kernels.hpp -- kernel argument blocks and kernel declarations shared
// between the .hip translation units and the host API.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snapmi.h"

namespace snapmi {

// Batch of raw streams to compress.  All pointers are device memory.
struct CompressArgs {
    const void *const *in_ptrs; // [n] stream i input
    const uint64_t *in_lens;    // [n]
    void *const *out_ptrs;      // [n] stream i output
    const uint64_t *out_caps;   // [n] or nullptr
    uint64_t *out_lens;         // [n]
    snapmi_error *errs;         // [n] or nullptr
    uint32_t *blk_first;        // [n+1] first global block index of stream i
    uint32_t *slot_first;       // [n+1] first scratch slot of stream i
    uint32_t *blk_size;         // [blocks] compressed bytes of each block
    uint64_t *blk_off;          // [blocks+1] exclusive scan of blk_size
    uint8_t *scratch;           // [slots * kSlotBytes]
    uint32_t n_streams;
    uint32_t host_blocks; // launch geometry computed from the host lengths
    uint32_t host_slots;
    uint32_t *ticket; // device-wide block ticket counter, zeroed per launch
    // lane-per-block match finder (k_match_blocks): token stream per block,
    // per-lane epoch-tagged hash tables in HBM
    unsigned long long *tokens; // [(blk_hi - blk_lo) * kMaxTokens]
    uint32_t *ntok;             // [blocks]
    uint32_t blk_lo, blk_hi;    // blocks this lane/encode launch covers
    unsigned long long *lane_tables; // [lanes * kMaxTable * 2] 16-byte entries
    uint32_t *lane_epochs;      // [lanes]
    uint32_t n_lanes;
    // experiment builds (-DSNAPMI_PROFILE) only: 16 u64 cycle counters
    unsigned long long *prof;
};

// wavefronts (= hash tables) per persistent compress workgroup: 5 x 32 KiB
// is all of a CU's LDS
constexpr uint32_t kCompressWaves = 5;
// token slots per block: at most 16385 tokens (every token but the last ends
// in a copy of >= 4 bytes), rounded up to whole 128-byte groups of 16 so a
// lane can write its tokens a full cache line at a time
constexpr uint32_t kMaxTokens = 16400;

// Batch of raw streams to decompress.
struct DecompressArgs {
    const void *const *in_ptrs;
    const uint64_t *in_lens;
    void *const *out_ptrs;    // nullptr for "lengths only"
    const uint64_t *out_caps; // [n] or nullptr when out_ptrs is nullptr
    uint64_t *out_lens;
    snapmi_error *errs; // [n] or nullptr
    // optional [n]: 1 = stored chunk (frame type 0x01): plain copy of the input
    const uint8_t *modes;
    uint32_t n_streams;
    // [n] stream indices, longest compressed stream first (k_plan_decompress)
    uint32_t *order;
    uint32_t *bucket_pos; // [64] scratch of k_plan_decompress
    // experiment builds (-DSNAPMI_PROFILE) only: 16 u64 cycle counters
    unsigned long long *prof;
};

__global__ void k_plan_compress(CompressArgs a);
__global__ void k_compress_blocks(CompressArgs a);
__global__ void k_match_blocks(CompressArgs a);
__global__ void k_encode_tokens(CompressArgs a);
__global__ void k_scan_sizes(CompressArgs a);
__global__ void k_compact(CompressArgs a);

__global__ void k_plan_decompress(DecompressArgs a);
__global__ void k_decompress_streams(DecompressArgs a);
__global__ void k_decompress_len(DecompressArgs a);

} // namespace snapmi
