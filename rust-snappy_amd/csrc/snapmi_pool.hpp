// snapmi - the host's arithmetic of the token path: how a batch's blocks are
// cut into launches and how many pages its token pool gets.  Plain functions
// of numbers (no HIP), so that tests/test_pool_cpu.py can check them on the
// CPU; snapmi_api.hip is their only user in the library.
#pragma once
#include <cstdint>

namespace snapmi {

// tokens per page, exceptions per page, and the most pages a block can fill
// (snapmi_kernels.hpp has the same numbers for the device; a static_assert in
// snapmi_api.hip ties them)
constexpr uint32_t kPoolTokPage = 512, kPoolExcPage = 256;
constexpr uint32_t kPoolPagesPerBlock = 33 + 4;

// The lane kernel runs over segments of the block list so that the token
// scratch stays bounded: at most `max_blocks` blocks a launch, and the
// launches of a batch equal in size - a short last launch has one block per
// lane or fewer and ends when its heaviest block does, as a full one would.
inline uint64_t segment_blocks(uint64_t blocks, uint64_t max_blocks)
{
    if (blocks <= max_blocks)
        return blocks;
    const uint64_t launches = (blocks + max_blocks - 1) / max_blocks;
    return (blocks + launches - 1) / launches;
}

// The worst case of one launch in pages, from the bytes of the batch's
// blocks (0: the caller does not say - full blocks): a block of n bytes has
// at most n / 4 + 1 tokens and n / 65 exceptions - n / 2 048 + n / 16 640 + 2
// pages - scaled to the launch's share of the blocks.
inline uint64_t pool_worst_pages(uint64_t block_bytes, uint64_t blocks,
                                 uint64_t seg_blocks)
{
    const uint64_t all = block_bytes ? block_bytes : blocks * 65536ull;
    const uint64_t seg =
        blocks ? (uint64_t)((double)all * seg_blocks / blocks) + 1 : 0;
    return seg / (kPoolTokPage * 4) + seg / (65 * kPoolExcPage) +
           2 * seg_blocks;
}

// Pages of a launch's pool: pct per cent of the worst case and what the
// launch keeps in hand on top (a page per lane, a run of 32 per lane
// wavefront); never under `floor_pages`; 100 per cent means "no block can
// spill": the bound of every block, whatever the launches' shares of the
// bytes.
inline uint64_t pool_pages(uint64_t block_bytes, uint64_t blocks,
                           uint64_t seg_blocks, uint32_t lanes, uint32_t pct,
                           uint64_t floor_pages)
{
    const uint64_t worst = pool_worst_pages(block_bytes, blocks, seg_blocks);
    const uint64_t in_hand = (uint64_t)lanes + lanes / 2;
    uint64_t pages = (worst * pct + 99) / 100 + in_hand;
    if (pages < floor_pages)
        pages = floor_pages;
    if (pages > worst + in_hand)
        pages = worst + in_hand;
    if (pct >= 100)
        pages = seg_blocks * kPoolPagesPerBlock + in_hand;
    return pages;
}

// What the pool's percentage becomes behind a launch of `of` blocks of which
// `spilled` found no page: unchanged up to a hundredth, a sixth more up to a
// tenth (the pool is nearly there), half as much again beyond.
inline uint32_t pool_grow(uint32_t now, uint64_t spilled, uint64_t of)
{
    if (!of || spilled * 100 <= of || now >= 100)
        return now;
    const uint32_t next =
        spilled * 10 > of ? now * 3 / 2 + 1 : now * 7 / 6 + 1;
    return next > 100 ? 100 : next;
}

} // namespace snapmi
