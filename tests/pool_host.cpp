// The host arithmetic of the token path (rust-snappy_amd/csrc/snapmi_pool.hpp)
// behind a C ABI for tests/test_pool_cpu.py.
#include "../rust-snappy_amd/csrc/snapmi_pool.hpp"

extern "C" {
uint64_t t_segment_blocks(uint64_t blocks, uint64_t max_blocks)
{
    return snapmi::segment_blocks(blocks, max_blocks);
}
uint64_t t_pool_worst_pages(uint64_t bytes, uint64_t blocks, uint64_t seg)
{
    return snapmi::pool_worst_pages(bytes, blocks, seg);
}
uint64_t t_pool_pages(uint64_t bytes, uint64_t blocks, uint64_t seg,
                      uint32_t lanes, uint32_t pct, uint64_t floor_pages)
{
    return snapmi::pool_pages(bytes, blocks, seg, lanes, pct, floor_pages);
}
uint32_t t_pool_grow(uint32_t now, uint64_t spilled, uint64_t of)
{
    return snapmi::pool_grow(now, spilled, of);
}
}
