// TEST INFRASTRUCTURE: the text of rust-snappy_amd/csrc/snapmi_tiny.hpp (the
// algorithm one GPU lane runs in k_compress_tiny / k_compress_small)
// instantiated over plain byte arrays, so that tests/test_tiny_lane_cpu.py can
// check it against the oracle without a GPU.  Every access is bounds-checked
// the way the device policies rely on (in32 inside the input, table indices
// inside the table of that length, positions below the length, output inside
// the bound the caller passes).  Never linked into the product library.
#include <stdint.h>
#include <string.h>

#include "snapmi_tiny.hpp"

namespace {
struct HostMem {
    const uint8_t *in;
    uint32_t n;
    uint8_t *out;
    uint32_t out_cap, entries;
    uint16_t table[16384];
    uint32_t bad; // bit per violated assumption
    uint32_t in8(uint32_t k)
    {
        if (k >= n) {
            bad |= 1;
            return 0;
        }
        return in[k];
    }
    uint32_t in32(uint32_t k)
    {
        if (k + 4 > n) {
            bad |= 2;
            return 0;
        }
        uint32_t v;
        memcpy(&v, in + k, 4);
        return v;
    }
    uint32_t tab(uint32_t h)
    {
        if (h >= entries) {
            bad |= 4;
            return 0;
        }
        return table[h];
    }
    void tab_set(uint32_t h, uint32_t v)
    {
        if (h >= entries || v >= n) {
            bad |= 8;
            return;
        }
        table[h] = (uint16_t)v;
    }
    void out8(uint32_t k, uint32_t v)
    {
        if (k >= out_cap || v > 255) {
            bad |= 16;
            return;
        }
        out[k] = (uint8_t)v;
    }
    void out32(uint32_t k, uint32_t v)
    {
        if ((k & 3) || k + 4 > out_cap) {
            bad |= 32;
            return;
        }
        memcpy(out + k, &v, 4);
    }
};
} // namespace

// out: out_cap bytes; returns the stream's length, or 0x80000000 | flags
extern "C" uint32_t tiny_lane_compress(const uint8_t *in, uint32_t n,
                                       uint8_t *out, uint32_t out_cap)
{
    if (n == 0 || n > 65536)
        return 0x80000000u;
    static thread_local HostMem m;
    m.in = in;
    m.n = n;
    m.out = out;
    m.out_cap = out_cap;
    m.entries = snapmi::tiny_table_size(n);
    m.bad = 0;
    memset(m.table, 0, sizeof m.table);
    const uint32_t d = snapmi::tiny_compress(m, n);
    return m.bad ? (0x80000000u | m.bad) : d;
}

// the output bound the device columns are sized by
extern "C" uint32_t tiny_lane_out_max(uint32_t n)
{
    if (n < snapmi::kTinyCompress)
        return n + (snapmi::kTinyOutMax - (snapmi::kTinyCompress - 1));
    if (n < snapmi::kSmallCompress)
        return n + snapmi::kSmallOutSlack;
    return 32 + n + n / 6; // the reference's bound
}
