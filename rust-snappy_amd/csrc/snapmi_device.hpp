// snapmi_device.hpp -- shared device-side helpers for the gfx950 Snappy codec.
//
// Everything here is written for CDNA4 wave64: one wavefront executes the
// sequential Snappy semantics of one block / one stream with its state in
// SGPRs (wave-uniform values), and uses the 64 lanes for the byte work
// (match extension, literal copies, back-reference copies).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "snapmi.h"

namespace snapmi {

constexpr uint32_t kWave = 64;
constexpr uint32_t kMaxBlock = 1u << 16;      // reference src/lib.rs:97
constexpr uint32_t kMaxTable = 1u << 14;      // reference src/compress.rs:11
constexpr uint32_t kInputMargin = 15;         // reference src/compress.rs:20
constexpr uint32_t kMinNonLiteral = 17;       // reference src/compress.rs:24
constexpr uint64_t kMaxInput = 0xFFFFFFFFull; // reference src/lib.rs:93
// max_compress_len(65536) = 76490 (reference src/frame.rs:12), rounded up to
// a multiple of 16 so scratch slots stay 16-byte aligned.
constexpr uint32_t kSlotBytes = 76496;

// Explicit address spaces.  Pointers fetched from the descriptor arrays are
// generic to the compiler; without these it emits flat_load/flat_store, which
// count against both vmcnt and lgkmcnt and so serialise LDS and HBM waits.
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef g_u8 *gptr;        // global (HBM) bytes
typedef const g_u8 *gcptr; // global (HBM) bytes, read only
typedef __attribute__((address_space(3))) uint16_t l_u16;
typedef l_u16 *lptr16; // LDS u16
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t uni(uint32_t v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

// v_readlane_b32 with a wave-uniform lane index.  The builtin returns int:
// always go through this wrapper so the value is never sign-extended.
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// gfx950 global memory takes unaligned dword accesses; the compiler emits a
// single global_load_dword / global_store_dword for these.
__device__ __forceinline__ uint32_t ld32u(gcptr p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ void st32u(gptr p, uint32_t v)
{
    __builtin_memcpy(p, &v, 4);
}

// 16 bytes from global memory at any alignment (one global_load_dwordx4)
__device__ __forceinline__ u32x4 ld128g(gcptr p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// The whole wave copies len bytes (a long literal), touching exactly
// from[0..len) and to[0..len): the destination is brought to a 16-byte
// boundary, then 16 bytes per lane, 1 KiB per instruction - and with kRows4
// four such rows in flight per trip, for a kernel that has the registers (the
// decoders run 8 waves per SIMD and get their loads in flight from that) -
// the rest bytewise.  Long literals are the whole of an incompressible
// stream.  The source may have any alignment.
template <bool kRows4>
__device__ __forceinline__ void wave_copy(gptr to, gcptr from, uint64_t len,
                                          uint32_t lane)
{
    uint32_t head = (uint32_t)((16 - ((uintptr_t)to & 15)) & 15);
    if (head > len)
        head = (uint32_t)len;
    if (lane < head)
        to[lane] = from[lane];
    const uint64_t body = (len - head) & ~15ull;
    gptr t = to + head;
    gcptr f = from + head;
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    uint64_t i = 16 * lane;
    for (; kRows4 && i + 3072 < body; i += 4096) {
        const u32x4 v0 = ld128g(f + i), v1 = ld128g(f + i + 1024),
                    v2 = ld128g(f + i + 2048), v3 = ld128g(f + i + 3072);
        *(g_u32x4 *)(t + i) = v0;
        *(g_u32x4 *)(t + i + 1024) = v1;
        *(g_u32x4 *)(t + i + 2048) = v2;
        *(g_u32x4 *)(t + i + 3072) = v3;
    }
    for (; i < body; i += 1024) {
        *(g_u32x4 *)(t + i) = ld128g(f + i);
    }
    const uint32_t tail = (uint32_t)(len - head - body);
    if (lane < tail)
        t[body + lane] = f[body + lane];
}

// Dword at base[pos..pos+4) where only base[0..avail) may be touched; bytes
// past `avail` read as zero.
__device__ __forceinline__ uint32_t ld32g(gcptr base, uint64_t pos,
                                          uint64_t avail)
{
    if (pos + 4 <= avail)
        return ld32u(base + pos);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 3; k++)
        if (pos + k < avail)
            v |= (uint32_t)base[pos + k] << (8 * k);
    return v;
}

// max_compress_len, reference src/compress.rs:42-53.
__host__ __device__ __forceinline__ uint64_t max_compress_len_u64(uint64_t n)
{
    if (n > kMaxInput)
        return 0;
    uint64_t m = 32 + n + n / 6;
    return m > kMaxInput ? 0 : m;
}

__host__ __device__ __forceinline__ uint32_t varint_len(uint64_t n)
{
    uint32_t k = 1;
    while (n >= 0x80) {
        n >>= 7;
        k++;
    }
    return k;
}

__device__ __forceinline__ void set_error(snapmi_error *errs, uint64_t i,
                                          int kind, uint64_t a, uint64_t b,
                                          uint64_t c)
{
    if (errs) {
        errs[i].kind = kind;
        errs[i].reserved = 0;
        errs[i].a = a;
        errs[i].b = b;
        errs[i].c = c;
    }
}

// A 256-byte window of a byte stream held in one VGPR: lane i owns the dword
// at stream offset base + 4*i.  A wave-uniform position inside the window is
// read with two v_readlane + one 64-bit scalar shift, i.e. without a memory
// round trip.  Refills are one coalesced 256-byte global load.
struct ByteWindow {
    gcptr src;          // stream start
    uint64_t avail;     // readable bytes from src
    uint64_t base;      // stream offset of lane 0's dword
    uint32_t v;         // this lane's dword

    __device__ __forceinline__ void init(gcptr s, uint64_t a)
    {
        src = s;
        avail = a;
        base = ~0ull;
        v = 0;
    }
    __device__ __forceinline__ void refill(uint64_t pos)
    {
        base = pos;
        v = ld32g(src, pos + 4 * (threadIdx.x & 63), avail);
    }
    // dwords idx, idx+1 (idx uniform)
    __device__ __forceinline__ uint64_t pair(uint32_t idx) const
    {
        uint32_t lo = rdlane(v, idx);
        uint32_t hi = rdlane(v, idx + 1);
        return ((uint64_t)hi << 32) | lo;
    }
    // u32 at uniform stream offset p (bytes past avail read as zero)
    __device__ __forceinline__ uint32_t get32(uint64_t p)
    {
        if (p < base || p - base > 248)
            refill(p);
        uint32_t o = (uint32_t)(p - base);
        return (uint32_t)(pair(o >> 2) >> (8 * (o & 3)));
    }
    // u64 at uniform stream offset p
    __device__ __forceinline__ uint64_t get64(uint64_t p)
    {
        if (p < base || p - base > 240)
            refill(p);
        uint32_t o = (uint32_t)(p - base);
        uint32_t i = o >> 2, r = 8 * (o & 3);
        uint32_t w0 = rdlane(v, i);
        uint32_t w1 = rdlane(v, i + 1);
        uint32_t w2 = rdlane(v, i + 2);
        uint64_t lo = ((uint64_t)w1 << 32) | w0;
        if (r == 0)
            return lo;
        return (lo >> r) | ((uint64_t)w2 << (64 - r));
    }
};

} // namespace snapmi
