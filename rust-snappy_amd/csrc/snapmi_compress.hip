// snapmi_compress.hip -- Snappy raw block compressor for gfx950 (CDNA4).
//
// What is computed is fixed by the reference (src/compress.rs: one greedy
// LZ77 parse per <=64 KiB block with a 16-bit hash table, the skip heuristic
// and the literal / copy-1 / copy-2 encoders) and must be bit-exact.  How it
// is computed is CDNA4-native.  Two match finders produce the same bytes:
// k_match_blocks + k_encode_tokens (one LANE per block, hash tables in HBM:
// the fast path for batches of thousands of blocks, described at its
// definition below) and k_compress_blocks (one WAVEFRONT per block, hash
// table in LDS: small batches, where the latency of a block matters):
//
//   * one wavefront owns one 64 KiB block at a time; blocks are independent
//     (fresh zeroed table per block, offsets never leave the block:
//     reference src/compress.rs:148,514-516), so the work list is simply
//     "all blocks of all streams of the batch";
//   * the u16 hash table (<=32 KiB) lives in LDS.  gfx950 hands out LDS in
//     1280-byte granules, so five separate 32 KiB workgroups do NOT fit a CU
//     (5 x 33280 > 163840; measured: tests/hw/lds_occupancy.hip).  The kernel
//     therefore runs ONE persistent workgroup of five wavefronts per CU that
//     owns all 160 KiB (5 x 32768 = 128 granules exactly); each wavefront
//     keeps its own table and pulls 64 KiB blocks from a device-wide ticket
//     counter until the batch is empty;
//   * the reference probes one position at a time (src/compress.rs:207-245).
//     Here the 64 lanes evaluate the next 64 positions of the reference's
//     probe schedule at once.  The sequential table semantics
//     ("candidate = table[h]; table[h] = s", in probe order) are reproduced
//     by ONE LDS atomic: ds_mskor_rtn_b32 replaces the 16-bit field and
//     returns the old dword, and gfx950 applies the lanes of one DS atomic
//     that hit the same address in ascending lane order (checked on hardware
//     by k_probe_lds_order below at every context creation - a context on a
//     device where it does not hold never launches this kernel - and, more
//     broadly, by tests/hw/lds_atomic_order.hip).  Entries written by lanes
//     past the first hit are rolled back;
//   * every lane compares 16 bytes at its candidate, so the first hit also
//     has its match length (reference extend_match, src/compress.rs:378-412);
//     longer matches continue 256 bytes per wave instruction;
//   * the step after a copy (reference :290-313: insert s-1, check s, fall
//     back to probing s+1...) is the same batch with two leading lanes, so
//     the kernel does one LDS atomic + one candidate gather per emitted copy;
//   * the match finder only records (literal, copy) tokens in two VGPRs, one
//     token per lane; every 64 tokens the wave encodes them all at once:
//     each lane sizes its own elements, a DPP wave scan places them, and the
//     lanes write their tags and copy their literal bytes in parallel.  No
//     store (and no vmcnt wait for one) sits on the per-copy dependency
//     chain; literals longer than 64 bytes are copied 1 KiB per instruction.
//
// Blocks 0 of every stream are written straight into the caller's output
// (after the varint); later blocks go to scratch slots and are moved into
// place by k_compact once the sizes are known (reference Encoder::compress
// concatenates them serially, src/compress.rs:128-153).
#include <type_traits>

#include "snapmi_device.hpp"
#include "snapmi_kernels.hpp"
#include "snapmi_span.hpp"

namespace snapmi {

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x, uint32_t shift)
{
    return (x * 0x1E35A7BDu) >> shift; // reference src/compress.rs:523-525
}

// Offsets of the reference's probe schedule from the position where the
// probe loop is (re)entered: skip = 32; step = skip >> 5; skip += step
// (reference src/compress.rs:204-211).  d[i] is the i-th probed offset.
struct DeltaTable {
    uint32_t d[448];
};
constexpr DeltaTable make_delta()
{
    DeltaTable t{};
    uint32_t skip = 32, p = 0;
    for (int i = 0; i < 448; i++) {
        t.d[i] = p < 0x100000u ? p : 0x100000u;
        const uint32_t step = skip >> 5;
        p += step;
        skip += step;
    }
    return t;
}
__device__ const DeltaTable kDelta = make_delta();

// 16 unaligned bytes.
struct B16 {
    uint32_t w[4];
};
__device__ __forceinline__ B16 ld128u(gcptr p)
{
    B16 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// the same loads from a copy of the block in LDS (k_compress_block_lds)
typedef __attribute__((address_space(3))) uint8_t l_u8c;
typedef const l_u8c *lcptr;
__device__ __forceinline__ B16 ld128u(lcptr p)
{
    B16 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
using snapmi::ld32u; // (the global-memory overload of snapmi_device.hpp)
__device__ __forceinline__ uint32_t ld32u(lcptr p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// number of equal leading bytes of two 16-byte values (0..16)
__device__ __forceinline__ uint32_t common16(const B16 &a, const B16 &b)
{
    const uint64_t d0 = (((uint64_t)(a.w[1] ^ b.w[1])) << 32) | (a.w[0] ^ b.w[0]);
    const uint64_t d1 = (((uint64_t)(a.w[3] ^ b.w[3])) << 32) | (a.w[2] ^ b.w[2]);
    // branch-free on purpose: both halves are always needed, so the compiler
    // keeps the candidate read as one 16-byte load
    const uint32_t m0 = d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u;
    const uint32_t m1 = d1 ? (uint32_t)__builtin_ctzll(d1) >> 3 : 8u;
    return m0 == 8 ? 8 + m1 : m0;
}

// ds_mskor_rtn_b32: MEM = (MEM & ~mask) | data, returns the old dword.
__device__ __forceinline__ uint32_t lds_mskor_rtn(uint32_t byte_addr,
                                                  uint32_t mask, uint32_t data)
{
    uint32_t old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old)
                 : "v"(byte_addr), "v"(mask), "v"(data)
                 : "memory");
    return old;
}

// DPP helpers: wave64 inclusive add-scan without LDS (row_shr within rows of
// 16, then row_bcast:15 / row_bcast:31 across rows).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    // lanes without a source (bound_ctrl) and rows outside ROW_MASK add 0
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(
        0, (int)v, CTRL, ROW_MASK, 0xF, false);
    return v + t;
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v = dpp_add<0x111, 0xF>(v); // row_shr:1
    v = dpp_add<0x112, 0xF>(v); // row_shr:2
    v = dpp_add<0x114, 0xF>(v); // row_shr:4
    v = dpp_add<0x118, 0xF>(v); // row_shr:8
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 -> rows 2,3
    return v;
}

// Tokens of the greedy parse, one per lane, encoded 64 at a time.
//   a = literal length | copy offset << 16
//   b = copy length    | literal start << 16      (copy length 0: no copy)
// A token with neither literal nor copy is never recorded, so the pattern
// "both zero" encodes the one length that does not fit 16 bits: a 65536-byte
// literal (a whole block without a single match).
// Encoded size of one token = literal of L bytes, then a copy of C bytes at
// offset O (C = 0: none): reference emit_literal (src/compress.rs:433-474)
// and emit_copy (:323-369).  The match finder adds these up per block so the
// encoder can write every block at its final position.
__device__ __forceinline__ uint32_t token_bytes(uint32_t L, uint32_t C,
                                                uint32_t O)
{
    const uint32_t lt = L == 0 ? 0 : (L <= 60 ? 1 : (L <= 256 ? 2 : 3));
    const uint32_t n64 = C >= 68 ? (C - 4) >> 6 : 0;
    const uint32_t rem = C - (n64 << 6);
    const uint32_t mid = rem > 64 ? 1 : 0;
    const uint32_t fin_len = rem - 60 * mid;
    const uint32_t fin = C == 0 ? 0 : (fin_len <= 11 && O <= 2047 ? 2 : 3);
    return lt + L + 3 * (n64 + mid) + fin;
}

struct TokenSink {
    gcptr src;          // block input
    uint32_t n;         // block length
    gptr dst;           // block output
    uint32_t d;         // bytes written so far
    uint32_t a, b;      // this lane's token
    uint32_t t;         // tokens pending (uniform)
    uint32_t lane;

    __device__ __forceinline__ void init(gcptr s, uint32_t len, gptr o,
                                         uint32_t l)
    {
        src = s;
        n = len;
        dst = o;
        d = 0;
        a = b = 0;
        t = 0;
        lane = l;
    }

    __device__ __forceinline__ void record(uint32_t lit_start,
                                           uint32_t lit_len, uint32_t offset,
                                           uint32_t copy_len)
    {
        const bool me = lane == t;
        a = me ? ((lit_len & 0xFFFFu) | (offset << 16)) : a;
        b = me ? (copy_len | (lit_start << 16)) : b;
        t++;
        if (t == kWave)
            flush();
    }

    // the same without the flush: for a caller that makes room itself
    // (k_compress_spans: at most 17 tokens per step, one flush site)
    __device__ __forceinline__ void record_nf(uint32_t lit_start,
                                              uint32_t lit_len,
                                              uint32_t offset,
                                              uint32_t copy_len)
    {
        const bool me = lane == t;
        a = me ? ((lit_len & 0xFFFFu) | (offset << 16)) : a;
        b = me ? (copy_len | (lit_start << 16)) : b;
        t++;
    }

    // Encode the pending tokens: reference emit_literal (src/compress.rs:
    // 433-474) and emit_copy / emit_copy2 (:323-369), one token per lane.
    __device__ __forceinline__ void flush()
    {
        const bool act = lane < t;
        const uint32_t C = act ? (b & 0xFFFFu) : 0; // copy length
        uint32_t L = act ? (a & 0xFFFFu) : 0;       // literal length
        if (act && L == 0 && C == 0)
            L = kMaxBlock;
        const uint32_t O = a >> 16;                 // copy offset
        const uint32_t P = b >> 16;                 // literal start
        // literal tag size (:436-463)
        const uint32_t lt = L == 0 ? 0 : (L <= 60 ? 1 : (L <= 256 ? 2 : 3));
        // copy pieces (:339-356): n64 x copy2(64), maybe copy2(60), then the
        // tail as copy1 or copy2
        const uint32_t n64 = C >= 68 ? (C - 4) >> 6 : 0;
        const uint32_t rem = C - (n64 << 6);
        const uint32_t mid = rem > 64 ? 1 : 0;
        const uint32_t fin_len = rem - 60 * mid;
        const bool c1 = fin_len <= 11 && O <= 2047;
        const uint32_t fin = C == 0 ? 0 : (c1 ? 2 : 3);
        const uint32_t size = lt + L + 3 * (n64 + mid) + fin;
        const uint32_t incl = wave_inclusive_scan(size);
        gptr o = dst + d + (incl - size);
        d += rdlane(incl, kWave - 1);
        t = 0;

        // literal tag
        if (lt) {
            const uint32_t n1 = L - 1;
            if (lt == 1) {
                o[0] = (uint8_t)(n1 << 2);
            } else {
                o[0] = lt == 2 ? (60u << 2) : (61u << 2);
                o[1] = (uint8_t)n1;
                if (lt == 3)
                    o[2] = (uint8_t)(n1 >> 8);
            }
        }
        o += lt;
        // literal bytes, one lane per literal while they are short: up to
        // four 16-byte pieces, ALL loads first (one memory latency for the
        // whole literal instead of one per 4 bytes: the encoder's passes are
        // a chain of such latencies), then the stores - whole pieces as one
        // 16-byte store, the last one dword- and bytewise
        gcptr in = src + P;
        const uint32_t Lp = (L + 15u) & ~15u;
        if (L && L <= 64 && P + Lp <= n) {
            u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
            v0 = ld128g(in);
            if (L > 16)
                v1 = ld128g(in + 16);
            if (L > 32)
                v2 = ld128g(in + 32);
            if (L > 48)
                v3 = ld128g(in + 48);
            typedef __attribute__((address_space(1))) u32x4 g_u32x4;
            typedef g_u32x4 __attribute__((aligned(1))) g_u32x4u;
            if (L >= 16)
                *(g_u32x4u *)o = v0;
            if (L >= 32)
                *(g_u32x4u *)(o + 16) = v1;
            if (L >= 48)
                *(g_u32x4u *)(o + 32) = v2;
            if (L >= 64)
                *(g_u32x4u *)(o + 48) = v3;
            // the last, partial piece (L & 15 bytes at o + (L & ~15))
            const uint32_t pk = L >> 4; // piece that holds it
            const u32x4 pv = pk == 0 ? v0 : (pk == 1 ? v1 : (pk == 2 ? v2 : v3));
            gptr po = o + (L & ~15u);
            const uint32_t r = L & 15u;
            if (r >= 4)
                st32u(po, pv.x);
            if (r >= 8)
                st32u(po + 4, pv.y);
            if (r >= 12)
                st32u(po + 8, pv.z);
            const uint32_t tl = r & 3u, tb = r & ~3u;
            const uint32_t ti = r >> 2; // dword holding the tail
            const uint32_t tw =
                ti == 0 ? pv.x : (ti == 1 ? pv.y : (ti == 2 ? pv.z : pv.w));
            if (tl >= 1)
                po[tb] = (uint8_t)tw;
            if (tl >= 2)
                po[tb + 1] = (uint8_t)(tw >> 8);
            if (tl >= 3)
                po[tb + 2] = (uint8_t)(tw >> 16);
        } else if (L && L <= 64) { // within 16 bytes of the block's end
            uint32_t i = 0;
            for (; i + 4 <= L; i += 4)
                st32u(o + i, ld32u(in + i));
            for (; i < L; i++)
                o[i] = in[i];
        }
        // copies
        gptr oc = o + L;
        const uint8_t olo = (uint8_t)O, ohi = (uint8_t)(O >> 8);
        for (uint32_t i = 0; i < n64; i++) {
            oc[0] = (uint8_t)((63u << 2) | 2u);
            oc[1] = olo;
            oc[2] = ohi;
            oc += 3;
        }
        if (mid) {
            oc[0] = (uint8_t)((59u << 2) | 2u);
            oc[1] = olo;
            oc[2] = ohi;
            oc += 3;
        }
        if (fin == 2) {
            oc[0] = (uint8_t)(((O >> 8) << 5) | ((fin_len - 4) << 2) | 1u);
            oc[1] = olo;
        } else if (fin == 3) {
            oc[0] = (uint8_t)(((fin_len - 1) << 2) | 2u);
            oc[1] = olo;
            oc[2] = ohi;
        }
        // long literals: the whole wave copies each (wave_copy)
        uint64_t longs = __ballot(L > 64);
        while (longs) {
            const uint32_t j = (uint32_t)__builtin_ctzll(longs);
            longs &= longs - 1;
            const uint32_t Lj = rdlane(L, j);
            const uint32_t Pj = rdlane(P, j);
            const uint64_t oj = ((uint64_t)rdlane((uint32_t)((uintptr_t)o >> 32), j) << 32) |
                                rdlane((uint32_t)(uintptr_t)o, j);
            wave_copy<true>((gptr)(uintptr_t)oj, src + Pj, Lj, lane);
        }
    }
};

// Continue a match past its first 16 bytes: common prefix of src[c..] and
// src[p..] bounded by the block end n, 256 bytes per wave instruction
// (reference extend_match, src/compress.rs:378-412).
template <class P>
__device__ __forceinline__ uint32_t extend_match(P src, uint32_t n,
                                                 uint32_t c, uint32_t p,
                                                 uint32_t lane)
{
    uint32_t len = 0;
    const uint32_t room = n - p;
    for (uint32_t pos = 0;; pos += 4 * kWave) {
        const uint32_t off = pos + 4 * lane;
        uint32_t eq = 0;
        if (off < room) {
            // dword loads clamped to the block; a clamped load is shifted so
            // the wanted bytes sit at the bottom (the rest is cut by `lim`)
            const uint32_t pa = c + off, pb = p + off;
            const uint32_t ca = pa < n - 4 ? pa : n - 4;
            const uint32_t cb = pb < n - 4 ? pb : n - 4;
            const uint32_t va = ld32u(src + ca) >> (8 * (pa - ca));
            const uint32_t vb = ld32u(src + cb) >> (8 * (pb - cb));
            const uint32_t x = va ^ vb;
            eq = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
            const uint32_t lim = room - off;
            eq = eq < lim ? eq : lim;
        }
        const uint64_t stop = __ballot(eq < 4);
        if (stop == 0) {
            len += 4 * kWave;
            continue;
        }
        const uint32_t f = (uint32_t)__builtin_ctzll(stop);
        return len + 4 * f + rdlane(eq, f);
    }
}


} // namespace

// One 64 KiB block, by one wavefront.  kLds: the match finder reads the block
// from a copy in LDS at `lblock` (filled here) instead of HBM / L2 - every
// load on the per-copy dependency chain is then an LDS access.
template <bool kLds>
__device__ __forceinline__ void compress_one_block(
    const CompressArgs &a, const uint32_t b, const uint32_t lane,
    const lptr16 table, const uint32_t tbase, const uint32_t c2,
    const uint32_t c3, const uint32_t cB, const uint32_t cBn,
    __attribute__((address_space(3))) uint8_t *lblock = nullptr)
{

    // stream lookup: blk_first[st] <= b < blk_first[st + 1]
    uint32_t lo = 0, hi = a.n_streams;
    // (a batch of one-block streams - pages, frame chunks: block b IS
    // stream b, and two loads say so instead of log2(n) dependent ones)
    if (b < a.n_streams && a.blk_first[b] == b && a.blk_first[b + 1] > b) {
        lo = b;
        hi = b + 1;
    }
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.blk_first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t st = lo;
    const uint32_t k = b - a.blk_first[st];
    const uint64_t total = a.in_lens[st];
    const uint64_t boff = (uint64_t)k * kMaxBlock;
    gcptr src = (gcptr)a.in_ptrs[st] + boff;
    const uint64_t avail = total - boff;
    const uint32_t n = avail < kMaxBlock ? (uint32_t)avail : kMaxBlock;

    gptr dst;
    if (k == 0) {
        // varint(total) then block 0, in place: reference
        // src/compress.rs:128 and src/bytes.rs:61-70
        dst = (gptr)a.out_ptrs[st];
        if (lane == 0) {
            uint64_t v = total;
            uint32_t i = 0;
            while (v >= 0x80) {
                dst[i++] = (uint8_t)v | 0x80;
                v >>= 7;
            }
            dst[i] = (uint8_t)v;
        }
        dst += varint_len(total);
    } else {
        const uint32_t slot = a.slot_first[st] + k - 1;
        if (slot >= a.host_slots)
            return; // stream rejected by k_plan_compress (E_ARGUMENT)
        dst = (gptr)a.scratch + (uint64_t)slot * kSlotBytes;
    }
    TokenSink out;
    out.init(src, n, dst, lane);

    if (n < kMinNonLiteral) { // reference src/compress.rs:140-146
        out.record(0, n, 0, 0);
        out.flush();
        if (lane == 0)
            a.blk_size[b] = out.d;
        return;
    }
    // what the match finder reads: the block itself, or its copy in LDS
    typename std::conditional<kLds, lcptr, gcptr>::type msrc;
    if constexpr (kLds) {
        // 1 KiB per wave instruction, coalesced; the tail bytewise
        typedef __attribute__((address_space(3))) u32x4 l_u32x4;
        typedef __attribute__((address_space(1))) u32x4 g_u32x4c;
        const uint32_t mis = (uint32_t)(uintptr_t)src & 15u; // 16-byte lines
        l_u32x4 *to = (l_u32x4 *)lblock;
        // LDS copy starts at the 16-byte line that holds src[0]
        const g_u32x4c *from = (const g_u32x4c *)(src - mis);
        const uint32_t lines = (n + mis + 15) / 16;
        for (uint32_t i = lane; i < lines; i += kWave)
            to[i] = from[i]; // (never past the line that holds the last byte)
        __builtin_amdgcn_wave_barrier();
        msrc = (lcptr)lblock + mis;
    } else {
        msrc = src;
    }

    // table sizing + zero fill: reference src/compress.rs:491-518
    uint32_t shift = 32 - 8, tsize = 256;
    while (tsize < kMaxTable && tsize < n) {
        shift--;
        tsize *= 2;
    }
    for (uint32_t i = 8 * lane; i < tsize; i += 8 * kWave)
        *(__attribute__((address_space(3))) u32x4 *)&table[i] =
            (u32x4){0, 0, 0, 0};
    // LDS operations of one wavefront execute in order: no barrier needed
    // between the zero fill and the first table access of this wavefront
    __builtin_amdgcn_wave_barrier();

    // reference Block::compress, src/compress.rs:195-317.
    //  chain == false: probing started at position 1 (block start);
    //  chain == true : a copy just ended at s (s < s_limit): lane 0 inserts
    //                  s-1, lane 1 is the check at s (:290-306), lanes 2..
    //                  are the probe loop restarted at s+1 (:310-312,:204).
    //  q: probes of the current run already done by earlier batches.
    const uint32_t s_limit = n - kInputMargin;
    uint32_t s = 0, next_emit = 0, q = 0;
    bool chain = false;
    // Register window of the input for hashing: wv0/1/2[lane] = the dword at
    // block offset wbase + {0,64,128} + lane.  After a copy the next batch's
    // hash inputs are a lane rotation of these (3 ds_bpermute), so the HBM/L2
    // round trip for "the 4 bytes at each probed position" leaves the chain.
    // wv3 is the prefetch slot: loaded one slide ahead of its first use.
    uint32_t wbase = 0x80000000u, wv0 = 0, wv1 = 0, wv2 = 0, wv3 = 0;
    const uint32_t cI = cB + 1; // offset of this lane's probe from s - 1
    PROF(
    uint64_t pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_last = __builtin_readcyclecounter();
    uint64_t n_batches = 0, n_copies = 0;
    )
    for (;;) {
        PROF(
        n_batches++;
        )
        TICK(0);
        uint32_t p, nextp;
        bool active = true, probe = true;
        const uint32_t run0 = chain ? s + 1 : 1; // where the probe run began
        if (q == 0) {
            if (!chain) {
                p = lane ? run0 + c2 : 0;
                nextp = run0 + c3;
                active = lane != 0; // lane 0 only fetches bytes 0..15
            } else {
                p = s + cB;
                nextp = s + cBn;
                probe = lane != 0;
            }
        } else {
            p = run0 + kDelta.d[q + lane];
            nextp = run0 + kDelta.d[q + lane + 1];
        }
        // reference: "if s_next > s_limit return done()", :212-214
        const bool valid = active && nextp <= s_limit;
        const uint32_t pc = p < n - 16 ? p : n - 16;
        uint32_t hx = 0; // the 4 bytes at p, for the hash
        if (chain && q == 0) {
            // from the register window: no memory round trip before the hash
            const uint32_t idx = s - 1 - wbase + cI; // < 192
            const int sel = (int)((idx & 63) << 2);
            const uint32_t g0 =
                (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)wv0);
            const uint32_t g1 =
                (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)wv1);
            const uint32_t g2 =
                (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)wv2);
            hx = idx < 64 ? g0 : (idx < 128 ? g1 : g2);
        }
        // issued after the window reads so that its latency overlaps the
        // hash, the LDS atomic and the candidate gather
        const B16 x = ld128u(msrc + pc);
        if (!(chain && q == 0)) {
            hx = x.w[0];
            // keep this a real (uniform) branch: as a select it would make
            // the hash wait for x on the window path too
            asm volatile("" : "+v"(hx));
        }
        TICK(1);
        const uint32_t h = hash32(hx, shift);
        uint32_t cand = 0;
        if (valid) {
            const uint32_t sh = (h & 1) * 16;
            const uint32_t old = lds_mskor_rtn(tbase + (h >> 1) * 4,
                                               0xFFFFu << sh, p << sh);
            cand = (old >> sh) & 0xFFFFu;
        }
        TICK(2);
        const B16 y = ld128u(msrc + cand);
        TICK(3);
        const uint32_t m = common16(x, y);
        // (cand < p always holds when the DS atomic applies lanes in ascending
        // order, which snapmi_ctx_create verifies; the test keeps the output
        // a valid stream even if that ever failed)
        const uint64_t hits = __ballot(valid && probe && m >= 4 && cand < p);
        if (hits == 0) {
            if (__ballot(active && !valid) != 0)
                break; // ran into s_limit: done
            q += q ? kWave : (chain ? kWave - 2 : kWave - 1);
            continue;
        }
        const uint32_t kh = (uint32_t)__builtin_ctzll(hits);
        const uint32_t pk = rdlane(p, kh);
        const uint32_t ck = rdlane(cand, kh);
        uint32_t len = rdlane(m, kh);
        // lanes past the hit never ran in the reference: give the table
        // back the value that was there before the first of them
        if (valid && lane > kh && cand <= pk)
            table[h] = (uint16_t)cand;

        TICK(4);
        TICK(5);
        if (len == 16)
            len += extend_match(msrc, n, ck + 16, pk + 16, lane);
        TICK(6);
        // literal next_emit..pk (reference :250-257) + copy (:272-273)
        out.record(next_emit, pk - next_emit, pk - ck, len);
        TICK(7);
        PROF(
        n_copies++;
        )
        s = pk + len;
        next_emit = s;
        chain = true;
        q = 0;
        // keep the window covering s-1 .. s+158: slide by 64 positions when
        // s-1 has moved past the first register, restart after a long jump
        {
            const uint32_t D = s - 1 - wbase;
            if (D >= 64) {
                if (D < 128) {
                    wv0 = wv1;
                    wv1 = wv2;
                    wv2 = wv3;
                    wbase += 64;
                    const uint32_t wp = wbase + 192 + lane;
                    wv3 = ld32u(msrc + (wp < n - 4 ? wp : n - 4));
                } else {
                    wbase = s - 1;
                    const uint32_t w0 = wbase + lane;
                    const uint32_t n4 = n - 4;
                    wv0 = ld32u(msrc + (w0 < n4 ? w0 : n4));
                    wv1 = ld32u(msrc + (w0 + 64 < n4 ? w0 + 64 : n4));
                    wv2 = ld32u(msrc + (w0 + 128 < n4 ? w0 + 128 : n4));
                    wv3 = ld32u(msrc + (w0 + 192 < n4 ? w0 + 192 : n4));
                }
            }
        }
        if (s >= s_limit) // reference :275-277
            break;
    }
    if (next_emit < n) // reference done(), src/compress.rs:417-426
        out.record(next_emit, n - next_emit, 0, 0);
    if (out.t)
        out.flush();
    if (lane == 0)
        a.blk_size[b] = out.d;
    PROF(
    TICK(8);
    if (lane == 0 && a.prof) {
        for (int i = 0; i < 9; i++)
            atomicAdd(&a.prof[i], (unsigned long long)pt[i]);
        atomicAdd(&a.prof[10], (unsigned long long)n_batches);
        atomicAdd(&a.prof[11], (unsigned long long)n_copies);
        atomicAdd(&a.prof[12], 1ull);
    }
    )
}

// Device-wide block ticket, one 64-bit word: low half = blocks claimed from
// the front of the list (lane-per-block kernel), high half = blocks claimed
// from the back (this wavefront-per-block kernel).  The two kernels run
// concurrently and meet in the middle; a claim is valid while front + back
// < nblocks.  Lane 0 takes the ticket, the wavefront shares it.  Returns the
// block index or 0xFFFFFFFF.
__device__ __noinline__ uint32_t next_ticket(uint32_t *ticket, uint32_t lane,
                                             uint32_t nblocks)
{
    uint32_t blk = 0;
    if (lane == 0) {
        const unsigned long long old =
            atomicAdd((unsigned long long *)ticket, 1ull << 32);
        const uint32_t front = (uint32_t)old, back = (uint32_t)(old >> 32);
        blk = (uint64_t)front + back < nblocks ? nblocks - 1 - back
                                               : 0xFFFFFFFFu;
    }
    return uni(blk);
}

// the same for a kernel that takes blocks from the front only
__device__ __noinline__ uint32_t next_front_ticket(uint32_t *ticket,
                                                   uint32_t lane)
{
    uint32_t t = 0;
    if (lane == 0)
        t = atomicAdd(ticket, 1u);
    return uni(t);
}
// the two-ended ticket over the blocks [lo, hi) of a segment: the next block
// from the BACK (the lane kernel counts the low word up from the front)
__device__ __noinline__ uint32_t next_back_ticket(uint32_t *ticket,
                                                  uint32_t lane, uint32_t lo,
                                                  uint32_t hi)
{
    uint32_t blk = 0;
    if (lane == 0) {
        const unsigned long long old =
            atomicAdd((unsigned long long *)ticket, 1ull << 32);
        const uint32_t front = (uint32_t)old, back = (uint32_t)(old >> 32);
        blk = hi > lo && (uint64_t)front + back < hi - lo ? hi - 1 - back
                                                          : 0xFFFFFFFFu;
    }
    return uni(blk);
}

// ---------------------------------------------------------------------
// Self-check run once per context (snapmi_ctx_create): k_compress_blocks
// needs one wave64 ds_mskor_rtn_b32 to apply lanes that hit the same address
// in ascending lane order - every lane must get back the value left by the
// nearest lower lane with the same slot, and the highest lane's value must
// be the one that stays.  256 patterns (all lanes one slot, heavy, some and
// rare duplicates), slots from a counter-based hash.  *bad != 0 afterwards
// means the property does not hold on this device: the context then never
// uses k_compress_blocks (the lane kernel does not depend on it).
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_probe_lds_order(uint32_t *bad)
{
    __shared__ uint16_t tab[2048];
    const uint32_t lane = threadIdx.x;
    const uint32_t tbase = (uint32_t)(uintptr_t)(lptr16)&tab[0];
    uint32_t fails = 0;
    for (uint32_t t = 0; t < 256; t++) {
        for (uint32_t i = lane; i < 2048; i += 64)
            tab[i] = (uint16_t)(0x8000u + i);
        __builtin_amdgcn_wave_barrier(); // one wave: its LDS ops are in order
        uint32_t r = (t * 64 + lane + 131u * blockIdx.x) * 2654435761u +
                     0x9E3779B9u;
        r ^= r >> 15;
        r *= 0x2C1B3C6Du;
        r ^= r >> 12;
        const uint32_t mode = t & 3;
        const uint32_t slot =
            mode == 0 ? 5u : (mode == 1 ? r % 8 : (mode == 2 ? r % 64 : r % 2048));
        const uint32_t sh = (slot & 1) * 16;
        const uint32_t old = lds_mskor_rtn(tbase + (slot >> 1) * 4,
                                           0xFFFFu << sh, (lane + 1) << sh);
        __builtin_amdgcn_wave_barrier();
        const uint32_t fin = tab[slot];
        uint32_t want = 0x8000u + slot, want_fin = 0;
#pragma unroll
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t sj = rdlane(slot, j);
            if (sj == slot) {
                if (j < lane)
                    want = j + 1;
                want_fin = j + 1;
            }
        }
        fails += ((old >> sh) & 0xFFFFu) != want;
        fails += fin != want_fin;
        __builtin_amdgcn_wave_barrier();
    }
    if (__ballot(fails != 0) != 0 && lane == 0)
        atomicOr(bad, 1u);
    // Second property (k_decompress_streams2): overlapping lanes of ONE plain
    // DS store are applied in ascending lane order too, so a lane may store a
    // whole 16 bytes for a shorter element and the lanes above it repair the
    // excess.  64 layouts: lane strides 1..23 and pseudo-random gaps, every
    // alignment.  bit 1 of *bad.
    {
        typedef __attribute__((address_space(3))) uint8_t l_u8;
        l_u8 *m = (l_u8 *)tab; // 4 KiB
        uint32_t sfail = 0;
        for (uint32_t t = 0; t < 64; t++) {
            const uint32_t gap =
                t < 23 ? t + 1
                       : 1 + ((lane * 2654435761u + t * 40503u) >> 7) % 20;
            uint32_t pos = wave_inclusive_scan(gap) + 3 * t;
            const uint32_t nextpos = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)(((lane + 1) & 63) << 2), (int)pos);
            u32x4 x;
            x.x = x.y = x.z = x.w = 0x01010101u * (lane + 1);
            __builtin_amdgcn_wave_barrier();
            __builtin_memcpy(m + pos, &x, 16);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t b = pos + k;
                if ((lane == 63 || b < nextpos) &&
                    m[b] != (uint8_t)(lane + 1))
                    sfail++;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ... and the decoders' multi-piece form: elements of 1..64 bytes,
        // whole 16-byte pieces, last piece first (four store instructions);
        // every byte must end up with the element that owns it.  The launch
        // fills the chip (32 waves per CU, each with LDS of its own), so the
        // property is checked under the LDS contention of the real kernels,
        // not by one idle wave.
        for (uint32_t t = 0; t < 48; t++) {
            uint32_t r = (t * 64 + lane + 977u * blockIdx.x) * 2654435761u;
            r ^= r >> 13;
            // mostly short elements, some of every length up to 64; the last
            // lanes short enough that 64 elements fit in the 4 KiB
            uint32_t len = (t & 1) ? 1 + (r >> 8) % 64 : 1 + (r >> 8) % 20;
            if (t >= 40)
                len = 17 + (r >> 8) % 48;
            const uint32_t end = wave_inclusive_scan(len) + t;
            const uint32_t pos = end - len;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t c = 48;; c -= 16) {
                if (c < len && pos + c + 16 <= 4096) {
                    u32x4 x;
                    x.x = x.y = x.z = x.w = 0x01010101u * (lane + 1);
                    __builtin_memcpy(m + pos + c, &x, 16);
                }
                if (c == 0)
                    break;
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = 0; k < len; k++)
                if (pos + k < 4096 - 16 && m[pos + k] != (uint8_t)(lane + 1))
                    sfail++;
            __builtin_amdgcn_wave_barrier();
        }
        if (__ballot(sfail != 0) != 0 && lane == 0)
            atomicOr(bad, 2u);
    }
}

// ---------------------------------------------------------------------
// Placement probe for the lane tables (snapmi_api.hip, lane_table_tries):
// the access pattern of k_match_blocks' probe rounds - every lane a chain of
// dependent random 16-byte reads, each followed by a write to the same
// entry, in its own 256 KiB table - without any of the parse.  Its duration
// on a candidate region ranks that region for the real kernel.  The caller
// zeroes the tables afterwards.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_probe_tables(unsigned long long *tables,
                                                     unsigned long long stride,
                                                     uint32_t steps,
                                                     uint32_t chunks,
                                                     uint32_t per_chunk)
{
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    // (the lane kernel's own lane -> table map: CompressArgs::lane_chunks)
    const uint32_t slot =
        chunks ? (gid % chunks) * per_chunk + gid / chunks : gid;
    g_u32x4 *t = (g_u32x4 *)tables + (uint64_t)slot * stride;
    uint32_t state = gid * 2654435761u + 12345u;
    for (uint32_t i = 0; i < steps; i++) {
        const uint32_t h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    if (state == 0x12345678u) // keep the chain alive
        t[0] = (u32x4){state, 0, 0, 0};
}

// n 16-byte entries to zero (the lane tables when they are mapped chunks:
// nothing but a kernel is known to reach every such mapping)
__global__ __launch_bounds__(256) void k_zero16(unsigned long long *p,
                                                unsigned long long n)
{
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    g_u32x4 *q = (g_u32x4 *)p;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 +
                                threadIdx.x;
         i < n; i += (unsigned long long)gridDim.x * 256)
        q[i] = (u32x4){0, 0, 0, 0};
}

#ifdef SNAPMI_TESTING // the one-copy-per-step kernels of rounds 1-3: the
                      // test build's cross-check (option span_kernel 0)
// ---------------------------------------------------------------------
// K1: persistent workgroups of five wavefronts, one 32 KiB table each.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(kCompressWaves * 64) void k_compress_blocks(
    CompressArgs a)
{
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kCompressWaves][kMaxTable];

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const lptr16 table = (lptr16)&tables[wave][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table; // LDS byte offset
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;

    // this lane's slice of the probe schedule
    const uint32_t c2 = lane >= 1 ? kDelta.d[lane - 1] : 0;
    const uint32_t c3 = kDelta.d[lane];
    // after a copy ending at s: lane 0 -> s-1, lane 1 -> s, lane j -> s+1+d[j-2]
    const uint32_t cB = lane >= 2 ? 1 + kDelta.d[lane - 2] : lane - 1;
    const uint32_t cBn = lane >= 2 ? 1 + c2 : 0; // offset of the next probe

    // one ticket per wavefront per block
    // (the helper's result comes back in a VGPR: re-pin it to an SGPR so the
    // whole block state stays scalar)
    uint32_t b = uni(next_ticket(a.ticket, lane, nblocks));
    while (b != 0xFFFFFFFFu) {
        if (lane == 0 && a.ntok)
            a.ntok[b] = 0xFFFFFFFFu; // encoded here, not by k_encode_tokens
        compress_one_block<false>(a, b, lane, table, tbase, c2, c3, cB, cBn);
        b = uni(next_ticket(a.ticket, lane, nblocks));
    }
}

// ---------------------------------------------------------------------
// K1 for the smallest batches: ONE wavefront per CU with the hash table AND
// the 64 KiB input block in LDS (32 KiB + 64 KiB + a line of slack: one such
// workgroup per CU).  Every load on the per-copy dependency chain - the hash
// inputs, the candidate gather, the match extension - is an LDS access, so a
// block finishes about twice as fast as in k_compress_blocks, where the
// gather goes to L2 / HBM.  It is what BASELINE.json's north_star describes
// ("one 64 KiB block per workgroup with the hash table and the input block
// staged in LDS"); with one block per CU in flight it is the right kernel
// only while the batch has no more blocks than a couple per CU (scalar calls,
// small frames): five tables per CU win from there, a lane per block beyond.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_compress_block_lds(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t table_mem[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint8_t block_mem[kMaxBlock + 32];

    const uint32_t lane = threadIdx.x;
    const lptr16 table = (lptr16)&table_mem[0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    const uint32_t c2 = lane >= 1 ? kDelta.d[lane - 1] : 0;
    const uint32_t c3 = kDelta.d[lane];
    const uint32_t cB = lane >= 2 ? 1 + kDelta.d[lane - 2] : lane - 1;
    const uint32_t cBn = lane >= 2 ? 1 + c2 : 0;
    uint32_t b = uni(next_ticket(a.ticket, lane, nblocks));
    while (b != 0xFFFFFFFFu) {
        if (lane == 0 && a.ntok)
            a.ntok[b] = 0xFFFFFFFFu;
        compress_one_block<true>(
            a, b, lane, table, tbase, c2, c3, cB, cBn,
            (__attribute__((address_space(3))) uint8_t *)block_mem);
        b = uni(next_ticket(a.ticket, lane, nblocks));
    }
}

#endif // SNAPMI_TESTING

// ---------------------------------------------------------------------
// K1s: wavefront per block, a WINDOW of 63 consecutive positions per step.
//
// k_compress_blocks pays one LDS atomic, one candidate gather and one ballot
// per emitted copy (~2 100 cycles, ~10 000 copies per 64 KiB of text).  Here
// a step fetches what the table holds and how many bytes match for EVERY
// position of a 63-byte window - whether the parse will look them up or not -
// and the sequential part of the reference (which positions are looked up,
// which are inserted: src/compress.rs:195-317) becomes a walk over results
// that are already in registers: span_walk (snapmi_span.hpp, the text
// tests/test_span_wave_cpu.py runs on the host, byte for byte).  Nine
// copies per step on text, ~1 100 steps per block instead of ~10 000.
//   * lane L holds position s - 1 + L; ONE lane-ordered ds_mskor_rtn_b32
//     writes every position of the window into the table and returns what was
//     there - the reference's candidate, or the position of a lower lane with
//     the same hash ("C bit");
//   * one gather of the 16 bytes at every candidate, one compare: hit mask
//     and match lengths of the whole window;
//   * the walk (scalar unit: ~15 instructions per copy) marks the positions
//     the reference really inserts; one lane-ordered ds_mskor_b32 then puts
//     every slot right (touched lanes write their position, untouched lanes
//     without a C bit give back what they displaced);
//   * a run of more than 32 misses continues with k_compress_blocks' schedule
//     step (64 probes of the growing stride per step, first hit wins); a match
//     of 16 bytes or more is finished by extend_match.
// Same persistent five-wave workgroups, same tables, same token encoder as
// k_compress_blocks; kLds as there (k_compress_span_lds: input block in LDS).
// ---------------------------------------------------------------------
__device__ __forceinline__ void lds_mskor(uint32_t byte_addr, uint32_t mask,
                                          uint32_t data)
{
    asm volatile("ds_mskor_b32 %0, %1, %2"
                 :
                 : "v"(byte_addr), "v"(mask), "v"(data)
                 : "memory");
}

namespace {
// The window kernel as a MATCH FINDER for k_encode_tokens (k_match_spans):
// tokens leave in that kernel's format (four bytes each and an exception
// list, snapmi_kernels.hpp) and what they will encode to is added up
// (token_bytes) - so
// that, as behind k_match_blocks, every block's size is known before a byte
// of it is written and the encoder can put it at its final position (no
// scratch slots, no k_compact).  Round 5: a step's tokens are stored by the
// lanes that found them, at their rank among the step's copies - no push into
// a token register (two ds_permute and their round trip per step), no flush:
// one predicated 8-byte store and a per-lane sum of sizes.
// How the match finders come by the pages of the token pool
// (CompressArgs::tok_pool).  Both are bound by the latency of their own
// dependent chains, and an atomic on the pool's one counter whose result is
// waited for where a page starts is one more link - for a whole wavefront,
// every 512 tokens of any of its lanes (the first form of this: + 6 % on
// k_match_both at cfg2).
//  - a window wavefront stages a block's tokens and moves them into pages at
//    the block's end (tok_commit_wave);
//  - a LANE has a page in hand; the lanes of a wavefront that have used
//    theirs get new ones where the round loop is convergent (its top), from a
//    run of kTokRun pages that the wavefront takes with one atomic, waited
//    for, every 32 pages (match_blocks).
// A lane's page in hand passes from block to block; a launch leaves a page or
// two per lane unused.
constexpr uint32_t kTokRun = 32, kNoPage = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t tok_page_ask(const CompressArgs &a)
{
    return atomicAdd(&a.tok_ctl[0], 1u);
}
// a lane's page in hand `spare` becomes slot `slot` of block b's page table
// - or, the pool having run out, the block is put on the spill list and gets
// the dump page.  `spilled`: it is on the list already.
__device__ __forceinline__ uint32_t tok_page_take(const CompressArgs &a,
                                                  uint32_t b, uint32_t slot,
                                                  uint32_t &spare,
                                                  bool &spilled)
{
    if (!spilled) {
        uint32_t p = spare;
        spare = kNoPage;
        // (the second page of one round - a token page and an exception
        // page, rare: asked for and waited for here)
        if (p == kNoPage)
            p = tok_page_ask(a);
        if (p < a.tok_pool_pages) {
            a.tok_pages[(uint64_t)(b - a.tok_base) * kPageTabStride + slot] =
                p;
            return p;
        }
        spilled = true;
        a.tok_ctl[kTokCtlList + atomicAdd(&a.tok_ctl[1], 1u)] = b;
    }
    return a.tok_pool_pages;
}

// ... for a window wavefront, whose step is bound by the instructions it
// issues and has no scalar register to spare (a comparison per step and a
// rarely taken branch for "this step opens a page" cost k_match_spans 11 %,
// in seven forms): the wavefront writes a block's tokens into a STAGING array
// of its own, as it did when every block had one, and at the block's end
// tok_commit_wave moves them into pages - as many as they need, consecutive,
// taken with one atomic.  Out of line, its arguments the fields it needs (a
// reference to the kernel's argument block would make the kernel keep a copy
// of it in scratch memory).  Returns what goes into CompressArgs::ntok.
__device__ __noinline__ uint32_t tok_commit_wave(
    const uint32_t *stage, uint32_t ntok, uint32_t nexc, uint32_t *ctl,
    uint32_t *tab, uint32_t *pool, uint32_t pool_pages, uint32_t b,
    uint32_t lane)
{
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    const uint32_t tp = (ntok + kTokPage - 1) / kTokPage,
                   ep = (nexc + kExcPage - 1) / kExcPage;
    uint32_t base = 0;
    if (lane == 0) {
        base = atomicAdd(&ctl[0], tp + ep);
        if (base + tp + ep > pool_pages) {
            ctl[kTokCtlList + atomicAdd(&ctl[1], 1u)] = b;
            base = kNoPage;
        }
    }
    base = uni(base);
    if (base == kNoPage)
        return kTokSpilled;
    if (lane < tp)
        tab[lane] = base + lane;
    if (lane < ep)
        tab[kTokPagesPerBlock + lane] = base + tp + lane;
    // (the wavefront's own stores of this block, read back: loads that see
    // the L2, where the atomic above has waited for them to arrive)
    const g_u64 *from = (const g_u64 *)stage;
    g_u64 *to = (g_u64 *)(pool + (uint64_t)base * kTokPage);
    const uint32_t words = (ntok + 1) / 2;
    for (uint32_t i = lane; i < words; i += 4 * kWave) {
        unsigned long long v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        v0 = __hip_atomic_load(from + i, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (i + kWave < words)
            v1 = __hip_atomic_load(from + i + kWave, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if (i + 2 * kWave < words)
            v2 = __hip_atomic_load(from + i + 2 * kWave, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if (i + 3 * kWave < words)
            v3 = __hip_atomic_load(from + i + 3 * kWave, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        to[i] = v0;
        if (i + kWave < words)
            to[i + kWave] = v1;
        if (i + 2 * kWave < words)
            to[i + 2 * kWave] = v2;
        if (i + 3 * kWave < words)
            to[i + 3 * kWave] = v3;
    }
    const g_u64 *efrom = (const g_u64 *)(stage + kMaxTokens);
    g_u64 *eto = (g_u64 *)(pool + (uint64_t)(base + tp) * kTokPage);
    for (uint32_t i = lane; i < nexc; i += kWave)
        eto[i] = __hip_atomic_load(efrom + i, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    return ntok;
}

struct TokenWriter {
    typedef __attribute__((address_space(1))) uint32_t g_u32;
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    g_u32 *tok;    // the wavefront's staging array: this block's tokens
    g_u64 *exc;    // ... and its exception list (tok_commit_wave)
    uint32_t ntok; // tokens stored so far (uniform)
    uint32_t nexc; // exceptions stored so far (uniform)
    uint32_t dsum; // encoded bytes of the tokens THIS LANE stored
    uint32_t d;    // finish(): encoded bytes of the block (uniform)
    uint32_t t;    // always 0 (TokenSink's interface: nothing is pending)
    uint32_t lane;

    __device__ __forceinline__ void init(const CompressArgs &a, uint32_t l)
    {
        tok = (g_u32 *)a.tok_stage +
              (uint64_t)uni(blockIdx.x * a.tok_stage_waves +
                            (threadIdx.x >> 6) - a.tok_stage_wave0) *
                  kTokStageWords;
        exc = (g_u64 *)(tok + kMaxTokens);
        ntok = 0;
        nexc = 0;
        dsum = 0;
        d = 0;
        t = 0;
        lane = l;
    }
    // the block's end: its tokens into pages; what goes into
    // CompressArgs::ntok
    __device__ __forceinline__ uint32_t commit(const CompressArgs &a,
                                               uint32_t b) const
    {
        return tok_commit_wave(
            (const uint32_t *)tok, ntok, nexc, a.tok_ctl,
            a.tok_pages + (uint64_t)(b - a.tok_base) * kPageTabStride,
            a.tok_pool, a.tok_pool_pages, b, lane);
    }
    // one token, the same in every lane
    __device__ __forceinline__ void record_nf(uint32_t lit_start,
                                              uint32_t lit_len,
                                              uint32_t offset,
                                              uint32_t copy_len)
    {
        (void)lit_start; // (k_encode_tokens adds the lengths up)
        const bool fits = tok_fits(lit_len, copy_len);
        if (lane == 0) {
            tok[ntok] = tok_pack(lit_len, copy_len, offset);
            if (!fits)
                exc[nexc] = tok_pack64(lit_len, copy_len, offset);
            dsum += token_bytes(lit_len, copy_len, offset);
        }
        ntok++;
        nexc += fits ? 0u : 1u;
    }
    // the copies of a window step: lane `mine` holds the rank-th of cnt
    // tokens, a copy of 4..15 bytes (one element: src/compress.rs:339-356).
    // Only the step's FIRST copy can come behind a literal of 1 024 bytes or
    // more (the literals between a window's copies are shorter than the
    // window): at most one exception per step, `first_long` (uniform) says so.
    __device__ __forceinline__ void put_step(bool mine, uint32_t rank,
                                             uint32_t cnt, uint32_t lit_len,
                                             uint32_t copy_len,
                                             uint32_t offset)
    {
        const bool big = mine && lit_len > 1023u;
        if (mine)
            tok[ntok + rank] = tok_pack(lit_len, copy_len, offset);
        if (big)
            exc[nexc] = tok_pack64(lit_len, copy_len, offset);
        const uint32_t lt =
            lit_len == 0 ? 0 : (lit_len <= 60 ? 1 : (lit_len <= 256 ? 2 : 3));
        const uint32_t fin = copy_len <= 11 && offset <= 2047 ? 2 : 3;
        dsum += mine ? lt + lit_len + fin : 0;
        ntok += cnt;
        nexc += __builtin_amdgcn_ballot_w64(big) ? 1u : 0u;
    }
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ void finish()
    {
        d = rdlane(wave_inclusive_scan(dsum), kWave - 1);
    }
};
struct SpanLanes {
    uint32_t mv, ov; // this lane's match length and exchanged table entry
    __device__ __forceinline__ uint32_t m(uint32_t l) const
    {
        return rdlane(mv, l);
    }
    __device__ __forceinline__ uint32_t old(uint32_t l) const
    {
        return rdlane(ov, l);
    }
};
template <class OUT> struct SpanSink {
    OUT *out;
    uint32_t emit; // where the pending literal starts
    __device__ __forceinline__ void token(uint32_t lit, uint32_t len,
                                          uint32_t off)
    {
        out->record_nf(emit, lit, off, len);
        emit += lit + len;
    }
};
} // namespace

// the wave of span_par_walk (snapmi_span.hpp) on the hardware
struct WaveDev {
    typedef uint32_t u32;
    typedef bool b1;
    uint32_t l;
    __device__ __forceinline__ u32 lane() const { return l; }
    __device__ __forceinline__ void count_cut() const {}
    __device__ __forceinline__ u32 sel(b1 c, u32 a, u32 b) const
    {
        return c ? a : b;
    }
    __device__ __forceinline__ b1 lt(u32 a, u32 b) const { return a < b; }
    __device__ __forceinline__ b1 ge(u32 a, u32 b) const { return a >= b; }
    __device__ __forceinline__ b1 eq(u32 a, u32 b) const { return a == b; }
    __device__ __forceinline__ b1 band(b1 a, b1 b) const { return a & b; }
    __device__ __forceinline__ b1 bor(b1 a, b1 b) const { return a | b; }
    __device__ __forceinline__ b1 bnot(b1 a) const { return !a; }
    __device__ __forceinline__ uint64_t ballot(b1 a) const
    {
        return __builtin_amdgcn_ballot_w64(a);
    }
    __device__ __forceinline__ u32 bperm(u32 idx, u32 v) const
    {
        return (u32)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)v);
    }
    __device__ __forceinline__ uint32_t readlane(u32 v, uint32_t i) const
    {
        return rdlane(v, i);
    }
    __device__ __forceinline__ b1 bit(uint64_t mask, u32 i) const
    {
        return ((uint32_t)(mask >> i) & 1u) != 0;
    }
    __device__ __forceinline__ u32 next_bit(uint64_t mask, u32 t) const
    {
        const uint64_t x = mask >> t;
        return x ? t + (u32)__builtin_ctzll(x) : 64u;
    }
    __device__ __forceinline__ u32 prev_bit(uint64_t mask, u32 t) const
    {
        const uint64_t x = mask & ((1ull << t) - 1);
        return x ? 63u - (u32)__builtin_clzll(x) : 64u;
    }
    __device__ __forceinline__ u32 shr_lo(uint64_t mask, u32 i) const
    {
        return (u32)(mask >> i);
    }
};

template <bool kLds, bool kTok = false>
__device__ __forceinline__ void compress_one_block_span(
    const CompressArgs &a, const uint32_t b, const uint32_t lane,
    const lptr16 table, const uint32_t tbase,
    __attribute__((address_space(3))) uint8_t *lblock = nullptr,
    const uint32_t tcap = kMaxTable)
{
    // stream lookup: blk_first[st] <= b < blk_first[st + 1]
    uint32_t lo_s = 0, hi_s = a.n_streams;
    // (a batch of one-block streams - pages, frame chunks: block b IS
    // stream b, and two loads say so instead of log2(n) dependent ones)
    if (b < a.n_streams && a.blk_first[b] == b && a.blk_first[b + 1] > b) {
        lo_s = b;
        hi_s = b + 1;
    }
    while (hi_s - lo_s > 1) {
        const uint32_t mid = (lo_s + hi_s) >> 1;
        if (a.blk_first[mid] <= b)
            lo_s = mid;
        else
            hi_s = mid;
    }
    const uint32_t st_i = lo_s;
    const uint32_t k = b - a.blk_first[st_i];
    const uint64_t total = a.in_lens[st_i];
    const uint64_t boff = (uint64_t)k * kMaxBlock;
    gcptr src = (gcptr)a.in_ptrs[st_i] + boff;
    const uint64_t avail = total - boff;
    const uint32_t n = avail < kMaxBlock ? (uint32_t)avail : kMaxBlock;
    if (n <= a.cls_lo || n > a.cls_hi)
        return; // another launch's block (CompressArgs::cls_lo)

    typename std::conditional<kTok, TokenWriter, TokenSink>::type out;
    if constexpr (kTok) {
        // match finder only: tokens for k_encode_tokens (which also writes
        // the varint of block 0)
        out.init(a, lane);
    } else {
        gptr dst;
        if (k == 0) {
            dst = (gptr)a.out_ptrs[st_i];
            if (lane == 0) { // varint(total): src/compress.rs:128
                uint64_t v = total;
                uint32_t i = 0;
                while (v >= 0x80) {
                    dst[i++] = (uint8_t)v | 0x80;
                    v >>= 7;
                }
                dst[i] = (uint8_t)v;
            }
            dst += varint_len(total);
        } else if (a.direct) {
            // (k_redo_spilled behind a lane-kernel batch: the sizes of the
            // blocks in front are known, as in k_encode_tokens)
            const uint32_t first = a.blk_first[st_i];
            const uint64_t nb = (total + kMaxBlock - 1) / kMaxBlock;
            if (first + nb > a.host_blocks)
                return; // rejected by k_plan_compress (E_ARGUMENT)
            dst = (gptr)a.out_ptrs[st_i] + varint_len(total) +
                  (a.blk_off[b] - a.blk_off[first]);
        } else {
            const uint32_t slot = a.slot_first[st_i] + k - 1;
            if (slot >= a.host_slots)
                return; // stream rejected by k_plan_compress (E_ARGUMENT)
            dst = (gptr)a.scratch + (uint64_t)slot * kSlotBytes;
        }
        out.init(src, n, dst, lane);
    }
    if (n < kMinNonLiteral) { // src/compress.rs:140-146
        out.record_nf(0, n, 0, 0);
        if constexpr (kTok)
            out.finish();
        else
            out.flush();
        uint32_t count = 0;
        if constexpr (kTok)
            count = out.commit(a, b);
        if (lane == 0) {
            a.blk_size[b] = out.d;
            if constexpr (kTok)
                a.ntok[b] = count;
        }
        return;
    }
    typename std::conditional<kLds, lcptr, gcptr>::type msrc;
    if constexpr (kLds) {
        typedef __attribute__((address_space(3))) u32x4 l_u32x4;
        typedef __attribute__((address_space(1))) u32x4 g_u32x4c;
        const uint32_t mis = (uint32_t)(uintptr_t)src & 15u;
        l_u32x4 *to = (l_u32x4 *)lblock;
        const g_u32x4c *from = (const g_u32x4c *)(src - mis);
        const uint32_t lines = (n + mis + 15) / 16;
        for (uint32_t i = lane; i < lines; i += kWave)
            to[i] = from[i];
        __builtin_amdgcn_wave_barrier();
        msrc = (lcptr)lblock + mis;
    } else {
        msrc = src;
    }
    // table sizing + zero fill: src/compress.rs:491-518
    // (tcap: the entries this kernel's table has room for - kMaxTable, or the
    // 8 192 of the small-block kernel, whose class never needs more)
    uint32_t shift = 32 - 8, tsize = 256;
    while (tsize < tcap && tsize < n) {
        shift--;
        tsize *= 2;
    }
    for (uint32_t i = 8 * lane; i < tsize; i += 8 * kWave)
        *(__attribute__((address_space(3))) u32x4 *)&table[i] =
            (u32x4){0, 0, 0, 0};
    __builtin_amdgcn_wave_barrier();

    const uint32_t s_limit = n - kInputMargin;
    const uint32_t n16 = n - 16, n4 = n - 4;
    SpanState st;
    st.s = 1;
    st.q = 0;
    st.chain = 0;
    st.next_emit = 0;
    uint32_t run0 = 1; // schedule steps: where the run began
    uint32_t dq = ~0u, d0 = 0, d1 = 0; // kDelta.d[dq + lane], [dq + lane + 1]
    // Register window of the input for the hashes (global input only): lane l
    // of wv[j] holds the dword at block offset wbase + 64 j + l; five
    // registers, so that a step may move s by up to 128 positions and the
    // next step's hash inputs are still a lane rotation (ds_bpermute) of
    // registers that arrived long ago.
    uint32_t wbase = 0, wv0 = 0, wv1 = 0, wv2 = 0, wv3 = 0, wv4 = 0;
    if constexpr (!kLds) {
        wv0 = ld32u(msrc + (lane < n4 ? lane : n4));
        wv1 = ld32u(msrc + (lane + 64 < n4 ? lane + 64 : n4));
        wv2 = ld32u(msrc + (lane + 128 < n4 ? lane + 128 : n4));
        wv3 = ld32u(msrc + (lane + 192 < n4 ? lane + 192 : n4));
        wv4 = ld32u(msrc + (lane + 256 < n4 ? lane + 256 : n4));
    }
    SpanSink<decltype(out)> sink;
    sink.out = &out;
    sink.emit = 0;
    PROF(
    uint64_t pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_last = __builtin_readcyclecounter();
    uint64_t n_batches = 0, n_copies = 0;
    )
    for (;;) {
        PROF(
        n_batches++;
        )
        TICK(0);
        // room for a step's tokens (at most 16 copies of a window + a long
        // match): the one place where tokens are encoded
        if constexpr (!kTok) {
            if (out.t + 17 > kWave)
                out.flush();
        }
        if (!st.chain && st.q >= kSpanRun) {
            // ---- schedule step (k_compress_blocks' batch for q > 0): lane l
            // is probe q + l of the run that began at run0
            // (this lane's two schedule entries were requested one step
            // ahead: a run of misses goes on with q + 64, and a load from the
            // table costs a step of incompressible data a fifth of its time)
            if (dq != st.q) {
                d0 = kDelta.d[st.q + lane];
                d1 = kDelta.d[st.q + lane + 1];
            }
            const uint32_t p = run0 + d0;
            const uint32_t nextp = run0 + d1;
            {
                const uint32_t qn = st.q + kWave + lane;
                d0 = kDelta.d[qn < 446 ? qn : 446];
                d1 = kDelta.d[qn < 446 ? qn + 1 : 447];
                dq = st.q + kWave;
            }
            const bool valid = nextp <= s_limit; // src/compress.rs:212-214
            const B16 x = ld128u(msrc + (p < n16 ? p : n16));
            const uint32_t h = hash32(x.w[0], shift);
            uint32_t cand = 0;
            if (valid) {
                const uint32_t sh = (h & 1) * 16;
                const uint32_t old = lds_mskor_rtn(tbase + (h >> 1) * 4,
                                                   0xFFFFu << sh, p << sh);
                cand = (old >> sh) & 0xFFFFu;
            }
            const B16 y = ld128u(msrc + cand);
            const uint32_t m = common16(x, y);
            const uint64_t hits = __ballot(valid && m >= 4 && cand < p);
            if (hits == 0) {
                if (__ballot(!valid) != 0)
                    break;
                st.q += kWave;
                continue;
            }
            const uint32_t kh = (uint32_t)__builtin_ctzll(hits);
            const uint32_t pk = rdlane(p, kh);
            const uint32_t ck = rdlane(cand, kh);
            uint32_t len = rdlane(m, kh);
            if (valid && lane > kh && cand <= pk)
                table[h] = (uint16_t)cand;
            if (len == 16)
                len += extend_match(msrc, n, ck + 16, pk + 16, lane);
            sink.token(pk - sink.emit, len, pk - ck);
            PROF(
            n_copies++;
            )
            st.s = pk + len;
            st.next_emit = st.s;
            st.chain = 1;
            st.q = 0;
            if (st.s >= s_limit) // src/compress.rs:275-277
                break;
        } else {
            // ---- window step: lane L = position s - 1 + L
            const uint32_t base = st.s;
            const uint32_t lo = base - st.chain; // first position exchanged
            const uint32_t P = base - 1 + lane;
            const bool active = lane ? P <= n16 : st.chain != 0;
            uint32_t hx;
            if constexpr (kLds) {
                hx = ld32u(msrc + (P < n4 ? P : n4));
            } else {
                const uint32_t idx = base - 1 - wbase + lane; // < 128
                const int sel = (int)((idx & 63) << 2);
                const uint32_t g0 =
                    (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)wv0);
                const uint32_t g1 =
                    (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)wv1);
                hx = idx < 64 ? g0 : g1;
            }
            const B16 x = ld128u(msrc + (P < n16 ? P : n16));
            TICK(1);
            const uint32_t h = hash32(hx, shift);
            const uint32_t sh = (h & 1) * 16;
            const uint32_t taddr = tbase + (h >> 1) * 4;
            uint32_t old = 0;
            if (active)
                old = (lds_mskor_rtn(taddr, 0xFFFFu << sh, P << sh) >> sh) &
                      0xFFFFu;
            TICK(2);
            const B16 y = ld128u(msrc + old);
            TICK(3);
            SpanLanes ln;
            ln.mv = common16(x, y);
            ln.ov = old;
            const bool cbit = active && old >= lo;
            const uint64_t hits = __builtin_amdgcn_ballot_w64(
                active && lane && ln.mv >= 4);
            const uint64_t cbits = __builtin_amdgcn_ballot_w64(cbit);
            uint64_t touched = 0;
            uint32_t at = 0, rc = kSpanCont;
            st.next_emit = sink.emit;
            // the fast walk (snapmi_span.hpp): the scalar unit follows the
            // chain of copies, the lanes derive everything else at once
            const WaveDev w{lane};
            bool fast = span_fast_ok_w(w, st, hits, n);
            TICK(13);
            if (fast) {
                // the lane-parallel walk (span_par_walk): the copies of the
                // step by pointer jumping, everything else per lane
                uint64_t vh;
                uint32_t lit;
                rc = span_par_walk(w, st, hits, ln.mv, old, cbit, sink.emit,
                                   vh, lit, touched, at);
                TICK(14);
                const uint32_t cnt = (uint32_t)__builtin_popcountll(vh);
                if (cnt) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                        (uint32_t)(vh >> 32),
                        __builtin_amdgcn_mbcnt_lo((uint32_t)vh, 0));
                    const bool is_vh = (vh >> lane) & 1;
                    if constexpr (kTok) {
                        out.put_step(is_vh, rank, cnt, lit, ln.mv, P - old);
                    } else {
                        // token of lane X goes to sink lane t + rank: a push
                        // (ds_permute); the other lanes push to a lane
                        // outside [t, t + cnt) whose value is not taken
                        const uint32_t to =
                            is_vh ? out.t + rank : (out.t + cnt) & 63u;
                        const uint32_t ta =
                            (lit & 0xFFFFu) | ((P - old) << 16);
                        const uint32_t tb = ln.mv | ((P - lit) << 16);
                        const uint32_t ra =
                            (uint32_t)__builtin_amdgcn_ds_permute(
                                (int)(to << 2), (int)ta);
                        const uint32_t rb =
                            (uint32_t)__builtin_amdgcn_ds_permute(
                                (int)(to << 2), (int)tb);
                        const bool got = lane - out.t < cnt;
                        out.a = got ? ra : out.a;
                        out.b = got ? rb : out.b;
                        out.t += cnt;
                    }
                }
            }
            if (!fast)
                rc = span_walk(st, hits, cbits, s_limit, ln, sink, touched,
                               at);
            TICK(4);
            // the table as the reference leaves it: one lane-ordered store
            {
                const bool t = (touched >> lane) & 1;
                if (active && (t || !cbit))
                    lds_mskor(taddr, 0xFFFFu << sh, (t ? P : old) << sh);
            }
            TICK(5);
            if (rc == kSpanLong) {
                const uint32_t pk = st.s, ck = rdlane(old, at);
                // (requesting the next 16 bytes of every 16-byte match under
                // the walk, so that a long match is measured to 32 bytes
                // without this round trip, was built and measured: 8 % SLOWER
                // on text and HTML alike - profiles/r5_span_ablations.txt)
                const uint32_t len =
                    16 + extend_match(msrc, n, ck + 16, pk + 16, lane);
                sink.token(pk - sink.emit, len, pk - ck);
                st.s = pk + len;
                st.chain = 1;
                st.q = 0;
                if (st.s >= s_limit)
                    break;
            } else if (rc == kSpanDone) {
                break;
            } else if (!st.chain && st.q >= kSpanRun) {
                run0 = st.s - st.q;
            }
            TICK(6);
        }
        // keep the register window covering s - 1 .. s + 126
        if constexpr (!kLds) {
            // (ONE slide per step and its load straight into wv4: a loop
            // here is unrolled by the compiler into register rotations that
            // wait for the load they have just issued - a memory round trip
            // per step, 770 of a step's 4 800 cycles in round 4's kernel.  A
            // step that went further - behind a long match - reloads)
            const uint32_t D = st.s - 1 - wbase;
            if (D >= 128) {
                wbase = st.s - 1;
                const uint32_t w0 = wbase + lane;
                wv0 = ld32u(msrc + (w0 < n4 ? w0 : n4));
                wv1 = ld32u(msrc + (w0 + 64 < n4 ? w0 + 64 : n4));
                wv2 = ld32u(msrc + (w0 + 128 < n4 ? w0 + 128 : n4));
                wv3 = ld32u(msrc + (w0 + 192 < n4 ? w0 + 192 : n4));
                wv4 = ld32u(msrc + (w0 + 256 < n4 ? w0 + 256 : n4));
            } else if (D >= 64) {
                wv0 = wv1;
                wv1 = wv2;
                wv2 = wv3;
                wv3 = wv4;
                wbase += 64;
                const uint32_t wp = wbase + 256 + lane;
                wv4 = ld32u(msrc + (wp < n4 ? wp : n4));
            }
        }
        TICK(7);
    }
    if (sink.emit < n) // done(): src/compress.rs:417-426
        out.record_nf(sink.emit, n - sink.emit, 0, 0);
    if constexpr (kTok) {
        out.finish();
    } else {
        if (out.t)
            out.flush();
    }
    uint32_t count = 0;
    if constexpr (kTok)
        count = out.commit(a, b);
    if (lane == 0) {
        a.blk_size[b] = out.d;
        if constexpr (kTok)
            a.ntok[b] = count;
    }
    PROF(
    TICK(8);
    if (lane == 0 && a.prof) {
        for (int i = 0; i < 9; i++)
            atomicAdd(&a.prof[i], (unsigned long long)pt[i]);
        for (int i = 13; i < 16; i++)
            atomicAdd(&a.prof[i], (unsigned long long)pt[i]);
        atomicAdd(&a.prof[10], (unsigned long long)n_batches);
        atomicAdd(&a.prof[11], (unsigned long long)n_copies);
        atomicAdd(&a.prof[12], 1ull);
    }
    )
}

// The window kernel as the match finder of the token path (k_scan_sizes +
// k_encode_tokens behind it, every block at its final position): what
// launch_compress runs instead of k_match_blocks where the lane kernel's
// tables buy nothing - batches that do not compress (every probe of a lane is
// an HBM transaction, here it is an LDS access: cfg5), and whenever the
// option says so.  Blocks [blk_lo, blk_hi) from the front of the ticket.
__global__ __launch_bounds__(kCompressWaves * 64) void k_match_spans(
    CompressArgs a)
{
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kCompressWaves][kMaxTable];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const lptr16 table = (lptr16)&tables[wave][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    // (the ticket through a helper that is not inlined: DESIGN section 5)
    uint32_t b = a.blk_lo + uni(next_front_ticket(a.ticket, lane));
    while (b < nblocks) {
        compress_one_block_span<false, true>(a, b, lane, table, tbase);
        b = a.blk_lo + uni(next_front_ticket(a.ticket, lane));
    }
}

// k_match_spans for blocks of at most 8 KiB: a table of 8 192 u16 is all the
// reference gives such a block (src/compress.rs:491-518), so ten wavefronts
// fit a CU where the 64 KiB kernel has room for five, and the window kernel
// is bound by the instructions and latencies of ONE wavefront per SIMD.  Only
// blocks of the launch's class (CompressArgs::cls_lo / cls_hi) are taken.
// 4 KiB of alice29.txt per stream, 1 GiB: 38 -> 70 GiB/s.  (Twenty wavefronts
// with 8 KiB tables for blocks of at most 4 KiB measured no better: from ten
// on the CU's one scalar unit is the limit, profiles/r5_small_blocks.txt.)
__global__ __launch_bounds__(kSmallTableWaves * 64) void k_match_spans_8k(
    CompressArgs a)
{
    constexpr uint32_t kEntries = 8192;
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kSmallTableWaves][kEntries];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const lptr16 table = (lptr16)&tables[wave][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    uint32_t b = a.blk_lo + uni(next_front_ticket(a.ticket, lane));
    while (b < nblocks) {
        compress_one_block_span<false, true>(a, b, lane, table, tbase,
                                             nullptr, kEntries);
        b = a.blk_lo + uni(next_front_ticket(a.ticket, lane));
    }
}

// What the batch compressed to, posted into pinned host memory for the NEXT
// batch's choice of match finder (launch_compress: a context whose data does
// not compress is better off with LDS tables).  Per slot: words 0..1 =
// compressed bytes of the blocks, 2..3 = input bytes of the batch, 4 = the
// batch's number (0 while the slot is being written).
__global__ __launch_bounds__(1024) void k_post_ratio(
    uint32_t *host_mapped, const uint64_t *blk_off, uint32_t blocks,
    const uint64_t *in_lens, uint32_t n_streams, uint32_t seq)
{
    __shared__ unsigned long long part[1024];
    unsigned long long sum = 0;
    for (uint32_t i = threadIdx.x; i < n_streams; i += 1024)
        sum += in_lens[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t w = 512; w; w >>= 1) {
        if (threadIdx.x < w)
            part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (two slots, taken in turn: the host reads the one whose number is
        // higher while the other is being written)
        uint32_t *slot = host_mapped + 8 * (seq & 1);
        const uint64_t total = blk_off[blocks];
        slot[4] = 0;
        __threadfence_system();
        slot[0] = (uint32_t)total;
        slot[1] = (uint32_t)(total >> 32);
        slot[2] = (uint32_t)part[0];
        slot[3] = (uint32_t)(part[0] >> 32);
        __threadfence_system();
        slot[4] = seq;
    }
}

// ---------------------------------------------------------------------
// The order of a window-kernel launch (round 6).  A block is one wavefront's
// from start to end, blocks differ in cost by what they hold (kppkn.gtb:
// 1 884 window steps a block, HTML 740, a PDF 100), and a launch of a few
// blocks per wavefront in ticket order ends when its last-started heavy block
// does: 256 MiB of the corpus round took 7.9 ms where the work divided by the
// wavefronts is 5 (list scheduling: makespan <= mean + the heaviest job).
// Longest-first needs the costs, and the only cheap predictor of a block's
// cost is a block of the same stream.  So: pass 1 = the FIRST block of every
// stream; whoever finishes a block posts its cycles per KiB for its stream.
// Pass 2 = the other blocks in stream order - but a wavefront that draws a
// block of a stream known to be light or middling (under 0.7 / 1.3 of the
// launch's running mean) puts it on a list instead and draws again: the heavy
// and the unknown run first, the middle list next, the light list last, and
// the launch ends with small jobs.  Every block is run exactly once; results
// do not depend on the order (blocks are independent:
// src/compress.rs:148,514-516).
//   sched[0] ticket  [1] blocks classified in pass 2  [2] blocks costed
//   [3] (unused)  [4..5] u64 sum of costs  [6] pushed M  [7] popped M
//   [8] pushed L  [9] popped L            then: cost[n_streams],
//   listM[slots], listL[slots] (kSchedEmpty = not written yet)
// ---------------------------------------------------------------------
namespace {
constexpr uint32_t kSchedEmpty = 0xFFFFFFFFu, kSchedHead = 16;
struct SpanSched {
    uint32_t *w;
    uint32_t n_streams, slots, nblocks;
    __device__ __forceinline__ uint32_t *cost() const { return w + kSchedHead; }
    __device__ __forceinline__ uint32_t *list(int which) const
    {
        return w + kSchedHead + n_streams + (which ? slots : 0);
    }
    // the stream of pass-2 entry j: slot_first[st] <= j < slot_first[st + 1]
    __device__ __forceinline__ uint32_t stream_of_slot(
        const uint32_t *slot_first, uint32_t j) const
    {
        uint32_t lo = 0, hi = n_streams;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (slot_first[mid] <= j)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    }
    // lane 0: the next block to run, or kSchedEmpty when the launch is over
    // (out of line, like next_ticket: an inlined ticket loop under `if (lane
    // == 0)` inside a persistent loop once compiled into a hang)
    __device__ __forceinline__ uint32_t next(const uint32_t *blk_first,
                                             const uint32_t *slot_first)
    {
        for (;;) {
            const uint32_t t = atomicAdd(&w[0], 1u);
            if (t >= n_streams + slots)
                break;
            if (t < n_streams) { // pass 1: the stream's first block
                const uint32_t b = blk_first[t];
                if (blk_first[t + 1] > b && b < nblocks)
                    return b;
                continue; // a stream without blocks
            }
            const uint32_t j = t - n_streams;
            const uint32_t st = stream_of_slot(slot_first, j);
            const uint32_t b = blk_first[st] + (j - slot_first[st]) + 1;
            uint32_t cls = 0; // 0 run now, 1 middle list, 2 light list
            if (b < nblocks) {
                const uint32_t c = __hip_atomic_load(
                    &cost()[st], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t done = __hip_atomic_load(
                    &w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c && done >= 64) {
                    const unsigned long long sum = __hip_atomic_load(
                        (unsigned long long *)&w[4], __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long mean = sum / done;
                    cls = 10ull * c < 7ull * mean    ? 2
                          : 10ull * c < 13ull * mean ? 1
                                                     : 0;
                }
                if (cls) {
                    const uint32_t i = atomicAdd(&w[cls == 1 ? 6 : 8], 1u);
                    __hip_atomic_store(&list(cls - 1)[i], b, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __threadfence();
            atomicAdd(&w[1], 1u); // this entry is classified
            if (b < nblocks && cls == 0)
                return b;
        }
        // the lists: middle first, then light; an entry may still be on its
        // way while pass 2 is being classified by others
        for (int which = 0; which < 2; which++) {
            for (;;) {
                const uint32_t i = atomicAdd(&w[which ? 9 : 7], 1u);
                if (i >= slots)
                    break;
                uint32_t b;
                for (;;) {
                    b = __hip_atomic_load(&list(which)[i], __ATOMIC_ACQUIRE,
                                          __HIP_MEMORY_SCOPE_AGENT);
                    if (b != kSchedEmpty)
                        break;
                    const uint32_t cl = __hip_atomic_load(
                        &w[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (cl >= slots &&
                        i >= __hip_atomic_load(&w[which ? 8 : 6],
                                               __ATOMIC_ACQUIRE,
                                               __HIP_MEMORY_SCOPE_AGENT))
                        break; // all of pass 2 is classified: none will come
                    __builtin_amdgcn_s_sleep(8);
                }
                if (b == kSchedEmpty)
                    break;
                return b;
            }
        }
        return kSchedEmpty;
    }
    // lane 0, behind a block of n bytes of stream st that took `cycles`
    __device__ __forceinline__ void post(uint32_t st, uint32_t n,
                                         unsigned long long cycles)
    {
        unsigned long long c = (cycles << 10) / (n ? n : 1);
        if (c > 0x7FFFFFFFull)
            c = 0x7FFFFFFFull;
        if (c == 0)
            c = 1;
        __hip_atomic_store(&cost()[st], (uint32_t)c, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd((unsigned long long *)&w[4], c);
        atomicAdd(&w[2], 1u);
    }
};
} // namespace

// (the whole wavefront calls, lane 0 draws - the shape of next_ticket, which
// is known to compile into what it says inside a persistent loop; the two
// arrays it reads, not the kernel's argument block: a reference to that makes
// the kernel keep a copy of it in scratch memory)
__device__ __noinline__ uint32_t span_sched_next(SpanSched sc,
                                                 const uint32_t *blk_first,
                                                 const uint32_t *slot_first,
                                                 uint32_t lane)
{
    uint32_t b = 0;
    if (lane == 0)
        b = sc.next(blk_first, slot_first);
    return uni(b);
}

__global__ __launch_bounds__(kCompressWaves * 64) void k_compress_spans(
    CompressArgs a)
{
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kCompressWaves][kMaxTable];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const lptr16 table = (lptr16)&tables[wave][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (a.sched) { // the order chosen as the launch goes (SpanSched)
        SpanSched sc;
        sc.w = a.sched;
        sc.n_streams = a.n_streams;
        sc.slots = a.slot_first[a.n_streams] < a.host_slots
                       ? a.slot_first[a.n_streams]
                       : a.host_slots;
        sc.nblocks = nblocks;
        for (;;) {
            const uint32_t b = uni(
                span_sched_next(sc, a.blk_first, a.slot_first, lane));
            if (b == kSchedEmpty)
                break;
            const unsigned long long t0 = __builtin_readcyclecounter();
            compress_one_block_span<false>(a, b, lane, table, tbase);
            if (lane == 0) {
                // (the block's stream and length once more: two loads and a
                // short search per block of ~4 M cycles)
                uint32_t lo = 0, hi = a.n_streams;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (a.blk_first[mid] <= b)
                        lo = mid;
                    else
                        hi = mid;
                }
                const uint64_t left =
                    a.in_lens[lo] - (uint64_t)(b - a.blk_first[lo]) * kMaxBlock;
                sc.post(lo, left < kMaxBlock ? (uint32_t)left : kMaxBlock,
                        __builtin_readcyclecounter() - t0);
            }
        }
        return;
    }
    uint32_t b = uni(next_ticket(a.ticket, lane, nblocks));
    while (b != 0xFFFFFFFFu) {
        if (lane == 0 && a.ntok)
            a.ntok[b] = 0xFFFFFFFFu; // encoded here, not by k_encode_tokens
        compress_one_block_span<false>(a, b, lane, table, tbase);
        b = uni(next_ticket(a.ticket, lane, nblocks));
    }
}

// The blocks of a token-path launch whose tokens found no page in the pool
// (CompressArgs::tok_pool), once more: the window kernel compresses them to
// where k_encode_tokens would have put them (its destination rules, with
// a.direct the final position - k_scan_sizes has run).  Launched behind every
// k_encode_tokens; a launch without spilled blocks costs its wavefronts one
// load.  h_stat (pinned host memory, may be null): what the next batch sizes
// its pool by - pages asked for, blocks spilled, blocks of the launch, seq.
__global__ __launch_bounds__(kCompressWaves * 64) void k_redo_spilled(
    CompressArgs a, uint32_t *h_stat, uint32_t seq)
{
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kCompressWaves][kMaxTable];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    const lptr16 table = (lptr16)&tables[wave][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t count = a.tok_ctl[1];
    if (count > a.blk_hi - a.blk_lo)
        count = a.blk_hi - a.blk_lo;
    if (h_stat && blockIdx.x == 0 && threadIdx.x == 0) {
        h_stat[0] = a.tok_ctl[0];
        h_stat[1] = count;
        h_stat[2] = a.blk_hi - a.blk_lo;
        __threadfence_system();
        h_stat[3] = seq;
    }
    if (count == 0)
        return;
    uint32_t i = uni(next_ticket(a.tok_ctl + 2, lane, count));
    while (i != 0xFFFFFFFFu) {
        const uint32_t b = uni(a.tok_ctl[kTokCtlList + i]);
        compress_one_block_span<false>(a, b, lane, table, tbase);
        i = uni(next_ticket(a.tok_ctl + 2, lane, count));
    }
}

// one block per CU, table and input block in LDS (the smallest batches)
__global__ __launch_bounds__(64) void k_compress_span_lds(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t table_mem[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint8_t block_mem[kMaxBlock + 32];
    const uint32_t lane = threadIdx.x;
    const lptr16 table = (lptr16)&table_mem[0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    uint32_t b = uni(next_ticket(a.ticket, lane, nblocks));
    while (b != 0xFFFFFFFFu) {
        if (lane == 0 && a.ntok)
            a.ntok[b] = 0xFFFFFFFFu;
        compress_one_block_span<true>(
            a, b, lane, table, tbase,
            (__attribute__((address_space(3))) uint8_t *)block_mem);
        b = uni(next_ticket(a.ticket, lane, nblocks));
    }
}

// ---------------------------------------------------------------------
// K1t: streams of fewer than 256 bytes, ONE PER LANE, input and table in LDS.
//
// A block of a couple of hundred bytes costs the lane kernel ~190 rounds of
// an HBM (or L2) round trip, and the wavefront kernels a whole wavefront for
// ~100 probes.  Its whole state is tiny, though: the input and the
// reference's smallest table (256 entries, src/compress.rs:491-518 - and a
// position fits a byte) are two columns of 64 dwords per lane, 32 KiB per
// wavefront, four wavefronts per CU.  Every read of the parse is then an LDS
// access of ~100 cycles and the lane runs the reference's loop as it is
// written (snapmi_tiny.hpp: the test suite runs the same text over byte
// arrays on the CPU, tests/test_tiny_lane_cpu.py).  The columns are dword-
// interleaved (byte k of lane l at ((k >> 2) * 64 + l) * 4 + (k & 3)), so
// lanes at the same position hit 64 different banks.  The output goes
// straight to the caller's buffer, element by element: it is never read
// back, and a third column (65 dwords per lane) meant three wavefronts per
// CU instead of four - 202 against 276 GiB/s on 200-byte streams
// (profiles/r3_tiny_output_direct.txt; SNAPMI_TINY_STAGE_OUT=1 builds that
// variant).
// k_plan_compress gives these streams no blocks (CompressArgs::small_limit),
// so the block kernels never see them.
// ---------------------------------------------------------------------
namespace {
#define SNAPMI_TINY_STAGE_OUT 0 // 1: an output column in LDS (experiment)
template <bool kStage> struct TinyColumns {
    typedef __attribute__((address_space(3))) uint32_t l_u32;
    typedef __attribute__((address_space(3))) uint8_t l_u8;
    l_u32 *in, *tb, *out; // this lane's columns: dword w at [w * 64]
    gptr gout;            // !kStage: the stream's place in the caller's buffer
    __device__ __forceinline__ static uint32_t at(uint32_t k)
    {
        return (k >> 2) * 256 + (k & 3); // byte offset inside the column
    }
    __device__ __forceinline__ uint32_t in8(uint32_t k) const
    {
        return ((const l_u8 *)in)[at(k)];
    }
    __device__ __forceinline__ uint32_t in32(uint32_t k) const
    {
        const l_u32 *w = in + (k >> 2) * 64; // k + 4 <= n <= 255: w[64] exists
        return __builtin_amdgcn_alignbyte(w[64], w[0], k & 3);
    }
    __device__ __forceinline__ uint32_t tab(uint32_t h) const
    {
        return ((const l_u8 *)tb)[at(h)];
    }
    __device__ __forceinline__ void tab_set(uint32_t h, uint32_t v)
    {
        ((l_u8 *)tb)[at(h)] = (uint8_t)v;
    }
    __device__ __forceinline__ void out8(uint32_t k, uint32_t v)
    {
        if (kStage)
            ((l_u8 *)out)[at(k)] = (uint8_t)v;
        else
            gout[k] = (uint8_t)v;
    }
    __device__ __forceinline__ void out32(uint32_t k, uint32_t v)
    {
        if (kStage)
            out[(k >> 2) * 64] = v;
        else
            st32u(gout + k, v);
    }
};
} // namespace

// stream i of the batch by this lane (k_compress_tiny: 64 consecutive streams
// per wavefront; k_seam_compress_tiny: one)
__device__ __forceinline__ void compress_tiny_stream(const CompressArgs &a,
                                                     const uint64_t i)
{
    constexpr bool kStage = SNAPMI_TINY_STAGE_OUT != 0;
    constexpr uint32_t kW = kTinyCompress / 4;         // dwords per column
    constexpr uint32_t kWo = kStage ? (kTinyOutMax + 3) / 4 : 1; // 65
    __shared__ uint32_t tin[kW * 64];
    __shared__ uint32_t ttab[kW * 64];
    __shared__ uint32_t tout[kWo * 64];
    const uint32_t lane = threadIdx.x;
    if (i >= a.n_streams)
        return;
    const uint64_t len = a.in_lens[i];
    if (len == 0 || len >= kTinyCompress || len >= a.small_limit)
        return; // not this kernel's
    if (a.out_caps && a.out_caps[i] < max_compress_len_u64(len))
        return; // BufferTooSmall, reported by k_plan_compress
    const uint32_t n = (uint32_t)len;
    typedef TinyColumns<kStage> Columns;
    Columns m;
    m.in = (typename Columns::l_u32 *)tin + lane;
    m.tb = (typename Columns::l_u32 *)ttab + lane;
    m.out = (typename Columns::l_u32 *)tout + lane;
    m.gout = (gptr)a.out_ptrs[i];
    gcptr src = (gcptr)a.in_ptrs[i];
    {
        // whole dwords; the last partial one bytewise (reads stay inside
        // the stream)
        uint32_t k = 0;
        for (; k + 4 <= n; k += 4)
            m.in[(k >> 2) * 64] = ld32u(src + k);
        uint32_t last = 0;
        for (uint32_t j = 0; k + j < n; j++)
            last |= (uint32_t)src[k + j] << (8 * j);
        m.in[(k >> 2) * 64] = last; // (k >> 2 <= 63)
    }
    if (n >= kMinNonLiteral)
        for (uint32_t w = 0; w < kW; w++) // fresh table: src/compress.rs:506
            m.tb[w * 64] = 0;
    const uint32_t d = tiny_compress(m, n);
    gptr dst = (gptr)a.out_ptrs[i];
    if (kStage) {
        uint32_t k = 0;
        for (; k + 4 <= d; k += 4)
            st32u(dst + k, m.out[(k >> 2) * 64]);
        const uint32_t last = m.out[(k >> 2) * 64];
        for (uint32_t j = 0; k + j < d; j++)
            dst[k + j] = (uint8_t)(last >> (8 * j));
    }
    a.out_lens[i] = d;
}

__global__ __launch_bounds__(64) void k_compress_tiny(CompressArgs a)
{
    compress_tiny_stream(a, (uint64_t)blockIdx.x * 64 + threadIdx.x);
}

// The libsnappy seam's single-launch path (snapmi_api.hip, seam_tiny): ONE
// stream of 1..255 bytes whose input and output lie in pinned host memory the
// device reaches over the link - no copy commands, no plan kernel, the
// stream's description in the kernel's arguments (a descriptor in host
// memory is a round trip over the link per dependent load), and the host
// learns of the end from *done (it polls it; a system-scope fence puts the
// output and the length in front of it).  All 64 lanes bring the input in -
// one load instruction, one round trip - and clear the table; lane 0 runs
// the reference's loop (snapmi_tiny.hpp) and stores the elements as they come.
__global__ __launch_bounds__(64) void k_seam_compress_tiny(
    const uint8_t *in, uint32_t n, uint8_t *out, unsigned long long *out_len,
    uint32_t *done, uint32_t seq)
{
    constexpr bool kStage = SNAPMI_TINY_STAGE_OUT != 0;
    constexpr uint32_t kW = kTinyCompress / 4; // dwords per column (64)
    constexpr uint32_t kWo = kStage ? (kTinyOutMax + 3) / 4 : 1;
    __shared__ uint32_t tin[kW * 64];
    __shared__ uint32_t ttab[kW * 64];
    __shared__ uint32_t tout[kWo * 64];
    const uint32_t lane = threadIdx.x;
    gcptr src = (gcptr)in;
    {
        // dword `lane` of the input into lane 0's column (reads stay inside
        // the stream: the last partial dword bytewise)
        const uint32_t k = 4 * lane;
        uint32_t v = 0;
        if (k + 4 <= n) {
            v = ld32u(src + k);
        } else {
            for (uint32_t j = 0; k + j < n; j++)
                v |= (uint32_t)src[k + j] << (8 * j);
        }
        tin[lane * 64] = v;
        ttab[lane * 64] = 0; // fresh table: src/compress.rs:506
    }
    __syncthreads();
    if (lane != 0)
        return;
    typedef TinyColumns<kStage> Columns;
    Columns m;
    m.in = (typename Columns::l_u32 *)tin;
    m.tb = (typename Columns::l_u32 *)ttab;
    m.out = (typename Columns::l_u32 *)tout;
    m.gout = (gptr)out;
    const uint32_t d = tiny_compress(m, n);
    *out_len = d;
    __threadfence_system();
    __hip_atomic_store(done, seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------
// K1s: streams of 256 .. 1023 (2047) bytes, a FEW per wavefront, input and
// table in LDS.
//
// The same idea one size class up.  A block of 1 KiB costs the lane kernel
// as many HBM table accesses per byte as a 64 KiB block (more: every block
// starts with an empty table), 24-28 ms per GiB whatever the size; its state,
// though, is 1 KiB of input + the reference's table for that length (1 024
// entries of u16) = 3 KiB, so 48 such streams fit a CU's LDS.  They are kept
// as contiguous per-lane regions, kL lanes of a wavefront each running the
// reference's loop on its own stream (snapmi_tiny.hpp) after all 64 lanes
// have brought the inputs in (16 bytes per lane and access).  The output goes
// straight to the caller's buffer, lane by lane: an output column in LDS
// costs a quarter of the streams per CU, and those are what the rate is made
// of (measured, profiles/r3_small_stream_variants.txt: 93.6 -> 116 GiB/s at
// 400 bytes, 43.5 -> 56.0 at 1 000; half the lanes per wavefront and twice
// the wavefronts, with or without the column: slower - the rounds are bound
// by instruction issue as much as by latency, and a wavefront issues for all
// of its lanes at once).  Classes by the stream's length: [256, 512) eight
// lanes per wavefront, [512, 1024) four, twelve wavefronts (12 KiB each) per
// CU.  [1024, 2048) with two lanes exists but is off by default: 26 GiB/s
// against the lane kernel's 35 (option small_stream_kernel = 2).
// A wavefront looks at 64 consecutive streams of the batch and works through
// the ones of its class kL at a time (no list, no ticket).
// ---------------------------------------------------------------------
namespace {
struct SmallRegion {
    typedef __attribute__((address_space(3))) uint8_t l_u8;
    typedef __attribute__((address_space(3), may_alias)) uint16_t l_u16s;
    l_u8 *in;
    l_u16s *tb;
    gptr out; // the stream's place in the caller's buffer
    __device__ __forceinline__ uint32_t in8(uint32_t k) const { return in[k]; }
    __device__ __forceinline__ uint32_t in32(uint32_t k) const
    {
        // (LDS takes unaligned dwords: tests/hw/lds_unaligned.hip)
        return ld32u((lcptr)(in + k));
    }
    __device__ __forceinline__ uint32_t tab(uint32_t h) const { return tb[h]; }
    __device__ __forceinline__ void tab_set(uint32_t h, uint32_t v)
    {
        tb[h] = (uint16_t)v;
    }
    __device__ __forceinline__ void out8(uint32_t k, uint32_t v)
    {
        out[k] = (uint8_t)v;
    }
    __device__ __forceinline__ void out32(uint32_t k, uint32_t v)
    {
        st32u(out + k, v);
    }
};

template <uint32_t kCap, uint32_t kL>
__device__ __forceinline__ void compress_small(const CompressArgs &a)
{
    // per-lane region: input, table (kCap entries)
    constexpr uint32_t kRegion = kCap + 2 * kCap;
    constexpr uint32_t kLo = kCap == 2 * kTinyCompress ? kTinyCompress
                                                       : kCap / 2;
    __shared__ __attribute__((aligned(16))) uint8_t mem[kL * kRegion];
    typedef SmallRegion::l_u8 l_u8;
    typedef __attribute__((address_space(3), may_alias)) u32x4 l_u32x4;
    l_u8 *const base = (l_u8 *)mem;
    const uint32_t lane = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * 64 + lane;
    uint64_t len = i < a.n_streams ? a.in_lens[i] : 0;
    if (len >= a.small_limit ||
        (a.out_caps && len && a.out_caps[i] < max_compress_len_u64(len)))
        len = 0; // the block kernels' / reported by k_plan_compress
    uint64_t M = __ballot(len >= kLo && len < kCap);
    while (M) {
        // lane j < kL takes the j-th stream of the class that is left
        uint32_t mine = 64; // position of this lane's stream in the wavefront
        {
            uint64_t m = M;
            for (uint32_t j = 0; j < kL && m; j++) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                if (lane == j)
                    mine = b;
            }
            M = m;
        }
        // bring the inputs in and clear the tables, all lanes at it
        uint32_t n = 0;
        for (uint32_t j = 0; j < kL; j++) {
            const uint32_t b = rdlane(mine, j);
            if (b == 64)
                break;
            const uint64_t st = (uint64_t)blockIdx.x * 64 + b;
            const uint32_t nj = uni((uint32_t)a.in_lens[st]);
            if (lane == j)
                n = nj;
            gcptr src = (gcptr)uni64((uint64_t)a.in_ptrs[st]);
            l_u8 *const r = base + j * kRegion;
            for (uint32_t k = lane * 16; k + 16 <= nj; k += 1024)
                *(l_u32x4 *)(r + k) = ld128g(src + k);
            if (lane < (nj & 15))
                r[(nj & ~15u) + lane] = src[(nj & ~15u) + lane];
            // (the table this length needs: 512 .. kCap entries of 2 bytes)
            uint32_t tbytes = 1024;
            while (tbytes < 2 * nj)
                tbytes *= 2;
            for (uint32_t k = lane * 16; k < tbytes; k += 1024)
                *(l_u32x4 *)(r + kCap + k) = (u32x4){0, 0, 0, 0};
        }
        // (one wavefront: its LDS accesses complete in order; the barriers
        // keep the compiler from moving them across)
        __syncthreads();
        if (mine != 64) {
            const uint64_t st = (uint64_t)blockIdx.x * 64 + mine;
            SmallRegion m;
            m.in = base + lane * kRegion;
            m.tb = (SmallRegion::l_u16s *)(m.in + kCap);
            m.out = (gptr)a.out_ptrs[st];
            a.out_lens[st] = tiny_compress(m, n);
        }
        __syncthreads(); // (the next round's inputs overwrite the regions)
    }
}
} // namespace
__global__ __launch_bounds__(64) void k_compress_small512(CompressArgs a)
{
    compress_small<512, 8>(a);
}
__global__ __launch_bounds__(64) void k_compress_small1k(CompressArgs a)
{
    compress_small<1024, 4>(a);
}
__global__ __launch_bounds__(64) void k_compress_small2k(CompressArgs a)
{
    compress_small<2048, 2>(a);
}

// ---------------------------------------------------------------------
// K1b: lane-per-block match finder.
//
// The wavefront-per-block kernel above is bound by a dependent chain per
// emitted copy and by "five tables per CU".  This kernel turns the problem
// around: every LANE runs the reference's sequential parse (src/compress.rs:
// 195-317) on its own block, with its own hash table in HBM, and a wave
// advances 64 independent chains per instruction.
//   * table entries are 16 bytes: the 12 input bytes at the position, the
//     position and an epoch.  The epoch means a table is never zeroed (an
//     entry of another epoch reads as position 0, the reference's fresh
//     table; epochs persist in the context across launches).  The stored
//     bytes decide hit or miss and measure matches shorter than 12 bytes
//     without touching the candidate's cache line;
//   * the loop is organised in ROUNDS of one memory round trip: every lane
//     issues the same three loads (16 B, 16 B, 4 B) whatever it is doing -
//     probing the table, inserting after a copy, or extending a long match
//     16 bytes further - then all lanes wait once and update their state
//     with ALU work only.  A lane never waits for another lane's match
//     extension, so the time of a wave is (rounds of its slowest lane) x
//     (one round trip), not the sum of every lane's serial loops;
//   * a lane that finishes its block takes the next one from the ticket
//     counter, so lanes stay busy whatever the mix of block costs;
//   * the lane only records tokens (literal length, copy length, offset);
//     k_encode_tokens below turns them into Snappy elements, one wavefront
//     per block, 64 tokens at a time.
// ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld32p(gcptr p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld64p(gcptr p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
namespace {
// kSpec: for launches of few blocks (snapmi_api.hip: at most 24 576), where
// what is waited for is the latency of a block's dependent rounds and the
// memory system is idle.  A probe's round also fetches the table entry of
// the probe that FOLLOWS IF IT MISSES - its position is known at the start
// of the round (src/compress.rs:207-216: the skip schedule; after a copy,
// s + 1) and its bytes are in the registers the window was read into - and a
// miss goes on to that probe in the same round.  Same table states in the
// same order (what this round wrote is forwarded), same tokens
// (tests/model_match_lane.py: the round order as a model, checked on the
// CPU); 27-39 % fewer rounds on text, each a fifth more expensive: 10-15 %
// off 2 048 .. 16 384 blocks.  A launch of 32 768 blocks and more is at the
// random-access rate of HBM even with one block per lane, and the extra
// table reads buy nothing there: the plain kernel.
// one wavefront's LDS: the lanes' input windows (two 128-byte lines of a
// lane's block) and token buffers (16 tokens = one 128-byte line: every store
// of a lane is its own DRAM transaction, and those are what bounds the kernel)
constexpr uint32_t kLaneRingWords = 64 * 64;
constexpr uint32_t kLaneTokWords = 64 * 16; // (in 8-byte words: 32 tokens a lane)
// lane: 0..63 of this wavefront; g: its lane id among all lanes of the launch
// (its table, its epoch)
template <bool kSpec>
__device__ __forceinline__ void match_blocks(
    const CompressArgs &a, const uint32_t lane, const uint32_t g,
    __attribute__((address_space(3))) uint32_t *const ring,
    __attribute__((address_space(3))) unsigned long long *const tokbuf)
{
    typedef __attribute__((address_space(3))) uint32_t l_tok;
    l_tok *const tbuf = (l_tok *)(tokbuf + lane * 16); // 32 tokens = 128 B
    typedef __attribute__((address_space(3))) uint32_t l_u32;
    typedef __attribute__((address_space(3))) u32x4 l_u32x4;
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    l_u32 *const win = ring + lane * 64;

    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u64x2 g_entry;
    // 16-byte entries: x = bytes 0..7 at the position, y = bytes 8..11 |
    // position << 32 | epoch << 48
    const uint32_t slot =
        a.lane_chunks ? (g % a.lane_chunks) * a.lane_per_chunk + g / a.lane_chunks
                      : g;
    g_entry *const tab =
        (g_entry *)a.lane_tables + (uint64_t)slot * a.lane_stride;
    unsigned long long epoch = a.lane_epochs[g]; // 16 bits used
    unsigned long long first8 = 0; // bytes 0..11 of the block (empty entry)
    uint32_t first4b = 0;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi; // this launch covers blocks [blk_lo, blk_hi)

    // what the lane does in the next round
    enum : uint32_t {
        kProbe = 0,  // look up position s (the skip loop, compress.rs:207-245)
        kChain = 1,  // after a copy: insert s-1, look up s (compress.rs:290-312)
        kExtend = 2, // compare 16 more bytes of an open match (compress.rs:378-412)
    };
    // per-lane block state
    bool have = false, out_of_work = false;
    uint32_t b = 0, n = 0, s_limit = 0, shift = 0;
    uint32_t mode = kProbe;
    uint32_t s = 0, s_next = 0, skip = 0, next_emit = 0;
    uint32_t mpos = 0, mcand = 0, p = 0, c = 0; // the open match
    uint32_t ntok = 0;
    uint32_t csize = 0; // encoded bytes of the block's tokens so far
    // input window: the bytes [hi - 256, hi) of the 128-byte aligned view of
    // the block (src_al = src - mis) are in `win`, at their offset mod 256
    uint32_t mis = 0, hi = 0;
    gcptr src = nullptr, src_al = nullptr;
    typedef __attribute__((address_space(1))) uint32_t g_tok;
    // the pages the block's tokens and exceptions are going to (the dump
    // page once the pool has run out: CompressArgs::tok_pool)
    uint32_t tpage = a.tok_pool_pages, epage = a.tok_pool_pages;
    bool spilled = false;
    uint32_t nexc = 0;
    // the lane's page in hand, and what is left of the wavefront's run
    // (uniform): see tok_page_ask
    uint32_t spare = kNoPage, run_next = 0, run_end = 0;
    // where token idx goes; the first token of a page takes the page
    auto tok_at = [&](uint32_t idx) -> g_tok * {
        if (idx % kTokPage == 0)
            tpage = tok_page_take(a, b, idx / kTokPage, spare, spilled);
        return (g_tok *)a.tok_pool + (uint64_t)tpage * kTokPage +
               idx % kTokPage;
    };
    auto exc_put = [&](unsigned long long v) {
        if (nexc % kExcPage == 0)
            epage = tok_page_take(a, b, kTokPagesPerBlock + nexc / kExcPage,
                                  spare, spilled);
        if (!spilled)
            ((g_u64 *)((g_tok *)a.tok_pool + (uint64_t)epage * kTokPage))
                [nexc % kExcPage] = v;
        nexc++;
    };

    for (;;) {
        // token pages for the lanes that used theirs in the last round (the
        // loop's top is convergent: the run's bounds stay uniform)
        {
            const uint64_t M_pg = __ballot(spare == kNoPage);
            if (M_pg) {
                const uint32_t cnt = (uint32_t)__builtin_popcountll(M_pg);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                    (uint32_t)(M_pg >> 32),
                    __builtin_amdgcn_mbcnt_lo((uint32_t)M_pg, 0));
                const uint32_t left = run_end - run_next;
                if (left < cnt) {
                    // what is left of the run goes to the first of them, a
                    // new run serves the others (lanes that work on blocks
                    // of one kind fill their pages in step: a remainder
                    // dropped here was a tenth of the pages of a batch
                    // sorted by file)
                    const uint32_t leader = (uint32_t)__builtin_ctzll(M_pg);
                    const uint32_t more = cnt - left;
                    const uint32_t take = more > kTokRun ? more : kTokRun;
                    uint32_t base = 0;
                    if (lane == leader)
                        base = atomicAdd(&a.tok_ctl[0], take);
                    base = rdlane(base, leader);
                    if (spare == kNoPage)
                        spare = rank < left ? run_next + rank
                                            : base + (rank - left);
                    run_next = base + more;
                    run_end = base + take;
                } else {
                    if (spare == kNoPage)
                        spare = run_next + rank;
                    run_next += cnt;
                }
            }
        }
        // Tickets are taken for the whole wavefront at once: the lanes that
        // need a block count themselves (ballot) and ONE of them adds the
        // count to the device-wide counter.  (One atomic per lane on one
        // address is 3.7 ns each, one after the other in the L2: 0.97 ms for
        // a launch of 262 144 tiny blocks, all of its duration.)
        const bool need = !have && !out_of_work;
        const uint64_t M_need = __ballot(need);
        unsigned long long tbase = 0;
        if (M_need) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(M_need);
            if (lane == leader)
                tbase = atomicAdd((unsigned long long *)a.ticket,
                                  (unsigned long long)__builtin_popcountll(
                                      M_need));
            tbase = ((unsigned long long)rdlane((uint32_t)(tbase >> 32),
                                                leader)
                     << 32) |
                    rdlane((uint32_t)tbase, leader);
        }
        if (need) {
            const unsigned long long old =
                tbase + __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(M_need >> 32),
                            __builtin_amdgcn_mbcnt_lo((uint32_t)M_need, 0));
            b = a.blk_lo + (uint32_t)old;
            if ((uint64_t)b + (uint32_t)(old >> 32) >= nblocks) {
                out_of_work = true;
            } else {
                // stream lookup: blk_first[st] <= b < blk_first[st + 1]
                uint32_t lo = 0, hi_st = a.n_streams;
                // (a batch of one-block streams - pages, frame chunks: block b IS
                // stream b, and two loads say so instead of log2(n) dependent ones)
                if (b < a.n_streams && a.blk_first[b] == b && a.blk_first[b + 1] > b) {
                    lo = b;
                    hi_st = b + 1;
                }
                while (hi_st - lo > 1) {
                    const uint32_t mid = (lo + hi_st) >> 1;
                    if (a.blk_first[mid] <= b)
                        lo = mid;
                    else
                        hi_st = mid;
                }
                const uint32_t k = b - a.blk_first[lo];
                const uint64_t total = a.in_lens[lo];
                const uint64_t boff = (uint64_t)k * kMaxBlock;
                src = (gcptr)a.in_ptrs[lo] + boff;
                n = total - boff < kMaxBlock ? (uint32_t)(total - boff)
                                             : kMaxBlock;
                spilled = false;
                ntok = 0;
                nexc = 0;
                csize = 0;
                next_emit = 0;
                have = true;
                if (n <= a.cls_lo || n > a.cls_hi) {
                    have = false; // another launch's block
                } else if (n < kMinNonLiteral) { // src/compress.rs:140-146
                    g_tok *const at = tok_at(0);
                    if (!spilled)
                        *at = tok_pack(n, 0, 0);
                    a.ntok[b] = spilled ? kTokSpilled : 1u;
                    a.blk_size[b] = token_bytes(n, 0, 0);
                    have = false;
                } else {
                    // fresh table = new epoch (src/compress.rs:491-518)
                    first8 = ld64p(src);
                    first4b = ld32p(src + 8);
                    epoch = (epoch + 1) & 0xFFFFu;
                    if (epoch == 0) { // wrapped: really clear this lane's table
                        for (uint32_t i = 0; i < kMaxTable; i++)
                            tab[i] = (u64x2){0, 0};
                        epoch = 1;
                    }
                    shift = 32 - 8;
                    uint32_t tsize = 256;
                    while (tsize < kMaxTable && tsize < n) {
                        shift--;
                        tsize *= 2;
                    }
                    s_limit = n - kInputMargin; // >= 2
                    // first trip of the skip loop: s = 1, s_next = 2
                    s = 1;
                    s_next = 2;
                    skip = 33;
                    mode = kProbe;
                    mis = (uint32_t)(uintptr_t)src & 127u;
                    src_al = src - mis;
                    hi = 0;
                }
            }
        }
        if (__ballot(have) == 0) {
            if (__ballot(!out_of_work) == 0)
                break;
            continue;
        }
        if (!have)
            continue;

        // ---- one round ----------------------------------------------------
        // The 17 bytes at pos-1 come from the window (pos = s, or p while a
        // match is open).  The next line is fetched while at least 64 bytes
        // of look-ahead remain, in the same round as the table access; only
        // a lane that jumped past its window (long match, large skip) has
        // to sit a round out.
        const uint32_t y0 = (mode == kExtend ? p : s) - 1 + mis;
        if (y0 >= hi)
            hi = y0 & ~127u; // jumped past the frontier: restart there
        const bool stall = y0 + 17 > hi;
        // (never fetch past the block's own last line: the next line may be
        // the first one behind the caller's allocation)
        const bool fill = y0 + 81 > hi && hi < ((n + mis + 127u) & ~127u);
        uint32_t q0, q1, q2, q3, r0, r1, r2, r3;
        {
            const uint32_t w0 = y0 >> 2, sh = y0 & 3;
            const uint32_t d0 = win[w0 & 63], d1 = win[(w0 + 1) & 63],
                           d2 = win[(w0 + 2) & 63], d3 = win[(w0 + 3) & 63],
                           d4 = win[(w0 + 4) & 63];
            q0 = __builtin_amdgcn_alignbyte(d1, d0, sh); // bytes at pos-1
            q1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            q2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            q3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            const uint32_t q4 = d4 >> (8 * sh);
            r0 = __builtin_amdgcn_alignbyte(q1, q0, 1); // bytes at pos
            r1 = __builtin_amdgcn_alignbyte(q2, q1, 1);
            r2 = __builtin_amdgcn_alignbyte(q3, q2, 1);
            r3 = __builtin_amdgcn_alignbyte(q4, q3, 1);
        }
        const uint32_t hprev = hash32(q0, shift), hcur = hash32(r0, shift);
        // the one random access of the round: table entry or candidate bytes
        gcptr pa = mode == kExtend ? src + c : (gcptr)(tab + hcur);
        pa = stall ? src_al + hi : pa;
        B16 A = ld128u(pa);
        // kSpec: the probe behind this one, delta bytes on (r3 ends at
        // pos + 15: its 12 bytes are in the registers for delta <= 3)
        uint32_t t0 = 0, t1 = 0, t2 = 0, h2 = 0;
        bool spec = false;
        B16 A2 = {{0, 0, 0, 0}};
        if (kSpec) {
            const uint32_t delta = mode == kChain ? 1 : s_next - s;
            spec = mode <= kChain && delta <= 3 && !stall;
            t0 = __builtin_amdgcn_alignbyte(r1, r0, delta);
            t1 = __builtin_amdgcn_alignbyte(r2, r1, delta);
            t2 = __builtin_amdgcn_alignbyte(r3, r2, delta);
            h2 = hash32(t0, shift);
            if (spec)
                A2 = ld128u((gcptr)(tab + h2));
        }
        // (the load is issued HERE, in front of the window's line: a compiler
        // barrier, or it may be sunk behind the fill's wait - two latencies
        // in every round that fills)
        asm volatile("" ::: "memory");
        if (fill) {
            const g_u32x4 *lp = (const g_u32x4 *)(src_al + hi);
            const u32x4 f0 = lp[0], f1 = lp[1], f2 = lp[2], f3 = lp[3],
                        f4 = lp[4], f5 = lp[5], f6 = lp[6], f7 = lp[7];
            l_u32x4 *dst = (l_u32x4 *)(win + ((hi >> 2) & 63));
            dst[0] = f0; dst[1] = f1; dst[2] = f2; dst[3] = f3;
            dst[4] = f4; dst[5] = f5; dst[6] = f6; dst[7] = f7;
            hi += 128;
        }
        // one wait for the whole round: A is materialised before the branch
        // that uses it (the fill's wait, when there was one, covered it)
        asm volatile("" : "+v"(A.w[0]));
        if (kSpec)
            asm volatile("" : "+v"(A2.w[0]));
        if (stall)
            continue;

        bool matched = false;  // a copy ends in this round at mend
        bool advance = false;  // next trip of the skip loop
        bool tail = false;     // open match within 16 bytes of the block end
        bool finished = false;
        uint32_t mend = 0;
        if (mode <= kChain) {
            if (mode == kChain) {
                // src/compress.rs:290-297: insert s-1 first; the lookup of s
                // must see it when both hash to the same slot
                const unsigned long long b8 =
                    ((unsigned long long)q1 << 32) | q0;
                const unsigned long long by =
                    (epoch << 48) | ((unsigned long long)(s - 1) << 32) | q2;
                tab[hprev] = (u64x2){b8, by};
                if (hprev == hcur) {
                    A.w[0] = q0;
                    A.w[1] = q1;
                    A.w[2] = q2;
                    A.w[3] = (uint32_t)(by >> 32);
                }
            }
            const unsigned long long p8 = ((unsigned long long)r1 << 32) | r0;
            const uint32_t p4 = r2;
            const bool live = (A.w[3] >> 16) == (uint32_t)epoch;
            const uint32_t cand = live ? A.w[3] & 0xFFFFu : 0;
            const unsigned long long c8 =
                live ? ((unsigned long long)A.w[1] << 32) | A.w[0] : first8;
            const uint32_t c4 = live ? A.w[2] : first4b;
            tab[hcur] = (u64x2){
                p8, (epoch << 48) | ((unsigned long long)s << 32) | p4};
            if ((uint32_t)c8 == r0) {
                // the entry holds the candidate's first 12 bytes, so most
                // matches are measured without touching the candidate's line
                const unsigned long long d8 = c8 ^ p8;
                const uint32_t d4 = c4 ^ p4;
                const uint32_t m =
                    d8 ? (uint32_t)__builtin_ctzll(d8) >> 3
                       : 8 + (d4 ? (uint32_t)__builtin_ctz(d4) >> 3 : 4);
                mpos = s;
                mcand = cand;
                if (m < 12) {
                    matched = true;
                    mend = s + m;
                } else {
                    p = s + 12;
                    c = cand + 12;
                    mode = kExtend;
                    tail = p + 16 > n;
                }
            } else {
                if (mode == kChain) { // src/compress.rs:310-312
                    s_next = s + 1;
                    skip = 32;
                }
                advance = true;
            }
        } else { // kExtend
            B16 Bv;
            Bv.w[0] = r0; Bv.w[1] = r1; Bv.w[2] = r2; Bv.w[3] = r3;
            const uint32_t m = common16(A, Bv);
            if (m < 16) {
                matched = true;
                mend = p + m;
            } else {
                p += 16;
                c += 16;
                tail = p + 16 > n;
            }
        }
        if (kSpec && advance && spec) {
            // the first probe missed: its advance (src/compress.rs:207-216),
            // then the probe at the new s with the entry fetched for it
            const uint32_t s_old = s;
            const bool was_chain = mode == kChain;
            s = s_next;
            const uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            mode = kProbe;
            advance = false;
            if (s_next > s_limit) {
                finished = true;
            } else {
                // what this round wrote is newer than what it read
                if (h2 == hcur) {
                    A2.w[0] = r0;
                    A2.w[1] = r1;
                    A2.w[2] = r2;
                    A2.w[3] = ((uint32_t)epoch << 16) | s_old;
                } else if (was_chain && h2 == hprev) {
                    A2.w[0] = q0;
                    A2.w[1] = q1;
                    A2.w[2] = q2;
                    A2.w[3] = ((uint32_t)epoch << 16) | (s_old - 1);
                }
                const unsigned long long p8 =
                    ((unsigned long long)t1 << 32) | t0;
                const uint32_t p4 = t2;
                const bool live = (A2.w[3] >> 16) == (uint32_t)epoch;
                const uint32_t cand = live ? A2.w[3] & 0xFFFFu : 0;
                const unsigned long long c8 =
                    live ? ((unsigned long long)A2.w[1] << 32) | A2.w[0]
                         : first8;
                const uint32_t c4 = live ? A2.w[2] : first4b;
                tab[h2] = (u64x2){
                    p8, (epoch << 48) | ((unsigned long long)s << 32) | p4};
                if ((uint32_t)c8 == t0) {
                    const unsigned long long d8 = c8 ^ p8;
                    const uint32_t d4 = c4 ^ p4;
                    const uint32_t m =
                        d8 ? (uint32_t)__builtin_ctzll(d8) >> 3
                           : 8 + (d4 ? (uint32_t)__builtin_ctz(d4) >> 3 : 4);
                    mpos = s;
                    mcand = cand;
                    if (m < 12) {
                        matched = true;
                        mend = s + m;
                    } else {
                        p = s + 12;
                        c = cand + 12;
                        mode = kExtend;
                        tail = p + 16 > n;
                    }
                } else {
                    advance = true;
                }
            }
        }
        if (tail) { // rare: finish the match bytewise up to the block end
            while (p < n && src[p] == src[c]) {
                p++;
                c++;
            }
            matched = true;
            mend = p;
        }
        bool flush = false;
        if (matched) {
            // token: literal next_emit..mpos, copy (mpos - mcand, mend - mpos)
            {
                // token_bytes(), short form for the usual token (literal of
                // at most 60 bytes, copy of at most 64): the round loop is
                // not free of VALU cost
                const uint32_t tl = mpos - next_emit, tc = mend - mpos,
                               to = mpos - mcand;
                tbuf[ntok & 31] = tok_pack(tl, tc, to);
                ntok++;
                if (tl > 60 || tc > 64) {
                    csize += token_bytes(tl, tc, to);
                    if (!tok_fits(tl, tc)) // rare: its numbers in full
                        exc_put(tok_pack64(tl, tc, to));
                } else {
                    csize += tl + 3 + (tl != 0) - (tc <= 11 && to <= 2047);
                }
            }
            flush = (ntok & 31) == 0;
            s = mend;
            next_emit = mend;
            mode = kChain;
            if (s >= s_limit) // src/compress.rs:275-277
                finished = true;
        }
        if (advance) { // src/compress.rs:207-216
            s = s_next;
            const uint32_t step = skip >> 5;
            s_next = s + step;
            skip += step;
            mode = kProbe;
            if (s_next > s_limit)
                finished = true;
        }
        if (finished)
            flush = flush || (ntok & 31) != 0;
        if (flush) { // the token group that holds token ntok-1
            g_u32x4 *to = (g_u32x4 *)tok_at((ntok - 1) & ~31u);
            const l_u32x4 *from = (const l_u32x4 *)tbuf;
            const u32x4 t0 = from[0], t1 = from[1], t2 = from[2], t3 = from[3],
                        t4 = from[4], t5 = from[5], t6 = from[6], t7 = from[7];
            // (a spilled block stores nothing: tens of thousands of lanes
            // writing one dump page wait for one another in its L2 channel -
            // 3.9 s for a launch of 64 000 spilled blocks)
            if (!spilled) {
                to[0] = t0; to[1] = t1; to[2] = t2; to[3] = t3;
                to[4] = t4; to[5] = t5; to[6] = t6; to[7] = t7;
            }
        }
        if (finished) { // done(): src/compress.rs:417-426
            if (next_emit < n) {
                const uint32_t tl = n - next_emit;
                // (behind a flushed group in that group's page; the first
                // token of a group of its own may open a page)
                g_tok *const at =
                    ntok % 32 ? (g_tok *)a.tok_pool +
                                    (uint64_t)tpage * kTokPage + ntok % kTokPage
                              : tok_at(ntok);
                if (!spilled)
                    *at = tok_pack(tl, 0, 0);
                ntok++;
                if (!tok_fits(tl, 0))
                    exc_put(tok_pack64(tl, 0, 0));
                csize += token_bytes(tl, 0, 0);
            }
            a.ntok[b] = spilled ? kTokSpilled : ntok;
            a.blk_size[b] = csize;
            have = false;
        }
    }
    a.lane_epochs[g] = epoch;
}
} // namespace

__global__ __launch_bounds__(64) void k_match_blocks(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t ring[kLaneRingWords];
    __shared__ __attribute__((aligned(16)))
    unsigned long long tokbuf[kLaneTokWords];
    match_blocks<false>(
        a, threadIdx.x, blockIdx.x * 64 + threadIdx.x,
        (__attribute__((address_space(3))) uint32_t *)ring,
        (__attribute__((address_space(3))) unsigned long long *)tokbuf);
}
__global__ __launch_bounds__(64) void k_match_blocks_spec(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t ring[kLaneRingWords];
    __shared__ __attribute__((aligned(16)))
    unsigned long long tokbuf[kLaneTokWords];
    match_blocks<true>(
        a, threadIdx.x, blockIdx.x * 64 + threadIdx.x,
        (__attribute__((address_space(3))) uint32_t *)ring,
        (__attribute__((address_space(3))) unsigned long long *)tokbuf);
}

// Both match finders on every CU (round 5): three wavefronts of the lane
// kernel - as many as reach the DRAM-transaction ceiling it is bound by
// (tests/hw/sweep.sh: 3 waves per CU 123 ms, 6 waves 122) - and two of the
// window kernel, which is bound by the instructions and latencies of a lone
// wavefront and touches HBM for little more than its input, in one persistent
// workgroup per CU: 3 x 24 KiB of lane windows and token lines + 2 x 32 KiB of
// tables.  One two-ended ticket: the lanes take blocks from the front of the
// segment, the windows from its back, until they meet; both write tokens for
// k_encode_tokens.  (Round 4 split the CUs between the two kernels and gained
// nothing; here the lanes keep every CU.)
__global__ __launch_bounds__(kBothWaves * 64) void k_match_both(CompressArgs a)
{
    __shared__ __attribute__((aligned(16)))
    uint32_t ring[kBothLaneWaves][kLaneRingWords];
    __shared__ __attribute__((aligned(16)))
    unsigned long long tokbuf[kBothLaneWaves][kLaneTokWords];
    __shared__ __attribute__((aligned(16)))
    uint16_t tables[kBothWaves - kBothLaneWaves][kMaxTable];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    if (wave < kBothLaneWaves) {
        match_blocks<false>(
            a, lane, (blockIdx.x * kBothLaneWaves + wave) * 64 + lane,
            (__attribute__((address_space(3))) uint32_t *)&ring[wave][0],
            (__attribute__((address_space(3))) unsigned long long *)
                &tokbuf[wave][0]);
        return;
    }
    const lptr16 table = (lptr16)&tables[wave - kBothLaneWaves][0];
    const uint32_t tbase = (uint32_t)(uintptr_t)table;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    for (;;) {
        const uint32_t b =
            uni(next_back_ticket(a.ticket, lane, a.blk_lo, nblocks));
        if (b == 0xFFFFFFFFu)
            break;
        compress_one_block_span<false, true>(a, b, lane, table, tbase);
    }
}

// ---------------------------------------------------------------------
// K1c: tokens -> Snappy elements, one wavefront per block, 64 tokens per
// pass (TokenSink::flush: reference emit_literal / emit_copy).
// ---------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_encode_tokens(CompressArgs a)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t b = a.blk_lo + blockIdx.x;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    if (b >= nblocks)
        return;
    if (a.ntok[b] >= kTokSpilled)
        return; // encoded by k_compress_blocks / left to k_redo_spilled
    uint32_t lo = 0, hi = a.n_streams;
    // (a batch of one-block streams - pages, frame chunks: block b IS
    // stream b, and two loads say so instead of log2(n) dependent ones)
    if (b < a.n_streams && a.blk_first[b] == b && a.blk_first[b + 1] > b) {
        lo = b;
        hi = b + 1;
    }
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.blk_first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t st = lo;
    const uint32_t k = b - a.blk_first[st];
    const uint64_t total = a.in_lens[st];
    const uint64_t boff = (uint64_t)k * kMaxBlock;
    gcptr src = (gcptr)a.in_ptrs[st] + boff;
    const uint32_t n =
        total - boff < kMaxBlock ? (uint32_t)(total - boff) : kMaxBlock;
    gptr dst;
    if (k == 0) {
        dst = (gptr)a.out_ptrs[st];
        if (lane == 0) { // varint(total): src/compress.rs:128
            uint64_t v = total;
            uint32_t i = 0;
            while (v >= 0x80) {
                dst[i++] = (uint8_t)v | 0x80;
                v >>= 7;
            }
            dst[i] = (uint8_t)v;
        }
        dst += varint_len(total);
    } else if (a.direct) {
        // the sizes of the blocks in front are known (k_match_blocks added
        // them up, k_scan_sizes ran over this segment): final position
        const uint32_t first = a.blk_first[st];
        const uint64_t nb = (total + kMaxBlock - 1) / kMaxBlock;
        if (first + nb > a.host_blocks)
            return; // rejected by k_plan_compress (E_ARGUMENT)
        dst = (gptr)a.out_ptrs[st] + varint_len(total) +
              (a.blk_off[b] - a.blk_off[first]);
    } else {
        const uint32_t slot = a.slot_first[st] + k - 1;
        if (slot >= a.host_slots)
            return;
        dst = (gptr)a.scratch + (uint64_t)slot * kSlotBytes;
    }
    TokenSink out;
    out.init(src, n, dst, lane);
    typedef __attribute__((address_space(1))) uint32_t g_tok;
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    // the block's page table (CompressArgs::tok_pool): a pass of 64 tokens
    // lies in one page
    // (lane i holds entry i: no load stands between a pass and its tokens)
    const uint32_t ptab =
        lane < kPageTabStride
            ? a.tok_pages[(uint64_t)(b - a.tok_base) * kPageTabStride + lane]
            : 0;
    const g_tok *const pool = (const g_tok *)a.tok_pool;
    const uint32_t count = a.ntok[b];
    uint32_t pos_base = 0, exc_base = 0;
    // (the next pass's tokens are loaded before this pass is encoded: one
    // memory latency less on the chain of every pass)
    uint32_t tnext =
        lane < count ? pool[(uint64_t)rdlane(ptab, 0) * kTokPage + lane] : 0;
    for (uint32_t t0 = 0; t0 < count; t0 += kWave) {
        const uint32_t m = count - t0 < kWave ? count - t0 : kWave;
        uint32_t L = 0, C = 0, O = 0;
        const uint32_t t = tnext;
        if (t0 + kWave < count) {
            const uint32_t pg = rdlane(ptab, (t0 + kWave) / kTokPage);
            if (t0 + kWave + lane < count)
                tnext = pool[(uint64_t)pg * kTokPage +
                             (t0 + kWave) % kTokPage + lane];
        }
        const uint32_t field = (t >> 10) & 63u;
        if (lane < m) {
            L = t & 1023u;
            C = field == kTokLiteral ? 0u : field + 4u;
            O = t >> 16;
        }
        // a token that did not fit four bytes: its numbers are in the
        // block's exception list, in the order of such tokens (rare: a pass
        // without one pays a ballot)
        const bool ex = lane < m && field == kTokException;
        const uint64_t E = __builtin_amdgcn_ballot_w64(ex);
        if (E) {
            // (every lane takes part in the exchange: its page table entry
            // may be the one an exception's lane needs)
            const uint32_t ei =
                exc_base + __builtin_amdgcn_mbcnt_hi(
                               (uint32_t)(E >> 32),
                               __builtin_amdgcn_mbcnt_lo((uint32_t)E, 0));
            const uint32_t eslot = ei / kExcPage < kExcPagesPerBlock
                                       ? ei / kExcPage
                                       : kExcPagesPerBlock - 1;
            const uint32_t epg = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)((kTokPagesPerBlock + eslot) << 2), (int)ptab);
            if (ex) {
                const unsigned long long f =
                    ((const g_u64 *)(pool + (uint64_t)epg * kTokPage))
                        [ei % kExcPage];
                L = (uint32_t)f & 0x1FFFFu;
                C = (uint32_t)(f >> 17) & 0xFFFFu;
                O = (uint32_t)(f >> 33) & 0xFFFFu;
            }
            exc_base += (uint32_t)__builtin_popcountll(E);
        }
        const uint32_t span = L + C;
        const uint32_t incl = wave_inclusive_scan(span);
        const uint32_t P = pos_base + incl - span; // literal start
        pos_base += rdlane(incl, kWave - 1);
        out.a = (L & 0xFFFFu) | (O << 16);
        out.b = C | (P << 16);
        out.t = m;
        out.flush();
    }
    if (lane == 0 && !a.direct)
        a.blk_size[b] = out.d;
}

// ---------------------------------------------------------------------
// Plan: per-stream validation + block table.  One workgroup of 1024 threads
// walks the streams 1024 at a time with a workgroup-wide exclusive scan.
// Reference checks: src/compress.rs:104-125.
// ---------------------------------------------------------------------
__device__ __forceinline__ uint2 wg_scan2(uint32_t x, uint32_t y,
                                          uint2 *wave_tot, uint2 *total)
{
    // inclusive scan inside the wave
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t sx = x, sy = y;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t tx = __shfl_up(sx, o), ty = __shfl_up(sy, o);
        if (lane >= o) {
            sx += tx;
            sy += ty;
        }
    }
    if (lane == 63)
        wave_tot[w] = make_uint2(sx, sy);
    __syncthreads();
    uint32_t bx = 0, by = 0, allx = 0, ally = 0;
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t i = 0; i < nw; i++) {
        const uint2 t = wave_tot[i];
        if (i < w) {
            bx += t.x;
            by += t.y;
        }
        allx += t.x;
        ally += t.y;
    }
    __syncthreads();
    *total = make_uint2(allx, ally);
    return make_uint2(bx + sx - x, by + sy - y); // exclusive
}

namespace {
// blocks of stream i, or 0 for a stream that is rejected, empty or tiny
// (reference src/compress.rs:104-125)
__device__ __forceinline__ uint32_t plan_blocks(const CompressArgs &a,
                                                uint32_t i, bool report)
{
    const uint64_t len = a.in_lens[i];
    const uint64_t need = max_compress_len_u64(len);
    if (report)
        a.out_lens[i] = 0;
    if (need == 0) {
        if (report)
            set_error(a.errs, i, SNAPMI_TOO_BIG, len, kMaxInput, 0);
        return 0;
    }
    if (a.out_caps && a.out_caps[i] < need) {
        if (report)
            set_error(a.errs, i, SNAPMI_BUFFER_TOO_SMALL, a.out_caps[i], need,
                      0);
        return 0;
    }
    if (len == 0) { // src/compress.rs:120-125
        if (report) {
            ((gptr)a.out_ptrs[i])[0] = 0;
            a.out_lens[i] = 1;
            set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
        }
        return 0;
    }
    if (report)
        set_error(a.errs, i, SNAPMI_OK, 0, 0, 0);
    if (len < a.small_limit)
        return 0; // k_compress_tiny's / _small's: no blocks, they write out_lens
    return (uint32_t)((len + kMaxBlock - 1) / kMaxBlock);
}
} // namespace

__global__ __launch_bounds__(1024) void k_plan_compress(CompressArgs a)
{
    __shared__ uint2 wave_tot[16];
    uint32_t carry_b = 0, carry_s = 0;
    for (uint32_t base = 0; base < a.n_streams; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t nb = i < a.n_streams ? plan_blocks(a, i, true) : 0;
        uint2 tot;
        const uint2 ex = wg_scan2(nb, nb ? nb - 1 : 0, wave_tot, &tot);
        uint32_t fb = carry_b + ex.x, fs = carry_s + ex.y;
        if (i < a.n_streams) {
            // The launch was sized from the host's copy of the lengths; a
            // stream that does not fit in it is rejected, never overrun.
            if (nb && ((uint64_t)fb + nb > a.host_blocks ||
                       (uint64_t)fs + (nb - 1) > a.host_slots))
                set_error(a.errs, i, SNAPMI_E_ARGUMENT, a.in_lens[i], 0, 0);
            a.blk_first[i] = fb;
            a.slot_first[i] = fs;
        }
        carry_b += tot.x;
        carry_s += tot.y;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.blk_first[a.n_streams] = carry_b;
        a.slot_first[a.n_streams] = carry_s;
    }
}

// Exclusive scan of blk_size (u32) into blk_off (u64) over the blocks
// [blk_lo, blk_hi), one workgroup; blk_off[blk_lo] carries over from the
// segment in front.
__global__ __launch_bounds__(1024) void k_scan_sizes(CompressArgs a)
{
    __shared__ uint64_t wave_tot[16];
    const uint32_t *blk_size = a.blk_size;
    uint64_t *blk_off = a.blk_off;
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t carry = a.blk_lo ? blk_off[a.blk_lo] : 0;
    __syncthreads(); // (blk_off[blk_lo] is rewritten below)
    for (uint32_t base = a.blk_lo; base < nblocks; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t x = i < nblocks ? blk_size[i] : 0;
        uint64_t sx = x;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(sx, o);
            if (lane >= o)
                sx += t;
        }
        if (lane == 63)
            wave_tot[w] = sx;
        __syncthreads();
        uint64_t before = 0, all = 0;
        for (uint32_t j = 0; j < (blockDim.x >> 6); j++) {
            const uint64_t t = wave_tot[j];
            if (j < w)
                before += t;
            all += t;
        }
        __syncthreads();
        if (i < nblocks)
            blk_off[i] = carry + before + sx - x;
        carry += all;
    }
    if (threadIdx.x == 0 && nblocks >= a.blk_lo)
        blk_off[nblocks] = carry;
}

// ---------------------------------------------------------------------
// The same two scans on many workgroups, for batches of many streams /
// blocks (10.7 M streams of 200 bytes: the one-workgroup kernels above took
// 27.8 ms and 41 x 0.46 ms of a 104 ms pass).  Three launches each: (a) every
// workgroup scans its 1024 items and leaves its total, (b) one workgroup
// scans the totals, (c) every workgroup adds its offset.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_plan_compress_a(CompressArgs a)
{
    __shared__ uint2 wave_tot[16];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nb = i < a.n_streams ? plan_blocks(a, i, true) : 0;
    uint2 tot;
    const uint2 ex = wg_scan2(nb, nb ? nb - 1 : 0, wave_tot, &tot);
    if (i < a.n_streams) {
        a.blk_first[i] = ex.x;  // within this workgroup, for now
        a.slot_first[i] = ex.y;
    }
    if (threadIdx.x == 0)
        a.plan_part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_plan_compress_b(CompressArgs a,
                                                          uint32_t nparts)
{
    __shared__ uint2 wave_tot[16];
    uint32_t carry_b = 0, carry_s = 0;
    for (uint32_t base = 0; base < nparts; base += blockDim.x) {
        const uint32_t j = base + threadIdx.x;
        const uint2 v = j < nparts ? a.plan_part[j] : make_uint2(0, 0);
        uint2 tot;
        const uint2 ex = wg_scan2(v.x, v.y, wave_tot, &tot);
        if (j < nparts)
            a.plan_part[j] = make_uint2(carry_b + ex.x, carry_s + ex.y);
        carry_b += tot.x;
        carry_s += tot.y;
    }
    if (threadIdx.x == 0) {
        a.blk_first[a.n_streams] = carry_b;
        a.slot_first[a.n_streams] = carry_s;
    }
}

__global__ __launch_bounds__(1024) void k_plan_compress_c(CompressArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_streams)
        return;
    const uint2 off = a.plan_part[blockIdx.x];
    const uint32_t fb = a.blk_first[i] + off.x, fs = a.slot_first[i] + off.y;
    const uint32_t nb = plan_blocks(a, i, false);
    // The launch was sized from the host's copy of the lengths; a stream
    // that does not fit in it is rejected, never overrun.
    if (nb && ((uint64_t)fb + nb > a.host_blocks ||
               (uint64_t)fs + (nb - 1) > a.host_slots))
        set_error(a.errs, i, SNAPMI_E_ARGUMENT, a.in_lens[i], 0, 0);
    a.blk_first[i] = fb;
    a.slot_first[i] = fs;
}

namespace {
__device__ __forceinline__ uint32_t scan_nblocks(const CompressArgs &a)
{
    uint32_t nblocks = a.blk_first[a.n_streams];
    if (nblocks > a.host_blocks)
        nblocks = a.host_blocks;
    if (nblocks > a.blk_hi)
        nblocks = a.blk_hi;
    return nblocks;
}
// exclusive scan of x over the workgroup (u64), total in *all
__device__ __forceinline__ uint64_t wg_scan64(uint64_t x, uint64_t *wave_tot,
                                              uint64_t *all)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t sx = x;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(sx, o);
        if (lane >= o)
            sx += t;
    }
    if (lane == 63)
        wave_tot[w] = sx;
    __syncthreads();
    uint64_t before = 0, tot = 0;
    for (uint32_t j = 0; j < (blockDim.x >> 6); j++) {
        const uint64_t t = wave_tot[j];
        if (j < w)
            before += t;
        tot += t;
    }
    __syncthreads();
    *all = tot;
    return before + sx - x;
}
} // namespace

// part64[0, nparts): workgroup totals, then offsets; part64[nparts]: the carry
// from the segment in front (blk_off[blk_lo] before it is rewritten)
__global__ __launch_bounds__(1024) void k_scan_sizes_a(CompressArgs a,
                                                       uint32_t nparts)
{
    __shared__ uint64_t wave_tot[16];
    const uint32_t nblocks = scan_nblocks(a);
    const uint32_t i = a.blk_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        a.scan_part[nparts] = a.blk_lo ? a.blk_off[a.blk_lo] : 0;
    __syncthreads();
    const uint64_t x = i < nblocks ? a.blk_size[i] : 0;
    uint64_t all;
    const uint64_t ex = wg_scan64(x, wave_tot, &all);
    if (i < nblocks)
        a.blk_off[i] = ex;
    if (threadIdx.x == 0)
        a.scan_part[blockIdx.x] = all;
}

__global__ __launch_bounds__(1024) void k_scan_sizes_b(CompressArgs a,
                                                       uint32_t nparts)
{
    __shared__ uint64_t wave_tot[16];
    const uint32_t nblocks = scan_nblocks(a);
    uint64_t carry = a.scan_part[nparts];
    for (uint32_t base = 0; base < nparts; base += blockDim.x) {
        const uint32_t j = base + threadIdx.x;
        const uint64_t v = j < nparts ? a.scan_part[j] : 0;
        uint64_t all;
        const uint64_t ex = wg_scan64(v, wave_tot, &all);
        if (j < nparts)
            a.scan_part[j] = carry + ex;
        carry += all;
    }
    if (threadIdx.x == 0 && nblocks >= a.blk_lo)
        a.blk_off[nblocks] = carry;
}

__global__ __launch_bounds__(1024) void k_scan_sizes_c(CompressArgs a)
{
    const uint32_t nblocks = scan_nblocks(a);
    const uint32_t i = a.blk_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblocks)
        a.blk_off[i] += a.scan_part[blockIdx.x];
}

// Direct encoding (k_encode_tokens wrote every block at its final position):
// what is left of k_compact is the compressed length of every stream.
__global__ __launch_bounds__(256) void k_stream_lens(CompressArgs a)
{
    const uint32_t st = blockIdx.x * 256 + threadIdx.x;
    if (st >= a.n_streams)
        return;
    const uint32_t first = a.blk_first[st];
    const uint32_t nb = a.blk_first[st + 1] - first;
    if (nb == 0 || (uint64_t)first + nb > a.host_blocks)
        return; // empty, in error or rejected: k_plan_compress wrote out_lens
    a.out_lens[st] = varint_len(a.in_lens[st]) +
                     (a.blk_off[first + nb] - a.blk_off[first]);
}

// ---------------------------------------------------------------------
// K3: move blocks 1.. of every stream from their scratch slot to
// out_ptrs[st] + varint + sum(size of earlier blocks); block 0's workgroup
// publishes the stream's compressed length.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(CompressArgs a)
{
    const uint32_t b = blockIdx.x;
    const uint32_t nblocks = a.blk_first[a.n_streams];
    if (b >= nblocks)
        return;
    uint32_t lo = 0, hi = a.n_streams;
    // (a batch of one-block streams - pages, frame chunks: block b IS
    // stream b, and two loads say so instead of log2(n) dependent ones)
    if (b < a.n_streams && a.blk_first[b] == b && a.blk_first[b + 1] > b) {
        lo = b;
        hi = b + 1;
    }
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.blk_first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t st = lo;
    const uint32_t first = a.blk_first[st];
    const uint32_t k = b - first;
    const uint64_t total = a.in_lens[st];
    const uint32_t vl = varint_len(total);
    const uint32_t nb = (uint32_t)((total + kMaxBlock - 1) / kMaxBlock);
    if ((uint64_t)first + nb > a.host_blocks ||
        (uint64_t)a.slot_first[st] + (nb - 1) > a.host_slots)
        return; // rejected by k_plan_compress (E_ARGUMENT), out_lens stays 0
    if (k == 0) {
        if (threadIdx.x == 0)
            a.out_lens[st] = vl + (a.blk_off[first + nb] - a.blk_off[first]);
        return;
    }
    gcptr from = (gcptr)a.scratch +
                 (uint64_t)(a.slot_first[st] + k - 1) * kSlotBytes;
    gptr to = (gptr)a.out_ptrs[st] + vl + (a.blk_off[b] - a.blk_off[first]);
    const uint32_t size = a.blk_size[b];
    // align the destination to 16 bytes, then 16-byte stores fed by
    // unaligned 16-byte loads (the slot is 16-aligned, `to` is arbitrary).
    uint32_t head = (uint32_t)((16 - ((uintptr_t)to & 15)) & 15);
    if (head > size)
        head = size;
    if (threadIdx.x < head)
        to[threadIdx.x] = from[threadIdx.x];
    const uint32_t body = (size - head) & ~15u;
    for (uint32_t i = 16 * threadIdx.x; i < body; i += 16 * blockDim.x) {
        u32x4 v;
        __builtin_memcpy(&v, from + head + i, 16);
        *(__attribute__((address_space(1))) u32x4 *)(to + head + i) = v;
    }
    const uint32_t tail = size - head - body;
    if (threadIdx.x < tail)
        to[head + body + threadIdx.x] = from[head + body + threadIdx.x];
}

} // namespace snapmi
