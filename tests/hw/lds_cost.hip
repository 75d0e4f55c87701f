// Hardware probe: what do the decoder's LDS instructions cost, in LDS-pipeline
// cycles per CU, when all 32 waves of a CU issue them (8 waves per SIMD, one
// 5 KiB workgroup per wave - the decoder's launch shape)?  The window loop of
// k_decompress_streams3 is budgeted in three currencies (VALU issue, SALU
// issue, LDS cycles); this measures the third.
//
//   ./lds_cost            prints, per pattern, ns per instruction per CU and
//                         the cycles that is at the measured clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) uint8_t l_u8;
constexpr int kIter = 4096;
struct B16 { uint64_t lo, hi; };
__device__ __forceinline__ void st128(l_u8 *p, const B16 &v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ void st64(l_u8 *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ B16 ld128(const l_u8 *p) { B16 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint64_t ld64(const l_u8 *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t ld32(const l_u8 *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }


enum Pat {
    P_BPERMUTE = 0,   // ds_bpermute_b32, select = lane + small
    P_PERMUTE,        // ds_permute_b32 (push), same selects
    P_W128_S5,        // ds_write_b128, lane stride 5 bytes (text-like pieces), 64 lanes
    P_W128_S5_45,     // same, 45 lanes active
    P_W128_S5_20,     // same, 20 lanes active
    P_W128_S16,       // ds_write_b128, stride 16, unaligned base
    P_W128_AL,        // ds_write_b128 aligned stride 16
    P_W64_S5,         // ds_write_b64 stride 5
    P_R64x2_RND,      // two ds_read_b64 at random unaligned ring addresses, 64 lanes
    P_R64x2_RND_30,   // 30 lanes
    P_R64x2_RND_6,    // 6 lanes
    P_R128_RND,       // ds_read_b128 unaligned random, 64 lanes
    P_R128_RND_30,
    P_R32_AL,         // ds_read_b32, aligned, consecutive (the flush)
    P_RU8_TAB,        // ds_read_u8 gather from a 256-byte table
    P_R32_TAB,        // ds_read_b32 gather from a 1 KiB table (256 dwords)
    P_R128_SEQ5,      // ds_read_b128, stride 5 (near copies read like they write)
    P_VALU,           // control: 8 dependent VALU adds (no LDS)
    P_BPERM8,         // 8 dependent ds_bpermute_b32 per iteration
    P_BPERM8_32,      // the same with 32 lanes active
    P_W8_SCATTER,     // ds_write_b8 to random bytes
    P_W128_DW,        // ds_write_b128, dword-aligned (stride 20)
    P_R128_DW,        // ds_read_b128, random dword-aligned
    P_SALU8,          // 8 dependent SALU adds
    P_VALU8_SALU8,    // 8 VALU + 8 SALU, independent chains
    P_READLANE8,      // 8 v_readlane + s_add chains
    P_COUNT
};
static const char *kNames[P_COUNT] = {
    "ds_bpermute_b32", "ds_permute_b32", "ds_write_b128 stride5 64 lanes",
    "ds_write_b128 stride5 45 lanes", "ds_write_b128 stride5 20 lanes",
    "ds_write_b128 stride16 unaligned", "ds_write_b128 stride16 aligned",
    "ds_write_b64 stride5", "2x ds_read_b64 random 64 lanes",
    "2x ds_read_b64 random 30 lanes", "2x ds_read_b64 random 6 lanes",
    "ds_read_b128 random 64 lanes", "ds_read_b128 random 30 lanes",
    "ds_read_b32 aligned consecutive", "ds_read_u8 table gather",
    "ds_read_b32 table gather", "ds_read_b128 stride5", "8 VALU adds (control)",
    "8x ds_bpermute_b32 (dependent)", "8x ds_bpermute_b32, 32 lanes", "ds_write_b8 scatter",
    "ds_write_b128 dword-aligned stride20", "ds_read_b128 random dword-aligned",
    "8 SALU adds", "8 VALU + 8 SALU", "8x (v_readlane + s_add)"};

template <int PAT> __global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t mem[4096 + 1024];
    l_u8 *m = (l_u8 *)mem;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 5120 / 4; i += 64)
        ((uint32_t *)mem)[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t rnd = (lane * 2654435761u + seed * 40503u + blockIdx.x * 977u);
    uint32_t acc = lane;
    B16 v = {rnd, lane};
    const uint32_t nact = PAT == P_W128_S5_45 ? 45
                          : PAT == P_W128_S5_20 ? 20
                          : (PAT == P_R64x2_RND_30 || PAT == P_R128_RND_30) ? 30
                          : PAT == P_R64x2_RND_6 ? 6 : 64;
    // active lanes spread over the wave like element starts are
    const bool act = ((lane * nact) >> 6) != (((lane + 1) * nact) >> 6) || nact == 64;
    for (int it = 0; it < kIter; it++) {
        rnd = rnd * 1664525u + 1013904223u;
        const uint32_t base = (it * 37) & 2047;
        if (PAT == P_BPERMUTE) {
            const int sel = (int)(((lane + 2 + ((rnd >> 20) & 3)) & 63) << 2);
            acc += (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)acc);
        } else if (PAT == P_PERMUTE) {
            const int sel = (int)(((lane + 2 + ((rnd >> 20) & 3)) & 63) << 2);
            acc += (uint32_t)__builtin_amdgcn_ds_permute(sel, (int)acc);
        } else if (PAT == P_W128_S5 || PAT == P_W128_S5_45 || PAT == P_W128_S5_20) {
            if (act) {
                v.lo += acc;
                st128(m + base + 5 * lane + 1, v);
            }
        } else if (PAT == P_W128_S16) {
            v.lo += acc;
            st128(m + base + 16 * lane + 3, v);
        } else if (PAT == P_W128_AL) {
            v.lo += acc;
            st128(m + (base & ~15u) + 16 * lane, v);
        } else if (PAT == P_W64_S5) {
            v.lo += acc;
            st64(m + base + 5 * lane + 1, v.lo);
        } else if (PAT == P_R64x2_RND || PAT == P_R64x2_RND_30 || PAT == P_R64x2_RND_6) {
            if (act) {
                const uint32_t p = (rnd >> 12) & 4095;
                const uint64_t a = ld64(m + p), b = ld64(m + ((p + 8) & 4095));
                acc += (uint32_t)a ^ (uint32_t)(b >> 32);
            }
        } else if (PAT == P_R128_RND || PAT == P_R128_RND_30) {
            if (act) {
                const uint32_t p = (rnd >> 12) & 4095;
                const B16 x = ld128(m + p);
                acc += (uint32_t)x.lo ^ (uint32_t)(x.hi >> 32);
            }
        } else if (PAT == P_R32_AL) {
            const uint32_t x = ld32(m + ((base & ~255u) + 4 * lane));
            acc += x;
        } else if (PAT == P_RU8_TAB) {
            acc += m[4096 + ((rnd >> 13) & 255)];
        } else if (PAT == P_R32_TAB) {
            const uint32_t x = ld32(m + 4096 + 4 * ((rnd >> 13) & 255));
            acc += x;
        } else if (PAT == P_R128_SEQ5) {
            const B16 x = ld128(m + base + 5 * lane + 1);
            acc += (uint32_t)x.lo ^ (uint32_t)(x.hi >> 32);
        } else if (PAT == P_BPERM8 || PAT == P_BPERM8_32) {
            if (PAT == P_BPERM8 || lane < 32) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(acc << 2), (int)acc) + 3;
            }
        } else if (PAT == P_W8_SCATTER) {
            m[(rnd >> 12) & 4095] = (uint8_t)acc;
        } else if (PAT == P_W128_DW) {
            v.lo += acc;
            st128(m + (base & ~3u) + 20 * lane, v);
        } else if (PAT == P_R128_DW) {
            const uint32_t p = (rnd >> 12) & 4092;
            const B16 x = ld128(m + p);
            acc += (uint32_t)x.lo ^ (uint32_t)(x.hi >> 32);
        } else if (PAT == P_SALU8) {
            uint32_t sa = __builtin_amdgcn_readfirstlane(acc);
#pragma unroll
            for (int j = 0; j < 8; j++)
                asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_u32 %0, %0, 7" : "+s"(sa) : : "scc");
            acc += sa;
        } else if (PAT == P_VALU8_SALU8) {
            uint32_t sa = __builtin_amdgcn_readfirstlane(acc);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_u32 %0, %0, 7" : "+s"(sa) : : "scc");
                asm volatile("v_mul_lo_u32 %0, %0, 3\n\tv_add_u32 %0, %0, %1" : "+v"(acc) : "v"(rnd));
            }
            acc += sa;
        } else if (PAT == P_READLANE8) {
            uint32_t sa = __builtin_amdgcn_readfirstlane(acc) & 63;
#pragma unroll
            for (int j = 0; j < 8; j++)
                sa = (sa + (uint32_t)__builtin_amdgcn_readlane((int)rnd, (int)sa)) & 63;
            acc += sa;
        } else {
            for (int j = 0; j < 8; j++)
                acc = acc * 3 + rnd;
        }
        asm volatile("" : "+v"(acc));
    }
    if (acc == 0x12345678u)
        out[0] = acc + (uint32_t)v.lo;
}

template <int PAT> static float run(uint32_t *d, int wgs)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(64), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(wgs), dim3(64), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int PAT> static void all(uint32_t *d, int cus, float *res)
{
    res[PAT] = run<PAT>(d, cus * 32);
    if constexpr (PAT + 1 < P_COUNT)
        all<PAT + 1>(d, cus, res);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    uint32_t *d;
    hipMalloc(&d, 64);
    float res[P_COUNT];
    all<0>(d, cus, res);
    const float ctl = res[P_VALU];
    printf("%d CUs, clock %.2f GHz, 32 waves per CU, %d iterations per wave\n", cus, ghz, kIter);
    printf("(each iteration also runs ~6 VALU of address arithmetic; the control row shows 8 VALU)\n");
    for (int p = 0; p < P_COUNT; p++) {
        // one instruction instance per wave per iteration; 32 waves share the CU's LDS
        const double ns = res[p] * 1e6 / ((double)kIter * 32);
        printf("%-36s %8.3f ms  %7.2f ns/instr/CU  %6.1f cycles\n", kNames[p], res[p], ns, ns * ghz);
    }
    (void)ctl;
    return 0;
}
