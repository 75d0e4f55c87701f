// Hardware probe: do unaligned LDS accesses (ds_read_b128 / b64 / b32 / u16 and
// ds_write_b128 / b64 / b32 / b16 at arbitrary byte addresses) behave bytewise?  The
// element-major decoder (k_decompress_streams2) reads 8 bytes and writes
// 8 / 4 / 2 / 1 bytes at unaligned ring positions.  hipcc emits single DS
// instructions for align-1 accesses on gfx950 (unaligned access mode); this
// checks the hardware agrees, for every address modulo 16.  Prints PASS/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) uint8_t l_u8;
__global__ void probe(uint32_t *bad)
{
    __shared__ __attribute__((aligned(16))) uint8_t mem[4096];
    l_u8 *m = (l_u8 *)mem;
    const uint32_t lane = threadIdx.x;
    uint32_t fails = 0;
    for (uint32_t rep = 0; rep < 16; rep++) {
        for (uint32_t i = lane; i < 4096; i += 64)
            m[i] = (uint8_t)(i * 7 + rep);
        __syncthreads();
        // reads: lane reads 8 / 4 / 2 bytes at 37 * lane + rep (all residues)
        const uint32_t a = (37 * lane + rep) & 4087;
        uint64_t v8; uint32_t v4; uint16_t v2;
        __builtin_memcpy(&v8, m + a, 8);
        __builtin_memcpy(&v4, m + a + 1, 4);
        __builtin_memcpy(&v2, m + a + 3, 2);
        uint64_t w8 = 0; uint32_t w4 = 0; uint16_t w2 = 0;
        for (int k = 7; k >= 0; k--) w8 = (w8 << 8) | (uint8_t)((a + k) * 7 + rep);
        for (int k = 3; k >= 0; k--) w4 = (w4 << 8) | (uint8_t)((a + 1 + k) * 7 + rep);
        for (int k = 1; k >= 0; k--) w2 = (uint16_t)((w2 << 8) | (uint8_t)((a + 3 + k) * 7 + rep));
        fails += v8 != w8;
        fails += v4 != w4;
        fails += v2 != w2;
        // 16 bytes at once (hipcc merges two 8-byte copies into ds_read_b128 /
        // ds_write_b128 whatever the alignment)
        struct { uint64_t lo, hi; } v16;
        __builtin_memcpy(&v16, m + a + 5, 16);
        uint64_t e0 = 0, e1 = 0;
        for (int k = 7; k >= 0; k--) e0 = (e0 << 8) | (uint8_t)((a + 5 + k) * 7 + rep);
        for (int k = 7; k >= 0; k--) e1 = (e1 << 8) | (uint8_t)((a + 13 + k) * 7 + rep);
        fails += v16.lo != e0;
        fails += v16.hi != e1;
        __syncthreads();
        {   // 16-byte write at 61 * lane + rep + 20 (inside the lane's 61)
            const uint32_t b16 = 61 * lane + rep + 20;
            struct { uint64_t lo, hi; } x16 = {0x1122334455667788ull ^ lane,
                                               0x99AABBCCDDEEFF00ull + lane};
            __builtin_memcpy(m + b16, &x16, 16);
            __syncthreads();
            for (int k = 0; k < 8; k++) fails += m[b16 + k] != (uint8_t)(x16.lo >> (8 * k));
            for (int k = 0; k < 8; k++) fails += m[b16 + 8 + k] != (uint8_t)(x16.hi >> (8 * k));
            fails += m[b16 + 16] != (uint8_t)((b16 + 16) * 7 + rep);
            fails += m[b16 - 1] != (uint8_t)((b16 - 1) * 7 + rep);
            __syncthreads();
            for (int k = 0; k < 16; k++) m[b16 + k] = (uint8_t)((b16 + k) * 7 + rep);
        }
        __syncthreads();
        // writes: lane owns 16 bytes at 61 * lane + rep; writes 8 + 4 + 2 + 1
        const uint32_t b = 61 * lane + rep;
        uint64_t x8 = 0x0807060504030201ull * (lane + 1);
        uint32_t x4 = 0xA1B2C3D4u + lane;
        uint16_t x2 = (uint16_t)(0xE5F6 + lane);
        __builtin_memcpy(m + b, &x8, 8);
        __builtin_memcpy(m + b + 8, &x4, 4);
        __builtin_memcpy(m + b + 12, &x2, 2);
        m[b + 14] = (uint8_t)(0x77 + lane);
        __syncthreads();
        for (int k = 0; k < 8; k++) fails += m[b + k] != (uint8_t)(x8 >> (8 * k));
        for (int k = 0; k < 4; k++) fails += m[b + 8 + k] != (uint8_t)(x4 >> (8 * k));
        for (int k = 0; k < 2; k++) fails += m[b + 12 + k] != (uint8_t)(x2 >> (8 * k));
        fails += m[b + 14] != (uint8_t)(0x77 + lane);
        // the byte behind a lane's 15 bytes belongs to nobody: untouched
        fails += m[b + 15] != (uint8_t)((b + 15) * 7 + rep);
        __syncthreads();
    }
    atomicAdd(bad, fails);
}
int main()
{
    uint32_t *d, h = 1;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("lds_unaligned: %u mismatches\n%s\n", h, h == 0 ? "PASS unaligned DS access is bytewise" : "FAIL");
    return h != 0;
}
