// Hardware probe: when the lanes of ONE wave64 plain LDS store instruction
// write overlapping byte ranges, which lane's bytes stay?  If stores are
// applied in ascending lane order (as DS atomics are, lds_atomic_order.hip),
// a lane may write a whole 16 bytes for an element of fewer bytes: the lanes
// above it (= the following elements) overwrite the excess.  Checked for
// ds_write_b128 / b64 (16 bytes as one memcpy, and as two 8-byte halves) with
// lane strides 1..23 and pseudo-random increasing positions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef __attribute__((address_space(3))) uint8_t l_u8;
__global__ void probe(uint32_t *bad, uint32_t *detail)
{
    __shared__ __attribute__((aligned(16))) uint8_t mem[8192];
    l_u8 *m = (l_u8 *)mem;
    const uint32_t lane = threadIdx.x;
    uint32_t fails = 0;
    for (uint32_t mode = 0; mode < 3; mode++) {
        uint32_t mf = 0;
        for (uint32_t t = 0; t < 64; t++) {
            for (uint32_t i = lane; i < 8192; i += 64)
                m[i] = 0xEE;
            __syncthreads();
            // increasing positions: stride t (1..23) or random gaps 1..20
            uint32_t gap = t < 23 ? t + 1 : 1 + ((lane * 2654435761u + t * 40503u) >> 7) % 20;
            uint32_t pos = gap;
            for (uint32_t o = 1; o < 64; o <<= 1) { // inclusive scan of gaps
                uint32_t v = __shfl_up(pos, o);
                if (lane >= o) pos += v;
            }
            pos += 3 * t; // varying alignment
            struct { uint64_t lo, hi; } x;
            x.lo = 0x0101010101010101ull * (lane + 1);
            x.hi = 0x0101010101010101ull * (lane + 1);
            const uint32_t width = mode == 0 ? 16 : (mode == 1 ? 8 : 4);
            if (mode == 0) {
                __builtin_memcpy(m + pos, &x, 16);
            } else if (mode == 1) {
                __builtin_memcpy(m + pos, &x.lo, 8);
            } else {
                uint32_t v4 = (uint32_t)x.lo;
                __builtin_memcpy(m + pos, &v4, 4);
            }
            __syncthreads();
            // expected under ascending lane order: byte b belongs to the
            // highest lane whose range covers it
            for (uint32_t k = 0; k < width; k++) {
                const uint32_t b = pos + k;
                const uint32_t nextpos = __shfl_down(pos, 1);
                // this lane's byte survives iff no higher lane covers it;
                // higher lanes start at >= nextpos (positions increase)
                const bool survives = lane == 63 || b < nextpos;
                if (survives && m[b] != (uint8_t)(lane + 1))
                    mf++;
            }
            __syncthreads();
        }
        atomicAdd(&detail[mode], mf);
        fails += mf;
    }
    atomicAdd(bad, fails);
}
int main()
{
    uint32_t *d, h = 1, det[3];
    hipMalloc(&d, 32);
    hipMemset(d, 0, 32);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, d + 1);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    hipMemcpy(det, d + 1, 12, hipMemcpyDeviceToHost);
    printf("mismatches: 16-byte stores %u, 8-byte %u, 4-byte %u\n", det[0], det[1], det[2]);
    printf("lds_write_order: %u mismatches\n%s\n", h,
           h == 0 ? "PASS overlapping stores of one instruction land in ascending lane order" : "FAIL");
    return h != 0;
}
