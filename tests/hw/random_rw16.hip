// Hardware probe: ceiling of "one dependent random 16-byte table access per
// lane per step" - the access pattern of k_match_blocks rounds.
//   mode 0: read only            e = tab[h]
//   mode 1: read + write same    e = tab[h]; tab[h] = v
//   mode 2: read + 2 writes      ... plus tab[h2] = v (another random slot)
// Every lane owns a 256 KiB table (16384 x 16 B) in HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(u32x4 *tables, unsigned *out, unsigned steps,
                                            int mode)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = tables + (size_t)gid * 16384;
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        if (mode >= 1)
            t[h] = (u32x4){state, i, h, gid};
        if (mode >= 2)
            t[((state ^ 0x9E3779B9u) * 0x85EBCA6Bu) >> 18] = (u32x4){i, state, gid, h};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
int main()
{
    const unsigned max_lanes = 256 * 10 * 64;
    u32x4 *tables; unsigned *out;
    hipMalloc(&tables, (size_t)max_lanes * 16384 * 16);
    hipMalloc(&out, max_lanes * 4);
    hipMemset(tables, 1, (size_t)max_lanes * 16384 * 16);
    for (int mode = 0; mode < 3; mode++)
        for (unsigned waves_per_cu : {1u, 2u, 3u, 5u, 7u, 10u}) {
            const unsigned lanes = 256 * waves_per_cu * 64;
            const unsigned steps = 3000;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, out, 100u, mode);
            hipEventRecord(a);
            hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, out, steps, mode);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d waves/CU %2u lanes %7u : %8.2f ms -> %.3e lane-steps/s (%.0f ns per step)\n",
                   mode, waves_per_cu, lanes, ms, (double)lanes * steps / (ms * 1e-3),
                   ms * 1e6 / steps);
        }
    return 0;
}
