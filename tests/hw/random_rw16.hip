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
                                            int mode, size_t stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = tables + (size_t)gid * stride;
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
int main(int argc, char **argv)
{
    // optional: GiB of memory allocated (and kept) BEFORE the tables, to move
    // the tables to another place in HBM (the rates are bimodal between
    // processes at large footprints; is it where the tables lie?)
    const size_t dummy_gib = argc > 1 ? (size_t)atoi(argv[1]) : 0;
    void *dummy = nullptr;
    if (dummy_gib) {
        hipMalloc(&dummy, dummy_gib << 30);
        hipMemset(dummy, 0, dummy_gib << 30);
    }
    const unsigned max_lanes = 256 * 10 * 64;
    // optional: spread the lanes' tables over `span_gib` of memory (default:
    // dense, 256 KiB apart = 40 GiB for 10 waves per CU)
    const size_t span_gib = argc > 2 ? (size_t)atoi(argv[2]) : 40;
    const size_t stride = (span_gib << 30) / max_lanes / 16 / 8 * 8; // entries
    u32x4 *tables; unsigned *out;
    hipMalloc(&tables, (size_t)max_lanes * stride * 16);
    hipMalloc(&out, max_lanes * 4);
    hipMemset(tables, 1, (size_t)max_lanes * stride * 16);
    for (int mode = 0; mode < 3; mode++)
        for (unsigned waves_per_cu : {1u, 2u, 3u, 5u, 7u, 10u}) {
            const unsigned lanes = 256 * waves_per_cu * 64;
            const unsigned steps = 3000;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, out, 100u, mode, stride);
            hipEventRecord(a);
            hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, out, steps, mode, stride);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d waves/CU %2u lanes %7u : %8.2f ms -> %.3e lane-steps/s (%.0f ns per step)\n",
                   mode, waves_per_cu, lanes, ms, (double)lanes * steps / (ms * 1e-3),
                   ms * 1e6 / steps);
        }
    return 0;
}
