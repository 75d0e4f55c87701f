// Hardware probe: how many "lane-private hash-table steps" per second can the
// memory system sustain?  Every LANE owns a 32 KiB u16 table and a 64 KiB
// input block in global memory and runs a dependent chain:
//   h = hash(state); cand = table[h]; table[h] = pos; x = load16(block+cand); state ^= x
// This is the access pattern of a thread-per-block Snappy match finder.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(64) void probe(unsigned short *tables, const unsigned char *blocks,
                                            unsigned *out, unsigned steps, unsigned lanes_total)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= lanes_total) return;
    unsigned short *t = tables + (size_t)gid * 16384;
    const unsigned char *b = blocks + (size_t)gid * 65536;
    unsigned state = gid * 2654435761u + 12345u, pos = 1;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const unsigned cand = t[h];
        t[h] = (unsigned short)pos;
        uint4 x;
        __builtin_memcpy(&x, b + (cand & 0xFFF0u), 16);
        unsigned y;
        __builtin_memcpy(&y, b + pos, 4);
        state = state * 1664525u + (x.x ^ x.y ^ x.z ^ x.w) + y + 1013904223u;
        pos = (pos + 3 + (state & 7)) & 0xFFFF;
    }
    out[gid] = state;
}
int main(int argc, char **argv)
{
    for (unsigned waves_per_cu : {1u, 2u, 4u, 8u, 16u, 32u}) {
        const unsigned lanes = 256 * waves_per_cu * 64;
        unsigned short *tables; unsigned char *blocks; unsigned *out;
        hipMalloc(&tables, (size_t)lanes * 32768);
        hipMalloc(&blocks, (size_t)lanes * 65536);
        hipMalloc(&out, lanes * 4);
        hipMemset(tables, 0, (size_t)lanes * 32768);
        hipMemset(blocks, 7, (size_t)lanes * 65536);
        const unsigned steps = 4000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, blocks, out, 200u, lanes);
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, tables, blocks, out, steps, lanes);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("waves/CU %2u  lanes %7u  tables %6.1f MiB  blocks %7.1f MiB : %8.2f ms  -> %.3e lane-steps/s  (%.1f ns per wave-step)\n",
               waves_per_cu, lanes, lanes * 32768.0 / 1048576, lanes * 65536.0 / 1048576, ms,
               (double)lanes * steps / (ms * 1e-3), ms * 1e6 / steps);
        hipFree(tables); hipFree(blocks); hipFree(out);
    }
    return 0;
}
