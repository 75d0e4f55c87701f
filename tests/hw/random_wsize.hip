// Hardware probe: does the size of a random table write change its cost?
// Every lane owns 16384 slots of STRIDE bytes; a step reads 16 B of slot h and
// writes W bytes of it (W = 8, 16, 32, 64; STRIDE = max(16, W)).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int W>
__global__ __launch_bounds__(64) void probe(unsigned char *tables, unsigned *out, unsigned steps)
{
    constexpr int STRIDE = W < 16 ? 16 : W;
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    unsigned char *t = tables + (size_t)gid * 16384 * STRIDE;
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        unsigned char *slot = t + (size_t)h * STRIDE;
        const u32x4 e = *(u32x4 *)slot;
        const u32x4 v = (u32x4){state, i, h, gid};
        if (W == 8) *(u32x2 *)slot = (u32x2){state, i};
        if (W >= 16) *(u32x4 *)slot = v;
        if (W >= 32) *(u32x4 *)(slot + 16) = v;
        if (W >= 64) { *(u32x4 *)(slot + 32) = v; *(u32x4 *)(slot + 48) = v; }
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
template <int W> void run(unsigned char *tables, unsigned *out)
{
    for (unsigned waves_per_cu : {2u, 5u, 10u}) {
        const unsigned lanes = 256 * waves_per_cu * 64, steps = 3000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(probe<W>, dim3(lanes / 64), dim3(64), 0, 0, tables, out, 100u);
        hipEventRecord(a);
        hipLaunchKernelGGL(probe<W>, dim3(lanes / 64), dim3(64), 0, 0, tables, out, steps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("write %2d B waves/CU %2u : %8.2f ms -> %.3e lane-steps/s\n", W, waves_per_cu, ms,
               (double)lanes * steps / (ms * 1e-3));
    }
}
int main()
{
    const size_t max_lanes = 256 * 10 * 64;
    unsigned char *tables; unsigned *out;
    hipMalloc(&tables, max_lanes * 16384 * 64);
    hipMalloc(&out, max_lanes * 4);
    hipMemset(tables, 1, max_lanes * 16384 * 64);
    run<8>(tables, out); run<16>(tables, out); run<32>(tables, out); run<64>(tables, out);
    return 0;
}
