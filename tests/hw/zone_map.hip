// Hardware probe: the rate of dependent random 16-byte read + write pairs
// (the lane kernel's table access, tests/hw/random_rw16.hip mode 1) by WHERE
// in the device's memory the tables lie.  65 536 tables of 256 KiB, dense
// (a 16 GiB window), 768 pairs per lane:
//   A. one allocation of nearly all free memory, the window moved through it
//      in steps of 16 GiB;
//   B. sixteen-GiB allocations made one after the other and all kept, each
//      probed as it comes (the order hipMalloc hands memory out in);
//   C. streaming by the same offsets: a device-to-device copy of 8 GiB.
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/zone_map tests/hw/zone_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(u32x4 *tables, unsigned *out,
                                            unsigned steps, size_t stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = tables + (size_t)gid * stride;
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run(void *base, unsigned *out)
{
    const unsigned lanes = 65536;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, (u32x4 *)base,
                       out, 64u, (size_t)16384);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, (u32x4 *)base,
                       out, 768u, (size_t)16384);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms;
}
int main()
{
    const size_t G = (size_t)1 << 30, W = 16 * G;
    unsigned *out;
    hipMalloc(&out, 65536 * 4);
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    printf("free %.1f GiB of %.1f\n", free_b / 1073741824.0, total_b / 1073741824.0);
    {
        size_t big = (free_b - 2 * G) / W * W;
        void *p = nullptr;
        while (big >= W && hipMalloc(&p, big) != hipSuccess) {
            (void)hipGetLastError();
            big -= W;
        }
        printf("A. one allocation of %zu GiB at %p: ms per window of 16 GiB, by offset\n",
               big / G, p);
        for (int rep = 0; rep < 2; rep++) {
            for (size_t off = 0; off + W <= big; off += W)
                printf("  +%3zu GiB %.2f", off / G, run((char *)p + off, out));
            printf("\n");
        }
        // C. streaming: a device-to-device copy of 8 GiB inside each window
        printf("C. the same allocation, hipMemcpyDtoD of 8 GiB inside each window: GB/s (read + write)\n");
        for (int rep = 0; rep < 2; rep++) {
            for (size_t off = 0; off + W <= big; off += W) {
                hipEvent_t a, b;
                hipEventCreate(&a);
                hipEventCreate(&b);
                hipEventRecord(a);
                hipMemcpyAsync((char *)p + off + 8 * G, (char *)p + off, 8 * G,
                               hipMemcpyDeviceToDevice, 0);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                printf("  +%3zu GiB %.0f", off / G, 2.0 * 8 * G / (ms * 1e-3) / 1e9);
                hipEventDestroy(a);
                hipEventDestroy(b);
            }
            printf("\n");
        }
        hipFree(p);
    }
    {
        printf("B. allocations of 16 GiB, one after the other, all kept: address, ms\n");
        std::vector<void *> held;
        for (;;) {
            void *p = nullptr;
            if (hipMalloc(&p, W) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            held.push_back(p);
            printf("  #%2zu %p %.2f\n", held.size(), p, run(p, out));
        }
        printf("   again, in the same order:");
        for (void *p : held)
            printf(" %.2f", run(p, out));
        printf("\n");
        for (void *p : held)
            hipFree(p);
    }
    return 0;
}
