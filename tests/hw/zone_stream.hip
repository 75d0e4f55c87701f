// Hardware probe (round 6): does STREAMING see the two parts of the device's
// memory that random access sees (tests/hw/zone_map.hip: dependent random
// 16-byte accesses run 30 % faster in the last third)?  cfg5 - a pure
// streaming path, 32 GiB of incompressible input - decompresses in 11.8 ms on
// some boxes and runs and in 14.0 ms on others (19 %), and nobody knew why.
// One allocation of nearly all free memory; per window of 16 GiB: a read-only
// kernel (16 bytes per lane per load, every CU), a write-only kernel, and a
// copy inside the window (8 GiB -> 8 GiB), GB/s each; then the copy between
// a window of the first part and one of the last.
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/zone_stream tests/hw/zone_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)
__global__ __launch_bounds__(256) void k_read(const u32x4 *p, size_t n, unsigned *out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (size_t)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        out[0] = 1;
}
__global__ __launch_bounds__(256) void k_write(u32x4 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store((u32x4){1, 2, 3, (unsigned)i}, p + i);
}
__global__ __launch_bounds__(256) void k_copy(const u32x4 *s, u32x4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
template <class F> static double timed(F f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipEventRecord(a));
    f();
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / 2 * 1e-3;
}
int main()
{
    const size_t G = (size_t)1 << 30, W = 16 * G;
    unsigned *out;
    CK(hipMalloc(&out, 4));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    size_t big = (free_b - 2 * G) / W * W;
    char *p = nullptr;
    while (big >= W && hipMalloc((void **)&p, big) != hipSuccess) {
        (void)hipGetLastError();
        big -= W;
    }
    printf("one allocation of %zu GiB\n", big / G);
    const dim3 grid(256 * 16), blk(256);
    for (int rep = 0; rep < 2; rep++) {
        printf("offset GiB: read GB/s, write GB/s, copy 8->8 GiB GB/s (read + write)\n");
        for (size_t off = 0; off + W <= big; off += W) {
            char *w = p + off;
            const double tr = timed([&] { hipLaunchKernelGGL(k_read, grid, blk, 0, 0, (const u32x4 *)w, W / 16, out); });
            const double tw = timed([&] { hipLaunchKernelGGL(k_write, grid, blk, 0, 0, (u32x4 *)w, W / 16); });
            const double tc = timed([&] { hipLaunchKernelGGL(k_copy, grid, blk, 0, 0, (const u32x4 *)w, (u32x4 *)(w + 8 * G), 8 * G / 16); });
            printf("  +%3zu: %5.0f %5.0f %5.0f\n", off / G, W / tr / 1e9, W / tw / 1e9, 2.0 * 8 * G / tc / 1e9);
        }
    }
    // a copy from the first part into the last and back
    {
        char *lo = p, *hi = p + big - W;
        const double t1 = timed([&] { hipLaunchKernelGGL(k_copy, grid, blk, 0, 0, (const u32x4 *)lo, (u32x4 *)hi, W / 16); });
        const double t2 = timed([&] { hipLaunchKernelGGL(k_copy, grid, blk, 0, 0, (const u32x4 *)hi, (u32x4 *)lo, W / 16); });
        printf("copy 16 GiB first -> last %.0f GB/s, last -> first %.0f GB/s (read + write)\n", 2.0 * W / t1 / 1e9, 2.0 * W / t2 / 1e9);
    }
    return 0;
}
