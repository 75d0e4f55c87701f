// Hardware probe (round 6): WHY do dependent random 16-byte read + write
// pairs run at 2.0e10/s in the first two thirds of the device's memory and
// at 2.6e10/s in the last third (tests/hw/zone_map.hip)?  If the slow part
// interleaves two ranks of the HBM stacks by some address bit, tables that
// keep that bit constant would run at the fast part's rate anywhere - inside
// a context's memory budget, with no filler allocation.
//
// The lane kernel's access (65 536 tables of 256 KiB = a dense 16 GiB
// window, 768 pairs per lane) with ONE ADDRESS BIT k HELD at 0 (or 1): the
// dense offset gets a constant bit inserted at position k, so the window
// spans 32 GiB and touches every second 2^k-byte piece of it.  k = 8 .. 34,
// in a window of the slow part and one of the fast part; then pairs of bits.
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/addr_bits tests/hw/addr_bits.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// up to two inserted bits (k2 > k1, positions in the OUTPUT address; k = 63:
// none); v1/v2 their values
__global__ __launch_bounds__(64) void probe(char *base, unsigned *out,
                                            unsigned steps, unsigned k1,
                                            unsigned v1, unsigned k2,
                                            unsigned v2)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        unsigned long long d = ((unsigned long long)gid << 18) | (h << 4);
        if (k1 < 63)
            d = ((d >> k1) << (k1 + 1)) | (d & ((1ull << k1) - 1)) |
                ((unsigned long long)v1 << k1);
        if (k2 < 63)
            d = ((d >> k2) << (k2 + 1)) | (d & ((1ull << k2) - 1)) |
                ((unsigned long long)v2 << k2);
        u32x4 *p = (u32x4 *)(base + d);
        const u32x4 e = *p;
        *p = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run(char *base, unsigned *out, unsigned k1, unsigned v1,
                 unsigned k2 = 63, unsigned v2 = 0)
{
    const unsigned lanes = 65536;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, base, out, 64u,
                       k1, v1, k2, v2);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(lanes / 64), dim3(64), 0, 0, base, out,
                       768u, k1, v1, k2, v2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms;
}
__global__ __launch_bounds__(64) void probe_stride(char *base, unsigned *out,
                                                   unsigned steps, size_t stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = (u32x4 *)(base + (size_t)gid * stride);
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run_stride(char *base, unsigned *out, size_t stride)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe_stride, dim3(1024), dim3(64), 0, 0, base, out, 64u, stride);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe_stride, dim3(1024), dim3(64), 0, 0, base, out, 768u, stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms;
}
int main(int argc, char **argv)
{
    const size_t G = (size_t)1 << 30;
    unsigned *out;
    hipMalloc(&out, 65536 * 4);
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    printf("free %.1f GiB of %.1f\n", free_b / 1073741824.0,
           total_b / 1073741824.0);
    size_t big = (free_b - 2 * G) / (16 * G) * (16 * G);
    char *p = nullptr;
    while (big >= 16 * G && hipMalloc((void **)&p, big) != hipSuccess) {
        (void)hipGetLastError();
        big -= 16 * G;
    }
    printf("one allocation of %zu GiB at %p\n", big / G, (void *)p);
    // the map again, by windows of 16 GiB (dense)
    printf("dense windows of 16 GiB by offset:");
    for (size_t off = 0; off + 16 * G <= big; off += 16 * G)
        printf(" %.2f", run(p + off, out, 63, 0));
    printf("\n");
    const size_t slow_off = 32 * G, fast_off = big - 32 * G;
    for (int zone = 0; zone < 2; zone++) {
        char *w = p + (zone ? fast_off : slow_off);
        printf("%s part, a window of 32 GiB at +%zu GiB: one address bit held (ms with 0 / with 1)\n",
               zone ? "fast" : "slow", (zone ? fast_off : slow_off) / G);
        printf("  dense (no bit held): %.2f %.2f\n", run(w, out, 63, 0),
               run(w + 16 * G, out, 63, 0));
        for (unsigned k = 8; k <= 34; k++)
            printf("  bit %2u: %.2f / %.2f\n", k, run(w, out, k, 0),
                   run(w, out, k, 1));
    }
    if (argc > 1 && !strcmp(argv[1], "nopairs")) goto after_pairs;
    // pairs of bits in the slow part (a 64 GiB window): the XOR of two bits
    // may be what selects; both held at 0
    {
        char *w = p + slow_off;
        printf("slow part, a window of 64 GiB at +%zu GiB: two bits held at 0 (rows k1, columns k2 = k1+1 ..)\n",
               slow_off / G);
        for (unsigned k1 = 8; k1 <= 33; k1 += 1) {
            printf("  k1 %2u:", k1);
            for (unsigned k2 = k1 + 1; k2 <= 35; k2++)
                printf(" %.2f", run(w, out, k1, 0, k2, 0));
            printf("\n");
            fflush(stdout);
        }
    }
after_pairs:
    // the library's own layout: a stride between two lanes' tables (16-byte
    // entries: 256 KiB dense .. 1 MiB), (A2) inside the big allocation at
    // the slow window, (B) in allocations of lanes x stride made one after
    // the other behind a 26 GiB stand-in for a batch, each kept
    static const unsigned kib[] = {256, 320, 384, 456, 512, 640, 768, 1024};
    printf("A2. strides inside the big allocation at +%zu GiB (ms):", slow_off / G);
    for (unsigned s : kib)
        printf("  %u KiB %.2f", s, run_stride(p + slow_off, out, (size_t)s << 10));
    printf("\n");
    hipFree(p);
    for (int rep = 0; rep < 2; rep++) {
        void *batch = nullptr;
        hipMalloc(&batch, 26 * G);
        std::vector<void *> held;
        printf("B%d. allocations of 65 536 x stride behind 26 GiB, each kept (address, ms):\n", rep);
        for (int i = 0; i < 8; i++) {
            const unsigned s = kib[rep ? 7 - i : i];
            void *r = nullptr;
            if (hipMalloc(&r, (size_t)65536 * ((size_t)s << 10)) != hipSuccess) {
                (void)hipGetLastError();
                printf("  %u KiB: no room\n", s);
                continue;
            }
            held.push_back(r);
            printf("  %4u KiB %p %.2f %.2f\n", s, r,
                   run_stride((char *)r, out, (size_t)s << 10),
                   run_stride((char *)r, out, (size_t)s << 10));
        }
        for (void *r : held)
            hipFree(r);
        hipFree(batch);
    }
    return 0;
}
