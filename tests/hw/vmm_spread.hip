// Hardware probe (round 6): can a context have the FAST rate of dependent
// random 16-byte read + write pairs (the lane kernel's table access) without
// holding more than its tables?  tests/hw/addr_bits.hip: 16 GiB of tables
// run at 2.0e10 pairs/s packed into the first two thirds of the device's
// memory and at 2.6e10/s when the same tables lie spread over 64 GiB of it.
// Here the spreading is done with the virtual-memory calls: 64 physical
// chunks of 1 GiB are created one after the other (hipMemCreate), sixteen of
// them are mapped and the other 48 given back.
//   Z. chunks 0..15 (packed in the order they were handed out), one range
//   X. chunks 0, 4, 8 .. 60 (spread over the 64 GiB that were handed out),
//      mapped into ONE contiguous 16 GiB range of addresses - the others
//      released BEFORE the probe runs
//   Y. chunks 0..15 mapped at every fourth GiB of a 64 GiB address range
//      (spread addresses, packed memory): is it the address that counts?
//   W. every chunk kept (64 GiB held), tables a MiB apart: the plain spread
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/vmm_spread tests/hw/vmm_spread.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)
// table of lane gid: chunk gid / per_chunk at base + chunk * chunk_stride,
// inside it (gid % per_chunk) * table_stride
__global__ __launch_bounds__(64) void probe(char *base, unsigned *out,
                                            unsigned steps, unsigned per_chunk,
                                            size_t chunk_stride,
                                            size_t table_stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = (u32x4 *)(base + (size_t)(gid / per_chunk) * chunk_stride +
                         (size_t)(gid % per_chunk) * table_stride);
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run(char *base, unsigned *out, unsigned per_chunk,
                 size_t chunk_stride, size_t table_stride)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 64u,
                       per_chunk, chunk_stride, table_stride);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 768u,
                       per_chunk, chunk_stride, table_stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms;
}
static double now()
{
    return std::chrono::duration<double>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}
int main()
{
    const size_t G = (size_t)1 << 30;
    unsigned *out;
    CK(hipMalloc(&out, 65536 * 4));
    void *batch = nullptr;
    CK(hipMalloc(&batch, 26 * G)); // a batch's own buffers lie in front
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop,
                                      hipMemAllocationGranularityRecommended));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("granularity %zu KiB, free %.1f GiB\n", gran >> 10, free_b / 1073741824.0);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int rep = 0; rep < 2; rep++) {
        const int N = 64;
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        double t0 = now();
        for (int i = 0; i < N; i++)
            CK(hipMemCreate(&h[i], G, &prop, 0));
        double t1 = now();
        printf("rep %d: 64 chunks of 1 GiB created in %.1f ms\n", rep, (t1 - t0) * 1e3);
        char *va = nullptr;
        // Z: chunks 0..15, one range
        CK(hipMemAddressReserve((void **)&va, 64 * G, 0, nullptr, 0));
        for (int i = 0; i < 16; i++)
            CK(hipMemMap(va + i * G, G, 0, h[i], 0));
        CK(hipMemSetAccess(va, 16 * G, &acc, 1));
        printf("  Z packed chunks 0..15, one range:          %.2f %.2f ms\n",
               run(va, out, 4096, G, 262144), run(va, out, 4096, G, 262144));
        CK(hipMemUnmap(va, 16 * G));
        // Y: chunks 0..15 at every fourth GiB of the range
        for (int i = 0; i < 16; i++) {
            CK(hipMemMap(va + 4 * i * G, G, 0, h[i], 0));
            CK(hipMemSetAccess(va + 4 * i * G, G, &acc, 1));
        }
        printf("  Y packed chunks at every fourth GiB:       %.2f %.2f ms\n",
               run(va, out, 4096, 4 * G, 262144), run(va, out, 4096, 4 * G, 262144));
        for (int i = 0; i < 16; i++)
            CK(hipMemUnmap(va + 4 * i * G, G));
        // W: all 64 chunks, tables a MiB apart
        for (int i = 0; i < N; i++)
            CK(hipMemMap(va + i * G, G, 0, h[i], 0));
        CK(hipMemSetAccess(va, 64 * G, &acc, 1));
        printf("  W all 64 chunks, tables a MiB apart:       %.2f %.2f ms\n",
               run(va, out, 1024, G, 1048576), run(va, out, 1024, G, 1048576));
        CK(hipMemUnmap(va, 64 * G));
        // X: chunks 0, 4, .. 60 in one range; the others released first
        t0 = now();
        for (int i = 0; i < N; i++)
            if (i % 4)
                CK(hipMemRelease(h[i]));
        for (int i = 0; i < 16; i++)
            CK(hipMemMap(va + i * G, G, 0, h[4 * i], 0));
        CK(hipMemSetAccess(va, 16 * G, &acc, 1));
        t1 = now();
        CK(hipMemGetInfo(&free_b, &total_b));
        printf("  (48 chunks released + 16 mapped in %.1f ms; free now %.1f GiB)\n",
               (t1 - t0) * 1e3, free_b / 1073741824.0);
        printf("  X spread chunks 0,4..60, one range:        %.2f %.2f ms\n",
               run(va, out, 4096, G, 262144), run(va, out, 4096, G, 262144));
        // ... and with other memory allocated into the gaps meanwhile
        void *gap = nullptr;
        CK(hipMalloc(&gap, 40 * G));
        printf("  X again, 40 GiB allocated behind it:       %.2f ms\n",
               run(va, out, 4096, G, 262144));
        CK(hipFree(gap));
        CK(hipMemUnmap(va, 16 * G));
        for (int i = 0; i < 16; i++)
            CK(hipMemRelease(h[4 * i]));
        CK(hipMemAddressFree(va, 64 * G));
    }
    // the same with chunks of 256 MiB (finer spread, 256 created, 64 kept)
    {
        const size_t C = 256u << 20;
        const int N = 256;
        std::vector<hipMemGenericAllocationHandle_t> h(N);
        double t0 = now();
        for (int i = 0; i < N; i++)
            CK(hipMemCreate(&h[i], C, &prop, 0));
        for (int i = 0; i < N; i++)
            if (i % 4)
                CK(hipMemRelease(h[i]));
        char *va = nullptr;
        CK(hipMemAddressReserve((void **)&va, 16 * G, 0, nullptr, 0));
        for (int i = 0; i < 64; i++)
            CK(hipMemMap(va + i * C, C, 0, h[4 * i], 0));
        CK(hipMemSetAccess(va, 16 * G, &acc, 1));
        double t1 = now();
        printf("256 chunks of 256 MiB, every fourth kept, one range (%.1f ms to build): %.2f %.2f ms\n",
               (t1 - t0) * 1e3, run(va, out, 1024, C, 262144), run(va, out, 1024, C, 262144));
        CK(hipMemUnmap(va, 16 * G));
        for (int i = 0; i < 64; i++)
            CK(hipMemRelease(h[4 * i]));
        CK(hipMemAddressFree(va, 16 * G));
    }
    // plain hipMalloc for comparison: 16 GiB packed, 64 GiB at a MiB apart
    {
        char *p = nullptr;
        CK(hipMalloc((void **)&p, 16 * G));
        printf("hipMalloc 16 GiB, packed:            %.2f ms\n", run(p, out, 65536, 0, 262144));
        CK(hipFree(p));
        CK(hipMalloc((void **)&p, 64 * G));
        printf("hipMalloc 64 GiB, a MiB per table:   %.2f ms\n", run(p, out, 65536, 0, 1048576));
        CK(hipFree(p));
    }
    return 0;
}
