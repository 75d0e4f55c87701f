// Hardware probe (round 6): 64 physical chunks of 1 GiB (hipMemCreate, one
// after the other, behind 26 GiB of hipMalloc) mapped into one 64 GiB range
// and ALL KEPT while the layouts below are probed - nothing is released
// meanwhile, so no wipe of released memory runs under a probe (it did in
// tests/hw/vmm_spread.hip: its second repetition measured the wipe).
// 65 536 tables of 256 KiB, 768 dependent random read + write pairs each:
//   packed in chunks 0..15 / 16..31 / 32..47 / 48..63
//   in every 4th chunk (0, 4 .. 60), in every 2nd chunk of the first 32
//   in the first 256 MiB of every chunk; a MiB per table over all 64
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/vmm_layouts tests/hw/vmm_layouts.hip
#include <hip/hip_runtime.h>
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
__global__ __launch_bounds__(64) void probe(char *base, unsigned *out,
                                            unsigned steps, unsigned per_group,
                                            size_t group_stride,
                                            size_t table_stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = (u32x4 *)(base + (size_t)(gid / per_group) * group_stride +
                         (size_t)(gid % per_group) * table_stride);
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run(char *base, unsigned *out, unsigned per_group,
                 size_t group_stride, size_t table_stride)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 64u,
                       per_group, group_stride, table_stride);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 768u,
                       per_group, group_stride, table_stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms;
}
int main()
{
    const size_t G = (size_t)1 << 30, M = (size_t)1 << 20;
    unsigned *out;
    CK(hipMalloc(&out, 65536 * 4));
    void *batch = nullptr;
    CK(hipMalloc(&batch, 26 * G));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const int N = 64;
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    for (int i = 0; i < N; i++)
        CK(hipMemCreate(&h[i], G, &prop, 0));
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, N * G, 0, nullptr, 0));
    for (int i = 0; i < N; i++)
        CK(hipMemMap(va + i * G, G, 0, h[i], 0));
    CK(hipMemSetAccess(va, N * G, &acc, 1));
    for (int rep = 0; rep < 2; rep++) {
        printf("packed in chunks 0..15 / 16..31 / 32..47 / 48..63: %.2f %.2f %.2f %.2f\n",
               run(va, out, 65536, 0, 262144), run(va + 16 * G, out, 65536, 0, 262144),
               run(va + 32 * G, out, 65536, 0, 262144), run(va + 48 * G, out, 65536, 0, 262144));
        printf("every 4th chunk (16 of 64):                 %.2f\n", run(va, out, 4096, 4 * G, 262144));
        printf("every 2nd chunk of 0..31 / of 32..63:       %.2f %.2f\n",
               run(va, out, 4096, 2 * G, 262144), run(va + 32 * G, out, 4096, 2 * G, 262144));
        printf("first 256 MiB of every chunk:               %.2f\n", run(va, out, 1024, G, 262144));
        printf("first 512 MiB of chunks 0..31:              %.2f\n", run(va, out, 2048, G, 262144));
        printf("a MiB per table, all 64 / 512 KiB, 0..31:   %.2f %.2f\n",
               run(va, out, 65536, 0, M), run(va, out, 65536, 0, M / 2));
    }
    // the same layouts in plain hipMalloc memory behind it (another place)
    char *p = nullptr;
    if (hipMalloc((void **)&p, 64 * G) == hipSuccess) {
        printf("hipMalloc 64 GiB: packed quarters %.2f %.2f %.2f %.2f, every 4th GiB %.2f, a MiB per table %.2f\n",
               run(p, out, 65536, 0, 262144), run(p + 16 * G, out, 65536, 0, 262144),
               run(p + 32 * G, out, 65536, 0, 262144), run(p + 48 * G, out, 65536, 0, 262144),
               run(p, out, 4096, 4 * G, 262144), run(p, out, 65536, 0, M));
    }
    return 0;
}
