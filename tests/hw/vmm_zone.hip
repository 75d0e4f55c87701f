// Hardware probe (round 6): WHICH memory do hipMemCreate (the virtual-memory
// calls) and hipMalloc hand out?  tests/hw/vmm_spread.hip found 16 GiB of
// tables from hipMemCreate running at the fast part's rate (1.93 ms) where
// hipMalloc's first 16 GiB run at the slow part's (2.54).  Here: regions of
// 16 GiB from either call made one after the other and all kept, each probed
// with the lane kernel's table access (65 536 tables of 256 KiB, 768
// dependent random 16-byte read + write pairs per lane).
// build: hipcc --offload-arch=gfx950 -O2 -o tests/hw/vmm_zone tests/hw/vmm_zone.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
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
                                            unsigned steps)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    u32x4 *t = (u32x4 *)(base + ((size_t)gid << 18));
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = t[h];
        t[h] = (u32x4){state, i, h, gid};
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}
static float run(char *base, unsigned *out)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 64u);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, base, out, 768u);
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
static double free_gib()
{
    size_t f = 0, t = 0;
    CK(hipMemGetInfo(&f, &t));
    return f / 1073741824.0;
}
int main(int argc, char **argv)
{
    const size_t G = (size_t)1 << 30, W = 16 * G;
    unsigned *out;
    CK(hipMalloc(&out, 65536 * 4));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    printf("free %.1f GiB\n", free_gib());
    struct Reg {
        hipMemGenericAllocationHandle_t h;
        char *va;
    };
    // 1. hipMemCreate regions, one after the other, all kept
    {
        std::vector<Reg> regs;
        printf("1. hipMemCreate, 16 GiB each, all kept (ms to create+map, probe ms, free GiB):\n");
        for (int i = 0; i < 17; i++) {
            Reg r;
            double t0 = now();
            if (hipMemCreate(&r.h, W, &prop, 0) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            CK(hipMemAddressReserve((void **)&r.va, W, 0, nullptr, 0));
            CK(hipMemMap(r.va, W, 0, r.h, 0));
            CK(hipMemSetAccess(r.va, W, &acc, 1));
            double t1 = now();
            regs.push_back(r);
            printf("  #%2d %p  %.1f ms  %.2f  %.1f\n", i + 1, (void *)r.va,
                   (t1 - t0) * 1e3, run(r.va, out), free_gib());
        }
        double t0 = now();
        for (Reg &r : regs) {
            CK(hipMemUnmap(r.va, W));
            CK(hipMemRelease(r.h));
            CK(hipMemAddressFree(r.va, W));
        }
        printf("  released in %.1f ms, free %.1f GiB", (now() - t0) * 1e3, free_gib());
        for (int k = 0; k < 6; k++) {
            usleep(500000);
            printf(" .. %.1f", free_gib());
        }
        printf("\n");
    }
    // 2. a mix: 26 GiB from hipMalloc (a batch), then ONE hipMemCreate region,
    // then hipMalloc regions until the device is full, then the first again
    {
        void *batch = nullptr;
        CK(hipMalloc(&batch, 26 * G));
        Reg r;
        double t0 = now();
        CK(hipMemCreate(&r.h, W, &prop, 0));
        CK(hipMemAddressReserve((void **)&r.va, W, 0, nullptr, 0));
        CK(hipMemMap(r.va, W, 0, r.h, 0));
        CK(hipMemSetAccess(r.va, W, &acc, 1));
        printf("2. behind 26 GiB of hipMalloc: hipMemCreate region (%.1f ms to make) %.2f %.2f ms\n",
               (now() - t0) * 1e3, run(r.va, out), run(r.va, out));
        std::vector<void *> held;
        printf("   hipMalloc regions of 16 GiB behind it:");
        for (;;) {
            void *p = nullptr;
            if (hipMalloc(&p, W) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            held.push_back(p);
            printf(" %.2f", run((char *)p, out));
        }
        printf("\n   the hipMemCreate region again: %.2f ms\n", run(r.va, out));
        for (void *p : held)
            CK(hipFree(p));
        CK(hipMemUnmap(r.va, W));
        CK(hipMemRelease(r.h));
        CK(hipMemAddressFree(r.va, W));
        CK(hipFree(batch));
    }
    // 3. stream-ordered allocation (hipMallocAsync: the runtime's pool)
    {
        usleep(3000000);
        void *p = nullptr;
        if (hipMallocAsync(&p, W, 0) == hipSuccess) {
            CK(hipStreamSynchronize(0));
            printf("3. hipMallocAsync 16 GiB: %.2f %.2f ms\n", run((char *)p, out), run((char *)p, out));
            CK(hipFreeAsync(p, 0));
            CK(hipStreamSynchronize(0));
        } else {
            (void)hipGetLastError();
            printf("3. hipMallocAsync failed\n");
        }
    }
    // 4. how long a release keeps the memory busy: 64 GiB from hipMalloc,
    // freed, then allocated again at once
    {
        usleep(3000000);
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, 64 * G));
        double t1 = now();
        CK(hipMemset(p, 1, 64 * G));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        CK(hipMalloc(&p, 64 * G));
        double t4 = now();
        CK(hipFree(p));
        printf("4. hipMalloc 64 GiB %.1f ms, memset %.1f ms, hipFree %.1f ms, hipMalloc again %.1f ms\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
    return 0;
}
