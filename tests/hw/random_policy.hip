// Hardware probe: does a cache policy (or an uncached allocation) change what
// a random 16-byte table access costs?  k_match_blocks' tables never hit in
// L2 (25 GB of tables, 32 MiB of L2), yet every table read is a 128-byte
// request to HBM (profiles/r2_pmc_requests.txt): 8x over-fetch.
// The access pattern of random_rw16 mode 1 (dependent read + write of the same
// entry), 6 waves per CU, tables 1 MiB apart; variants:
//   0 default            1 nt load + nt store     2 sc0 sc1 load + store
//   3 sc1 load + store   4 nt sc0 sc1             5..: the same on memory
//   from hipExtMallocWithFlags(hipDeviceMallocUncached)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 g_u32x4;

template <int P> __device__ __forceinline__ u32x4 ld(g_u32x4 *p)
{
    u32x4 v;
    if (P == 0) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P> __device__ __forceinline__ void st(g_u32x4 *p, u32x4 v)
{
    if (P == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

template <int P>
__global__ __launch_bounds__(64) void probe(u32x4 *tables, unsigned *out, unsigned steps, size_t stride)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    g_u32x4 *t = (g_u32x4 *)tables + (size_t)gid * stride;
    unsigned state = gid * 2654435761u + 12345u;
    for (unsigned i = 0; i < steps; i++) {
        const unsigned h = (state * 0x1E35A7BDu) >> 18;
        const u32x4 e = ld<P>(t + h);
        st<P>(t + h, (u32x4){state, i, h, gid});
        state = state * 1664525u + (e.x ^ e.y ^ e.z ^ e.w) + 1013904223u;
    }
    out[gid] = state;
}

template <int P> void run(const char *name, u32x4 *tables, unsigned *out, size_t stride, unsigned lanes)
{
    const unsigned steps = 3000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<P>, dim3(lanes / 64), dim3(64), 0, 0, tables, out, 100u, stride);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<P>, dim3(lanes / 64), dim3(64), 0, 0, tables, out, steps, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.2f ms -> %.3e lane-steps/s\n", name, ms, (double)lanes * steps / (ms * 1e-3));
}

int main()
{
    const unsigned lanes = 256 * 6 * 64;
    const size_t stride = (1u << 20) / 16; // entries: tables 1 MiB apart
    const size_t bytes = (size_t)lanes * stride * 16;
    unsigned *out; hipMalloc(&out, lanes * 4);
    for (int uncached = 0; uncached < 2; uncached++) {
        u32x4 *tables = nullptr;
        hipError_t e = uncached ? hipExtMallocWithFlags((void **)&tables, bytes, hipDeviceMallocUncached)
                                : hipMalloc((void **)&tables, bytes);
        if (e != hipSuccess) { printf("alloc %d failed: %s\n", uncached, hipGetErrorString(e)); continue; }
        hipMemset(tables, 1, bytes);
        hipDeviceSynchronize();
        const char *m = uncached ? "uncached alloc" : "hipMalloc";
        char nm[64];
        snprintf(nm, 64, "%s default", m);      run<0>(nm, tables, out, stride, lanes);
        snprintf(nm, 64, "%s nt", m);           run<1>(nm, tables, out, stride, lanes);
        snprintf(nm, 64, "%s sc0 sc1", m);      run<2>(nm, tables, out, stride, lanes);
        snprintf(nm, 64, "%s sc1", m);          run<3>(nm, tables, out, stride, lanes);
        snprintf(nm, 64, "%s sc0 sc1 nt", m);   run<4>(nm, tables, out, stride, lanes);
        snprintf(nm, 64, "%s default again", m); run<0>(nm, tables, out, stride, lanes);
        hipFree(tables);
    }
    return 0;
}
