// Hardware probe: in what order does one wave64 LDS atomic instruction apply
// lanes that hit the same address?  The batched match finder relies on
// "ascending lane order" (lane j sees the value left by the nearest lower lane
// with the same address).  Run on the GPU: prints PASS/FAIL counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned mskor_rtn(unsigned addr, unsigned mask, unsigned data)
{
    unsigned old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old) : "v"(addr), "v"(mask), "v"(data) : "memory");
    return old;
}

__global__ void probe(const unsigned *slots, unsigned *out_xchg, unsigned *out_mskor, int trials)
{
    __shared__ unsigned lds[1024];
    __shared__ unsigned short tab[2048];
    const unsigned lane = threadIdx.x;
    for (int t = 0; t < trials; t++) {
        for (unsigned i = lane; i < 1024; i += 64) lds[i] = 0xAAAA0000u + i;
        for (unsigned i = lane; i < 2048; i += 64) tab[i] = (unsigned short)(0x8000u + i);
        __syncthreads();
        unsigned slot = slots[t * 64 + lane];           // 0..2047
        // 32-bit exchange on lds[slot & 1023]
        unsigned old = __hip_atomic_exchange(&lds[slot & 1023], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        out_xchg[t * 64 + lane] = old;
        // 16-bit field replace on tab[slot] via ds_mskor_rtn_b32
        unsigned byte_addr = (unsigned)(size_t)(&tab[0]) + (slot >> 1) * 4;
        unsigned sh = (slot & 1) * 16;
        unsigned o2 = mskor_rtn(byte_addr, 0xFFFFu << sh, (lane + 1) << sh);
        out_mskor[t * 64 + lane] = (o2 >> sh) & 0xFFFF;
        __syncthreads();
        // final state check value: read back
        out_mskor[trials * 64 + t * 64 + lane] = tab[slot];
        __syncthreads();
    }
}

int main()
{
    const int trials = 4096;
    std::vector<unsigned> slots(trials * 64);
    srand(1);
    for (int t = 0; t < trials; t++) {
        int mode = t % 4;
        for (int l = 0; l < 64; l++) {
            unsigned s;
            if (mode == 0) s = 5;                          // all same
            else if (mode == 1) s = rand() % 8;            // heavy dups
            else if (mode == 2) s = rand() % 64;           // some dups
            else s = rand() % 2048;                        // rare dups
            slots[t * 64 + l] = s;
        }
    }
    unsigned *d_slots, *d_x, *d_m;
    hipMalloc(&d_slots, slots.size() * 4);
    hipMalloc(&d_x, slots.size() * 4);
    hipMalloc(&d_m, slots.size() * 8);
    hipMemcpy(d_slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_slots, d_x, d_m, trials);
    hipDeviceSynchronize();
    std::vector<unsigned> x(slots.size()), m(slots.size() * 2);
    hipMemcpy(x.data(), d_x, x.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), d_m, m.size() * 4, hipMemcpyDeviceToHost);
    long bad_x = 0, bad_m = 0, bad_f = 0;
    for (int t = 0; t < trials; t++) {
        for (int l = 0; l < 64; l++) {
            unsigned s = slots[t * 64 + l];
            // expected under ascending-lane order
            unsigned ex = 0xAAAA0000u + (s & 1023), em = 0x8000u + s;
            for (int j = 0; j < l; j++) {
                if ((slots[t * 64 + j] & 1023) == (s & 1023)) ex = j;
                if (slots[t * 64 + j] == s) em = j + 1;
            }
            unsigned ef = 0;
            for (int j = 0; j < 64; j++) if (slots[t * 64 + j] == s) ef = j + 1;
            if (x[t * 64 + l] != ex) bad_x++;
            if (m[t * 64 + l] != em) bad_m++;
            if (m[trials * 64 + t * 64 + l] != ef) bad_f++;
        }
    }
    printf("lds_atomic_order: xchg mismatches=%ld mskor mismatches=%ld final-state mismatches=%ld of %d\n",
           bad_x, bad_m, bad_f, trials * 64);
    printf("%s\n", (bad_x == 0 && bad_m == 0 && bad_f == 0) ? "PASS ascending-lane order" : "FAIL");
    return 0;
}
