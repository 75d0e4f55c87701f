// Hardware probe: workgroups of 64 threads resident per CU as a function of
// the dynamic LDS size (occupancy API + a timing census).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void spin(unsigned long long *out, unsigned iters)
{
    extern __shared__ unsigned lds[];
    unsigned v = threadIdx.x;
    lds[threadIdx.x] = v;
    for (unsigned i = 0; i < iters; i++) v = v * 1664525u + lds[(v >> 8) & 63];
    if (v == 0xdeadbeef) out[0] = v;
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlock %zu\n",
           p.multiProcessorCount, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
    for (int lds : {16384, 24576, 28672, 30720, 32000, 32256, 32512, 32768, 36864, 40960, 53248, 65536}) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, 64, lds);
        // census by timing: grid = k * CUs blocks, find the k where time doubles
        float t[10];
        for (int k = 1; k <= 8; k++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(spin, dim3(k * p.multiProcessorCount), dim3(64), lds, 0, d, 20000u);
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(k * p.multiProcessorCount), dim3(64), lds, 0, d, 200000u);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&t[k], a, b);
        }
        printf("lds %6d: api %d blocks/CU; ms at k=1..8:", lds, nb);
        for (int k = 1; k <= 8; k++) printf(" %.2f", t[k]);
        printf("\n");
    }
    return 0;
}
