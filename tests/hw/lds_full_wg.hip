// Hardware probe: can ONE workgroup of 320 threads own all 163840 B of LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(320) void full(unsigned *out)
{
    __shared__ unsigned short t[5][16384];
    unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (unsigned i = l; i < 16384; i += 64) t[w][i] = (unsigned short)(i + w);
    __syncthreads();
    unsigned acc = 0;
    for (unsigned i = l; i < 16384; i += 64) acc += t[(w + 1) % 5][i];
    atomicAdd(out, acc);
}
int main()
{
    unsigned *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(full, dim3(512), dim3(320), 0, 0, d);
    hipError_t e = hipGetLastError();
    printf("launch: %s\n", hipGetErrorString(e));
    e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("sum %u\n", h);
    return 0;
}
