// pmc_calib.hip -- known-byte-count kernels to calibrate the rocprofv3 memory counters on gfx950
// (FETCH_SIZE / WRITE_SIZE / TCC_EA0_*_DRAM_32B), per access width.  Measurement tool, not product.
// Every kernel moves exactly BYTES = 512 MiB in and 512 MiB out (larger than the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void __launch_bounds__(256) copy_k(const T *__restrict__ in, T *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int main()
{
    const size_t bytes = 512ull << 20;
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy_k<uint4>, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16);
        hipLaunchKernelGGL(copy_k<uint32_t>, dim3(bytes / 4 / 256), dim3(256), 0, 0, (const uint32_t *)a, (uint32_t *)b, bytes / 4);
        hipLaunchKernelGGL(copy_k<uint8_t>, dim3(bytes / 256), dim3(256), 0, 0, (const uint8_t *)a, (uint8_t *)b, bytes);
    }
    hipDeviceSynchronize();
    printf("each kernel: %zu bytes read + %zu bytes written\n", bytes, bytes);
    return 0;
}
