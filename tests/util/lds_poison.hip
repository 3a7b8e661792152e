// Test utility (not part of the product): fills the LDS of every CU -- and a chunk of freshly allocated device memory -- with NaN
// bit patterns, the state a GPU may be in after another tenant.  A kernel that multiplies an operand it never wrote by a zero
// weight (0 x NaN = NaN) passes on a clean device and fails after this.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void poison_kernel(unsigned* sink)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = 0x7fc00000u + (unsigned)i % 7u;
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = lds[blockIdx.x % 1000];
}
extern "C" int poison_lds(int repeats)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 4096 * 4) != hipSuccess) return -1;
    for (int r = 0; r < repeats; ++r) hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(1024), 160 * 1024, 0, sink);
    // device memory: allocate, fill with NaN, free -- later hipMallocs of the library may land on it
    void* big = nullptr;
    if (hipMalloc(&big, (size_t)1 << 30) == hipSuccess) { hipMemset(big, 0xff, (size_t)1 << 30); hipDeviceSynchronize(); hipFree(big); }
    const hipError_t e = hipDeviceSynchronize();
    hipFree(sink);
    return e == hipSuccess ? 0 : -2;
}
