// Diagnostic microbenchmark (not part of the product): what fp32 MFMA rate does an MI355X sustain in
//  (a) a pure v_mfma_f32_32x32x2f32 loop, (b) the same loop fed by the halo kernel's LDS read pattern
// at the halo kernel's launch shape (248 blocks x 512 threads / 496 x 256) and burst length (~150 us) and for a long run.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int LDS>
__global__ __launch_bounds__(512, 2) void peak_kernel(float* out, int iters, float a0)
{
    __shared__ __attribute__((aligned(16))) float sm[(340 + 256) * 36];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < (340 + 256) * 36; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        sm[i] = LDS == 2 ? ((float)(h & 0xffffff) / 8388608.f - 1.f) : a0 * (float)(i & 15);      // LDS==2: random operands in [-1,1)
    }
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    v4f af = {a0, a0 + 1.f, a0 + 2.f, a0 + 3.f};
    v4f bf[4] = {af, af * 2.f, af * 3.f, af * 4.f};
    const float* a_base = sm + ((wave % 8) * 34 + (lane & 31)) * 36 + (lane >> 5) * 4;
    const float* b_base = sm + (340 + (lane & 31)) * 36 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (LDS) {
                af = *reinterpret_cast<const v4f*>(a_base + kk * 8 + (it & 1) * 36);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const v4f*>(b_base + j * 32 * 36 + kk * 8 + (it & 1) * 36);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[j].w, acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[t] = s;
}

template <int LDS>
static void run(const char* name, int grid, int block, int iters, int reps)
{
    float* out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(peak_kernel<LDS>, dim3(grid), dim3(block), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(peak_kernel<LDS>, dim3(grid), dim3(block), 0, 0, out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * (block / 64) * iters * 64.0 * 4096.0 * reps;
    printf("%-34s grid=%4d block=%3d iters=%6d reps=%3d  %8.1f us/launch  %7.1f TFLOP/s\n", name, grid, block, iters, reps, ms * 1e3 / reps, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    // 64 MFMAs of 64 cycles per iteration per wave; 2 waves per SIMD -> 8192 cycles per iteration per CU
    run<0>("pure mfma, burst", 248, 512, 33, 50);
    run<1>("mfma + lds reads, burst", 248, 512, 33, 50);
    run<0>("pure mfma, burst, 256 CUs", 256, 512, 33, 50);
    run<1>("mfma + lds reads, burst, 2x4 waves", 496, 256, 33, 50);
    run<0>("pure mfma, sustained", 248, 512, 3300, 10);
    run<1>("mfma + lds reads, sustained", 248, 512, 3300, 10);
    run<0>("pure mfma, sustained, 256 CUs", 256, 512, 3300, 10);
    run<2>("random operands + lds, burst", 248, 512, 33, 50);
    run<2>("random operands + lds, sustained", 248, 512, 3300, 10);
    run<2>("random operands + lds, sust., 256", 256, 512, 3300, 10);
    run<1>("constant operands + lds, sust., 256", 256, 512, 3300, 10);
    return 0;
}
