// Diagnostic microbenchmark (not part of the product): how much does each kind of side work inside an MFMA loop cost on
// gfx950?  Loop = the halo kernel's K step (64 x v_mfma_f32_32x32x2f32 per wave, 20 ds_read_b128), 8 waves per CU.
// FLAGS: 1 = ~64 VALU ops / step, 2 = 3 ds_write_b128 / step, 4 = 3 global_load_dwordx4 / step (L2-resident),
//        8 = one s_barrier / step, 16 = VALU work placed after the MFMAs instead of between them
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(512, 2) void mix_kernel(float* out, const float* __restrict__ src, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NSM = (2 * 340 + 3 * 128) * 36;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < NSM; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        sm[i] = (float)(h & 0xffffff) / 8388608.f - 1.f;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* a_base = sm + (wave * 34 + (lane & 31)) * 36 + (lane >> 5) * 4;
    const float* b_base = sm + (680 + (lane & 31)) * 36 + (lane >> 5) * 4;
    float* wdst = sm + (680 + 256 + (t >> 3)) * 36 + (t & 7) * 4;             // third weight buffer: never read
    const float* g = src + (size_t)(blockIdx.x & 63) * 16384 + t * 4;
    v4f gl[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float va = (float)t, vb = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        const int tog = (it & 1) * 36;
        if (F & 4) {
#pragma unroll
            for (int q = 0; q < 3; ++q) gl[q] = *reinterpret_cast<const v4f*>(g + q * 2048 + (it & 3) * 8192 / 4);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const v4f af = *reinterpret_cast<const v4f*>(a_base + kk * 8 + tog);
            v4f bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const v4f*>(b_base + j * 32 * 36 + kk * 8 + tog);
            if ((F & 1) && !(F & 16)) {
#pragma unroll
                for (int q = 0; q < 16; ++q) va = fmaf(va, vb, 0.5f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[j].w, acc[j], 0, 0, 0);
            }
            if ((F & 1) && !(F & 16)) {
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
            }
            if ((F & 8) && kk == 1) __syncthreads();
        }
        if ((F & 1) && (F & 16)) {
#pragma unroll
            for (int q = 0; q < 64; ++q) va = fmaf(va, vb, 0.5f);
        }
        if (F & 2) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { v4f w = gl[q]; w.x += va; *reinterpret_cast<v4f*>(wdst + q * 64 * 36) = w; }
        }
    }
    float s = va + gl[0].x + gl[1].y + gl[2].z;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[t] = s;
}

template <int F>
static void run(const char* name, float* out, const float* src)
{
    const int grid = 248, iters = 1500, reps = 5;
    const size_t lds = (size_t)(2 * 340 + 3 * 128) * 36 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(mix_kernel<F>, dim3(grid), dim3(512), lds, 0, out, src, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mix_kernel<F>, dim3(grid), dim3(512), lds, 0, out, src, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us_step = ms * 1e3 / reps / iters;
    printf("%-52s %6.3f us/step  %6.1f TFLOP/s  (%s)\n", name, us_step, 248.0 * 8 * 64 * 4096 / (us_step * 1e-6) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main()
{
    float* out; float* src; hipMalloc(&out, 4096); hipMalloc(&src, 64 * 16384 * 4 + 65536); hipMemset(src, 0, 64 * 16384 * 4 + 65536);
    run<0>("mfma + 20 ds_read_b128", out, src);
    run<1>("+ 64 VALU between the MFMAs", out, src);
    run<17>("+ 64 VALU after the MFMAs", out, src);
    run<2>("+ 3 ds_write_b128", out, src);
    run<4>("+ 3 global_load_dwordx4", out, src);
    run<6>("+ 3 global loads -> 3 ds_write", out, src);
    run<8>("+ barrier", out, src);
    run<14>("+ loads, writes, barrier", out, src);
    run<15>("+ loads, writes, barrier, VALU between", out, src);
    run<31>("+ loads, writes, barrier, VALU after", out, src);
    return 0;
}
