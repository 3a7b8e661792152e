// warp_ref_launcher.hip -- TEST INFRASTRUCTURE ONLY (checker; never linked or loaded by the product).
//
// Builds the reference's OWN warp kernel for gfx950 so that A2 ("stn" semantics, the reference's GPU
// path) is pinned on reference-compiled code instead of on a restatement:
//   REF_KERNEL_FILE = lines 1-109 of /root/reference/stnbdhw/BilinearSamplerBDHW.cu (stride_t, getTopLeft,
//   between, toAddress, the __global__ kernel), cut out by oracle/Makefile at build time into a scratch file
//   that is deleted again after the compile -- the reference source is never copied into this repository.
//   Its `#include "utils.h"` (Torch/Lua glue, not used by those lines) resolves to an empty stub.
// Everything below line 109 of the reference file is the THC / Lua launcher (:111-200); the 12 lines here
// replace it with the same launch geometry (:119-120) and the strides a contiguous BDHW tensor reports
// (:124-137; BilinearSamplerBDHW.lua:26-42 asserts contiguity).
//
// Two builds (oracle/Makefile): libwarp_ref.so with the compiler's default floating-point contraction
// (what nvcc's default -fmad=true does to the reference build: CMakeLists.txt sets no flag) and
// libwarp_ref_nofma.so with -ffp-contract=off (the expression exactly as written, :103-106).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include REF_KERNEL_FILE

extern "C" int warp_ref_bdhw(const float* img, const float* flow, float* out, int B, int C, int H, int W, int Ho, int Wo,
                             void* stream)
{
    dim3 blocks(C, Ho * ((Wo + 511) / 512), B);      // BilinearSamplerBDHW.cu:119
    dim3 threads(32, 16);                            // :120
    const stride_t si{C * H * W, H * W, W, 1}, sg{2 * Ho * Wo, Ho * Wo, Wo, 1}, so{C * Ho * Wo, Ho * Wo, Wo, 1};
    hipLaunchKernelGGL(BilinearSamplerBDHW_bilinearSamplingFromGrid, blocks, threads, 0, (hipStream_t)stream,
                       const_cast<float*>(img), si, const_cast<float*>(flow), sg, out, so, C, H, W, Ho, Wo);
    return (int)hipGetLastError();
}
