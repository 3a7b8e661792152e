// consistency_pixel.h -- one pixel of checkConsistency (consistencyChecker/consistencyChecker.cpp:100-128), BIT-EXACT with the
// reference binary (x86-64 SSE2 scalar math, no FMA contraction).  Shared by consistency_kernel (kernels_consistency.hip) and the
// fused check + certainty + erosion tile kernel (kernels_frame.hip); both translation units are compiled with -ffp-contract=off:
// every fp32 / fp64 operation below is written with the exact promotions the C++ expressions of the reference imply and must
// round once per operation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fav {

__device__ __forceinline__ uint8_t consistency_pixel(const float2* f1, const float2* f2, const float* structure, const float* avg_ptr,
                                                     int ax, int ay, int W, int H)
{
    const size_t i = (size_t)ay * W + ax;
    const float2 fa = f1[i];                           // (u, v) of flow1 at a
    const float bx = (float)ax + fa.x;                 // :102
    const float by = (float)ay + fa.y;                 // :103
    // :104-109 `x1 < 0 || x2 >= W || y1 < 0 || y2 >= H` evaluated on the floats: the same predicate for every finite in-range
    // value (floor(b) < 0 <=> b < 0; floor(b) + 1 >= W <=> b >= W - 1), and for |flow| >= 2^31, +-inf and NaN -- where the
    // reference's cvttsd2si yields INT_MIN, i.e. "x1 < 0" -- it also gives 0 without the GPU's saturating conversion
    // (INT_MAX + 1 wraps, passes the integer test and gathers out of bounds).
    if (!(bx >= 0.f) || !(by >= 0.f) || bx >= (float)(W - 1) || by >= (float)(H - 1)) return 0;
    const int x1 = (int)floorf(bx), y1 = (int)floorf(by);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const float alphaX = bx - (float)x1, alphaY = by - (float)y1;             // :110
    const float2 p11 = f2[(size_t)y1 * W + x1], p21 = f2[(size_t)y1 * W + x2];
    const float2 p12 = f2[(size_t)y2 * W + x1], p22 = f2[(size_t)y2 * W + x2];
    const double omx = 1.0 - (double)alphaX, omy = 1.0 - (double)alphaY;
    // :111-116  float = double*float + float*float
    float a = (float)(omx * (double)p11.x + (double)(alphaX * p21.x));
    float b = (float)(omx * (double)p12.x + (double)(alphaX * p22.x));
    const float u = (float)(omy * (double)a + (double)(alphaY * b));
    a = (float)(omx * (double)p11.y + (double)(alphaX * p21.y));
    b = (float)(omx * (double)p12.y + (double)(alphaX * p22.y));
    const float v = (float)(omy * (double)a + (double)(alphaY * b));
    const float cx = bx + u, cy = by + v;                                     // :117-118
    const float u2 = fa.x, v2 = fa.y;
    float structureTerm = 0.f;
    if (structure != nullptr) {                                               // :122-124
        const float savg = *avg_ptr;
        const float h = savg / 2.0f - structure[i];
        structureTerm = 4.0f / savg * (h > 0.0f ? h : 0.0f);
    }
    const float ex = cx - (float)ax, ey = cy - (float)ay;
    const float lhs = ex * ex + ey * ey;
    const float mag = ((u2 * u2 + v2 * v2) + u * u) + v * v;
    const double rhs = (0.01 * (double)mag + (double)structureTerm) + (double)0.5f;   // :125
    return ((double)lhs >= rhs) ? 0 : 255;
}

}  // namespace fav
