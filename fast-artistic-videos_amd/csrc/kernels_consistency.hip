// kernels_consistency.hip -- forward-backward flow consistency mask on gfx950, BIT-EXACT with the
// reference's consistencyChecker binary (consistencyChecker/consistencyChecker.cpp:80-134, built by
// consistencyChecker/Makefile:2 for x86-64: SSE2 scalar math, no FMA contraction).
//
// This translation unit MUST be compiled with -ffp-contract=off: every fp32/fp64 operation below
// is written with the exact promotions the C++ expressions of the reference imply, and must round
// once per operation.
//
// 3-argument mode: one lane per pixel, coalesced float2 reads of flow1, four float2 gathers of
// flow2 (L2-served: the flow is locally smooth), one byte written.  17 algorithmic bytes / pixel.
// 4-argument mode adds the image-structure term (computeCorners :39-78): gradient + second-moment
// (parallel), the two recursive smoothing passes (one lane per row / per column, sequential along
// the line exactly like CFilter.h:1416-1464), eigenvalue, CMatrix::normalize with its order
// dependent min/max quirk (CMatrix.h:721-737, reproduced with an exact parallel formulation) and
// CMatrix::avg, an order-dependent fp32 running sum (CMatrix.h:1245-1251) reproduced by a single
// wave adding in index order.
#include "fav_internal.h"

namespace fav {
namespace {

__global__ __launch_bounds__(256) void consistency_kernel(const float2* f1, const float2* f2, const float* structure,
                                                          const float* avg_ptr, uint8_t* out, int W, int H)
{
    const int ay = blockIdx.y, ax = blockIdx.x * 256 + threadIdx.x;
    if (ax >= W) return;
    const size_t i = (size_t)ay * W + ax;
    const float2 fa = f1[i];                           // (u, v) of flow1 at a
    const float bx = (float)ax + fa.x;                 // :102
    const float by = (float)ay + fa.y;                 // :103
    const int x1 = (int)floorf(bx), y1 = (int)floorf(by);
    const int x2 = x1 + 1, y2 = y1 + 1;
    if (x1 < 0 || x2 >= W || y1 < 0 || y2 >= H) { out[i] = 0; return; }       // :108-109
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
    out[i] = ((double)lhs >= rhs) ? 0 : 255;
}

// ------------------------------------------------------------------------------------------------
// 4-argument mode (structure map)
// ------------------------------------------------------------------------------------------------
struct IIR { float k, pm, pp, e2, a2; };

// gradient [-0.5,0,0.5] with edge-repeating mirror (CFilter.h:600-611,1499-1578), second-moment sums
// over the 3 colour planes in plane order (consistencyChecker.cpp:54-60)
__global__ __launch_bounds__(256) void moments_kernel(const uint8_t* rgb_hwc, float* dxx, float* dyy, float* dxy, int W, int H)
{
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int xm = x - 1 < 0 ? 0 : x - 1, xp = x + 1 >= W ? W - 1 : x + 1;
    const int ym = y - 1 < 0 ? 0 : y - 1, yp = y + 1 >= H ? H - 1 : y + 1;
    float sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int c = 0; c < 3; ++c) {
        const float l = (float)rgb_hwc[((size_t)y * W + xm) * 3 + c], r = (float)rgb_hwc[((size_t)y * W + xp) * 3 + c];
        const float up = (float)rgb_hwc[((size_t)ym * W + x) * 3 + c], dn = (float)rgb_hwc[((size_t)yp * W + x) * 3 + c];
        const float mid = (float)rgb_hwc[((size_t)y * W + x) * 3 + c];
        float dx = 0.f; dx += -0.5f * l; dx += 0.0f * mid; dx += 0.5f * r;
        float dy = 0.f; dy += -0.5f * up; dy += 0.0f * mid; dy += 0.5f * dn;
        sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
    }
    const size_t i = (size_t)y * W + x;
    dxx[i] = sxx; dyy[i] = syy; dxy[i] = sxy;
}

// one lane per line; `n` samples with element stride `es`, line stride `ls`; scratch holds v1.
// Arithmetic order exactly as CFilter.h:1426-1437 / 1451-1462.
__global__ __launch_bounds__(64) void iir_kernel(float* plane0, size_t plane_stride, float* scratch0, int nlines, int n,
                                                 int es, int ls, IIR c)
{
    const int line = blockIdx.x * 64 + threadIdx.x;
    if (line >= nlines || n < 2) return;
    float* m = plane0 + (size_t)blockIdx.y * plane_stride + (size_t)line * ls;
    float* v1 = scratch0 + (size_t)blockIdx.y * plane_stride + (size_t)line * ls;
#define M_(i) m[(size_t)(i) * es]
#define V1_(i) v1[(size_t)(i) * es]
    float m0 = M_(0), m1 = M_(1);
    float a0 = (0.5f - c.k * c.pm) * m0;
    float a1 = c.k * (m1 + c.pm * m0) + (c.a2 - c.e2) * a0;
    V1_(0) = a0; V1_(1) = a1;
    float mp = m1;
    int x = 2;
    for (; x + 8 <= n; x += 8) {          // loads are independent of the recurrence: fetch 8 samples, then run the chain
        float mv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) mv[q] = M_(x + q);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float a = c.k * (mv[q] + c.pm * mp) + c.a2 * a1 - c.e2 * a0;
            V1_(x + q) = a; a0 = a1; a1 = a; mp = mv[q];
        }
    }
    for (; x < n; ++x) {
        const float mx = M_(x);
        const float a = c.k * (mx + c.pm * mp) + c.a2 * a1 - c.e2 * a0;
        V1_(x) = a; a0 = a1; a1 = a; mp = mx;
    }
    // backward sweep: keep the ORIGINAL m(x+1), m(x+2) in registers while m is overwritten
    const float ml = M_(n - 1);
    float b1 = (0.5f + c.k * c.pm) * ml;                                   // v2(n-1)
    float b0 = c.k * ((c.pp - c.e2) * ml) + (c.a2 - c.e2) * b1;            // v2(n-2)
    float mo1 = M_(n - 2);                                                 // original m(n-2)
    float mo2 = ml;                                                        // original m(n-1)
    M_(n - 1) = V1_(n - 1) + b1;
    M_(n - 2) = V1_(n - 2) + b0;
    // now b0 = v2(x+1), b1 = v2(x+2) for x = n-3; mo1 = m(x+1), mo2 = m(x+2)
    int xb = n - 3;
    for (; xb - 7 >= 0; xb -= 8) {
        float mv[8], vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { mv[q] = M_(xb - q); vv[q] = V1_(xb - q); }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float bv = c.k * (c.pp * mo1 - c.e2 * mo2) + c.a2 * b0 - c.e2 * b1;
            M_(xb - q) = vv[q] + bv;
            b1 = b0; b0 = bv; mo2 = mo1; mo1 = mv[q];
        }
    }
    for (; xb >= 0; --xb) {
        const float mx = M_(xb);
        const float bv = c.k * (c.pp * mo1 - c.e2 * mo2) + c.a2 * b0 - c.e2 * b1;
        M_(xb) = V1_(xb) + bv;
        b1 = b0; b0 = bv; mo2 = mo1; mo1 = mx;
    }
#undef M_
#undef V1_
}

__global__ __launch_bounds__(256) void eigen_kernel(const float* dxx, const float* dyy, const float* dxy, float* corners,
                                                    size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = dxx[i], b = dxy[i], c = dyy[i];
    const float temp = (float)(0.5 * (double)(a + c));                      // consistencyChecker.cpp:73
    const float temp2 = temp * temp + b * b - a * c;
    corners[i] = temp2 < 0.0f ? 0.0f : temp - sqrtf(temp2);
}

// CMatrix::normalize's scan (CMatrix.h:727-729):
//     if (v > cmax) cmax = v; else if (v < cmin) cmin = v;        cmax0 = -30000, cmin0 = +30000
// cmax = max(-30000, max_i v_i).  An element updates cmin only when it is NOT a strict running
// maximum, i.e. v_i <= max(-30000, v_0..v_{i-1}).  Both are computed exactly in parallel:
// pass 1: per-block maxima; pass 2: exclusive prefix max over blocks (single block) then, per block,
// an in-block exclusive running max and the min over non-record elements.
constexpr int NB = 1024;   // elements per block in the normalize scans

__global__ __launch_bounds__(256) void blockmax_kernel(const float* v, size_t n, float* bmax)
{
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * NB;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < NB; j += 256)
        if (base + j < n) m = fmaxf(m, v[base + j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) bmax[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// single block: exclusive prefix max of bmax (seeded with -30000) -> bpre; total max -> mm[0]
__global__ __launch_bounds__(1024) void prefixmax_kernel(const float* bmax, int nb, float* bpre, float* mm)
{
    __shared__ float sh[1024];
    float carry = -30000.0f;
    for (int base = 0; base < nb; base += 1024) {
        const int j = base + threadIdx.x;
        const float own = j < nb ? bmax[j] : -INFINITY;
        sh[threadIdx.x] = own;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {           // inclusive Hillis-Steele max scan
            const float t = threadIdx.x >= o ? sh[threadIdx.x - o] : -INFINITY;
            __syncthreads();
            sh[threadIdx.x] = fmaxf(sh[threadIdx.x], t);
            __syncthreads();
        }
        const float excl = threadIdx.x > 0 ? fmaxf(carry, sh[threadIdx.x - 1]) : carry;
        if (j < nb) bpre[j] = excl;
        const float tot = fmaxf(carry, sh[1023]);
        __syncthreads();
        carry = tot;
    }
    if (threadIdx.x == 0) mm[0] = carry;              // cmax
}

__global__ __launch_bounds__(256) void quirkmin_kernel(const float* v, size_t n, const float* bpre, float* bmin)
{
    __shared__ float sh[NB];
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * NB;
    for (int j = threadIdx.x; j < NB; j += 256) sh[j] = base + j < n ? v[base + j] : -INFINITY;
    __syncthreads();
    // each thread owns 4 consecutive elements; running max of everything before them
    const int j0 = threadIdx.x * 4;
    float pre = -INFINITY;
    for (int j = 0; j < j0; ++j) pre = fmaxf(pre, sh[j]);   // small (<= 1020 LDS reads); exactness over speed
    pre = fmaxf(pre, bpre[blockIdx.x]);
    float mn = 30000.0f;
    for (int j = j0; j < j0 + 4; ++j) {
        if (base + j < n) {
            const float x = sh[j];
            if (x > pre) pre = x; else if (x < mn) mn = x;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) bmin[blockIdx.x] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void minreduce_kernel(const float* bmin, int nb, float* mm)
{
    __shared__ float red[4];
    float mn = 30000.0f;
    for (int j = threadIdx.x; j < nb; j += 256) mn = fminf(mn, bmin[j]);
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) mm[1] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));   // cmin
}

__global__ __launch_bounds__(256) void normalize_kernel(float* v, size_t n, const float* mm)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float cmax = mm[0], cmin = mm[1];
    float t = cmax - cmin;
    if (t == 0.f) t = 1.f; else t = (1.0f - 0.0f) / t;
    float x = v[i];
    x -= cmin; x *= t; x += 0.0f;
    v[i] = x;
}

// CMatrix::avg: fp32 running sum in index order, then / size (CMatrix.h:1245-1251).  The additions are inherently
// sequential (each one rounds).  One wave runs the chain: 2048-element chunks are staged into LDS with coalesced loads
// (prefetched a chunk ahead in registers), and the chain reads them back as broadcast float4s, so the only dependent
// instruction per element is the v_add itself.  ~1.7 ms for 1280x720; it runs on the stream's side queue, overlapped with
// the network of the previous frame.
__global__ __launch_bounds__(64) void avg_kernel(const float* v, int n, float* avg_out)
{
    constexpr int CH = 2048;                        // elements per chunk (8 float4 per lane)
    __shared__ __attribute__((aligned(16))) float buf[2][CH];
    const int lane = threadIdx.x;
    float acc = 0.f;
    const int nfull = n / CH;
    const float4* v4 = reinterpret_cast<const float4*>(v);      // v is 256-byte aligned (workspace carve)
    float4 r[8];
    if (nfull > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = v4[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&buf[0][(i * 64 + lane) * 4]) = r[i];
    }
    int cur = 0;
    for (int c = 0; c < nfull; ++c) {
        const bool more = c + 1 < nfull;
        if (more) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = v4[(size_t)(c + 1) * (CH / 4) + i * 64 + lane];
        }
        const float4* b4 = reinterpret_cast<const float4*>(buf[cur]);
#pragma unroll 16
        for (int i = 0; i < CH / 4; ++i) {
            const float4 q = b4[i];                 // same address in every lane: LDS broadcast
            acc += q.x; acc += q.y; acc += q.z; acc += q.w;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&buf[cur ^ 1][(i * 64 + lane) * 4]) = r[i];
        }
        cur ^= 1;
    }
    for (int base = nfull * CH; base < n; base += 64) {
        const float x = base + lane < n ? v[base + lane] : 0.f;
        const int m = n - base < 64 ? n - base : 64;
        for (int j = 0; j < m; ++j) acc += __shfl(x, j);
    }
    if (lane == 0) *avg_out = acc / (float)n;
}

// recursiveSmoothX through LDS: a block owns 64 rows; 64x64 tiles are moved with coalesced row-major accesses and each
// lane walks ITS row inside the tile (LDS pitch 65: conflict-free column walk), so the sequential recurrences of
// CFilter.h:1426-1437 keep their exact order while global memory sees full lines.
// [R][C] -> [C][R] per plane (32x32 LDS tiles, both sides coalesced); lets the X smoothing pass run as the coalesced
// one-lane-per-line kernel on the transposed planes without changing a single floating-point operation
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, size_t plane_stride, int R, int C)
{
    __shared__ float tl[32][33];
    const float* ip = in + (size_t)blockIdx.z * plane_stride;
    float* op = out + (size_t)blockIdx.z * plane_stride;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tl[j][tx] = ip[(size_t)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) op[(size_t)(c0 + j) * R + r0 + tx] = tl[tx][j];
}

__global__ __launch_bounds__(256) void iir_x_kernel(float* plane0, size_t plane_stride, float* scratch0, int H, int W, IIR c)
{
    __shared__ float tile[64 * 65];
    __shared__ float tv1[64 * 65];
    const int t = threadIdx.x, lane = t & 63, tr = t >> 6;        // 4 waves move the tiles, wave 0 runs the recurrences
    const int row0 = blockIdx.x * 64;
    const int nrows = min(64, H - row0);
    float* mp = plane0 + (size_t)blockIdx.y * plane_stride;
    float* v1p = scratch0 + (size_t)blockIdx.y * plane_stride;
    const bool act = t < nrows;                                   // t < 64: this lane owns row t
    if (W < 2) return;
    float a0 = 0.f, a1 = 0.f, mprev = 0.f;
    // forward sweep: v1
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int nc = min(64, W - x0);
        for (int r = tr; r < nrows; r += 4)
            if (lane < nc) tile[r * 65 + lane] = mp[(size_t)(row0 + r) * W + x0 + lane];
        __syncthreads();
        if (act)
#pragma unroll 8
            for (int j = 0; j < nc; ++j) {
                const int x = x0 + j;
                const float mx = tile[t * 65 + j];
                float a;
                if (x == 0) a = (0.5f - c.k * c.pm) * mx;
                else if (x == 1) a = c.k * (mx + c.pm * mprev) + (c.a2 - c.e2) * a1;
                else a = c.k * (mx + c.pm * mprev) + c.a2 * a1 - c.e2 * a0;
                a0 = a1; a1 = a; mprev = mx;
                tile[t * 65 + j] = a;
            }
        __syncthreads();
        for (int r = tr; r < nrows; r += 4)
            if (lane < nc) v1p[(size_t)(row0 + r) * W + x0 + lane] = tile[r * 65 + lane];
        __syncthreads();
    }
    // backward sweep: v2, m = v1 + v2 (the original m(x+1), m(x+2) are carried in registers)
    float b0 = 0.f, b1 = 0.f, mo1 = 0.f, mo2 = 0.f;
    const int last_x0 = ((W - 1) / 64) * 64;
    for (int x0 = last_x0; x0 >= 0; x0 -= 64) {
        const int nc = min(64, W - x0);
        for (int r = tr; r < nrows; r += 4)
            if (lane < nc) {
                tile[r * 65 + lane] = mp[(size_t)(row0 + r) * W + x0 + lane];
                tv1[r * 65 + lane] = v1p[(size_t)(row0 + r) * W + x0 + lane];
            }
        __syncthreads();
        if (act)
#pragma unroll 8
            for (int j = nc - 1; j >= 0; --j) {
                const int x = x0 + j;
                const float mx = tile[t * 65 + j];
                float bv;
                if (x == W - 1) bv = (0.5f + c.k * c.pm) * mx;
                else if (x == W - 2) bv = c.k * ((c.pp - c.e2) * mo1) + (c.a2 - c.e2) * b0;
                else bv = c.k * (c.pp * mo1 - c.e2 * mo2) + c.a2 * b0 - c.e2 * b1;
                tile[t * 65 + j] = tv1[t * 65 + j] + bv;
                b1 = b0; b0 = bv; mo2 = mo1; mo1 = mx;
            }
        __syncthreads();
        for (int r = tr; r < nrows; r += 4)
            if (lane < nc) mp[(size_t)(row0 + r) * W + x0 + lane] = tile[r * 65 + lane];
        __syncthreads();
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

size_t structure_workspace_bytes(int W, int H)
{
    const size_t n = (size_t)W * H;
    const size_t nb = (n + NB - 1) / NB;
    // 3 planes + 3 scratch planes + corners + 3 transposed planes + bmax + bpre + bmin + mm(2) + avg(1)
    return align_up(n * 4, 256) * 10 + align_up(nb * 4, 256) * 3 + 256;
}

static void iir_constants(float sigma, IIR& c)
{
    // CFilter.h:1419-1425; NMath::Pi is a float (NMath.cpp:9); host libm (same as the reference)
    const float Pi = 3.1415926536f;
    const float aAlpha = (float)(2.5 / (double)(sqrtf(Pi) * sigma));
    const float aExp = expf(-aAlpha);
    const float aExpSqr = aExp * aExp;
    c.a2 = (float)(2.0 * (double)aExp);
    c.k = (float)((1.0 - (double)aExp) * (1.0 - (double)aExp) / (1.0 + 2.0 * (double)aAlpha * (double)aExp - (double)aExpSqr));
    c.pm = (float)((double)aExp * ((double)aAlpha - 1.0));
    c.pp = (float)((double)aExp * ((double)aAlpha + 1.0));
    c.e2 = aExpSqr;
}

int launch_structure(const uint8_t* rgb_hwc, int W, int H, void* ws, size_t ws_bytes, const float** structure_out,
                     const float** avg_out, hipStream_t st)
{
    FAV_REQUIRE(ws != nullptr && ws_bytes >= structure_workspace_bytes(W, H), "consistency: workspace too small");
    FAV_REQUIRE(W >= 2 && H >= 2, "consistency: structure mode needs W,H >= 2");
    const size_t n = (size_t)W * H, ps = align_up(n * 4, 256) / 4;
    const int nb = (int)((n + NB - 1) / NB);
    float* planes = static_cast<float*>(ws);          // dxx, dyy, dxy
    float* scratch = planes + 3 * ps;
    float* corners = scratch + 3 * ps;
    float* tmp3 = corners + ps;
    float* bmax = tmp3 + 3 * ps;
    float* bpre = bmax + align_up((size_t)nb * 4, 256) / 4;
    float* bmin = bpre + align_up((size_t)nb * 4, 256) / 4;
    float* mm = bmin + align_up((size_t)nb * 4, 256) / 4;   // [0]=cmax [1]=cmin [2]=avg
    IIR c; iir_constants(3.0f, c);                          // main(): computeCorners(image, &structure, 3.0f)
    const dim3 g2((W + 255) / 256, H);
    hipLaunchKernelGGL(moments_kernel, g2, dim3(256), 0, st, rgb_hwc, planes, planes + ps, planes + 2 * ps, W, H);
    // recursiveSmoothX then Y on dxx, dyy, dxy (:62-67); planes are independent => blockIdx.y = plane
    // recursiveSmoothX: transpose -> one lane per (former) row walking coalesced memory -> transpose back
    hipLaunchKernelGGL(transpose_kernel, dim3((W + 31) / 32, (H + 31) / 32, 3), dim3(256), 0, st, planes, tmp3, ps, H, W);
    hipLaunchKernelGGL(iir_kernel, dim3((H + 63) / 64, 3), dim3(64), 0, st, tmp3, ps, scratch, H, W, H, 1, c);
    hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (W + 31) / 32, 3), dim3(256), 0, st, tmp3, planes, ps, W, H);
    hipLaunchKernelGGL(iir_kernel, dim3((W + 63) / 64, 3), dim3(64), 0, st, planes, ps, scratch, W, H, W, 1, c);
    hipLaunchKernelGGL(eigen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, planes, planes + ps, planes + 2 * ps,
                       corners, n);
    hipLaunchKernelGGL(blockmax_kernel, dim3(nb), dim3(256), 0, st, corners, n, bmax);
    hipLaunchKernelGGL(prefixmax_kernel, dim3(1), dim3(1024), 0, st, bmax, nb, bpre, mm);
    hipLaunchKernelGGL(quirkmin_kernel, dim3(nb), dim3(256), 0, st, corners, n, bpre, bmin);
    hipLaunchKernelGGL(minreduce_kernel, dim3(1), dim3(256), 0, st, bmin, nb, mm);
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, corners, n, mm);
    hipLaunchKernelGGL(avg_kernel, dim3(1), dim3(64), 0, st, corners, (int)n, mm + 2);
    FAV_LAUNCH_CHECK("structure kernels");
    *structure_out = corners;
    *avg_out = mm + 2;
    return FAV_OK;
}

int launch_consistency(const float* f1_flo, const float* f2_flo, const float* structure, const float* avg, uint8_t* out,
                       int W, int H, hipStream_t st)
{
    hipLaunchKernelGGL(consistency_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st,
                       reinterpret_cast<const float2*>(f1_flo), reinterpret_cast<const float2*>(f2_flo), structure, avg, out,
                       W, H);
    FAV_LAUNCH_CHECK("consistency_kernel");
    return FAV_OK;
}

}  // namespace fav
