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
// CMatrix::avg, an order-dependent fp32 running sum (CMatrix.h:1245-1251) evaluated exactly AND in parallel: inside one binade
// the running sum is an integer multiple of its ulp, so every addend is a two-state (parity) transducer and the chain is a scan of
// transducer compositions (avg_scan_kernel below; pinned against the scalar loop by test_sequential_sum_bit_exact).
#include "fav_internal.h"
#include "consistency_pixel.h"

namespace fav {
namespace {

__global__ __launch_bounds__(256) void consistency_kernel(const float2* f1, const float2* f2, const float* structure,
                                                          const float* avg_ptr, uint8_t* out, int W, int H)
{
    const int ay = blockIdx.y, ax = blockIdx.x * 256 + threadIdx.x;
    if (ax >= W) return;
    out[(size_t)ay * W + ax] = consistency_pixel(f1, f2, structure, avg_ptr, ax, ay, W, H);
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
__global__ __launch_bounds__(256) void iir_kernel(float* plane0, size_t plane_stride, float* scratch0, int nlines, int n,
                                                 int es, int ls, IIR c)
{
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
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
    // The loads are independent of the recurrence, the kernel is latency-bound (one wave per 64 lines, ~35 waves in all):
    // samples travel in groups of G, and the next group is already in flight while the chain runs over the current one.
    constexpr int G = 32;
    if (x + G <= n) {
        float cur[G], nxt[G];
#pragma unroll
        for (int q = 0; q < G; ++q) cur[q] = M_(x + q);
        for (; x + G <= n; x += G) {
            const bool more = x + 2 * G <= n;
            if (more) {
#pragma unroll
                for (int q = 0; q < G; ++q) nxt[q] = M_(x + G + q);
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const float a = c.k * (cur[q] + c.pm * mp) + c.a2 * a1 - c.e2 * a0;
                V1_(x + q) = a; a0 = a1; a1 = a; mp = cur[q];
            }
#pragma unroll
            for (int q = 0; q < G; ++q) cur[q] = nxt[q];
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
    if (xb - (G - 1) >= 0) {
        float cm[G], cv[G], nm[G], nv[G];
#pragma unroll
        for (int q = 0; q < G; ++q) { cm[q] = M_(xb - q); cv[q] = V1_(xb - q); }
        for (; xb - (G - 1) >= 0; xb -= G) {
            const bool more = xb - (2 * G - 1) >= 0;
            if (more) {
#pragma unroll
                for (int q = 0; q < G; ++q) { nm[q] = M_(xb - G - q); nv[q] = V1_(xb - G - q); }
            }
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const float bv = c.k * (c.pp * mo1 - c.e2 * mo2) + c.a2 * b0 - c.e2 * b1;
                M_(xb - q) = cv[q] + bv;
                b1 = b0; b0 = bv; mo2 = mo1; mo1 = cm[q];
            }
#pragma unroll
            for (int q = 0; q < G; ++q) { cm[q] = nm[q]; cv[q] = nv[q]; }
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

// CMatrix::avg: fp32 running sum in index order, then / size (CMatrix.h:1245-1251).  Every addition rounds, so the value
// depends on the order -- but not on more than that: while the running sum s stays inside one binade [2^k, 2^(k+1)) it is an
// integer multiple S of ulp = 2^(k-23), and adding x >= 0 gives S' = S + floor(q) + [frac(q) > 1/2] + [frac(q) == 1/2 and
// S + floor(q) odd] with q = x / ulp (round-to-nearest-even, q exact in fp64).  The only state an element needs from its
// predecessors is therefore the PARITY of S: each element is a 2-state transducer parity -> (increment, parity), and
// transducers compose associatively.  avg_scan_kernel (one 1024-thread block) evaluates the sum window by window (16384
// elements): per-thread composition of 16 transducers, a block-wide scan of the compositions (wave shuffles + one LDS hop),
// a replay with the true
// start parity that also finds the first element whose addition could leave the binade.  That element is added natively
// (one real fp32 add, so the rounding at the coarser ulp is the hardware's), and the next window starts behind it in the new
// binade.  Elements that break the premise (negative, non-finite, or the sum still zero / denormal) are added natively too;
// if a window makes little progress the rest of it runs as a plain sequential chain out of LDS, which bounds the worst case
// at the old one-wave chain's speed.  Bit-identical to the scalar loop for every input (tests: ties, mixed magnitudes,
// negatives, zeros); ~0.4 ms instead of 2.9 ms at 1280x720.
// Increments are only needed exactly while the sum stays inside the binade, i.e. below 2^24 - S <= 2^23: everything is
// 32-bit integer arithmetic with sums saturating at CAP (saturating addition of non-negative numbers is associative).
constexpr int XCAP = 1 << 28;
struct Xd { int d0, d1, pp; };       // saturated increment for start parity 0 / 1; pp bit 0 / 1 = final parity for start 0 / 1

__device__ __forceinline__ int sat_add(int a, int b) { const int r = a + b; return r < XCAP ? r : XCAP; }

// element x against a running sum with exponent field e: q = x / ulp(s) = m * 2^(ex - e) (m = x's 24-bit mantissa).
// f = floor(q) (XCAP if the element cannot be handled inside the binade: negative, non-finite, q >= 2^24), g = frac > 1/2,
// tie = frac == 1/2
__device__ __forceinline__ void elem_class(float x, int e, int& f, int& g, int& tie)
{
    const unsigned b = __float_as_uint(x);
    int ex = (int)(b >> 23) & 255;
    unsigned m = b & 0x7FFFFFu;
    if (ex) m |= 0x800000u; else ex = 1;                     // denormal: no hidden bit, exponent of 2^-126
    const int k = e - ex;                                    // q = m >> k
    const bool bad = (b >> 31) | (ex == 255) | (k < 0 && m != 0);
    const int ks = k < 0 ? 0 : (k > 25 ? 25 : k);            // k >= 25: q < 1/2
    const unsigned r = m & ((1u << ks) - 1u), half = ks ? 1u << (ks - 1) : 0xFFFFFFFFu;
    f = bad ? XCAP : (int)(m >> ks);
    g = r > half; tie = (r == half) & (k <= 24);
    if (k > 24) { g = 0; }
}

__device__ __forceinline__ Xd xd_then(const Xd& a, const Xd& b)      // composition "a, then b"
{
    Xd r;
    const int a0 = a.pp & 1, a1 = (a.pp >> 1) & 1;
    r.d0 = sat_add(a.d0, a0 ? b.d1 : b.d0);
    r.d1 = sat_add(a.d1, a1 ? b.d1 : b.d0);
    r.pp = ((b.pp >> a0) & 1) | (((b.pp >> a1) & 1) << 1);
    return r;
}

__device__ __forceinline__ Xd xd_shfl_up(const Xd& v, int off)
{
    Xd r; r.d0 = __shfl_up(v.d0, off); r.d1 = __shfl_up(v.d1, off); r.pp = __shfl_up(v.pp, off);
    return r;
}

__global__ __launch_bounds__(1024) void avg_scan_kernel(const float* v, int n, float* avg_out, float* sum_out)
{
    constexpr int NT = 1024, E = 16, WIN = NT * E, NW = NT / 64;
    __shared__ int wD[2][NW], wPP[NW];                       // per-wave totals, then their exclusive scan
    __shared__ __attribute__((aligned(16))) float sX[WIN];
    __shared__ int s_cross, s_i, s_stretch;
    __shared__ float s_sum;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const Xd ident = {0, 0, 2};                              // parity 0 -> 0, 1 -> 1
    if (t == 0) { s_sum = 0.f; s_i = 0; s_stretch = 256; }
    __syncthreads();
    for (;;) {
        const float s = s_sum; const int i = s_i;
        if (i >= n) break;
        __syncthreads();                                   // everyone has read the state
        if (t == 0) s_cross = WIN;
        const unsigned bits = __float_as_uint(s);
        const int e = (int)(bits >> 23) & 255;
        const bool ok = s > 0.f && e >= 1 && e <= 254;     // normal positive running sum
        const int M = (int)((bits & 0x7FFFFFu) | 0x800000u);
        const int limit = (1 << 24) - M;                   // an element may leave the binade if D + f + 1 >= limit
        const int base = i + t * E;
        float x[E];
        if (base + E <= n && (base & 3) == 0) {
#pragma unroll
            for (int j = 0; j < E; j += 4) { const float4 q = *reinterpret_cast<const float4*>(v + base + j); x[j] = q.x; x[j + 1] = q.y; x[j + 2] = q.z; x[j + 3] = q.w; }
        } else {
#pragma unroll
            for (int j = 0; j < E; ++j) x[j] = base + j < n ? v[base + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < E; j += 4) *reinterpret_cast<float4*>(&sX[t * E + j]) = make_float4(x[j], x[j + 1], x[j + 2], x[j + 3]);
        // pass 1: this thread's E elements as one transducer
        int d0 = 0, d1 = 0, p0 = 0, p1 = 1;
        int cls[E];                                         // f | g << 29 | tie << 30, reused by the replay
#pragma unroll
        for (int j = 0; j < E; ++j) {
            int f, g, tie; elem_class(x[j], e, f, g, tie);
            cls[j] = f | (g << 29) | (tie << 30);
            const int fo = f & 1;
            const int dl0 = f + g + (tie & (p0 ^ fo)), dl1 = f + g + (tie & (p1 ^ fo));
            d0 = sat_add(d0, dl0); p0 = (p0 + dl0) & 1;
            d1 = sat_add(d1, dl1); p1 = (p1 + dl1) & 1;
        }
        const Xd me = {d0, d1, p0 | (p1 << 1)};
        // inclusive scan inside the wave (shuffles), wave totals through LDS, exclusive scan of the 16 totals by wave 0
        Xd inc = me;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const Xd a = xd_shfl_up(inc, off); if (lane >= off) inc = xd_then(a, inc); }
        if (lane == 63) { wD[0][wave] = inc.d0; wD[1][wave] = inc.d1; wPP[wave] = inc.pp; }
        __syncthreads();
        if (wave == 0) {
            const bool in = lane < NW;
            Xd wi = ident;
            if (in) { wi.d0 = wD[0][lane]; wi.d1 = wD[1][lane]; wi.pp = wPP[lane]; }
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) { const Xd a = xd_shfl_up(wi, off); if (lane >= off) wi = xd_then(a, wi); }
            Xd ex = xd_shfl_up(wi, 1);                       // exclusive: the waves before this one
            if (lane == 0) ex = ident;
            if (in) { wD[0][lane] = ex.d0; wD[1][lane] = ex.d1; wPP[lane] = ex.pp; }
        }
        __syncthreads();
        // exclusive prefix of this thread for the true start parity
        const int par0 = M & 1;
        Xd pre = {wD[0][wave], wD[1][wave], wPP[wave]};      // the waves before
        {
            Xd exl = xd_shfl_up(inc, 1);                     // the lanes before, inside the wave
            if (lane == 0) exl = ident;
            pre = xd_then(pre, exl);
        }
        int drun = par0 ? pre.d1 : pre.d0;
        int p = (pre.pp >> par0) & 1;
        // pass 2: replay with the true parity; first element that may leave the binade (or breaks the premise)
        int mycross = E, dcross = drun;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int f = cls[j] & ((1 << 29) - 1), g = (cls[j] >> 29) & 1, tie = (cls[j] >> 30) & 1;
            const bool hit = mycross == E && (!ok || drun + f + 1 >= limit);
            if (hit) { mycross = j; dcross = drun; }
            const int dl = f + g + (tie & (p ^ (f & 1)));
            drun = sat_add(drun, dl); p = (p + dl) & 1;
        }
        if (mycross == E) dcross = drun;
        if (mycross < E) atomicMin(&s_cross, t * E + mycross);
        __syncthreads();
        const int jc = s_cross;                            // window-relative index of the first native element, WIN if none
        const int owner = jc < WIN ? jc / E : NT - 1;
        if (t == owner) {
            float sn = ok ? ldexpf((float)(M + dcross), e - 150) : s;      // exact: M + dcross < 2^24
            int in = i + WIN;
            if (jc < WIN) {
                // the native element; after a window with little progress (premise broken: zero / negative / non-finite
                // values) also a stretch of plain chain out of LDS, doubling while that keeps happening -- the worst case
                // degrades to the sequential chain, not below it
                int stretch = 1;
                if (jc < 512) { stretch = s_stretch; s_stretch = stretch < WIN ? stretch * 2 : WIN; } else s_stretch = 256;
                const int stop = jc + stretch < WIN ? jc + stretch : WIN;
                int q = jc;
                for (; q < stop && (q & 3); ++q) sn += sX[q];
#pragma unroll 4
                for (; q + 4 <= stop; q += 4) { const float4 w = *reinterpret_cast<const float4*>(&sX[q]); sn += w.x; sn += w.y; sn += w.z; sn += w.w; }
                for (; q < stop; ++q) sn += sX[q];
                in = i + stop;
            } else s_stretch = 256;
            s_sum = sn; s_i = in < n ? in : n;
        }
        __syncthreads();
    }
    if (t == 0) { const float s = s_sum; if (sum_out) *sum_out = s; if (avg_out) *avg_out = s / (float)n; }
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
    // One lane per line, so a pass is 34 / 60 waves of a 130-240 us dependent chain.  As blocks of ONE wave they spread over as many
    // CUs as are free at that moment, and each of them then keeps a whole-CU block of the network's persistent grids waiting; as
    // blocks of four waves (one per SIMD: the look-ahead register sets of the chain need more than 256 registers) they sit on 9 / 15 CUs
    static const int ib = getenv("FAV_IIR_BLOCK") ? std::max(64, std::min(256, atoi(getenv("FAV_IIR_BLOCK")) / 64 * 64)) : 256;      // (tuning: read once)
    hipLaunchKernelGGL(iir_kernel, dim3((H + ib - 1) / ib, 3), dim3(ib), 0, st, tmp3, ps, scratch, H, W, H, 1, c);
    hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (W + 31) / 32, 3), dim3(256), 0, st, tmp3, planes, ps, W, H);
    hipLaunchKernelGGL(iir_kernel, dim3((W + ib - 1) / ib, 3), dim3(ib), 0, st, planes, ps, scratch, W, H, W, 1, c);
    hipLaunchKernelGGL(eigen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, planes, planes + ps, planes + 2 * ps,
                       corners, n);
    hipLaunchKernelGGL(blockmax_kernel, dim3(nb), dim3(256), 0, st, corners, n, bmax);
    hipLaunchKernelGGL(prefixmax_kernel, dim3(1), dim3(1024), 0, st, bmax, nb, bpre, mm);
    hipLaunchKernelGGL(quirkmin_kernel, dim3(nb), dim3(256), 0, st, corners, n, bpre, bmin);
    hipLaunchKernelGGL(minreduce_kernel, dim3(1), dim3(256), 0, st, bmin, nb, mm);
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, corners, n, mm);
    hipLaunchKernelGGL(avg_scan_kernel, dim3(1), dim3(1024), 0, st, corners, (int)n, mm + 2, static_cast<float*>(nullptr));
    FAV_LAUNCH_CHECK("structure kernels");
    *structure_out = corners;
    *avg_out = mm + 2;
    return FAV_OK;
}

int launch_sequential_sum(const float* x, size_t n, float* sum_out, hipStream_t st)
{
    FAV_REQUIRE(n > 0 && n < (1ull << 31), "sequential sum: bad length");
    hipLaunchKernelGGL(avg_scan_kernel, dim3(1), dim3(1024), 0, st, x, (int)n, static_cast<float*>(nullptr), sum_out);
    FAV_LAUNCH_CHECK("avg_scan_kernel");
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
