// kernels_vr.hip -- elementwise / gather kernels of the 360-degree cube-map orchestration:
//   fast_artistic_video_vr.lua:130-152 (rotations, combineSides), :204-237 (border certainty), :239-302 (border prior and
//   its blend with the flow-warped previous face), :454-509 (post-blend of the six faces), :511-559 (median, cube map,
//   equirectangular input strip); utils.lua:151-159 (median filter); core.lua:108-117,161-171 (input assembly with a
//   ready-made prior and the uniform-random occlusion fill).
// All of it is HBM-bound planar fp32 [3][H][W] work, one lane per pixel, consecutive lanes on consecutive x.  The
// perspective / equirectangular warps themselves are the A2 kernel (kernels_frame.hip: warp_kernel) fed with static maps.
// Compiled with -ffp-contract=off: every expression rounds like the CPU restatement (oracle/vr_oracle.py).
#include "fav_internal.h"

namespace fav {
namespace {

// dst[c][i][j] of the rotated image (fast_artistic_video_vr.lua:130-144); src is [3][H][W]
//   1: rotate90      -> [3][W][H], dst[i][j] = src[j][W-1-i]
//   2: rotateMinus90 -> [3][W][H], dst[i][j] = src[H-1-j][i]
//   3: rotate180     -> [3][H][W], dst[i][j] = src[H-1-i][W-1-j]
__device__ __forceinline__ size_t rot_src(int mode, int i, int j, int H, int W)
{
    if (mode == 1) return (size_t)j * W + (W - 1 - i);
    if (mode == 2) return (size_t)(H - 1 - j) * W + i;
    if (mode == 3) return (size_t)(H - 1 - i) * W + (W - 1 - j);
    return (size_t)i * W + j;
}

__global__ __launch_bounds__(256) void rotate_kernel(const float* src, float* dst, int H, int W, int mode)
{
    const int DH = (mode == 1 || mode == 2) ? W : H, DW = (mode == 1 || mode == 2) ? H : W;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= DW) return;
    const size_t n = (size_t)H * W, s = rot_src(mode, i, j, H, W), d = (size_t)i * DW + j;
    (void)DH;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * n + d] = src[c * n + s];
}

// acc = (first ? 0 : acc) + (div ? w / div : w)        -- combineSides :146-152 / border sums :262-279
__global__ __launch_bounds__(256) void accum_kernel(float* acc, const float* w, const float* div, size_t n, int first)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float dv = div ? div[i] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = div ? w[c * n + i] / dv : w[c * n + i];
        acc[c * n + i] = first ? v : acc[c * n + i] + v;
    }
}

// func_load_cert :204-237: border certainty = max of the masks of the already stylised neighbours; max with the frame's mask
__global__ __launch_bounds__(256) void vr_cert_kernel(const uint8_t* cert_u8, const float* m0, const float* m1, const float* m2,
                                                      const float* m3, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float b = 0.f;
    if (m0) b = fmaxf(b, m0[i]);
    if (m1) b = fmaxf(b, m1[i]);
    if (m2) b = fmaxf(b, m2[i]);
    if (m3) b = fmaxf(b, m3[i]);
    out[i] = cert_u8 ? fmaxf((float)cert_u8[i] / 255.f, b) : b;       // image.load(…, 1): byte / 255
}

// :281-293: mask = max(g, ceil(g) * (1 - cert)) * m;  out = warped * (1 - mask) + border * mask
__global__ __launch_bounds__(256) void vr_prior_kernel(const float* lfw, const float* border, const float* grad, const float* cert,
                                                       const float* m, const float* m2, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i], ci = 1.f - cert[i];
    const float mk = m2 ? m[i] + m2[i] : m[i];
    const float mask = fmaxf(g, ceilf(g) * ci) * mk, anti = 1.f - mask;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * n + i] = lfw[c * n + i] * anti + border[c * n + i] * mask;
}

// blend_other_sides :454-509: out = seg * anti + borders * g   (anti = fp32(1 - g) computed by the host in double)
__global__ __launch_bounds__(256) void vr_blend_kernel(const float* seg, const float* borders, const float* g, const float* anti,
                                                       float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * n + i] = seg[c * n + i] * anti[i] + borders[c * n + i] * g[i];
}

// utils.median_filter (utils.lua:151-159): r x r windows, no padding -> [3][H-r+1][W-r+1], lower median.
// r = 3 (the reference's default): 19-exchange min/max network on registers (exact value of the 5th smallest);
// r = 5: partial selection sort (rare).
#define FAV_CSWAP(a, b) { const float lo_ = fminf(a, b); b = fmaxf(a, b); a = lo_; }
__global__ __launch_bounds__(256) void median3_kernel(const float* src, float* dst, int H, int W)
{
    const int OW = W - 2, OH = H - 2;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    if (x >= OW) return;
    const float* p = src + ((size_t)c * H + y) * W + x;
    float p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[W], p4 = p[W + 1], p5 = p[W + 2], p6 = p[2 * W], p7 = p[2 * W + 1], p8 = p[2 * W + 2];
    FAV_CSWAP(p1, p2) FAV_CSWAP(p4, p5) FAV_CSWAP(p7, p8) FAV_CSWAP(p0, p1) FAV_CSWAP(p3, p4) FAV_CSWAP(p6, p7)
    FAV_CSWAP(p1, p2) FAV_CSWAP(p4, p5) FAV_CSWAP(p7, p8) FAV_CSWAP(p0, p3) FAV_CSWAP(p5, p8) FAV_CSWAP(p4, p7)
    FAV_CSWAP(p3, p6) FAV_CSWAP(p1, p4) FAV_CSWAP(p2, p5) FAV_CSWAP(p4, p7) FAV_CSWAP(p4, p2) FAV_CSWAP(p6, p4)
    FAV_CSWAP(p4, p2)
    dst[((size_t)c * OH + y) * OW + x] = p4;
}
#undef FAV_CSWAP

__global__ __launch_bounds__(256) void median_kernel(const float* src, float* dst, int H, int W, int r)
{
    const int OW = W - r + 1, OH = H - r + 1;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    if (x >= OW) return;
    float v[25];
    const float* p = src + (size_t)c * H * W;
    int n = 0;
    for (int dy = 0; dy < r; ++dy)
        for (int dx = 0; dx < r; ++dx) v[n++] = p[(size_t)(y + dy) * W + x + dx];
    const int k = (n - 1) / 2;
    for (int a = 0; a <= k; ++a) {               // partial selection sort up to the median
        int mi = a;
        for (int b = a + 1; b < n; ++b) if (v[b] < v[mi]) mi = b;
        const float t = v[a]; v[a] = v[mi]; v[mi] = t;
    }
    dst[((size_t)c * OH + y) * OW + x] = v[k];
}

// copies a (cropped, rotated) face into a wide strip [3][SH][SW] at column x0: strip[y][x0 + x] = rot(crop(face))[y][x]
__global__ __launch_bounds__(256) void strip_kernel(const float* face, int FH, int FW, int cy, int cx, int CH, int CW, int mode,
                                                    float* strip, int SH, int SW, int x0)
{
    const int DW = (mode == 1 || mode == 2) ? CH : CW;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= DW) return;
    const size_t s = rot_src(mode, i, j, CH, CW);            // index inside the crop, row pitch CW
    const int sy = (int)(s / CW) + cy, sx = (int)(s % CW) + cx;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        strip[((size_t)c * SH + i) * SW + x0 + j] = face[((size_t)c * FH + sy) * FW + sx];
}

__device__ __forceinline__ int reflect_i(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// run_next_image / run_image input assembly (core.lua:121-138,161-171) with a ready-made prior:
//   ch 0..2 pre(content), ch 3..5 fill + pre(prior) * cert, ch 6 cert, ch 7 zero; reflection padded NHWC8.
//   prior == null: no prior (cert = 0): ch 3..5 = fill, ch 6 = 0.
__global__ __launch_bounds__(256) void prep_prior_kernel(const uint8_t* frame_hwc, const float* prior, const float* cert,
                                                         int fill_random, unsigned seed, unsigned index, int H, int W, int pad,
                                                         float* in8)
{
    const int Wp = W + 2 * pad;
    const int yp = blockIdx.y, xp = blockIdx.x * 256 + threadIdx.x;
    if (xp >= Wp) return;
    const int y = reflect_i(yp - pad, H), x = reflect_i(xp - pad, W);
    const size_t i = (size_t)y * W + x, n = (size_t)H * W;
    const uint8_t* px = frame_hwc + i * 3;
    const float mean[3] = {103.939f, 116.779f, 123.68f};               // BGR (preprocess.lua:48)
    float o[8];
    const float cv = prior ? cert[i] : 0.f;
    const float cinv = (cv + -1.f) * -1.f;                              // core.lua:111
#pragma unroll
    for (int b = 0; b < 3; ++b) {                                       // b: BGR channel, source RGB channel 2 - b
        o[b] = ((float)px[2 - b] / 255.f) * 255.f - mean[b];
        float fill = 0.f;
        if (fill_random) fill = (fill_uniform(seed, index, 2 - b, y, x) * 255.f - mean[b]) * cinv;
        const float pm = prior ? (prior[(size_t)(2 - b) * n + i] * 255.f - mean[b]) * cv : 0.f;
        o[3 + b] = fill + pm;
    }
    o[6] = cv; o[7] = 0.f;
    float4* dst = reinterpret_cast<float4*>(in8 + ((size_t)yp * Wp + xp) * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// .flo payload [H][W][2] (u = dx, v = dy) -> the Lua loader's [2][H][W] with [0] = dy, [1] = dx (flowFileLoader.lua:27-29)
__global__ __launch_bounds__(256) void flo_to_lua_kernel(const float2* flo, float* lua, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 f = flo[i];
    lua[i] = f.y; lua[n + i] = f.x;
}

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

int launch_vr_rotate(const float* src, float* dst, int H, int W, int mode, hipStream_t st)
{
    const int DH = (mode == 1 || mode == 2) ? W : H, DW = (mode == 1 || mode == 2) ? H : W;
    hipLaunchKernelGGL(rotate_kernel, dim3((DW + 255) / 256, DH), dim3(256), 0, st, src, dst, H, W, mode);
    FAV_LAUNCH_CHECK("rotate_kernel");
    return FAV_OK;
}

int launch_vr_accum(float* acc, const float* w, const float* div, size_t n, int first, hipStream_t st)
{
    hipLaunchKernelGGL(accum_kernel, grid1(n), dim3(256), 0, st, acc, w, div, n, first);
    FAV_LAUNCH_CHECK("accum_kernel");
    return FAV_OK;
}

int launch_vr_cert(const uint8_t* cert_u8, const float* m0, const float* m1, const float* m2, const float* m3, float* out, size_t n,
                   hipStream_t st)
{
    hipLaunchKernelGGL(vr_cert_kernel, grid1(n), dim3(256), 0, st, cert_u8, m0, m1, m2, m3, out, n);
    FAV_LAUNCH_CHECK("vr_cert_kernel");
    return FAV_OK;
}

int launch_vr_prior(const float* lfw, const float* border, const float* grad, const float* cert, const float* m, const float* m2,
                    float* out, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(vr_prior_kernel, grid1(n), dim3(256), 0, st, lfw, border, grad, cert, m, m2, out, n);
    FAV_LAUNCH_CHECK("vr_prior_kernel");
    return FAV_OK;
}

int launch_vr_blend(const float* seg, const float* borders, const float* g, const float* anti, float* out, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(vr_blend_kernel, grid1(n), dim3(256), 0, st, seg, borders, g, anti, out, n);
    FAV_LAUNCH_CHECK("vr_blend_kernel");
    return FAV_OK;
}

int launch_vr_median(const float* src, float* dst, int H, int W, int r, hipStream_t st)
{
    FAV_REQUIRE(r >= 1 && r <= 5 && (r & 1) && H >= r && W >= r, "median filter: window %d unsupported (odd, <= 5)", r);
    if (r == 3) hipLaunchKernelGGL(median3_kernel, dim3((W - 2 + 255) / 256, H - 2, 3), dim3(256), 0, st, src, dst, H, W);
    else hipLaunchKernelGGL(median_kernel, dim3((W - r + 1 + 255) / 256, H - r + 1, 3), dim3(256), 0, st, src, dst, H, W, r);
    FAV_LAUNCH_CHECK("median_kernel");
    return FAV_OK;
}

int launch_vr_strip(const float* face, int FH, int FW, int cy, int cx, int CH, int CW, int mode, float* strip, int SH, int SW,
                    int x0, hipStream_t st)
{
    const int DH = (mode == 1 || mode == 2) ? CW : CH, DW = (mode == 1 || mode == 2) ? CH : CW;
    FAV_REQUIRE(cy >= 0 && cx >= 0 && cy + CH <= FH && cx + CW <= FW && DH <= SH && x0 + DW <= SW, "strip: crop outside the face");
    hipLaunchKernelGGL(strip_kernel, dim3((DW + 255) / 256, DH), dim3(256), 0, st, face, FH, FW, cy, cx, CH, CW, mode, strip, SH,
                       SW, x0);
    FAV_LAUNCH_CHECK("strip_kernel");
    return FAV_OK;
}

int launch_vr_flo_to_lua(const float* flo_uv, float* lua_dydx, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL(flo_to_lua_kernel, grid1(n), dim3(256), 0, st, reinterpret_cast<const float2*>(flo_uv), lua_dydx, n);
    FAV_LAUNCH_CHECK("flo_to_lua_kernel");
    return FAV_OK;
}

int launch_vr_prep(const uint8_t* frame_hwc, const float* prior, const float* cert, int fill_random, unsigned seed, unsigned index,
                   int H, int W, int pad, float* in8, hipStream_t st)
{
    hipLaunchKernelGGL(prep_prior_kernel, dim3((W + 2 * pad + 255) / 256, H + 2 * pad), dim3(256), 0, st, frame_hwc, prior, cert,
                       fill_random, seed, index, H, W, pad, in8);
    FAV_LAUNCH_CHECK("prep_prior_kernel");
    return FAV_OK;
}

}  // namespace fav
