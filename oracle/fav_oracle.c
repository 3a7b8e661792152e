/*
 * fav_oracle.c -- CPU restatement of the fast-artistic-videos per-frame hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (fast-artistic-videos_amd/) never links, imports or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference tree).  Items marked [recalled] restate un-vendored Torch7 packages
 * (nn / image) whose sources are not in the reference tree: they are the documented
 * published semantics of those packages and are cross-checked against PyTorch-CPU
 * in tests/ -- parity for them is "unpinned" (no reference golden vectors exist).
 * The consistency mask IS pinned: it is compared byte-for-byte against the reference's
 * own consistencyChecker compiled into oracle/_ref/ (see oracle/Makefile).
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared  (no -ffast-math, no FMA
 * contraction: the mask arithmetic must round exactly like the reference's x86-64 SSE2 build,
 * consistencyChecker/Makefile:2).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread cap for the small test problems: with hundreds of host threads the fork/join of every tiny loop dominates */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * A2  warp, GPU semantics ("stn"): stnbdhw/BilinearSamplerBDHW.cu:48-109
 *   img  [C][H][W], flow [2][Ho][Wo] with flow[0]=dy, flow[1]=dx (BilinearSamplerBDHW.cu:72-73,
 *   flowFileLoader.lua:27-29), out [C][Ho][Wo].  Each of the 4 taps is zeroed individually when
 *   its integer coordinates fall outside the image (:92-101).
 * ------------------------------------------------------------------------------------------ */
/* float -> int as the GPU the reference kernel runs on converts (v_cvt_i32_f32 on gfx950, cvt.rzi.s32.f32 on NVIDIA):
 * saturating, NaN -> 0.  (x86's cvttss2si returns INT_MIN for all of these, C leaves them undefined.)  Pinned by
 * tests/golden/warp_*.npz = outputs of the reference's own kernel (oracle/_ref/libwarp_ref*.so) on such flows. */
static int gpu_f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* exact_f32 = 0: the sum :103-106 evaluated in double and rounded once (the reference's nvcc / hipcc default build may
 *                contract the fp32 expression into FMAs either way: this is the value both builds approximate);
 * exact_f32 = 1: the fp32 expression exactly as written, no contraction (this file is compiled with -ffp-contract=off):
 *                bit-identical to the reference kernel built with -ffp-contract=off (libwarp_ref_nofma.so). */
static void warp_stn_impl(const float* img, const float* flow, float* out, int C, int H, int W, int Ho, int Wo, int exact_f32)
{
#pragma omp parallel for
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
            float yf = flow[(size_t)0 * Ho * Wo + (size_t)y * Wo + x] + (float)y;   /* :72 */
            float xf = flow[(size_t)1 * Ho * Wo + (size_t)y * Wo + x] + (float)x;   /* :73 */
            int x0 = gpu_f2i(floorf(xf));            /* getTopLeft :13-23 */
            int y0 = gpu_f2i(floorf(yf));
            float wx = 1.0f - (xf - (float)x0);
            float wy = 1.0f - (yf - (float)y0);
            int x1 = (int)((unsigned)x0 + 1u), y1 = (int)((unsigned)y0 + 1u);       /* wraps like the hardware add */
            int xin0 = (x0 >= 0 && x0 <= W - 1), xin1 = (x1 >= 0 && x1 <= W - 1);
            int yin0 = (y0 >= 0 && y0 <= H - 1), yin1 = (y1 >= 0 && y1 <= H - 1);
            for (int c = 0; c < C; ++c) {
                const float* p = img + (size_t)c * H * W;
                float tl = (xin0 && yin0) ? p[(size_t)y0 * W + x0] : 0.0f;           /* :86-101 */
                float tr = (xin1 && yin0) ? p[(size_t)y0 * W + x1] : 0.0f;
                float bl = (xin0 && yin1) ? p[(size_t)y1 * W + x0] : 0.0f;
                float br = (xin1 && yin1) ? p[(size_t)y1 * W + x1] : 0.0f;
                float r;
                if (exact_f32) {
                    r = wx * wy * tl + (1.0f - wx) * wy * tr + wx * (1.0f - wy) * bl + (1.0f - wx) * (1.0f - wy) * br;
                } else {
                    double v = (double)wx * wy * tl + (double)(1.0f - wx) * wy * tr
                             + (double)wx * (1.0f - wy) * bl + (double)(1.0f - wx) * (1.0f - wy) * br;
                    r = (float)v;
                }
                out[(size_t)c * Ho * Wo + (size_t)y * Wo + x] = r;
            }
        }
}

void orc_warp_stn(const float* img, const float* flow, float* out, int C, int H, int W, int Ho, int Wo)
{
    warp_stn_impl(img, flow, out, C, H, W, Ho, Wo, 0);
}

void orc_warp_stn_f32(const float* img, const float* flow, float* out, int C, int H, int W, int Ho, int Wo)
{
    warp_stn_impl(img, flow, out, C, H, W, Ho, Wo, 1);
}

/* ------------------------------------------------------------------------------------------
 * A2  warp, CPU semantics ("cpu"): fast_artistic_video/utils.lua:147
 *   image.warp(img, flow, 'bilinear', true, 'pad', 0)   [recalled: torch `image` package,
 *   generic/image.c image_(Main_warp), offset_mode=1, clamp_mode=pad, pad_value=0]
 *   Whole pixel = 0 when the sample point is outside [0,H-1]x[0,W-1]; otherwise bilinear with
 *   the +1 neighbours clamped to the last row/column.
 * ------------------------------------------------------------------------------------------ */
void orc_warp_cpu(const float* img, const float* flow, float* out,
                  int C, int H, int W, int Ho, int Wo)
{
#pragma omp parallel for
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
            float iy = (float)y + flow[(size_t)y * Wo + x];
            float ix = (float)x + flow[(size_t)Ho * Wo + (size_t)y * Wo + x];
            size_t o = (size_t)y * Wo + x;
            if (iy < 0 || iy > (float)(H - 1) || ix < 0 || ix > (float)(W - 1)) {
                for (int c = 0; c < C; ++c) out[(size_t)c * Ho * Wo + o] = 0.0f;
                continue;
            }
            long x_nw = (long)floorf(ix), y_nw = (long)floorf(iy);
            long x_e = x_nw + 1, y_s = y_nw + 1;
            float nw = ((float)x_e - ix) * ((float)y_s - iy);
            float ne = (ix - (float)x_nw) * ((float)y_s - iy);
            float sw = ((float)x_e - ix) * (iy - (float)y_nw);
            float se = (ix - (float)x_nw) * (iy - (float)y_nw);
            long xe = x_e < W - 1 ? x_e : W - 1, ys = y_s < H - 1 ? y_s : H - 1;
            for (int c = 0; c < C; ++c) {
                const float* p = img + (size_t)c * H * W;
                double v = (double)p[y_nw * W + x_nw] * nw + (double)p[y_nw * W + xe] * ne
                         + (double)p[ys * W + x_nw] * sw + (double)p[ys * W + xe] * se;
                out[(size_t)c * Ho * Wo + o] = (float)v;
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * A4  structure map: consistencyChecker/consistencyChecker.cpp:39-78 (computeCorners) followed by
 *     CMatrix::normalize(0,1) (CMatrix.h:721-737) as main() does (:155-159).
 *   rgb: planar u8 [3][H][W] (CTensor::readFromPPM de-interleaves, CTensor.h:925-937).
 *   All arithmetic fp32 in the reference's evaluation order; sequential (order dependent).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float k, pm, pp, e2, a2; } orc_iir_t;

/* recursive-filter constants: CFilter.h:1419-1425 (types exactly as written there:
 * NMath::Pi is a float (NMath.cpp:9), sqrt/exp resolve to the float overloads) */
void orc_iir_constants(float sigma, float* out5)
{
    const float Pi = 3.1415926536f;
    float aAlpha = (float)(2.5 / (double)(sqrtf(Pi) * sigma));
    float aExp = expf(-aAlpha);
    float aExpSqr = aExp * aExp;
    float a2Exp = (float)(2.0 * (double)aExp);
    float k = (float)((1.0 - (double)aExp) * (1.0 - (double)aExp)
                      / (1.0 + 2.0 * (double)aAlpha * (double)aExp - (double)aExpSqr));
    float aPreMinus = (float)((double)aExp * ((double)aAlpha - 1.0));
    float aPrePlus = (float)((double)aExp * ((double)aAlpha + 1.0));
    out5[0] = k; out5[1] = aPreMinus; out5[2] = aPrePlus; out5[3] = aExpSqr; out5[4] = a2Exp;
}

/* one line of recursiveSmoothX / recursiveSmoothY: CFilter.h:1426-1437 */
static void iir_line(float* m, int n, int stride, const orc_iir_t* c, float* v1, float* v2)
{
#define M(i) m[(size_t)(i) * stride]
    if (n < 2) { /* the reference indexes out of bounds here; keep the line unchanged */ return; }
    v1[0] = (0.5f - c->k * c->pm) * M(0);
    v1[1] = c->k * (M(1) + c->pm * M(0)) + (c->a2 - c->e2) * v1[0];
    for (int x = 2; x < n; ++x)
        v1[x] = c->k * (M(x) + c->pm * M(x - 1)) + c->a2 * v1[x - 1] - c->e2 * v1[x - 2];
    v2[n - 1] = (0.5f + c->k * c->pm) * M(n - 1);
    v2[n - 2] = c->k * ((c->pp - c->e2) * M(n - 1)) + (c->a2 - c->e2) * v2[n - 1];
    for (int x = n - 3; x >= 0; --x)
        v2[x] = c->k * (c->pp * M(x + 1) - c->e2 * M(x + 2)) + c->a2 * v2[x + 1] - c->e2 * v2[x + 2];
    for (int x = 0; x < n; ++x) M(x) = v1[x] + v2[x];
#undef M
}

void orc_corners(const uint8_t* rgb, float* structure, int W, int H)
{
    size_t n = (size_t)W * H;
    float* dxx = (float*)calloc(n, sizeof(float));
    float* dyy = (float*)calloc(n, sizeof(float));
    float* dxy = (float*)calloc(n, sizeof(float));
    /* gradient: CDerivative(3) = [-0.5, 0, 0.5] (CFilter.h:600-611), borders mirrored with the
     * edge pixel repeated (CFilter.h:1507-1512,1551-1556); second-moment sums :54-60 */
    for (int c = 0; c < 3; ++c) {
        const uint8_t* p = rgb + (size_t)c * n;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int xm = x - 1 < 0 ? 0 : x - 1, xp = x + 1 >= W ? W - 1 : x + 1;
                int ym = y - 1 < 0 ? 0 : y - 1, yp = y + 1 >= H ? H - 1 : y + 1;
                float dx = 0.0f, dy = 0.0f;
                dx += -0.5f * (float)p[(size_t)y * W + xm]; dx += 0.0f * (float)p[(size_t)y * W + x];
                dx += 0.5f * (float)p[(size_t)y * W + xp];
                dy += -0.5f * (float)p[(size_t)ym * W + x]; dy += 0.0f * (float)p[(size_t)y * W + x];
                dy += 0.5f * (float)p[(size_t)yp * W + x];
                size_t i = (size_t)y * W + x;
                dxx[i] += dx * dx; dyy[i] += dy * dy; dxy[i] += dx * dy;
            }
    }
    float cst[5]; orc_iir_constants(3.0f, cst);
    orc_iir_t c = { cst[0], cst[1], cst[2], cst[3], cst[4] };
    int mx = W > H ? W : H;
    float* v1 = (float*)malloc(sizeof(float) * mx);
    float* v2 = (float*)malloc(sizeof(float) * mx);
    float* planes[3] = { dxx, dyy, dxy };          /* :62-67 order: dxx X,Y; dyy X,Y; dxy X,Y */
    for (int p = 0; p < 3; ++p) {
        for (int y = 0; y < H; ++y) iir_line(planes[p] + (size_t)y * W, W, 1, &c, v1, v2);
        for (int x = 0; x < W; ++x) iir_line(planes[p] + x, H, W, &c, v1, v2);
    }
    /* smallest eigenvalue :69-77 */
    for (size_t i = 0; i < n; ++i) {
        float a = dxx[i], b = dxy[i], cc = dyy[i];
        float temp = (float)(0.5 * (double)(a + cc));
        float temp2 = temp * temp + b * b - a * cc;
        structure[i] = temp2 < 0.0f ? 0.0f : temp - sqrtf(temp2);
    }
    /* normalize(0,1) with initial min/max -30000/+30000 and the else-if quirk: CMatrix.h:70,721-737 */
    float cmin = 30000.0f, cmax = -30000.0f;
    for (size_t i = 0; i < n; ++i) {
        if (structure[i] > cmax) cmax = structure[i];
        else if (structure[i] < cmin) cmin = structure[i];
    }
    float t = cmax - cmin;
    if (t == 0) t = 1; else t = (1.0f - 0.0f) / t;
    for (size_t i = 0; i < n; ++i) {
        float v = structure[i];
        v -= cmin; v *= t; v += 0.0f;
        structure[i] = v;
    }
    free(dxx); free(dyy); free(dxy); free(v1); free(v2);
}

/* sequential fp32 running sum / size: CMatrix::avg, CMatrix.h:1245-1251 */
float orc_avg(const float* m, int W, int H)
{
    float a = 0;
    int n = W * H;
    for (int i = 0; i < n; ++i) a += m[i];
    return a / n;
}

/* ------------------------------------------------------------------------------------------
 * A3  forward-backward consistency: consistencyChecker/consistencyChecker.cpp:80-134
 *   flow1, flow2: planar [2][H][W], plane 0 = u (x), plane 1 = v (y)  (readMiddlebury :29-33,
 *   CTensor.h:1001-1007).  structure: normalised corner map or NULL (3-arg mode).
 *   out: H*W bytes in {0,255}; reliable starts at 255 (:151); the motion-edge branch :129-132
 *   only stores 255 over 255 and is therefore omitted; clip(0,255) (:169) is a no-op.
 *   float/double promotions exactly as the C++ expressions imply.
 * ------------------------------------------------------------------------------------------ */
void orc_consistency(const float* flow1, const float* flow2, const float* structure,
                     uint8_t* out, int W, int H)
{
    size_t n = (size_t)W * H;
    const float* u1p = flow1; const float* v1p = flow1 + n;
    const float* u2p = flow2; const float* v2p = flow2 + n;
    float structureAvg = 0;
    if (structure) structureAvg = orc_avg(structure, W, H);
    for (int ay = 0; ay < H; ++ay)
        for (int ax = 0; ax < W; ++ax) {
            size_t i = (size_t)ay * W + ax;
            float bx = ax + u1p[i];
            float by = ay + v1p[i];
            int x1 = (int)floor(bx);
            int y1 = (int)floor(by);
            int x2 = x1 + 1, y2 = y1 + 1;
            if (x1 < 0 || x2 >= W || y1 < 0 || y2 >= H) { out[i] = 0; continue; }
            float alphaX = bx - x1; float alphaY = by - y1;
            float a = (1.0 - alphaX) * u2p[(size_t)y1 * W + x1] + alphaX * u2p[(size_t)y1 * W + x2];
            float b = (1.0 - alphaX) * u2p[(size_t)y2 * W + x1] + alphaX * u2p[(size_t)y2 * W + x2];
            float u = (1.0 - alphaY) * a + alphaY * b;
            a = (1.0 - alphaX) * v2p[(size_t)y1 * W + x1] + alphaX * v2p[(size_t)y1 * W + x2];
            b = (1.0 - alphaX) * v2p[(size_t)y2 * W + x1] + alphaX * v2p[(size_t)y2 * W + x2];
            float v = (1.0 - alphaY) * a + alphaY * b;
            float cx = bx + u;
            float cy = by + v;
            float u2 = u1p[i];
            float v2 = v1p[i];
            float structureTerm = 0;
            if (structure) {
                float h = structureAvg / 2.0f - structure[i];
                structureTerm = 4.0f / structureAvg * (h > 0.0f ? h : 0.0f);
            }
            if (((cx - ax) * (cx - ax) + (cy - ay) * (cy - ay))
                >= 0.01 * (u2 * u2 + v2 * v2 + u * u + v * v) + structureTerm + 0.5f)
                out[i] = 0;
            else
                out[i] = 255;
        }
}

/* ------------------------------------------------------------------------------------------
 * A5  min filter: fast_artistic_video/utils.lua:161-169
 *   1 - maxpool_{r x r, stride 1, pad r/2}(1 - cert)  [recalled: nn.SpatialMaxPooling pads with
 *   -inf, i.e. the window is truncated at the borders]
 * ------------------------------------------------------------------------------------------ */
void orc_min_filter(const float* cert, float* out, int H, int W, int r)
{
    int p = r / 2;
#pragma omp parallel for
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float m = -INFINITY;
            for (int dy = -p; dy < r - p; ++dy)
                for (int dx = -p; dx < r - p; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    float v = cert[(size_t)yy * W + xx] * -1.0f + 1.0f;   /* MulConstant(-1), AddConstant(1) */
                    if (v > m) m = v;
                }
            out[(size_t)y * W + x] = m * -1.0f + 1.0f;
        }
}

/* ------------------------------------------------------------------------------------------
 * A6  VGG pre/de-process: fast_artistic_video/preprocess.lua:48,57-62,66-71
 *   pre: RGB[0,1] -> BGR * 255 - mean(103.939,116.779,123.68);  de: (x + mean)/255 -> RGB
 * ------------------------------------------------------------------------------------------ */
static const float VGG_MEAN[3] = { 103.939f, 116.779f, 123.68f };

void orc_preprocess(const float* rgb, float* bgr, int H, int W)
{
    size_t n = (size_t)H * W;
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < n; ++i) bgr[c * n + i] = rgb[(2 - c) * n + i] * 255.0f - VGG_MEAN[c];
}

void orc_deprocess(const float* bgr, float* rgb, int H, int W)
{
    size_t n = (size_t)H * W;
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < n; ++i) rgb[c * n + i] = (bgr[(2 - c) * n + i] + VGG_MEAN[2 - c]) / 255.0f;
}

/* ------------------------------------------------------------------------------------------
 * A7  7-channel input assembly: fast_artistic_video_core.lua:161-173 (run_next_image) with
 *     fill_occlusions = vgg-mean (generate_fill :108-117 returns zeros) and :133-138 (first frame)
 *   frame_rgb [3][H][W] in [0,1]; warped_rgb [3][H][W] (NULL for the first frame);
 *   cert [H][W] (already min-filtered; NULL for the first frame) -> in7 [7][H][W]
 * ------------------------------------------------------------------------------------------ */
void orc_assemble(const float* frame_rgb, const float* warped_rgb, const float* cert,
                  float* in7, int H, int W)
{
    size_t n = (size_t)H * W;
    orc_preprocess(frame_rgb, in7, H, W);
    if (!warped_rgb) { memset(in7 + 3 * n, 0, 4 * n * sizeof(float)); return; }
    orc_preprocess(warped_rgb, in7 + 3 * n, H, W);
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < n; ++i) in7[(3 + c) * n + i] = in7[(3 + c) * n + i] * cert[i] + 0.0f;
    memcpy(in7 + 6 * n, cert, n * sizeof(float));
}

/* ------------------------------------------------------------------------------------------
 * A8  network layers [recalled: Torch7 nn; forward semantics only], NCHW, N = 1
 * ------------------------------------------------------------------------------------------ */

/* nn.SpatialConvolution (models_video.lua:20,32,80,93): cross-correlation, zero padding,
 * weight [Cout][Cin][kH][kW], out = floor((in + 2p - k)/s) + 1.  fp32 data, fp64 accumulation. */
void orc_conv2d(const float* in, int Cin, int H, int W, const float* weight, const float* bias,
                int Cout, int kH, int kW, int sH, int sW, int pH, int pW, float* out)
{
    int OH = (H + 2 * pH - kH) / sH + 1, OW = (W + 2 * pW - kW) / sW + 1;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * OW);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < OH; ++oy) {
                for (int ox = 0; ox < OW; ++ox) acc[ox] = bias ? (double)bias[co] : 0.0;
                for (int ci = 0; ci < Cin; ++ci)
                    for (int ky = 0; ky < kH; ++ky) {
                        int iy = oy * sH + ky - pH;
                        if (iy < 0 || iy >= H) continue;
                        const float* row = in + ((size_t)ci * H + iy) * W;
                        const float* wr = weight + (((size_t)co * Cin + ci) * kH + ky) * kW;
                        for (int kx = 0; kx < kW; ++kx) {
                            double w = wr[kx];
                            /* ox range with 0 <= ox*sW + kx - pW < W */
                            int lo = pW - kx; lo = lo <= 0 ? 0 : (lo + sW - 1) / sW;
                            int hi = (W - 1 + pW - kx) / sW; if (hi > OW - 1) hi = OW - 1;
                            const float* src = row + kx - pW;
                            if (sW == 1) for (int ox = lo; ox <= hi; ++ox) acc[ox] += w * (double)src[ox];
                            else for (int ox = lo; ox <= hi; ++ox) acc[ox] += w * (double)src[(size_t)ox * sW];
                        }
                    }
                float* o = out + ((size_t)co * OH + oy) * OW;
                for (int ox = 0; ox < OW; ++ox) o[ox] = (float)acc[ox];
            }
        free(acc);
    }
}

/* nn.SpatialFullConvolution(nIn, nOut, kW, kH, dW, dH, padW, padH, adjW, adjH) (models_video.lua:88,102) [recalled: Torch7 nn]:
 * transposed convolution, weight [Cin][Cout][kH][kW]; out size = (in-1)*s - 2p + k + adj;
 * out[co][iy*s - p + ky][ix*s - p + kx] += in[ci][iy][ix] * w[ci][co][ky][kx].  fp64 accumulation. */
void orc_full_conv2d(const float* in, int Cin, int H, int W, const float* weight, const float* bias,
                     int Cout, int k, int s, int p, int adj, float* out)
{
    int OH = (H - 1) * s - 2 * p + k + adj, OW = (W - 1) * s - 2 * p + k + adj;
#pragma omp parallel for
    for (int co = 0; co < Cout; ++co) {
        double* acc = (double*)malloc(sizeof(double) * OH * OW);
        for (int i = 0; i < OH * OW; ++i) acc[i] = bias ? (double)bias[co] : 0.0;
        for (int ci = 0; ci < Cin; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double x = in[((size_t)ci * H + iy) * W + ix];
                    for (int ky = 0; ky < k; ++ky) {
                        int oy = iy * s - p + ky; if (oy < 0 || oy >= OH) continue;
                        for (int kx = 0; kx < k; ++kx) {
                            int ox = ix * s - p + kx; if (ox < 0 || ox >= OW) continue;
                            acc[(size_t)oy * OW + ox] += x * (double)weight[(((size_t)ci * Cout + co) * k + ky) * k + kx];
                        }
                    }
                }
        for (int i = 0; i < OH * OW; ++i) out[(size_t)co * OH * OW + i] = (float)acc[i];
        free(acc);
    }
}

/* nn.SpatialBatchNormalization in evaluate mode (models_video.lua:24,35,126; model:evaluate() core.lua:47) [recalled]:
 * y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta.  In place; relu fuses a following nn.ReLU. */
void orc_batchnorm_eval(float* x, int C, int H, int W, const float* mean, const float* var, const float* gamma,
                        const float* beta, float eps, int relu)
{
    size_t n = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        float* p = x + (size_t)c * n;
        double invstd = 1.0 / sqrt((double)var[c] + (double)eps);
        float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
        for (size_t i = 0; i < n; ++i) {
            float v = (float)(((double)p[i] - (double)mean[c]) * invstd) * g + b;
            p[i] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
}

/* nn.InstanceNormalization (InstanceNormalization.lua:33-53): SpatialBatchNormalization in
 * training mode over a 1 x (N*C) x H x W view => per-channel mean and BIASED variance over H*W,
 * y = (x - mean) / sqrt(var + eps) * gamma + beta; two-pass, fp64 accumulators [recalled].
 * relu != 0 fuses the following nn.ReLU(true) (models_video.lua:26,129). In place. */
void orc_instnorm(float* x, int C, int H, int W, const float* gamma, const float* beta,
                  float eps, int relu)
{
    size_t n = (size_t)H * W;
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        float* p = x + (size_t)c * n;
        double s = 0; for (size_t i = 0; i < n; ++i) s += p[i];
        double mean = s / (double)n;
        double q = 0; for (size_t i = 0; i < n; ++i) { double d = p[i] - mean; q += d * d; }
        double invstd = 1.0 / sqrt(q / (double)n + (double)eps);
        float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
        for (size_t i = 0; i < n; ++i) {
            float v = (float)(((double)p[i] - mean) * invstd) * g + b;
            p[i] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
}

/* nn.SpatialReflectionPadding(l,r,t,b) (train_video.lua:319-325): mirror, edge not repeated */
void orc_reflect_pad(const float* in, int C, int H, int W, int l, int r, int t, int b, float* out)
{
    int OH = H + t + b, OW = W + l + r;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < OH; ++y) {
            int sy = y - t; if (sy < 0) sy = -sy; if (sy >= H) sy = 2 * (H - 1) - sy;
            for (int x = 0; x < OW; ++x) {
                int sx = x - l; if (sx < 0) sx = -sx; if (sx >= W) sx = 2 * (W - 1) - sx;
                out[((size_t)c * OH + y) * OW + x] = in[((size_t)c * H + sy) * W + sx];
            }
        }
}

/* nn.SpatialUpSamplingNearest(s) (models_video.lua:98) */
void orc_upsample_nearest(const float* in, int C, int H, int W, int s, float* out)
{
    int OH = H * s, OW = W * s;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x)
                out[((size_t)c * OH + y) * OW + x] = in[((size_t)c * H + y / s) * W + x / s];
}

/* nn.ShaveImage(s) + nn.CAddTable (ShaveImage.lua:9-16, models_video.lua:44-51):
 * out[C][H-2s][W-2s] = block + skip[:, s:H-s, s:W-s]   (block is [C][H-2s][W-2s]) */
void orc_shave_add(const float* block, const float* skip, int C, int H, int W, int s, float* out)
{
    int OH = H - 2 * s, OW = W - 2 * s;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x)
                out[((size_t)c * OH + y) * OW + x] =
                    block[((size_t)c * OH + y) * OW + x] + skip[((size_t)c * H + y + s) * W + x + s];
}

/* nn.Tanh + nn.MulConstant(k) (models_video.lua:135-136); TotalVariation forward = identity
 * (TotalVariation.lua:12-15) */
void orc_tanh_mul(float* x, size_t n, float k)
{
    for (size_t i = 0; i < n; ++i) x[i] = (float)tanh((double)x[i]) * k;
}

/* image.save [recalled]: clamp to [0,1], * 255, truncate to byte.  planar [3][H][W] -> HWC u8 */
void orc_to_u8_hwc(const float* rgb, uint8_t* out, int H, int W)
{
    size_t n = (size_t)H * W;
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            float v = rgb[c * n + i];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            out[i * 3 + c] = (uint8_t)(v * 255.0f);
        }
}
