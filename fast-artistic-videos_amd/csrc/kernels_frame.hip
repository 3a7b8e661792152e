// kernels_frame.hip -- per-frame gather / elementwise kernels for gfx950 (HBM-bound work):
//   A2  optical-flow warp            (stnbdhw/BilinearSamplerBDHW.cu:48-109, utils.lua:141-149)
//   A5  certainty erosion            (utils.lua:161-169)
//   A6  VGG pre-processing           (preprocess.lua:48,57-62)
//   A7  7-channel input assembly     (fast_artistic_video_core.lua:133-138,161-171)
//   A9  image.save quantisation      (fast_artistic_video.lua:160-170)
// plus the fused form used by the per-frame pipeline: warp + preprocess + mask + concat +
// nn.SpatialReflectionPadding written straight into the network's padded NHWC8 input.
// One lane per output pixel, consecutive lanes on consecutive x => coalesced streams for every
// dense operand; the bilinear taps are gathers served by L2 (the flow is locally smooth).
// Compiled with -ffp-contract=off so the fp32 expressions round like the CPU restatement.
#include "fav_internal.h"
#include "consistency_pixel.h"

namespace fav {
namespace {

__device__ __forceinline__ int reflect(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// bilinear weights/taps shared by all channels of one output pixel
struct Taps {
    int o00, o01, o10, o11;     // linear offsets y*W+x (clamped so they are always addressable)
    float w00, w01, w10, w11;   // the weights as the reference forms them (NOT zeroed for taps outside the image)
    unsigned in;                // bit k set: tap k lies inside the image; otherwise its VALUE is 0 (BilinearSamplerBDHW.cu:86-101)
};

// FAV_BORDER_STN: BilinearSamplerBDHW.cu:13-23,72-73,92-106.  yf = dy + y, xf = dx + x.
// Bit-exact with the reference kernel compiled for gfx950 (oracle/_ref/libwarp_ref_nofma.so, tests/golden/warp_*.npz): the
// same float -> int conversion (saturating, NaN -> 0), the weights left as computed and the out-of-image VALUES zeroed, so
// that a non-finite weight meets a zero exactly as in the reference (inf x 0 = NaN for infinite flows).
__device__ __forceinline__ Taps taps_stn(float yf, float xf, int H, int W)
{
    Taps t;
    const int x0 = (int)floorf(xf), y0 = (int)floorf(yf);
    const float wx = 1.f - (xf - (float)x0), wy = 1.f - (yf - (float)y0);
    const int x1 = (int)((unsigned)x0 + 1u), y1 = (int)((unsigned)y0 + 1u);          // INT_MAX + 1 wraps like the hardware add
    const bool xi0 = x0 >= 0 && x0 <= W - 1, xi1 = x1 >= 0 && x1 <= W - 1;
    const bool yi0 = y0 >= 0 && y0 <= H - 1, yi1 = y1 >= 0 && y1 <= H - 1;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
    t.o00 = yc0 * W + xc0; t.o01 = yc0 * W + xc1; t.o10 = yc1 * W + xc0; t.o11 = yc1 * W + xc1;
    t.w00 = wx * wy;
    t.w01 = (1.f - wx) * wy;
    t.w10 = wx * (1.f - wy);
    t.w11 = (1.f - wx) * (1.f - wy);
    t.in = (unsigned)(xi0 && yi0) | ((unsigned)(xi1 && yi0) << 1) | ((unsigned)(xi0 && yi1) << 2) | ((unsigned)(xi1 && yi1) << 3);
    return t;
}

// FAV_BORDER_CPU: image.warp(..., 'bilinear', true, 'pad', 0) [Torch7 `image`, recalled]
__device__ __forceinline__ Taps taps_cpu(float iy, float ix, int H, int W)
{
    Taps t;
    if (iy < 0.f || iy > (float)(H - 1) || ix < 0.f || ix > (float)(W - 1)) {
        t.o00 = t.o01 = t.o10 = t.o11 = 0;
        t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
        t.in = 0u;
        return t;
    }
    const int xw = (int)floorf(ix), yn = (int)floorf(iy);
    const int xe = xw + 1, ys = yn + 1;
    t.w00 = ((float)xe - ix) * ((float)ys - iy);
    t.w01 = (ix - (float)xw) * ((float)ys - iy);
    t.w10 = ((float)xe - ix) * (iy - (float)yn);
    t.w11 = (ix - (float)xw) * (iy - (float)yn);
    const int xec = min(xe, W - 1), ysc = min(ys, H - 1);
    t.o00 = yn * W + xw; t.o01 = yn * W + xec; t.o10 = ysc * W + xw; t.o11 = ysc * W + xec;
    t.in = 15u;
    return t;
}

__device__ __forceinline__ Taps make_taps(int border, float yf, float xf, int H, int W)
{
    return border == FAV_BORDER_CPU ? taps_cpu(yf, xf, H, W) : taps_stn(yf, xf, H, W);
}

__device__ __forceinline__ float sample(const float* plane, const Taps& t)
{
    // summation order of BilinearSamplerBDHW.cu:103-106; taps outside the image contribute weight x 0 (:86-101)
    const float v00 = (t.in & 1u) ? plane[t.o00] : 0.f, v01 = (t.in & 2u) ? plane[t.o01] : 0.f;
    const float v10 = (t.in & 4u) ? plane[t.o10] : 0.f, v11 = (t.in & 8u) ? plane[t.o11] : 0.f;
    return t.w00 * v00 + t.w01 * v01 + t.w10 * v10 + t.w11 * v11;
}

// the same four values with the two taps of a row fetched as ONE 8-byte load when they are neighbours in memory (everywhere but at a
// clamped border): half the gather instructions of the fused input assembly, whose three planes share the taps (round 6)
__device__ __forceinline__ float sample_pairs(const float* plane, const Taps& t)
{
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    float v00, v01, v10, v11;
    if (t.o01 == t.o00 + 1) { const f2u a = *reinterpret_cast<const f2u*>(plane + t.o00); v00 = a.x; v01 = a.y; }
    else { v00 = plane[t.o00]; v01 = plane[t.o01]; }
    if (t.o11 == t.o10 + 1) { const f2u b = *reinterpret_cast<const f2u*>(plane + t.o10); v10 = b.x; v11 = b.y; }
    else { v10 = plane[t.o10]; v11 = plane[t.o11]; }
    v00 = (t.in & 1u) ? v00 : 0.f; v01 = (t.in & 2u) ? v01 : 0.f; v10 = (t.in & 4u) ? v10 : 0.f; v11 = (t.in & 8u) ? v11 : 0.f;
    return t.w00 * v00 + t.w01 * v01 + t.w10 * v10 + t.w11 * v11;
}

__global__ __launch_bounds__(256) void warp_kernel(const float* img, const float* flow, float* out, int C, int H, int W,
                                                   int Ho, int Wo, int border)
{
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= Wo) return;
    const size_t on = (size_t)Ho * Wo;
    const float* fl = flow + (size_t)b * 2 * on;
    const float yf = fl[(size_t)y * Wo + x] + (float)y;        // flow[:,0] = dy  (BilinearSamplerBDHW.cu:72)
    const float xf = fl[on + (size_t)y * Wo + x] + (float)x;   // flow[:,1] = dx  (:73)
    const Taps t = make_taps(border, yf, xf, H, W);
    const float* ib = img + (size_t)b * C * H * W;
    float* ob = out + (size_t)b * C * on + (size_t)y * Wo + x;
    for (int c = 0; c < C; ++c) ob[(size_t)c * on] = sample(ib + (size_t)c * H * W, t);
}

// certainty from the checker's PGM byte: image.load(...,1) = byte/255; -invert_occlusion and
// -fix_occlusions of fast_artistic_video.lua:79-86,99-112
__device__ __forceinline__ float cert_from_byte(uint8_t mb, const float2* bw_flo, int invert, int fix_occ, int border, int y, int x, int H, int W)
{
    const size_t i = (size_t)y * W + x;
    float c = (float)mb / 255.f;
    if (invert) c = (c + -1.f) * -1.f;
    if (fix_occ) {
        const float2 f = bw_flo[i];
        const Taps t = make_taps(border, f.y + (float)y, f.x + (float)x, H, W);
        // warp of an all-ones image (fast_artistic_video.lua:79-86): weight x (1 | 0), the reference's expression
        float ones = t.w00 * ((t.in & 1u) ? 1.f : 0.f) + t.w01 * ((t.in & 2u) ? 1.f : 0.f) + t.w10 * ((t.in & 4u) ? 1.f : 0.f) +
                     t.w11 * ((t.in & 8u) ? 1.f : 0.f);
        ones = ones + -0.5f;
        const float sg = ones > 0.f ? 1.f : (ones < 0.f ? -1.f : 0.f);
        c *= fmaxf(sg, 0.f);
    }
    return c;
}

__device__ __forceinline__ float cert_from_mask(const uint8_t* mask, const float2* bw_flo, int invert, int fix_occ, int border, int y, int x, int H, int W)
{
    return cert_from_byte(mask[(size_t)y * W + x], bw_flo, invert, fix_occ, border, y, x, H, W);
}

// utils.min_filter (utils.lua:161-169): 1 - maxpool_{r x r, stride 1, pad r/2}(1 - cert); max-pooling pads with -inf, i.e. the
// windows are truncated at the borders.  max is exact and order-free, so the r x r window is evaluated separably on an LDS
// tile (rows, then columns): 2r instead of r*r taps, each input element read from memory once per tile.
constexpr int MF_TX = 64, MF_TY = 16, MF_RMAX = 15;
// FROM_MASK: the tile is filled from the checker's mask byte with the certainty options applied on the way in (cert_from_mask; 1.5x of that cheap work is repeated in the tiles' halos) -- one launch and one 3.7 MB plane less per frame
// FROM_MASK == 2: the tile is filled from the two FLOWS -- the forward-backward check itself (consistency_pixel, bit-exact) runs on
// every tile position, the mask byte of the tile's own pixels is written out (tests, -temporal_eval_file, the VR path read it), the
// certainty options and the erosion follow as above: the mask, the certainty and the eroded certainty of a frame in ONE launch
// (the check is recomputed 1.5x in the tiles' halos; it replaces a 7.4 us kernel and a kernel boundary)
template <int FROM_MASK>
__global__ __launch_bounds__(256) void min_filter_kernel(const float* cert, const uint8_t* mask, const float2* bw_flo, int invert, int fix_occ,
                                                         int border, float* out, int H, int W, int r,
                                                         const float2* fw_flo = nullptr, const float* structure = nullptr, const float* avg_ptr = nullptr,
                                                         uint8_t* mask_out = nullptr)
{
    __shared__ float a[MF_TY + MF_RMAX - 1][MF_TX + MF_RMAX];        // 1 - cert, -inf outside the image
    __shared__ float b[MF_TY + MF_RMAX - 1][MF_TX + 1];              // row maxima
    const int t = threadIdx.x, p = r / 2;
    const int x0 = blockIdx.x * MF_TX, y0 = blockIdx.y * MF_TY;
    const int TW = MF_TX + r - 1, THh = MF_TY + r - 1;
    for (int e = t; e < TW * THh; e += 256) {
        const int ly = e / TW, lx = e - ly * TW;
        const int yy = y0 + ly - p, xx = x0 + lx - p;
        float v = -INFINITY;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            float c;
            if (FROM_MASK == 2) {
                const uint8_t mb = consistency_pixel(bw_flo, fw_flo, structure, avg_ptr, xx, yy, W, H);
                if (ly >= p && ly < p + MF_TY && lx >= p && lx < p + MF_TX) mask_out[(size_t)yy * W + xx] = mb;      // the tile's own pixels
                c = cert_from_byte(mb, bw_flo, invert, fix_occ, border, yy, xx, H, W);
            } else if (FROM_MASK == 1) {
                c = cert_from_mask(mask, bw_flo, invert, fix_occ, border, yy, xx, H, W);
            } else {
                c = cert[(size_t)yy * W + xx];
            }
            v = c * -1.f + 1.f;                                      // MulConstant(-1), AddConstant(1)
        }
        a[ly][lx] = v;
    }
    __syncthreads();
    for (int e = t; e < MF_TX * THh; e += 256) {
        const int ly = e / MF_TX, lx = e - ly * MF_TX;
        float m = -INFINITY;
        for (int d = 0; d < r; ++d) m = fmaxf(m, a[ly][lx + d]);
        b[ly][lx] = m;
    }
    __syncthreads();
    for (int e = t; e < MF_TX * MF_TY; e += 256) {
        const int ly = e / MF_TX, lx = e - ly * MF_TX;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        float m = -INFINITY;
        for (int d = 0; d < r; ++d) m = fmaxf(m, b[ly + d][lx]);
        out[(size_t)y * W + x] = m * -1.f + 1.f;
    }
}

__device__ __forceinline__ float vgg_mean(int c) { return c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f); }

__global__ __launch_bounds__(256) void assemble_kernel(const float* frame, const float* warped, const float* cert,
                                                       float* in7, int H, int W)
{
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float cv = (warped != nullptr) ? cert[i] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        in7[c * n + i] = frame[(2 - c) * n + i] * 255.f - vgg_mean(c);                // preprocess.lua:57-62
        float pr = 0.f;
        if (warped != nullptr) pr = (warped[(2 - c) * n + i] * 255.f - vgg_mean(c)) * cv + 0.f;   // core:166-170
        in7[(3 + c) * n + i] = pr;
    }
    in7[6 * n + i] = cv;                                                             // core:171 / :136
}

// Fused A2 + A6 + A7 + reflection pad -> padded NHWC8 network input [H+2p][W+2p][8]
//   ch 0..2 content (BGR, mean-subtracted), ch 3..5 masked warped prior, ch 6 certainty, ch 7 zero
// prev_rgb: [3][Hs][Ws] -- the previous OUTPUT, which is larger than the frame when H or W is not a multiple of 4; the warp samples it
// on the flow's H x W grid (BilinearSamplerBDHW.lua:71), bounds and pitch are the source's
// the eight floats of network-input pixel (y, x) of the UNPADDED frame (shared by prep_input_kernel and check_prep_kernel: one body,
// one rounding).  cv: the pixel's eroded certainty (ignored without a previous frame); byte01: the block's table of byte / 255
__device__ __forceinline__ void prep_pixel(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float2* bw_flo, float cv_in,
                                           int border, int H, int W, int y, int x, const float* byte01, int fill_random, unsigned seed, unsigned index,
                                           float4& lo, float4& hi)
{
    const size_t i = (size_t)y * W + x;
    const uint8_t* px = frame_hwc + i * 3;
    const float rgb[3] = {byte01[px[0]], byte01[px[1]], byte01[px[2]]};
    lo.x = rgb[2] * 255.f - 103.939f;
    lo.y = rgb[1] * 255.f - 116.779f;
    lo.z = rgb[0] * 255.f - 123.68f;
    // generate_fill (core.lua:108-117): 0 (vgg-mean) or pre(u) * (1 - cert) with u the documented counter RNG
    float fb = 0.f, fg = 0.f, fr = 0.f;
    const float cv = prev_rgb != nullptr ? cv_in : 0.f;
    if (fill_random) {
        const float cinv = (cv + -1.f) * -1.f;
        fb = (fill_uniform(seed, index, 2, y, x) * 255.f - 103.939f) * cinv;
        fg = (fill_uniform(seed, index, 1, y, x) * 255.f - 116.779f) * cinv;
        fr = (fill_uniform(seed, index, 0, y, x) * 255.f - 123.68f) * cinv;
    }
    if (prev_rgb != nullptr) {
        const float2 f = bw_flo[i];                                     // .flo payload: (u, v) = (dx, dy)
        const Taps t = make_taps(border, f.y + (float)y, f.x + (float)x, Hs, Ws);
        const size_t ns = (size_t)Hs * Ws;
        const float wr = sample_pairs(prev_rgb, t), wg = sample_pairs(prev_rgb + ns, t), wb = sample_pairs(prev_rgb + 2 * ns, t);
        lo.w = fb + (wb * 255.f - 103.939f) * cv;                       // torch.add(fill, prev_warped_masked), core:169
        hi.x = fg + (wg * 255.f - 116.779f) * cv;
        hi.y = fr + (wr * 255.f - 123.68f) * cv;
        hi.z = cv;
    } else {
        lo.w = fb; hi.x = fg; hi.y = fr; hi.z = 0.f;                    // core:133-138: fill only, zero mask
    }
    hi.w = 0.f;
}

__global__ __launch_bounds__(256) void prep_input_kernel(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws,
                                                         const float2* bw_flo, const float* cert, int border, int H,
                                                         int W, int pad, float* in8, int fill_random, unsigned seed, unsigned index, int* q0_out)
{
    // (the XCD this queue deals block 0 of a launch to: the look-ahead mask's long-lived blocks place themselves by it, kernels_consistency.hip)
    if (q0_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        *q0_out = (int)(xcc & 7u);
    }
    // image.load: byte / 255 -- a correctly rounded division (a dozen instructions); the 256 possible quotients are formed once per block
    __shared__ float byte01[256];
    byte01[threadIdx.x] = (float)threadIdx.x / 255.f;
    __syncthreads();
    const int Wp = W + 2 * pad;
    const int yp = blockIdx.y, xp = blockIdx.x * 256 + threadIdx.x;
    if (xp >= Wp) return;
    const int y = reflect(yp - pad, H), x = reflect(xp - pad, W);
    float4 lo, hi;
    prep_pixel(frame_hwc, prev_rgb, Hs, Ws, bw_flo, prev_rgb != nullptr ? cert[(size_t)y * W + x] : 0.f, border, H, W, y, x, byte01, fill_random, seed, index, lo, hi);
    float4* o = reinterpret_cast<float4*>(in8 + ((size_t)yp * Wp + xp) * 8);
    o[0] = lo; o[1] = hi;
}

// Round 5: forward-backward check + certainty options + erosion (min_filter_kernel<2>) AND the input assembly (prep_input_kernel) of a
// frame in ONE launch: a tile of 64 x 16 frame pixels (+ the erosion's halo) runs the check into LDS, erodes it there, and every own
// pixel -- its eroded certainty in a register -- is assembled and written to its place in the padded network input and to the places
// the reflection padding copies it to (nn.SpatialReflectionPadding: a pixel within `pad` of an edge appears up to three times per axis).
// One launch and one certainty round trip less per frame; the same bytes: the check is consistency_pixel, the assembly prep_pixel.
__global__ __launch_bounds__(256) void check_prep_kernel(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float2* bw_flo,
                                                         const float2* fw_flo, const float* structure, const float* avg_ptr, uint8_t* mask_out,
                                                         float* cert_out, int invert, int fix_occ, int border, int r, int H, int W, int pad,
                                                         float* in8, int fill_random, unsigned seed, unsigned index)
{
    __shared__ float a[MF_TY + MF_RMAX - 1][MF_TX + MF_RMAX];        // 1 - cert, -inf outside the image
    __shared__ float b[MF_TY + MF_RMAX - 1][MF_TX + 1];              // row maxima
    __shared__ float byte01[256];
    const int t = threadIdx.x, p = r / 2;
    byte01[t] = (float)t / 255.f;
    const int x0 = blockIdx.x * MF_TX, y0 = blockIdx.y * MF_TY;
    const int TW = MF_TX + r - 1, THh = MF_TY + r - 1;
    for (int e = t; e < TW * THh; e += 256) {
        const int ly = e / TW, lx = e - ly * TW;
        const int yy = y0 + ly - p, xx = x0 + lx - p;
        float v = -INFINITY;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const uint8_t mb = consistency_pixel(bw_flo, fw_flo, structure, avg_ptr, xx, yy, W, H);
            if (ly >= p && ly < p + MF_TY && lx >= p && lx < p + MF_TX) mask_out[(size_t)yy * W + xx] = mb;      // the tile's own pixels
            const float c = cert_from_byte(mb, bw_flo, invert, fix_occ, border, yy, xx, H, W);
            v = c * -1.f + 1.f;                                      // MulConstant(-1), AddConstant(1)
        }
        a[ly][lx] = v;
    }
    __syncthreads();
    for (int e = t; e < MF_TX * THh; e += 256) {
        const int ly = e / MF_TX, lx = e - ly * MF_TX;
        float m = -INFINITY;
        for (int d = 0; d < r; ++d) m = fmaxf(m, a[ly][lx + d]);
        b[ly][lx] = m;
    }
    __syncthreads();
    const int Wp = W + 2 * pad;
    for (int e = t; e < MF_TX * MF_TY; e += 256) {
        const int ly = e / MF_TX, lx = e - ly * MF_TX;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        float m = -INFINITY;
        for (int d = 0; d < r; ++d) m = fmaxf(m, b[ly + d][lx]);
        const float cv = m * -1.f + 1.f;
        cert_out[(size_t)y * W + x] = cv;
        float4 lo, hi;
        prep_pixel(frame_hwc, prev_rgb, Hs, Ws, bw_flo, cv, border, H, W, y, x, byte01, fill_random, seed, index, lo, hi);
        // the pixel's places in the padded input: row y + pad, and -- reflect(yp - pad, H) == y -- pad - y above the image (1 <= y <= pad),
        // pad + 2 (H - 1) - y below it (H - 1 - pad <= y <= H - 2); columns likewise
        int yt[3], xt[3], ny = 0, nx = 0;
        yt[ny++] = y + pad; if (y >= 1 && y <= pad) yt[ny++] = pad - y; if (y <= H - 2 && y >= H - 1 - pad) yt[ny++] = pad + 2 * (H - 1) - y;
        xt[nx++] = x + pad; if (x >= 1 && x <= pad) xt[nx++] = pad - x; if (x <= W - 2 && x >= W - 1 - pad) xt[nx++] = pad + 2 * (W - 1) - x;
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) {
                float4* o = reinterpret_cast<float4*>(in8 + ((size_t)yt[i] * Wp + xt[j]) * 8);
                o[0] = lo; o[1] = hi;
            }
    }
}

// A padding layer inside the network (models_video.lua:12-16,27-31,70-77: padding_type reflect / replicate puts one in front of every
// convolution): out[y][x][:] = in[map(y - pt)][map(x - pl)][:], NHWC, 16 bytes per lane.  mode 0 = nn.SpatialReflectionPadding (mirror
// without repeating the edge pixel), mode 1 = nn.SpatialReplicationPadding (repeat the edge pixel) [both `nn`, recalled; the oracle
// pins them on numpy's 'reflect' / 'edge' modes = torch.nn.functional.pad's].  `ups`: the source is still to be x2 nearest-upsampled
// (nn.SpatialUpSamplingNearest pending on the tensor): logical pixel (i, j) lives at (i >> 1, j >> 1).
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float4* in, int Hl, int Wl, int in_pitch, int C4, int ups, float4* out, int Ho, int Wo, int pl, int pt, int mode)
{
    const size_t total = (size_t)Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const size_t px = i / C4;
        const int x = (int)(px % Wo), y = (int)(px / Wo);
        int sy = y - pt, sx = x - pl;
        if (mode == 0) { if (sy < 0) sy = -sy; if (sy >= Hl) sy = 2 * (Hl - 1) - sy; if (sx < 0) sx = -sx; if (sx >= Wl) sx = 2 * (Wl - 1) - sx; }
        else { sy = sy < 0 ? 0 : (sy >= Hl ? Hl - 1 : sy); sx = sx < 0 ? 0 : (sx >= Wl ? Wl - 1 : sx); }
        sy >>= ups; sx >>= ups;
        out[i] = in[((size_t)sy * in_pitch + sx) * C4 + c];
    }
}

// test view of the fused input: interior of the padded NHWC8 buffer -> planar [7][H][W] (fav_stream_get_input_f32)
__global__ __launch_bounds__(256) void unpad_input_kernel(const float* in8, int H, int W, int pad, float* in7)
{
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    const float4* p = reinterpret_cast<const float4*>(in8 + ((size_t)(y + pad) * (W + 2 * pad) + (x + pad)) * 8);
    const float4 lo = p[0], hi = p[1];
    in7[i] = lo.x; in7[n + i] = lo.y; in7[2 * n + i] = lo.z; in7[3 * n + i] = lo.w;
    in7[4 * n + i] = hi.x; in7[5 * n + i] = hi.y; in7[6 * n + i] = hi.z;
}

// the tail of a host-ordered look-ahead: everything before it on its queue is complete when the host reads the value
__global__ void store_flag_kernel(uint32_t* flag_host, uint32_t value)
{
    __threadfence_system();
    *reinterpret_cast<volatile uint32_t*>(flag_host) = value;
    __threadfence_system();
}

// image.save: clamp to [0,1], *255, truncate [Torch7 `image`, recalled]; planar float RGB -> HWC u8
__global__ __launch_bounds__(256) void quantize_kernel(const float* rgb, uint8_t* out, int H, int W)
{
    const size_t n = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = rgb[c * n + i];
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i * 3 + c] = (uint8_t)(v * 255.f);
    }
}

// temporal-consistency term of func_eval (fast_artistic_video.lua:128-151): sum over c,y,x of
// (warp(prev)[c] * cert - cur[c] * cert)^2, fp64 partial sums per block (the caller adds them in block order and divides
// by 3*H*W = nn.MSECriterion's mean)
__global__ __launch_bounds__(256) void temporal_loss_kernel(const float* prev_rgb, const float* cur_rgb, const float2* bw_flo,
                                                            const uint8_t* cert_u8, int border, int H, int W, double* partial)
{
    __shared__ double red[256];
    const size_t n = (size_t)H * W;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const float2 f = bw_flo[i];
        const Taps t = make_taps(border, f.y + (float)y, f.x + (float)x, H, W);
        const float cv = (float)cert_u8[i] / 255.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = sample(prev_rgb + c * n, t) * cv, b = cur_rgb[c * n + i] * cv;
            const float d = a - b;
            acc += (double)d * (double)d;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

}  // namespace

int launch_warp(const float* img, const float* flow, float* out, int B, int C, int H, int W, int Ho, int Wo, int border,
                hipStream_t st)
{
    hipLaunchKernelGGL(warp_kernel, dim3((Wo + 255) / 256, Ho, B), dim3(256), 0, st, img, flow, out, C, H, W, Ho, Wo,
                       border);
    FAV_LAUNCH_CHECK("warp_kernel");
    return FAV_OK;
}

int launch_min_filter_f32(const float* cert, float* out, int H, int W, int r, hipStream_t st)
{
    FAV_REQUIRE(r >= 1 && r <= MF_RMAX, "min filter: window %d unsupported (1..%d)", r, MF_RMAX);
    hipLaunchKernelGGL(min_filter_kernel<0>, dim3((W + MF_TX - 1) / MF_TX, (H + MF_TY - 1) / MF_TY), dim3(256), 0, st, cert, nullptr, nullptr, 0, 0, 0, out, H, W, r,
                       nullptr, nullptr, nullptr, nullptr);
    FAV_LAUNCH_CHECK("min_filter_kernel");
    return FAV_OK;
}

int launch_assemble(const float* frame, const float* warped, const float* cert, float* in7, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(((size_t)H * W + 255) / 256)), dim3(256), 0, st, frame, warped,
                       cert, in7, H, W);
    FAV_LAUNCH_CHECK("assemble_kernel");
    return FAV_OK;
}

int launch_cert_prepare(const uint8_t* mask, const float* backward_flo, int invert, int fix_occ, int border, int r,
                        float* cert_tmp, float* cert, int H, int W, hipStream_t st)
{
    FAV_REQUIRE(r >= 1 && r <= MF_RMAX, "min filter: window %d unsupported (1..%d)", r, MF_RMAX);
    (void)cert_tmp;                           // (the un-eroded certainty is no longer materialised: the erosion reads the mask byte)
    hipLaunchKernelGGL(min_filter_kernel<1>, dim3((W + MF_TX - 1) / MF_TX, (H + MF_TY - 1) / MF_TY), dim3(256), 0, st, nullptr, mask,
                       reinterpret_cast<const float2*>(backward_flo), invert, fix_occ, border, cert, H, W, r, nullptr, nullptr, nullptr, nullptr);
    FAV_LAUNCH_CHECK("min_filter_kernel");
    return FAV_OK;
}

// forward-backward check + certainty options + erosion of one frame in one launch (A3/A4 + fav.lua:99-112 + A5)
int launch_check_cert(const float* backward_flo, const float* forward_flo, const float* structure, const float* avg, uint8_t* mask_out,
                      int invert, int fix_occ, int border, int r, float* cert, int H, int W, hipStream_t st)
{
    FAV_REQUIRE(r >= 1 && r <= MF_RMAX, "min filter: window %d unsupported (1..%d)", r, MF_RMAX);
    hipLaunchKernelGGL(min_filter_kernel<2>, dim3((W + MF_TX - 1) / MF_TX, (H + MF_TY - 1) / MF_TY), dim3(256), 0, st, nullptr, nullptr,
                       reinterpret_cast<const float2*>(backward_flo), invert, fix_occ, border, cert, H, W, r,
                       reinterpret_cast<const float2*>(forward_flo), structure, avg, mask_out);
    FAV_LAUNCH_CHECK("min_filter_kernel<check>");
    return FAV_OK;
}

// check + certainty options + erosion + input assembly of one frame in one launch (check_prep_kernel); needs a previous frame
int launch_check_prep(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float* backward_flo, const float* forward_flo,
                      const float* structure, const float* avg, uint8_t* mask_out, float* cert, int invert, int fix_occ, int border, int r,
                      int H, int W, int pad, float* in8, hipStream_t st, int fill_random, unsigned seed, unsigned index)
{
    FAV_REQUIRE(r >= 1 && r <= MF_RMAX, "min filter: window %d unsupported (1..%d)", r, MF_RMAX);
    FAV_REQUIRE(prev_rgb != nullptr && pad < H && pad < W, "fused check + input assembly: needs a previous frame and pad < H, W");
    hipLaunchKernelGGL(check_prep_kernel, dim3((W + MF_TX - 1) / MF_TX, (H + MF_TY - 1) / MF_TY), dim3(256), 0, st, frame_hwc, prev_rgb, Hs, Ws,
                       reinterpret_cast<const float2*>(backward_flo), reinterpret_cast<const float2*>(forward_flo), structure, avg, mask_out, cert,
                       invert, fix_occ, border, r, H, W, pad, in8, fill_random, seed, index);
    FAV_LAUNCH_CHECK("check_prep_kernel");
    return FAV_OK;
}

int launch_prep_input(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float* backward_flo, const float* cert,
                      int border, int H, int W, int pad, float* in8, hipStream_t st, int fill_random, unsigned seed, unsigned index, int* q0_out)
{
    hipLaunchKernelGGL(prep_input_kernel, dim3((W + 2 * pad + 255) / 256, H + 2 * pad), dim3(256), 0, st, frame_hwc,
                       prev_rgb, Hs, Ws, reinterpret_cast<const float2*>(backward_flo), cert, border, H, W, pad, in8, fill_random, seed, index, q0_out);
    FAV_LAUNCH_CHECK("prep_input_kernel");
    return FAV_OK;
}

int launch_pad_nhwc(const float* in, int Hp, int Wp, int in_pitch, int C, int ups, float* out, int pl, int pr, int pt, int pb, int mode, hipStream_t st)
{
    FAV_REQUIRE(in && out && C > 0 && (C & 3) == 0 && pl >= 0 && pr >= 0 && pt >= 0 && pb >= 0 && (mode == 0 || mode == 1) && (ups == 0 || ups == 1), "padding layer: bad argument");
    const int Hl = Hp << ups, Wl = Wp << ups;
    FAV_REQUIRE(mode == 1 || (pl < Wl && pr < Wl && pt < Hl && pb < Hl), "reflection padding must be smaller than the tensor");
    const int Ho = Hl + pt + pb, Wo = Wl + pl + pr;
    const size_t total = (size_t)Ho * Wo * (C / 4);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(in), Hl, Wl, in_pitch, C / 4, ups,
                       reinterpret_cast<float4*>(out), Ho, Wo, pl, pt, mode);
    FAV_LAUNCH_CHECK("pad_nhwc_kernel");
    return FAV_OK;
}

int launch_store_flag(uint32_t* flag_host, uint32_t value, hipStream_t st)
{
    hipLaunchKernelGGL(store_flag_kernel, dim3(1), dim3(1), 0, st, flag_host, value);
    FAV_LAUNCH_CHECK("store_flag_kernel");
    return FAV_OK;
}

int launch_unpad_input(const float* in8, int H, int W, int pad, float* in7, hipStream_t st)
{
    hipLaunchKernelGGL(unpad_input_kernel, dim3((unsigned)(((size_t)H * W + 255) / 256)), dim3(256), 0, st, in8, H, W, pad, in7);
    FAV_LAUNCH_CHECK("unpad_input_kernel");
    return FAV_OK;
}

int launch_temporal_loss(const float* prev_rgb, const float* cur_rgb, const float* backward_flo, const uint8_t* cert_u8, int border,
                         int H, int W, double* partial256, hipStream_t st)
{
    hipLaunchKernelGGL(temporal_loss_kernel, dim3(256), dim3(256), 0, st, prev_rgb, cur_rgb, reinterpret_cast<const float2*>(backward_flo),
                       cert_u8, border, H, W, partial256);
    FAV_LAUNCH_CHECK("temporal_loss_kernel");
    return FAV_OK;
}

int launch_quantize_rgb8(const float* rgb_planar, uint8_t* out_hwc, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)(((size_t)H * W + 255) / 256)), dim3(256), 0, st, rgb_planar,
                       out_hwc, H, W);
    FAV_LAUNCH_CHECK("quantize_kernel");
    return FAV_OK;
}

}  // namespace fav
