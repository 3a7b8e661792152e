// kernels_up2.hip -- U2 + c3s1-64 (nn.SpatialUpSamplingNearest(2) followed by the 3x3 128 -> 64 convolution, models_video.lua:123-128)
// computed on the PHYSICAL pixels of its input.
//
// The upsampled image holds every pixel 2 x 2 times, so a 3x3 window covers only 2 x 2 distinct input pixels: for each of the four
// output phases (py, px) = (row parity, column parity) the layer is a 2x2 convolution on the physical image with merged weights
// (up2_pack.h) -- 4 multiply-adds per output and channel pair instead of 9, i.e. 2.25x fewer matrix instructions than the
// halo-resident kernel spends on the upsampled image, with the same operands (only the weight sums are re-associated).
//
// Block = 8 waves on one CU, tile = 8 x 32 physical pixels (-> 16 x 64 output pixels x 64 channels):
//   * wave = (phase, 32-channel tile): 4 phases x 2 = 8 waves.  A wave owns ALL 256 pixels of the tile for its phase and channel
//     tile: 8 accumulators (one per tile row, 128 registers).  Every weight fragment is then used by 8 MFMAs in a row and by no
//     other wave, so weights go global -> registers (packed in fragment order on the host, one 1 KiB load per 32 MFMAs, prefetched
//     one step ahead): no LDS ring, no barrier for weights
//   * input: per 32-channel slice the (8+2) x (32+2) halo sits in LDS (pixel pitch 36 floats, two buffers); a wave's four taps are
//     the halo shifted by (py + a, px + b): all fragment addresses are one per-lane base + immediates.  Staging: six 16-byte
//     pieces per thread and slice, one in flight at a time; raw buffer loads return zero outside the image, the pending
//     InstanceNorm/ReLU of the input is applied on the way (0.14 vector-ALU instructions per MFMA) and masked there
//   * ONE barrier per slice (512 MFMAs per wave)
//   * epilogue: bias, NHWC stores to (2y + py, 2x + px), per-tile InstanceNorm partials (mean, M2, count) merged over the four
//     phase waves of a channel tile
#include <algorithm>
#include <cstdlib>

#include "fav_internal.h"
#include "up2_pack.h"

namespace fav {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

constexpr int LDSS = 36;                     // pixel pitch in floats
constexpr int U2_HW = 34, U2_HP = 10 * 34;   // halo: 10 rows x 34 pixels
constexpr int U2_HPP = 384;                  // padded to 6 pieces x 512 threads / 8 chunks (pixels 340..383 are scratch)
constexpr int U2_HB = U2_HPP * LDSS;         // floats per halo buffer (55 296 B)

struct Up2Args {
    const float* in; const float* wpk; const float* bias; const float* scale1; const float* shift1;
    float* out; float2* partials; int* counts;
    int PH, PW, IWp, CIN, tiles_x, tiles_y, relu1;
    // conv3_up2w_kernel<true> only (more than 64 filters): the output channels are computed in `groups` groups of 64, a work item =
    // (tile, group), group-minor; COUTP = the channel pitch of out / partials; wpk holds one packed block (conv_up2w_pack) per group
    int COUTP, groups;
};

// AFF: the input carries a pending per-channel scale/shift (+ReLU) -- U2 is followed by InstanceNormalization + ReLU in the reference's
// architecture strings (models_video.lua:94-98,119-131), so the canonical layer sees IN(join) through the upsampling.  Zero padding
// applies AFTER that transform: out-of-image pieces are masked, not transformed.
template <bool AFF>
__global__ __launch_bounds__(512, 2) void conv3_up2_kernel(const Up2Args p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hs = smem;                  // [2][U2_HB]
    float* const aff = smem + 2 * U2_HB;     // [2][CIN]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int py = wave >> 2, px = (wave >> 1) & 1, nt = wave & 1;
    const int CIN = p.CIN, nslices = CIN >> 5, nsteps = nslices * 16;      // a step = (tap, group of 8 channels): 32 MFMAs per wave
    const int m = lane & 31, h = lane >> 5;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, nsteps * 8192, 0x00020000);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.PH * p.IWp * CIN * 4, 0x00020000);
    const int wlo = lane * 16, wso = wave * 1024;                          // weights: lane * 16 + [step * 8192 + wave * 1024]

    // fragments: tile row j, tap (a, b), channel group kg -> halo pixel (py + a + j, px + b + m), channels kg * 8 + 4 h ..
    const float* const ab = Hs + ((py * U2_HW) + px + m) * LDSS + 4 * h;
    // staging pieces e = t + 512 i: halo pixel e >> 3 (row-major 10 x 34; pixels >= 340 are scratch), 16-byte chunk e & 7
    const int c4 = t & 7;
    float* const hst = Hs + (t >> 3) * LDSS + c4 * 4;                      // piece i: + 64 i pixels

    if (AFF) { for (int i = t; i < CIN; i += 512) { aff[i] = p.scale1[i]; aff[CIN + i] = p.shift1[i]; } __syncthreads(); }
    const float lo1 = (AFF && p.relu1) ? 0.f : -INFINITY;
    const float* const affr = aff + c4 * 4;

    const int ntiles = p.tiles_x * p.tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int sy0 = ty * 8, sx0 = tx * 32;
        int ho[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pix = (t >> 3) + 64 * i, hy = (pix * 1928) >> 16, hx = pix - hy * U2_HW;      // pix / 34 (pix < 384)
            const int sy = sy0 - 1 + hy, sx = sx0 - 1 + hx;
            const bool v = pix < U2_HP && (unsigned)sy < (unsigned)p.PH && (unsigned)sx < (unsigned)p.PW;
            ho[i] = v ? ((sy * p.IWp + sx) * CIN + c4 * 4) * 4 : -16;                               // -16: past the buffer -> reads as zero
        }

        v4f hq;                            // the halo piece in flight
        v4f fa[2][4], fb[2];
#define U2_LOAD_H(i_, slice_) { hq = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[i_], (slice_) * 128, 0)); }
#define U2_XF(v_, i_, slice_)                                                                       \
        { if (AFF) { const v4f sc_ = *reinterpret_cast<const v4f*>(affr + (slice_) * 32), sh_ = *reinterpret_cast<const v4f*>(affr + CIN + (slice_) * 32); \
                     const float mk_ = ho[i_] >= 0 ? 1.f : 0.f;                                      \
                     v_.x = fmaxf(fmaf(v_.x, sc_.x, sh_.x), lo1) * mk_; v_.y = fmaxf(fmaf(v_.y, sc_.y, sh_.y), lo1) * mk_; \
                     v_.z = fmaxf(fmaf(v_.z, sc_.z, sh_.z), lo1) * mk_; v_.w = fmaxf(fmaf(v_.w, sc_.w, sh_.w), lo1) * mk_; } }
#define U2_STORE_H(i_, buf_, slice_) { U2_XF(hq, i_, slice_); *reinterpret_cast<v4f*>(hst + (buf_) * U2_HB + (i_) * 64 * LDSS) = hq; }
#define U2_LOAD_B(set_, step_) { fb[set_] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, wso + (step_) * 8192, 0)); }
        // A fragments of half-step (tap TP_, group KG_, row half HF_) from halo buffer par_
#define U2_READ_A(set_, par_, TP_, KG_, HF_)                                                        \
        { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                          \
              fa[set_][jj] = *reinterpret_cast<const v4f*>(ab + (par_) * U2_HB + ((((TP_) >> 1) + 4 * (HF_) + jj) * U2_HW + ((TP_) & 1)) * LDSS + (KG_) * 8); }
#define U2_MFMA(aset_, bset_, HF_)                                                                  \
        { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) acc[4 * (HF_) + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[aset_][jj].x, fb[bset_].x, acc[4 * (HF_) + jj], 0, 0, 0); \
          _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) acc[4 * (HF_) + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[aset_][jj].y, fb[bset_].y, acc[4 * (HF_) + jj], 0, 0, 0); \
          _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) acc[4 * (HF_) + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[aset_][jj].z, fb[bset_].z, acc[4 * (HF_) + jj], 0, 0, 0); \
          _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) acc[4 * (HF_) + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[aset_][jj].w, fb[bset_].w, acc[4 * (HF_) + jj], 0, 0, 0); }
#define U2_FENCE() __builtin_amdgcn_sched_barrier(0)

        // prologue: slice 0 -> buffer 0 (all six pieces in flight at once: the accumulators are not live yet)
        {
            v4f q0[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) q0[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[i], 0, 0));
            U2_LOAD_B(0, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) { U2_XF(q0[i], i, 0); *reinterpret_cast<v4f*>(hst + i * 64 * LDSS) = q0[i]; }
        }
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __syncthreads();

        // K loop.  Per slice 16 steps (4 taps x 4 channel groups) of two half-steps (tile rows 0-3, 4-7), 16 MFMAs each:
        //   half-step q: the A fragments of q + 1 are read while q's MFMAs run; at the first half of a step the NEXT step's weight
        //   fragment is requested; pieces of the next slice's halo: piece i requested at step 2 i, first half, stored at step 2 i + 1,
        //   second half (buffer par ^ 1, last read before the previous slice's barrier)
        for (int s = 0; s < nslices; ++s) {
            const int par = s & 1;
            const int sn = min(s + 1, nslices - 1);
            U2_READ_A(0, par, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int tp = st >> 2, kg = st & 3;
                const int gstep = s * 16 + st;
                // first half
                U2_READ_A(1, par, tp, kg, 1);
                U2_LOAD_B((st + 1) & 1, min(gstep + 1, nsteps - 1));
                if ((st & 1) == 0 && st < 12) U2_LOAD_H(st >> 1, sn);
                U2_FENCE(); U2_MFMA(0, st & 1, 0); U2_FENCE();
                // second half
                if (st < 15) U2_READ_A(0, par, (st + 1) >> 2, (st + 1) & 3, 0);
                if ((st & 1) == 1 && st < 12) U2_STORE_H(st >> 1, par ^ 1, sn);
                U2_FENCE(); U2_MFMA(1, st & 1, 1); U2_FENCE();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
#undef U2_LOAD_H
#undef U2_STORE_H
#undef U2_XF
#undef U2_LOAD_B
#undef U2_READ_A
#undef U2_MFMA
#undef U2_FENCE

        // ---- epilogue: acc[j][r] = output (2 (sy0 + j) + py, 2 (sx0 + mi) + px), channel nt * 32 + n;  mi = (r & 3) + 8 (r >> 2) + 4 h
        const int n = lane & 31, co = nt * 32 + n;
        const float bv = p.bias[co];
        const int OW = 2 * p.PW;
        float sm = 0.f; int nv = 0;
        const bool full = sy0 + 8 <= p.PH && sx0 + 32 <= p.PW;             // (220 of the 230 tiles at 1280x720)
        char* const ob = reinterpret_cast<char*>(p.out + ((size_t)(2 * sy0 + py) * OW + 2 * (sx0 + 4 * h) + px) * 64 + co);
        const size_t jpitch = (size_t)2 * OW * 256;                        // two output rows, 64 channels of 4 bytes
        if (full) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[j][r] + bv;
                    acc[j][r] = v;
                    *reinterpret_cast<float*>(ob + j * jpitch + ((r & 3) + 8 * (r >> 2)) * 512) = v;
                    sm += v;
                }
            nv = 128;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sy = sy0 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sx = sx0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = acc[j][r] + bv;
                    acc[j][r] = v;
                    if (sy < p.PH && sx < p.PW) { *reinterpret_cast<float*>(ob + j * jpitch + ((r & 3) + 8 * (r >> 2)) * 512) = v; sm += v; ++nv; }
                }
            }
        }
        if (p.partials != nullptr) {
            float2* stt = reinterpret_cast<float2*>(Hs);             // [8 waves][32]
            int* wn = reinterpret_cast<int*>(Hs + 2 * 8 * 32);         // [8]
            const int nw = nv + __shfl_xor(nv, 32);
            const float ssum = sm + __shfl_xor(sm, 32);
            const float mu = nw ? ssum / (float)nw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sy = sy0 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sx = sx0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float d = acc[j][r] - mu;
                    if (sy < p.PH && sx < p.PW) q = fmaf(d, d, q);
                }
            }
            q += __shfl_xor(q, 32);
            if (lane < 32) stt[wave * 32 + lane] = make_float2(mu, q);
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < 64) {
                // channel t = ntc * 32 + nn: exact merge (Chan et al.) of the four phase waves 2 ph + ntc
                const int ntc = t >> 5, nn = t & 31;
                int cnt = 0; float s1 = 0.f;
                for (int ph = 0; ph < 4; ++ph) { const int w = 2 * ph + ntc; cnt += wn[w]; s1 += (float)wn[w] * stt[w * 32 + nn].x; }
                const float mean = cnt ? s1 / (float)cnt : 0.f;
                float m2 = 0.f;
                for (int ph = 0; ph < 4; ++ph) { const int w = 2 * ph + ntc; const float d = stt[w * 32 + nn].x - mean; m2 += stt[w * 32 + nn].y + (float)wn[w] * d * d; }
                p.partials[(size_t)tile * 64 + t] = make_float2(mean, m2);
                if (t == 0) p.counts[tile] = cnt;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same layer with F(2x2,3x3) minimal filtering on top of the upsampling identity (up2_pack.h, second half): the transform line
// i = 2 vanishes on repeated rows, NINE positions remain, each a [pixels] x [CIN] x [64] GEMM whose operand is a difference of
// neighbouring physical pixels -- 2.25 multiply-adds per output instead of 4 (phase merge above) or 9 (direct form).
//
// Block = 12 waves (three per SIMD, 168 registers each), tile = 4 rows x 32 physical pixels: wave = (transform row i in {0, 1, 3},
// tile row) owning the three positions (i, 0), (i, 1), (i, 3) for the row's 32 pixels and all 64 output channels: 6 accumulator
// tiles.  The three waves of a SIMD (w, w + 4, w + 8) are one of each kind.  Per 8-channel group a wave reads two halo rows x
// three columns, forms its line  L = Ra - kappa Rb  (kappa = 0 for i = 1) and the three column operands (5 vector instructions
// per register, 0.83 per MFMA); weights are packed per position in fragment order and go global -> registers, reloaded in place
// right after their last use (the four waves of a kind share the L1 lines).  Output transform: every wave folds its columns
// (Z_i0 = M_i0 + M_i1, Z_i1 = M_i1 - M_i3), the kinds 0 and 3 hand their two folds over through LDS, the kind-1 wave adds
// Y_0b = Z_0b + Z_1b, Y_1b = Z_1b - Z_3b, bias, stores the 2 x 2 output pixels and takes the statistics.
// ------------------------------------------------------------------------------------------------
constexpr int UW_HW = 34, UW_HP = 6 * 34;      // halo 6 x 34 pixels
constexpr int UW_HPP = 288;                    // padded to 3 pieces x 768 threads / 8 chunks
constexpr int UW_HB = UW_HPP * LDSS;
static_assert(2 * 288 * LDSS <= 8 * 64 * 2 * LDSS, "halo buffers inside the exchange area");
constexpr int UW_ZS = 8 * 64 * 2 * LDSS;       // exchange: [kind 0|3 x 4 rows][64 channels][2 folds][36] floats (147 KB)

template <bool WIDE>
__global__ __launch_bounds__(768) void conv3_up2w_kernel(const Up2Args p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hs = smem;                  // [2][UW_HB]; the epilogue's exchange area overlays it
    float* const aff = smem + UW_ZS;         // [2][CIN]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kind = wave >> 2, row = wave & 3;                            // kind 0, 1, 2 <-> transform row i = 0, 1, 3
    const int CIN = p.CIN, nslices = CIN >> 5, nkg = CIN >> 3;
    const int m = lane & 31, h = lane >> 5;
    const int ngrp = WIDE ? p.groups : 1, PITCH = WIDE ? p.COUTP : 64;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, nkg * 18 * 1024 * ngrp, 0x00020000);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.PH * p.IWp * CIN * 4, 0x00020000);
    const int wlo = lane * 16, wso0 = kind * 3 * 2048;                     // weights: lane * 16 + [group * nkg * 18432 + kg * 18432 + (3 kind + jj) * 2048 + nt * 1024]
    // operand rows: kind 0: halo rows (row, row + 1), L = Ra - Rb;  kind 1: (row + 1, row + 1), L = Ra;  kind 2: (row + 1, row + 2), L = Ra - Rb
    const float kap = kind == 1 ? 0.f : 1.f;
    const float* const ra = Hs + ((row + (kind == 0 ? 0 : 1)) * UW_HW + m) * LDSS + 4 * h;
    const float* const rb = Hs + ((row + (kind == 0 ? 1 : kind == 1 ? 1 : 2)) * UW_HW + m) * LDSS + 4 * h;
    const int c4 = t & 7;
    float* const hst = Hs + (t >> 3) * LDSS + c4 * 4;                       // piece i: + 96 i pixels
    for (int i = t; i < CIN; i += 768) { aff[i] = p.scale1[i]; aff[CIN + i] = p.shift1[i]; }
    __syncthreads();
    const float lo1 = p.relu1 ? 0.f : -INFINITY;
    const float* const affr = aff + c4 * 4;

    const int ntiles = p.tiles_x * p.tiles_y * ngrp;      // work items: (pixel tile) * groups + group
    for (int item = blockIdx.x; item < ntiles; item += gridDim.x) {
        const int tile = WIDE ? item / ngrp : item, grp = WIDE ? item - tile * ngrp : 0;
        const int wso = wso0 + grp * nkg * 18432;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int sy0 = ty * 4, sx0 = tx * 32;
        int ho[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pix = (t >> 3) + 96 * i, hy = (pix * 1928) >> 16, hx = pix - hy * UW_HW;      // pix / 34 (pix < 288)
            const int sy = sy0 - 1 + hy, sx = sx0 - 1 + hx;
            const bool v = pix < UW_HP && (unsigned)sy < (unsigned)p.PH && (unsigned)sx < (unsigned)p.PW;
            ho[i] = v ? ((sy * p.IWp + sx) * CIN + c4 * 4) * 4 : -16;
        }
#define UW_XF(v_, i_, slice_)                                                                       \
        { const v4f sc_ = *reinterpret_cast<const v4f*>(affr + (slice_) * 32), sh_ = *reinterpret_cast<const v4f*>(affr + CIN + (slice_) * 32); \
          const float mk_ = ho[i_] >= 0 ? 1.f : 0.f;                                                \
          v_.x = fmaxf(fmaf(v_.x, sc_.x, sh_.x), lo1) * mk_; v_.y = fmaxf(fmaf(v_.y, sc_.y, sh_.y), lo1) * mk_; \
          v_.z = fmaxf(fmaf(v_.z, sc_.z, sh_.z), lo1) * mk_; v_.w = fmaxf(fmaf(v_.w, sc_.w, sh_.w), lo1) * mk_; }
#define UW_LOAD_B(jj_, kg_)                                                                         \
        { const int so_ = wso + (kg_) * 18432 + (jj_) * 2048;                                       \
          fb[jj_][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so_, 0)); \
          fb[jj_][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo + 1024, so_, 0)); }
#define UW_READ_R(par_, kg_)                                                                        \
        { _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                        \
              Ra[dx] = *reinterpret_cast<const v4f*>(ra + (par_) * UW_HB + dx * LDSS + (kg_) * 8);  \
              Rb[dx] = *reinterpret_cast<const v4f*>(rb + (par_) * UW_HB + dx * LDSS + (kg_) * 8); } }

        v4f fb[3][2], Ra[3], Rb[3], V[3];
        f32x16 acc[3][2];
        // prologue: slice 0 -> buffer 0
        {
            v4f q0[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) q0[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[i], 0, 0));
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) UW_LOAD_B(jj, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) { UW_XF(q0[i], i, 0); *reinterpret_cast<v4f*>(hst + i * 96 * LDSS) = q0[i]; }
        }
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][nt][r] = 0.f;
        __syncthreads();
        UW_READ_R(0, 0);

        for (int s = 0; s < nslices; ++s) {
            const int par = s & 1;
            const int sn = min(s + 1, nslices - 1);
            v4f hq;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int kgg = s * 4 + kg;
                {   // this wave's line, then its three column operands
                    v4f L[3];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        L[dx].x = fmaf(-kap, Rb[dx].x, Ra[dx].x); L[dx].y = fmaf(-kap, Rb[dx].y, Ra[dx].y);
                        L[dx].z = fmaf(-kap, Rb[dx].z, Ra[dx].z); L[dx].w = fmaf(-kap, Rb[dx].w, Ra[dx].w);
                    }
                    V[0] = L[0] - L[1]; V[1] = L[1]; V[2] = L[1] - L[2];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kg < 3) { UW_READ_R(par, kg + 1); }
                if (kg < 3) hq = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[kg], sn * 128, 0));   // one halo piece of the next slice per group
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[jj][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[jj][st], fb[jj][nt][st], acc[jj][nt], 0, 0, 0);
                    UW_LOAD_B(jj, min(kgg + 1, nkg - 1));      // reloaded in place: needed again one group from now
                }
                if (kg < 3) { UW_XF(hq, kg, sn); *reinterpret_cast<v4f*>(hst + (par ^ 1) * UW_HB + kg * 96 * LDSS) = hq; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            UW_READ_R(par ^ 1, 0);
        }
#undef UW_XF
#undef UW_LOAD_B
#undef UW_READ_R
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                    // (the prefetched reads above are done: the exchange area may overwrite the halo buffers)

        // ---- output transform.  acc[jj][nt][r] = M[i][j] of physical pixel (sy0 + row, sx0 + mi), channel nt * 32 + n.
        // Column fold in every wave: Z0 = M_i0 + M_i1, Z1 = M_i1 - M_i3.  Kinds 0 and 2 publish [slot = (kind ? 4 : 0) + row][channel][fold][pixel]
        const int n = lane & 31;
        if (kind != 1) {
            float* const zw = Hs + (((kind ? 4 : 0) + row) * 64 + n) * 2 * LDSS + 4 * h;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4f z0, z1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        z0[e] = acc[0][nt][r] + acc[1][nt][r]; z1[e] = acc[1][nt][r] - acc[2][nt][r];
                    }
                    *reinterpret_cast<v4f*>(zw + nt * 32 * 2 * LDSS + 8 * g) = z0;
                    *reinterpret_cast<v4f*>(zw + nt * 32 * 2 * LDSS + LDSS + 8 * g) = z1;
                }
        }
        __syncthreads();
        const int OW = 2 * p.PW, sy = sy0 + row;
        float2* stt = reinterpret_cast<float2*>(smem + UW_ZS + 2 * CIN);      // [4 rows][64] (behind the affine table)
        int* wn = reinterpret_cast<int*>(smem + UW_ZS + 2 * CIN + 2 * 4 * 64);  // [4]
        if (kind == 1) {
            const float* const z0r = Hs + ((0 + row) * 64 + n) * 2 * LDSS + 4 * h;      // kind 0 (i = 0)
            const float* const z3r = Hs + ((4 + row) * 64 + n) * 2 * LDSS + 4 * h;      // kind 2 (i = 3)
            float sm[2] = {0.f, 0.f}; int nv = 0;
            float y3[2][16];                   // the fourth output of every pixel (the other three replace the accumulators)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int co = grp * 64 + nt * 32 + n;
                const float bv = p.bias[co];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f a0 = *reinterpret_cast<const v4f*>(z0r + nt * 32 * 2 * LDSS + 8 * g), a1 = *reinterpret_cast<const v4f*>(z0r + nt * 32 * 2 * LDSS + LDSS + 8 * g);
                    const v4f c0 = *reinterpret_cast<const v4f*>(z3r + nt * 32 * 2 * LDSS + 8 * g), c1 = *reinterpret_cast<const v4f*>(z3r + nt * 32 * 2 * LDSS + LDSS + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const int sx = sx0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float b0 = acc[0][nt][r] + acc[1][nt][r], b1 = acc[1][nt][r] - acc[2][nt][r];      // Z_1b of this wave
                        const float y00 = a0[e] + b0 + bv, y01 = a1[e] + b1 + bv, y10 = b0 - c0[e] + bv, y11 = b1 - c1[e] + bv;
                        acc[0][nt][r] = y00; acc[1][nt][r] = y01; acc[2][nt][r] = y10; y3[nt][r] = y11;      // kept for the statistics
                        if (sy < p.PH && sx < p.PW) {
                            float* o = p.out + ((size_t)(2 * sy) * OW + 2 * sx) * PITCH + co;
                            o[0] = y00; o[PITCH] = y01; o[(size_t)OW * PITCH] = y10; o[(size_t)OW * PITCH + PITCH] = y11;
                            sm[nt] += (y00 + y01) + (y10 + y11);
                            if (nt == 0) nv += 4;
                        }
                    }
                }
            }
            if (p.partials != nullptr) {
                const int nw = nv + __shfl_xor(nv, 32);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float ssum = sm[nt] + __shfl_xor(sm[nt], 32);
                    const float mu = nw ? ssum / (float)nw : 0.f;
                    float q = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int sx = sx0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (sy < p.PH && sx < p.PW) {
                            const float d0 = acc[0][nt][r] - mu, d1 = acc[1][nt][r] - mu, d2 = acc[2][nt][r] - mu, d3 = y3[nt][r] - mu;
                            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
                        }
                    }
                    q += __shfl_xor(q, 32);
                    if (lane < 32) stt[row * 64 + nt * 32 + lane] = make_float2(mu, q);
                }
                if (lane == 0) wn[row] = nw;
            }
        }
        __syncthreads();
        if (p.partials != nullptr && t < 64) {
            int cnt = 0; float s1 = 0.f;
            for (int w = 0; w < 4; ++w) { cnt += wn[w]; s1 += (float)wn[w] * stt[w * 64 + t].x; }
            const float mean = cnt ? s1 / (float)cnt : 0.f;
            float m2 = 0.f;
            for (int w = 0; w < 4; ++w) { const float d = stt[w * 64 + t].x - mean; m2 += stt[w * 64 + t].y + (float)wn[w] * d * d; }
            p.partials[(size_t)tile * PITCH + grp * 64 + t] = make_float2(mean, m2);
            if (t == 0 && grp == 0) p.counts[tile] = cnt;
        }
        __syncthreads();            // the exchange area / statistics scratch are free for the next tile's halo
    }
}

}  // namespace

static bool up2_winograd() { static const bool w = diag_env("FAV_UP2_PHASES") == nullptr; return w; }      // (tuning: read once)
bool conv3_up2_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups)
{
    return k == 3 && stride == 1 && pad == 1 && ups == 1 && stages == 1 && cin_pitch % 32 == 0 && cin_pitch >= 32 && cin_pitch <= 256 &&
           cout == coutp && (cout == 64 || (up2_winograd() && cout % 64 == 0 && cout <= 512));      // the nine-position kernel: any number of 64-filter groups
}
int conv3_up2_tiles(int OH, int OW) { return ((OH / 2 + (up2_winograd() ? 3 : 7)) / (up2_winograd() ? 4 : 8)) * ((OW / 2 + 31) / 32); }

int launch_conv3_up2(const ConvLaunch& c, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3_up2_eligible(c.CIN, c.COUT, c.COUTp, c.KH, c.stride, c.pad, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && !c.stuff && wpk,
                "upsampled 3x3 conv: not eligible");
    FAV_REQUIRE((c.IH & 1) == 0 && (c.IW & 1) == 0 && c.OH == c.IH && c.OW == c.IW, "upsampled 3x3 conv: bad geometry");
    FAV_REQUIRE((long long)(c.IH / 2 + 1) * c.IWp * c.CIN < (1ll << 29), "upsampled 3x3 conv: tensor too large for 32-bit byte offsets");
    Up2Args a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.relu1 = c.pre.relu1; a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.PH = c.IH / 2; a.PW = c.IW / 2; a.IWp = c.IWp; a.CIN = c.CIN;
    const bool wg = up2_winograd();
    a.tiles_x = (a.PW + 31) / 32; a.tiles_y = wg ? (a.PH + 3) / 4 : (a.PH + 7) / 8;
    const size_t lds = wg ? (size_t)(UW_ZS + 2 * c.CIN + 2 * 4 * 64 + 8) * sizeof(float)          // exchange area (the halo buffers live inside it) | affine table | statistics
                          : (size_t)(2 * U2_HB + 2 * c.CIN) * sizeof(float);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_up2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_up2w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_up2w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        cus[dv] = prop_cus;
    }
    a.COUTP = c.COUTp; a.groups = c.COUTp / 64;
    FAV_REQUIRE((long long)(c.OH + 1) * c.OW * c.COUTp < (1ll << 31), "upsampled 3x3 conv: output too large");
    const int tiles = a.tiles_x * a.tiles_y * a.groups;
    const int grid = std::min(tiles, std::max(1, cus[dv] - c.reserve_cus));
    // (every U2 of the reference's builder is followed by a normalisation: stages == 1)
    if (wg && a.groups > 1) { hipLaunchKernelGGL(conv3_up2w_kernel<true>, dim3(grid), dim3(768), lds, st, a); }      // (wpk: the groups' nine-position blocks only)
    else if (wg) { a.wpk = wpk + conv_up2_packed_floats(c.CIN); hipLaunchKernelGGL(conv3_up2w_kernel<false>, dim3(grid), dim3(768), lds, st, a); }
    else hipLaunchKernelGGL(conv3_up2_kernel<true>, dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv3_up2_kernel");
    return FAV_OK;
}

}  // namespace fav
