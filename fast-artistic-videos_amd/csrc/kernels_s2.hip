// kernels_s2.hip -- the two 3x3 STRIDE-2 layers (d64: 32 -> 64 at 1360x800, d128: 64 -> 128 at 680x400; models_video.lua:88-92,
// nn.SpatialConvolution(.., 3, 3, 2, 2, 1, 1)) as a halo-resident implicit GEMM in the style of the round-2 kernels: weights packed in
// fragment order on the host and read global -> registers, the operand double-buffered in LDS, ONE barrier per K chunk, no
// stream-K hand-off (units are whole tiles, nothing is assumed co-resident).
//
//   * tile = TR output rows x 32 output columns x all output channels on one block of 8 waves; wave = (tile nt of 32 output channels,
//     output row): NTC x TR = 8 -- d64: 2 x 4 (2200 tiles at 1280x720 input), d128: 4 x 2 (1100 tiles: short tiles so that the 256
//     CUs finish within 5 rounds of 17 us instead of 3 rounds of 35 us).  One accumulator tile (16 registers) per wave
//   * K runs in CHUNKS of 16 input channels: the (2 TR + 1) x 65 pixel halo of a chunk is 64 bytes per pixel, stored at a pitch of 80
//     bytes (5 sixteen-byte slots: odd, so the 16 lanes a ds_read_b128 serves together hit 16 different slots) as an EVEN-column
//     plane (33 per row) and an ODD-column plane (32 per row): for every tap the 32 lanes of a wave (32 consecutive output columns
//     = input columns 2 m + kx) read 32 consecutive pixels of one plane with immediate offsets.  Two buffers of 47 KB (TR = 4): the
//     next chunk -- of this tile or of the block's next tile -- is requested while the first half of the current one is multiplied
//     and committed (pending InstanceNorm / ReLU of the producer applied, zero padding masked) during the second half
//   * a chunk = 2 channel groups x 9 taps x 4 MFMAs; the nine weight fragments of a group (one 16-byte load per lane each) are
//     requested one group ahead (two register sets); the four row waves of a channel tile read the same fragments (L1)
//   * epilogue per tile: bias, NHWC store, per-wave (mean, M2, count); the merge of the TR rows and the write of the tile's
//     InstanceNorm partial ride on the NEXT chunk's barrier -- no barrier of their own
#include <algorithm>
#include <cstdlib>

#include "fav_internal.h"
#include "s2_pack.h"

namespace fav {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

constexpr int SW_P = 20;                    // pixel pitch in floats: 16 channels + 4
constexpr int SW_EW = 33, SW_OW = 32;       // pixels per row of the even / odd column plane

struct S2wArgs {
    const float* in; const float* wpk; const float* bias; const float* scale1; const float* shift1;
    float* out; float2* partials; int* counts;
    int IH, IW, IWp, CIN, OH, OW, pad, tiles_x, tiles_y, stages, relu1;
    // WIDE instantiations only (more than 128 filters): the output channels are computed in `groups` groups of 32 NTC, a work item =
    // (tile, group), group-minor; COUTP = the channel pitch of out / partials; wpk holds one packed block (conv_s2w_pack) per group.
    // The grid is a multiple of `groups`: a persistent block stays with one group (its weight prefetch runs across tiles)
    int COUTP, groups;
};

template <int NTC, int TR> struct SwGeo {
    static constexpr int NW = NTC * TR;                     // waves per block: one per (tile of 32 output channels, output row)
    static constexpr int NTH = 64 * NW;
    static constexpr int HR = 2 * TR + 1;                   // halo rows
    static constexpr int EP = HR * SW_EW;                   // pixels in the even plane
    static constexpr int HP = EP + HR * SW_OW;              // halo pixels
    static constexpr int PS = NTH / 4;                      // pixels staged per pass of the block (four 16-byte pieces per pixel)
    static constexpr int NPC = (HP + PS - 1) / PS;          // pieces per thread and chunk
    static constexpr int HB = NPC * PS * SW_P;              // floats per buffer (pixels HP .. NPC * PS - 1 are scratch)
};

template <int NTC, int TR, bool WIDE = false>
__global__ __launch_bounds__(64 * NTC * TR, NTC * TR / 4) void conv3s2w_kernel(const S2wArgs p)
{
    using G = SwGeo<NTC, TR>;
    static_assert(G::NW % 4 == 0 && G::NW >= 8 && G::NW <= 16, "whole waves per SIMD");
    constexpr int COUT = NTC * 32;
    constexpr int EP = G::EP, HP = G::HP, NPC = G::NPC, HB = G::HB, PS = G::PS, NTH = G::NTH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Hs = smem;                                             // [2][HB]
    float* const aff = smem + 2 * HB;                                   // [2][CIN]
    float2* const stt = reinterpret_cast<float2*>(aff + 2 * p.CIN);    // [TR][COUT]
    int* const wn = reinterpret_cast<int*>(stt + TR * COUT);            // [TR]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nt = wave % NTC, row = wave / NTC;
    const int m = lane & 31, h = lane >> 5;
    const int CIN = p.CIN, nch = CIN >> 4;
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ngrp = WIDE ? p.groups : 1, grp = WIDE ? lb % p.groups : 0;      // (gridDim.x % groups == 0: every item of this block is of group grp)
    const int PITCH = WIDE ? p.COUTP : COUT;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk) + (size_t)grp * (nch * 18 * NTC * 256), 0, nch * 18 * NTC * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.IH * p.IWp * CIN * 4, 0x00020000);
    const int wlo = lane * 16, wnt = nt * 1024;                         // weights: lane * 16 + [((chunk * 2 + kg) * 9 + tap) * NTC * 1024 + nt * 1024]

    // fragments of tap (ky, kx), channel group kg: even plane for kx = 0 / 2 (shifted by one pixel), odd plane for kx = 1
    const float* const aE = Hs + ((2 * row) * SW_EW + m) * SW_P + 4 * h;
    const float* const aO = Hs + (EP + (2 * row) * SW_OW + m) * SW_P + 4 * h;
#define SW_AOFF(T_) (((T_) % 3 == 1) ? ((T_) / 3) * SW_OW * SW_P : (((T_) / 3) * SW_EW + ((T_) % 3 == 2 ? 1 : 0)) * SW_P)
#define SW_ABASE(T_) (((T_) % 3 == 1) ? aO : aE)

    // staging piece i of this thread: halo pixel p0 + PS i (plane-major), 16-byte chunk c4; its place inside the halo is fixed
    const int c4 = t & 3, p0 = t >> 2;
    float* const hst = Hs + p0 * SW_P + c4 * 4;
    int hyx[NPC], hbase[NPC];                                           // position inside the halo; byte offset from the halo's origin pixel
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int pe = p0 + PS * i;
        int hy = 0x7000, hx = 0;                                        // pixels past the halo (scratch): never inside the image
        if (pe < EP) { hy = pe / SW_EW; hx = 2 * (pe - hy * SW_EW); }
        else if (pe < HP) { const int q = pe - EP; hy = q / SW_OW; hx = 2 * (q - hy * SW_OW) + 1; }
        hyx[i] = hy << 16 | hx;
        hbase[i] = ((pe < HP ? (hy * p.IWp + hx) * p.CIN : 0) + c4 * 4) * 4;
    }
    for (int i = t; i < CIN; i += NTH) { aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f; }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float* const affr = aff + c4 * 4;

    const int ntiles = p.tiles_x * p.tiles_y * ngrp;      // work items: tile = (pixel tile) * groups + group
    int tile = lb;
    if (tile >= ntiles) return;

    int ho[NPC]; float hm[NPC];
    v4f hq[NPC];
    // interior tiles (the halo lies inside the image: all but the frame of border tiles): one add per piece, no masks
    bool hmask = false;
#define SW_TILE_SETUP(tile_)                                                                        \
    {   const int pt_ = WIDE ? (tile_) / ngrp : (tile_);                                            \
        const int ty_ = pt_ / p.tiles_x, tx_ = pt_ - ty_ * p.tiles_x;                               \
        const int iy0_ = 2 * ty_ * TR - p.pad, ix0_ = 2 * tx_ * 32 - p.pad;                         \
        hmask = iy0_ < 0 || ix0_ < 0 || iy0_ + 2 * TR >= p.IH || ix0_ + 64 >= p.IW;                  \
        if (!hmask) {                                                                               \
            const int org_ = (iy0_ * p.IWp + ix0_) * CIN * 4;                                       \
            _Pragma("unroll") for (int i = 0; i < NPC; ++i) ho[i] = hbase[i] + org_;                \
        } else {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < NPC; ++i) {                                       \
                const int iy = iy0_ + (hyx[i] >> 16), ix = ix0_ + (hyx[i] & 0xffff);                \
                const bool v = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);   \
                ho[i] = ((v ? (iy * p.IWp + ix) * CIN : 0) + c4 * 4) * 4;                           \
                hm[i] = v ? 1.f : 0.f;                                                              \
            } } }
#define SW_LOAD_H(chunk_)                                                                           \
    { _Pragma("unroll") for (int i = 0; i < NPC; ++i) hq[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[i], (chunk_) * 64, 0)); }
#define SW_COMMIT(chunk_, par_)                                                                     \
    {   const v4f sc_ = *reinterpret_cast<const v4f*>(affr + (chunk_) * 16), sh_ = *reinterpret_cast<const v4f*>(affr + CIN + (chunk_) * 16); \
        _Pragma("unroll") for (int i = 0; i < NPC; ++i) {                                           \
            hq[i].x = fmaxf(fmaf(hq[i].x, sc_.x, sh_.x), lo1); hq[i].y = fmaxf(fmaf(hq[i].y, sc_.y, sh_.y), lo1); \
            hq[i].z = fmaxf(fmaf(hq[i].z, sc_.z, sh_.z), lo1); hq[i].w = fmaxf(fmaf(hq[i].w, sc_.w, sh_.w), lo1); } \
        if (hmask) { _Pragma("unroll") for (int i = 0; i < NPC; ++i) hq[i] *= hm[i]; }              /* zero padding applies after the transform */ \
        _Pragma("unroll") for (int i = 0; i < NPC; ++i) *reinterpret_cast<v4f*>(hst + (par_) * HB + i * PS * SW_P) = hq[i]; }
    v4f fb[2][9], fa[3];
#define SW_LOAD_B(set_, chunk_, kg_)                                                                \
    {   const int so_ = (((chunk_) * 2 + (kg_)) * 9) * NTC * 1024 + wnt;                            \
        _Pragma("unroll") for (int tp = 0; tp < 9; ++tp)                                            \
            fb[set_][tp] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so_ + tp * NTC * 1024, 0)); }
#define SW_READ_A(slot_, T_, kg_, par_) { fa[slot_] = *reinterpret_cast<const v4f*>(SW_ABASE(T_) + SW_AOFF(T_) + (kg_) * 8 + (par_) * HB); }
    // one channel group: nine taps x four MFMA steps; the fragments of taps 0 and 1 were read by the caller, tap T + 2 is read at tap T
    // (NXT_: after the ninth tap of group 0 the first two taps of group 1 follow)
#define SW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SW_MFMA4(a_, b_)                                                                            \
    {   acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_).x, (b_).x, acc, 0, 0, 0);                   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_).y, (b_).y, acc, 0, 0, 0);                   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_).z, (b_).z, acc, 0, 0, 0);                   \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_).w, (b_).w, acc, 0, 0, 0); }
#define SW_TAPS(set_, kg_, par_, T0_, T1_, NXT_)                                                    \
    {   _Pragma("unroll") for (int tp = (T0_); tp < (T1_); ++tp) {                                  \
            if (tp + 2 < 9) { SW_READ_A_DYN((tp + 2) % 3, tp + 2, kg_, par_); }                      \
            else if (NXT_) { SW_READ_A_DYN((tp + 2) % 3, tp + 2 - 9, 1, par_); }                    \
            SW_MFMA4(fa[tp % 3], fb[set_][tp]);                                                     \
        } }
#define SW_GROUP(set_, kg_, par_, NXT_) SW_TAPS(set_, kg_, par_, 0, 9, NXT_)
    // (tap index is a compile-time constant after unrolling; the helper picks plane and offset from it)
#define SW_READ_A_DYN(slot_, T_, kg_, par_)                                                         \
    { const int T__ = (T_); const float* b__ = (T__ % 3 == 1) ? aO : aE;                            \
      const int o__ = (T__ % 3 == 1) ? (T__ / 3) * SW_OW * SW_P : ((T__ / 3) * SW_EW + (T__ % 3 == 2 ? 1 : 0)) * SW_P; \
      fa[slot_] = *reinterpret_cast<const v4f*>(b__ + o__ + (kg_) * 8 + (par_) * HB); }

    // ---- prologue: first chunk of the first tile -> buffer 0 (exposed latency, once per block)
    SW_TILE_SETUP(tile);
    SW_LOAD_H(0);
    SW_LOAD_B(0, 0, 0);
    __syncthreads();                        // transform table
    SW_COMMIT(0, 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    SW_READ_A_DYN(0, 0, 0, 0); SW_READ_A_DYN(1, 1, 0, 0);
    int par = 0, pend = -1;
    int ptile = WIDE ? tile / ngrp : tile;
    int oy0 = (ptile / p.tiles_x) * TR, ox0 = (ptile - (ptile / p.tiles_x) * p.tiles_x) * 32;
    const int n = lane & 31, co = nt * 32 + n;
    const float bv = p.bias[grp * COUT + co];
    for (;;) {
        const int ntile = tile + (int)gridDim.x;
        for (int chunk = 0; chunk < nch; ++chunk) {
            const bool last = chunk == nch - 1;
            const int nchunk = last ? 0 : chunk + 1;
            if (last && ntile < ntiles) SW_TILE_SETUP(ntile);           // (ho / hm now describe the tile being fetched)
            // Global requests are spread over the taps (one weight fragment per tap, the halo pieces with the first taps): eight waves
            // asking for 14 KB each right behind the barrier would keep the matrix pipes waiting for the address unit.
            // (after the block's last chunk the requests fetch a chunk nobody reads: no branches in the loop body)
            const int so1 = ((chunk * 2 + 1) * 9) * NTC * 1024 + wnt, so0 = ((nchunk * 2) * 9) * NTC * 1024 + wnt;
            // first channel group; requested: the weights of the second, the next chunk's halo
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                fb[1][tp] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so1 + tp * NTC * 1024, 0));
                if (tp < NPC) hq[tp] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho[tp], nchunk * 64, 0));
                if (tp + 2 < 9) { SW_READ_A_DYN((tp + 2) % 3, tp + 2, 0, par); } else { SW_READ_A_DYN((tp + 2) % 3, tp + 2 - 9, 1, par); }
                SW_FENCE(); SW_MFMA4(fa[tp % 3], fb[0][tp]); SW_FENCE();
            }
            // second channel group; requested: the weights of the next chunk's first group; the halo is committed behind tap 2
#pragma unroll
            for (int tp = 0; tp < 7; ++tp) {
                fb[0][tp] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so0 + tp * NTC * 1024, 0));
                if (tp + 2 < 9) { SW_READ_A_DYN((tp + 2) % 3, tp + 2, 1, par); }
                SW_FENCE(); SW_MFMA4(fa[tp % 3], fb[1][tp]); SW_FENCE();
                if (tp == 2) { __builtin_amdgcn_sched_barrier(0); SW_COMMIT(nchunk, par ^ 1); __builtin_amdgcn_sched_barrier(0); }
            }
            // the reads of taps 7 and 8 are issued: the chunk's barrier; the first two fragments of the next chunk are requested from
            // the other buffer while taps 7 and 8 multiply out of registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            {
                const v4f a7_ = fa[7 % 3], a8_ = fa[8 % 3];
                fb[0][7] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so0 + 7 * NTC * 1024, 0));
                fb[0][8] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, so0 + 8 * NTC * 1024, 0));
                SW_READ_A_DYN(0, 0, 0, par ^ 1); SW_READ_A_DYN(1, 1, 0, par ^ 1);
                SW_FENCE(); SW_MFMA4(a7_, fb[1][7]); SW_MFMA4(a8_, fb[1][8]); SW_FENCE();
            }
            if (pend >= 0) {
                // InstanceNorm partial of the previous tile: exact merge (Chan et al.) of its TR row waves
                if (t < COUT) {
                    int cnt = 0; float s1 = 0.f;
                    for (int w = 0; w < TR; ++w) { cnt += wn[w]; s1 += (float)wn[w] * stt[w * COUT + t].x; }
                    const float mean = cnt ? s1 / (float)cnt : 0.f;
                    float m2 = 0.f;
                    for (int w = 0; w < TR; ++w) { const float d = stt[w * COUT + t].x - mean; m2 += stt[w * COUT + t].y + (float)wn[w] * d * d; }
                    p.partials[(size_t)pend * PITCH + grp * COUT + t] = make_float2(mean, m2);
                    if (t == 0 && grp == 0) p.counts[pend] = cnt;
                }
                pend = -1;
            }
            par ^= 1;
        }
        // ---- epilogue of the tile: acc[r] = output (oy0 + row, ox0 + mi), channel co; mi = (r & 3) + 8 (r >> 2) + 4 h
        {
            const int oy = oy0 + row;
            float sm = 0.f; int nv = 0;
            float* const ob = p.out + ((size_t)oy * p.OW + ox0 + 4 * h) * PITCH + grp * COUT + co;
            const bool full = oy0 + TR <= p.OH && ox0 + 32 <= p.OW;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[r] + bv;
                    acc[r] = v;
                    ob[(size_t)((r & 3) + 8 * (r >> 2)) * PITCH] = v;
                    sm += v;
                }
                nv = 16;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mi = (r & 3) + 8 * (r >> 2);
                    const float v = acc[r] + bv;
                    acc[r] = v;
                    if (oy < p.OH && ox0 + 4 * h + mi < p.OW) { ob[(size_t)mi * PITCH] = v; sm += v; ++nv; }
                }
            }
            if (p.partials != nullptr) {
                const int nw = nv + __shfl_xor(nv, 32);
                const float ssum = sm + __shfl_xor(sm, 32);
                const float mu = nw ? ssum / (float)nw : 0.f;
                float q = 0.f;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = acc[r] - mu; q = fmaf(d, d, q); }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int mi = (r & 3) + 8 * (r >> 2);
                        const float d = acc[r] - mu;
                        if (oy < p.OH && ox0 + 4 * h + mi < p.OW) q = fmaf(d, d, q);
                    }
                }
                q += __shfl_xor(q, 32);
                if (lane < 32) stt[row * COUT + co] = make_float2(mu, q);
                if (lane == 0 && nt == 0) wn[row] = nw;
                pend = ptile;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        tile = ntile;
        if (tile >= ntiles) break;
        ptile = WIDE ? tile / ngrp : tile;
        oy0 = (ptile / p.tiles_x) * TR; ox0 = (ptile - (ptile / p.tiles_x) * p.tiles_x) * 32;
    }
    if (pend >= 0) {
        __syncthreads();
        if (t < COUT) {
            int cnt = 0; float s1 = 0.f;
            for (int w = 0; w < TR; ++w) { cnt += wn[w]; s1 += (float)wn[w] * stt[w * COUT + t].x; }
            const float mean = cnt ? s1 / (float)cnt : 0.f;
            float m2 = 0.f;
            for (int w = 0; w < TR; ++w) { const float d = stt[w * COUT + t].x - mean; m2 += stt[w * COUT + t].y + (float)wn[w] * d * d; }
            p.partials[(size_t)pend * PITCH + grp * COUT + t] = make_float2(mean, m2);
            if (t == 0 && grp == 0) p.counts[pend] = cnt;
        }
    }
#undef SW_AOFF
#undef SW_ABASE
#undef SW_TILE_SETUP
#undef SW_LOAD_H
#undef SW_COMMIT
#undef SW_LOAD_B
#undef SW_READ_A
#undef SW_GROUP
#undef SW_TAPS
#undef SW_MFMA4
#undef SW_FENCE
#undef SW_READ_A_DYN
}

template <int NTC, int TR, bool WIDE = false>
int launch_s2w_t(const S2wArgs& a, int reserve_cus, hipStream_t st)
{
    const auto kern = conv3s2w_kernel<NTC, TR, WIDE>;
    const size_t lds = (size_t)(2 * SwGeo<NTC, TR>::HB + 2 * a.CIN) * sizeof(float) + (size_t)TR * NTC * 32 * sizeof(float2) + 16 * sizeof(int);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        cus[dv] = prop_cus;
    }
    const int ngrp = WIDE ? a.groups : 1;
    const int tiles = a.tiles_x * a.tiles_y * ngrp;
    // (Rounds 2-5 launched a layer of many short tiles one tile per block next to the look-ahead queue: a persistent block that shared its
    //  CU with a side-queue kernel fell behind and its statically assigned tiles became the launch's tail.  Since round 6 the mask's
    //  long-lived kernels sit on the CUs this grid leaves free (kernels_consistency.hip, xcd_share) and the persistent form is the
    //  faster one again: d64 100 us against 137 us, profiles/r8l_4arg_cu_filling_ab.log.  FAV_S2W_TILE_GRID restores the old form.)
    const int slots = std::max(1, cus[dv] - reserve_cus);
    static const bool tile_grid = diag_env("FAV_S2W_TILE_GRID") != nullptr;      // (tuning: read once)
    int grid = (tile_grid && reserve_cus > 0 && tiles > 4 * slots) ? tiles : std::min(tiles, slots);
    if (WIDE) grid = std::max(ngrp, grid / ngrp * ngrp);      // a block stays with one group
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SwGeo<NTC, TR>::NTH), lds, st, a);
    FAV_LAUNCH_CHECK("conv3s2w_kernel");
    return FAV_OK;
}

}  // namespace

bool conv3s2w_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups)
{
    return k == 3 && stride == 2 && pad <= 1 && ups == 0 && stages <= 1 && cin_pitch % 16 == 0 && cin_pitch >= 32 && cin_pitch <= 512 &&
           cout == coutp && (coutp == 64 || (coutp % 128 == 0 && coutp <= 1024));      // 64, or any number of 128-filter groups
}
// tile rows: d64 4 (8 waves); d128 3 (12 waves: 737 tiles = 3 rounds on 256 CUs at 1280x720; with 2 rows 1100 tiles = 5 rounds where 4.3 would do)
// (measured and not kept: d64 with 6 rows on 12 waves -- 104 us against 98.8 us with 4 rows on 8: profiles/r02y_s2w_rows_ab.log)
static int s2w_rows(int coutp) { static const int r128 = diag_env("FAV_S2W_ROWS128") ? atoi(diag_env("FAV_S2W_ROWS128")) : 3; return coutp == 64 ? 4 : (r128 == 2 && coutp == 128 ? 2 : 3); }      // (tuning: read once)
int conv3s2w_tiles(int OH, int OW, int coutp) { const int tr = s2w_rows(coutp); return ((OH + tr - 1) / tr) * ((OW + 31) / 32); }

int launch_conv3s2w(const ConvLaunch& c, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3s2w_eligible(c.CIN, c.COUT, c.COUTp, c.KH, c.stride, c.pad, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && !c.stuff && wpk,
                "stride-2 conv: not eligible");
    FAV_REQUIRE((long long)(c.IH + 1) * c.IWp * c.CIN < (1ll << 29), "stride-2 conv: tensor too large for 32-bit byte offsets");
    S2wArgs a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.stages = c.pre.stages; a.relu1 = c.pre.relu1;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.CIN = c.CIN; a.OH = c.OH; a.OW = c.OW; a.pad = c.pad;
    const int tr = s2w_rows(c.COUTp);
    a.tiles_x = (c.OW + 31) / 32; a.tiles_y = (c.OH + tr - 1) / tr;
    a.COUTP = c.COUTp; a.groups = c.COUTp > 128 ? c.COUTp / 128 : 1;
    FAV_REQUIRE((long long)(c.OH + 1) * c.OW * c.COUTp < (1ll << 31), "stride-2 conv: output too large");
    if (a.groups > 1) return launch_s2w_t<4, 3, true>(a, c.reserve_cus, st);
    return c.COUTp == 64 ? launch_s2w_t<2, 4>(a, c.reserve_cus, st) : tr == 2 ? launch_s2w_t<4, 2>(a, c.reserve_cus, st) : launch_s2w_t<4, 3>(a, c.reserve_cus, st);
}

}  // namespace fav
