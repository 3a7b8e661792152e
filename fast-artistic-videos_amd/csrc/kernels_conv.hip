// kernels_conv.hip -- the transformer network's device code for gfx950 (MI355X, CDNA4).
//
// nn.SpatialConvolution (models_video.lua:20,32,80,93) as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain), with
//   * nn.InstanceNormalization + nn.ReLU of the PRODUCING layer applied on load (per-channel
//     scale/shift [+ReLU], up to two stacked stages), zero padding applied after the transform,
//   * nn.SpatialUpSamplingNearest(2) folded into the gather index map,
//   * bias add, the raw NHWC store and the InstanceNorm statistics of the OUTPUT (per-tile mean and
//     M2, merged later in fp64) in the epilogue,
//   * for the last layer: nn.Tanh, nn.MulConstant and the VGG de-processing in the epilogue.
// Activations are NHWC (channels-last) so that a K-slice (32 consecutive input channels of one
// filter tap) is one contiguous 128-byte line.  GEMM view: M = output pixels, N = output channels,
// K = taps * Cin.
//
// Tiling: 4 or 8 waves of 64 lanes.  Block tile 128 (M) x BN (N) x 32 (K), LDS double-buffered
// [rows][36] (row stride 36 floats makes the 16-byte fragment reads conflict-free), register-staged
// global->LDS copies issued two K-steps ahead and written to LDS in chunks interleaved with the MFMAs.
#include <algorithm>
#include <cstdlib>

#include "fav_internal.h"

namespace fav {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // native vector: struct float4 copies lower to memcpy through scratch

namespace {

// Launch-time facts that are cached per kernel instantiation are kept PER DEVICE (function attributes and CU counts belong to
// the device that is current at the launch): a single process may drive several GPUs.
constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

constexpr int BM = CONV_BM;   // 128 output pixels per block
constexpr int BK = 32;        // K elements per step
constexpr int LDSS = 36;      // LDS row stride in floats (144 B: 16-B aligned, conflict-free b128 reads)

struct ConvArgs {
    const float* in; const float* wgt; const float* bias;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    float* out; float2* partials; float* out_planar; float* out_raw;
    int IH, IW, IWp, ups, stuff, CIN;
    int COUT, COUTp, KH, KW, stride, pad, Kpad, OH, OW;
    int stages, relu1, relu2, final_mode;
    int cin_shift, kw_magic, ntaps_magic;
    float tanh_mul;
    float* sk_ws; unsigned* sk_flags; unsigned sk_epoch;     // stream-K hand-off (null = data-parallel)
    unsigned* sk_err;        // host-mapped word set when a hand-off wait times out (null = not reported)
    int reserve_cus;
};

// 16-byte write-through store (sc1): the stream-K partial tiles are published with these + `s_waitcnt vmcnt(0)` + an sc1 flag
// store, instead of plain stores + an agent-scope release fence (which writes back the whole XCD L2's dirty lines, i.e. also the
// output tiles other blocks are storing at that moment): MI355X_MICROARCH.md "publish-large" row, 8.2 -> 3.0 us per 64 KB.
__device__ __forceinline__ void store16_wt(void* p, v4f v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Per-tile InstanceNorm partials with ONE barrier: every wave reduces its own 32 pixels x 32 channels accumulator tile(s) to
// (mean, M2, count) in registers (two half-wave shuffles), the NW waves' results meet in LDS, and thread c merges the waves of
// channel c exactly (Chan et al.): mean = sum n_w mean_w / n, M2 = sum M2_w + n_w (mean_w - mean)^2.  (Before: block-wide
// sum -> barrier -> mean -> barrier -> M2 -> barrier -> barrier, four barriers per tile on an 8-wave block.)
__device__ __forceinline__ float2 merge_wave_stats(const float2* st, const int* wn, int NW, int pitch, int c, int* n_out)
{
    int n = 0; float s = 0.f;
    for (int w = 0; w < NW; ++w) { n += wn[w]; s += (float)wn[w] * st[w * pitch + c].x; }
    const float mean = n ? s / (float)n : 0.f;
    float m2 = 0.f;
    for (int w = 0; w < NW; ++w) { const float d = st[w * pitch + c].x - mean; m2 += st[w * pitch + c].y + (float)wn[w] * d * d; }
    *n_out = n;
    return make_float2(mean, m2);
}

// branch-free form used inside the MFMA loop: lo = 0 for ReLU, -inf for none; identity = scale 1, shift 0
__device__ __forceinline__ float4 affine4_lo(float4 v, const float* sc, const float* sh, float lo)
{
    const float4 s = *reinterpret_cast<const float4*>(sc);
    const float4 b = *reinterpret_cast<const float4*>(sh);
    v.x = fmaxf(fmaf(v.x, s.x, b.x), lo); v.y = fmaxf(fmaf(v.y, s.y, b.y), lo);
    v.z = fmaxf(fmaf(v.z, s.z, b.z), lo); v.w = fmaxf(fmaf(v.w, s.w, b.w), lo);
    return v;
}

__device__ __forceinline__ float4 affine4(float4 v, const float* sc, const float* sh, int relu)
{
    const float4 s = *reinterpret_cast<const float4*>(sc);
    const float4 b = *reinterpret_cast<const float4*>(sh);
    v.x = fmaf(v.x, s.x, b.x); v.y = fmaf(v.y, s.y, b.y); v.z = fmaf(v.z, s.z, b.z); v.w = fmaf(v.w, s.w, b.w);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

// One kernel, two work distributions:
//   SK = false  data-parallel: one block per output tile (grid = m-tiles x n-tiles);
//   SK = true   stream-K: a fixed grid (2 blocks per CU) splits the flattened (tile, K-step) space evenly.
//               A block's range is [tail of a tile shared with the previous block][whole tiles][head of a
//               tile shared with the next block].  The block holding the HEAD (k = 0) of a split tile owns its
//               epilogue and runs it LAST in its own timeline; the blocks holding the remaining K ranges
//               write their partial accumulators FIRST in their timelines, publish a flag (agent-scope
//               release), and the owner picks them up after an agent-scope acquire.  This removes the
//               "515 tiles on 512 slots" quantisation that cost the residual layers ~20 %.
template <int BN, int WM, int WN, bool SK = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8) ? 4 : 2) void conv_mfma_kernel(const ConvArgs p)   // 2 blocks per CU
{
    constexpr int NT = 64 * WM * WN;           // threads per block (4 or 8 waves)
    constexpr int RP = NT / 8;                 // tile rows staged per pass (8 threads x 16 B per 32-wide K slice)
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AROWS = BM / RP, BROWS = (BN + RP - 1) / RP;
    static_assert(BM % RP == 0 && (BN % RP == 0 || BN < RP), "staging layout");
    static_assert(AROWS <= 4 && BROWS <= 4 && BK == 32, "load/store chunks are tied to the 4 MFMA groups of a K-step");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM*LDSS]
    float* Bs = smem + 2 * BM * LDSS;         // [2][BN*LDSS]
    float* aff = Bs + 2 * BN * LDSS;          // [4][CIN]: scale1, shift1, scale2, shift2

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int M = p.OH * p.OW;
    const int CIN = p.CIN;
    const int ntaps = p.KH * p.KW;
    const int nsteps = p.Kpad / BK;
    const int mtiles = (M + BM - 1) / BM, ntiles = p.COUTp / BN;

    // XCD-aware block order: the dispatcher places block b on XCD b % 8 (observed; used for L2 locality only).
    // Give every XCD a contiguous range of logical blocks so the halo rows of neighbouring tiles hit its L2
    // (and stream-K hand-offs mostly stay inside one XCD).
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }

    // transform tables: always two stages in the loop (identity = scale 1, shift 0, no ReLU floor), so the
    // K loop carries no data-dependent or uniform branches and the scheduler can interleave it with the MFMAs
    for (int i = t; i < CIN; i += NT) {
        aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f;
        aff[2 * CIN + i] = p.stages >= 2 ? p.scale2[i] : 1.f; aff[3 * CIN + i] = p.stages >= 2 ? p.shift2[i] : 0.f;
    }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float lo2 = (p.stages >= 2 && p.relu2) ? 0.f : -INFINITY;
    __syncthreads();

    // per-thread staging assignment: row r0 + RP*i of the tile, 16-byte chunk c4 of the 32-wide K slice
    const int c4 = t & 7, r0 = t >> 3;
    // fragment read bases: lane l supplies row (l&31) and the k pair {r, 4+r} selected by (l>>5)
    const int frag_off = (lane & 31) * LDSS + (lane >> 5) * 4;
    // C/D layout of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31, rbase = 4 * (lane >> 5);

    long long u = 0, u_end = nsteps;
    if (SK) {
        const long long U = (long long)mtiles * ntiles * nsteps;
        u = U * lb / gridDim.x; u_end = U * (lb + 1) / gridDim.x;
    }

    while (u < u_end) {
        int mblock, nblock, k0 = 0, k1 = nsteps;
        if (SK) {
            const int tl = (int)(u / nsteps);
            k0 = (int)(u - (long long)tl * nsteps);
            k1 = (int)(u_end - u) < nsteps - k0 ? k0 + (int)(u_end - u) : nsteps;
            mblock = tl / ntiles; nblock = tl - mblock * ntiles;
        } else {
            mblock = lb; nblock = blockIdx.y;
        }
        u += k1 - k0;

        int iy0[AROWS], ix0[AROWS];
        bool rv[AROWS];
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int m = mblock * BM + r0 + RP * i;
            rv[i] = m < M;
            const int mm = rv[i] ? m : 0;
            const int oy = mm / p.OW, ox = mm - oy * p.OW;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
        }
        const float* wrow = p.wgt + (size_t)(nblock * BN + r0) * p.Kpad + c4 * 4;

        // Two register sets: while the MFMAs of step s run, the global loads of step s+2 are in flight (set L)
        // and the data of step s+1 (set S, loaded one step earlier) is transformed and written to the other
        // LDS buffer in chunks interleaved with the four MFMA groups, so that VALU / LDS-store work executes
        // in the shadow of the 64-cycle matrix instructions instead of in a phase of its own.
        float4 ra0[AROWS], ra1[AROWS];
        v4f rb0[BROWS], rb1[BROWS];
        bool va0[AROWS], va1[AROWS];
        int ci0 = 0, ci1 = 0;

// tap decomposition for K-step s_ and this thread's 16-byte chunk.  CIN is a power of two (checked on the
// host), so k -> (tap, ci) is a shift; tap -> (ky, kx) uses a 16.16 reciprocal of KW (exact for tap < 4096).
#define FAV_TAP_SETUP(X, s_)                                                                                \
        int ky##X, kx##X; bool tv##X;                                                                       \
        {                                                                                                   \
            int tap_;                                                                                       \
            if (p.cin_shift >= 5) {       /* K order (channel slice, tap, 32 ch): consecutive steps re-read  */ \
                const int cs_ = ((s_) * p.ntaps_magic) >> 16;   /* the same lines shifted by one tap (L1 reuse) */ \
                tap_ = (s_) - cs_ * ntaps;                                                                  \
                ci##X = cs_ * BK + c4 * 4;                                                                  \
            } else {                      /* K order (tap, ci): several taps per 32-wide slice               */ \
                const int kb_ = (s_) * BK + c4 * 4;                                                         \
                tap_ = kb_ >> p.cin_shift;                                                                  \
                ci##X = kb_ & (CIN - 1);                                                                    \
            }                                                                                               \
            ky##X = (tap_ * p.kw_magic) >> 16; kx##X = tap_ - ky##X * p.KW;                                 \
            tv##X = tap_ < ntaps;                                                                           \
        }
// global -> registers (set X), chunk q_ of 4: A row q_ and B row q_ of K-step s_ (raw values; the transform
// is applied when they are written to LDS)
#define FAV_LOAD_CHUNK(X, s_, q_)                                                                           \
        {                                                                                                   \
            if ((q_) < AROWS) {                                                                             \
                constexpr int i_ = (q_) < AROWS ? (q_) : 0;                                                 \
                const int iy_ = iy0[i_] + ky##X, ix_ = ix0[i_] + kx##X;                                     \
                va##X[i_] = rv[i_] & tv##X & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW) & (((iy_ | ix_) & p.stuff) == 0); \
                const int off_ = va##X[i_] ? ((iy_ >> p.ups) * p.IWp + (ix_ >> p.ups)) * CIN + ci##X : 0;   \
                ra##X[i_] = *reinterpret_cast<const float4*>(p.in + off_);   /* 32-bit element offset */     \
            }                                                                                               \
            if ((q_) < BROWS) {                                                                             \
                constexpr int j_ = (q_) < BROWS ? (q_) : 0;                                                 \
                if (BN >= RP || r0 < BN) rb##X[j_] = *reinterpret_cast<const v4f*>(wrow + (RP * j_) * p.Kpad + (s_) * BK); \
            }                                                                                               \
        }
#define FAV_LOAD_STEP(X, s_)                                                                                \
        { FAV_TAP_SETUP(X, s_); FAV_LOAD_CHUNK(X, s_, 0); FAV_LOAD_CHUNK(X, s_, 1); FAV_LOAD_CHUNK(X, s_, 2); FAV_LOAD_CHUNK(X, s_, 3); }
// registers (set X) -> LDS buffer buf_, chunk q_ of 4: A row q_ (pending transform of the producer: IN
// scale/shift [+ReLU], two stages; then zero for padding / out-of-range rows) and B row q_
#define FAV_STORE_CHUNK(X, buf_, q_)                                                                        \
        {                                                                                                   \
            if ((q_) < AROWS) {                                                                             \
                float4 v_ = ra##X[(q_) < AROWS ? (q_) : 0];                                                 \
                v_ = affine4_lo(v_, aff + ci##X, aff + CIN + ci##X, lo1);                                   \
                v_ = affine4_lo(v_, aff + 2 * CIN + ci##X, aff + 3 * CIN + ci##X, lo2);                     \
                const float m_ = va##X[(q_) < AROWS ? (q_) : 0] ? 1.f : 0.f;                                \
                v_.x *= m_; v_.y *= m_; v_.z *= m_; v_.w *= m_;                                             \
                *reinterpret_cast<float4*>(As + (buf_) * BM * LDSS + (r0 + RP * (q_)) * LDSS + c4 * 4) = v_; \
            }                                                                                               \
            if ((q_) < BROWS && (BN >= RP || r0 < BN))                                                      \
                *reinterpret_cast<v4f*>(Bs + (buf_) * BN * LDSS + (r0 + RP * (q_)) * LDSS + c4 * 4) = rb##X[(q_) < BROWS ? (q_) : 0]; \
        }
// one group of MFMAs: fragment step kk_ of the current LDS buffer
#define FAV_MFMA_GROUP(kk_)                                                                                 \
        {                                                                                                   \
            float4 af[TM], bf[TN];                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDSS + (kk_) * 8); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * LDSS + (kk_) * 8); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0); \
                }                                                                                           \
        }
// one K-step: LDS[cur] holds step s, set S holds step s+1 (raw), set L is free.
#define FAV_STEP(L, S, do_load_, do_store_)                                                                 \
        {                                                                                                   \
            const float* a_base = As + cur * BM * LDSS + (wm * TM * 32) * LDSS + frag_off;                  \
            const float* b_base = Bs + cur * BN * LDSS + (wn * TN * 32) * LDSS + frag_off;                  \
            FAV_TAP_SETUP(L, s + 2);                                                                        \
            FAV_MFMA_GROUP(0); if (do_load_) FAV_LOAD_CHUNK(L, s + 2, 0); if (do_store_) FAV_STORE_CHUNK(S, cur ^ 1, 0); \
            FAV_MFMA_GROUP(1); if (do_load_) FAV_LOAD_CHUNK(L, s + 2, 1); if (do_store_) FAV_STORE_CHUNK(S, cur ^ 1, 1); \
            FAV_MFMA_GROUP(2); if (do_load_) FAV_LOAD_CHUNK(L, s + 2, 2); if (do_store_) FAV_STORE_CHUNK(S, cur ^ 1, 2); \
            FAV_MFMA_GROUP(3); if (do_load_) FAV_LOAD_CHUNK(L, s + 2, 3); if (do_store_) FAV_STORE_CHUNK(S, cur ^ 1, 3); \
            __syncthreads();                                                                                \
            cur ^= 1; ++s;                                                                                  \
        }

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        int cur = 0, s = k0;
        FAV_LOAD_STEP(0, k0);
        FAV_STORE_CHUNK(0, 0, 0); FAV_STORE_CHUNK(0, 0, 1); FAV_STORE_CHUNK(0, 0, 2); FAV_STORE_CHUNK(0, 0, 3);
        if (k0 + 1 < k1) FAV_LOAD_STEP(1, k0 + 1);
        __syncthreads();

        while (s + 3 < k1) {          // steady state: no guards, two steps per trip (register sets swap roles)
            FAV_STEP(0, 1, true, true);
            FAV_STEP(1, 0, true, true);
        }
        if (s < k1) FAV_STEP(0, 1, s + 2 < k1, s + 1 < k1);
        if (s < k1) FAV_STEP(1, 0, s + 2 < k1, s + 1 < k1);
        if (s < k1) FAV_STEP(0, 1, s + 2 < k1, s + 1 < k1);

#undef FAV_LOAD_STEP
#undef FAV_LOAD_CHUNK
#undef FAV_TAP_SETUP
#undef FAV_STORE_CHUNK
#undef FAV_MFMA_GROUP
#undef FAV_STEP

        // ------------------------------------------------------------ stream-K hand-off
        constexpr int NV4 = TM * TN * 4;                     // float4 per thread in a partial tile
        if (SK && k0 > 0) {
            // contributor: dump the partial accumulators, publish (agent-scope release), next segment
            float4* slot = reinterpret_cast<float4*>(p.sk_ws) + (size_t)lb * NV4 * NT + t;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        slot[(size_t)((i * TN + j) * 4 + q) * NT] =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.sk_flags + lb, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            continue;
        }
        if (SK && k1 < nsteps) {
            // owner of a split tile: the remaining K ranges were computed by the following logical blocks at
            // the very start of their timelines; wait for each (one lane polls, relaxed), acquire, accumulate
            const long long U = (long long)mtiles * ntiles * nsteps;
            int covered = k1;
            for (int nb = lb + 1; covered < nsteps && nb < (int)gridDim.x; ++nb) {
                const long long nu0 = U * nb / gridDim.x, nu1 = U * (nb + 1) / gridDim.x;
                const int span = (int)((nu1 - nu0) < (long long)(nsteps - covered) ? (nu1 - nu0) : (nsteps - covered));
                if (t == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.sk_flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) { if (p.sk_err) __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // bounded, and reported to the host
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const float4* slot = reinterpret_cast<const float4*>(p.sk_ws) + (size_t)nb * NV4 * NT + t;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = slot[(size_t)((i * TN + j) * 4 + q) * NT];
                            acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                        }
                covered += span;
            }
        }

        // ------------------------------------------------------------ epilogue
        const int m_wave = mblock * BM + wm * TM * 32;
        const int n_wave = nblock * BN + wn * TN * 32;
        if (p.final_mode) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_wave + j * 32 + col;
                if (n < p.COUT && n < 3) {
                    const float bv = p.bias[n];
                    const float mean = n == 0 ? 103.939f : (n == 1 ? 116.779f : 123.68f);      // preprocess.lua:48 (BGR)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                            if (m < M) {
                                const float v = tanhf(acc[i][j][r] + bv) * p.tanh_mul;          // models_video.lua:135-136
                                if (p.out_raw) p.out_raw[(size_t)n * M + m] = v;
                                if (p.out_planar) p.out_planar[(size_t)(2 - n) * M + m] = (v + mean) / 255.f;  // preprocess.lua:66-71
                            }
                        }
                }
            }
        } else {
            float* red = smem;               // [WM][BN] (LDS is free again: the K loop ended on a barrier)
            float* mean_s = smem + WM * BN;  // [BN]
            const int cnt = min(BM, M - mblock * BM);
            float lsum[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_wave + j * 32 + col;
                const float bv = p.bias[n];
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                        const float v = acc[i][j][r] + bv;
                        acc[i][j][r] = v;
                        if (m < M) {
                            if (n < p.COUT) p.out[(size_t)m * p.COUT + n] = v;
                            sm += v;
                        }
                    }
                lsum[j] = sm;
            }
            if (p.partials != nullptr) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float sm = lsum[j] + __shfl_xor(lsum[j], 32);
                    if (lane < 32) red[wm * BN + (wn * TN + j) * 32 + lane] = sm;
                }
                __syncthreads();
                if (t < BN) {
                    float sm = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) sm += red[w * BN + t];
                    mean_s[t] = sm / (float)cnt;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float mu = mean_s[(wn * TN + j) * 32 + col];
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                            const float d = acc[i][j][r] - mu;
                            if (m < M) q = fmaf(d, d, q);
                        }
                    q += __shfl_xor(q, 32);
                    if (lane < 32) red[wm * BN + (wn * TN + j) * 32 + lane] = q;
                }
                __syncthreads();
                if (t < BN) {
                    float q = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) q += red[w * BN + t];
                    p.partials[(size_t)mblock * p.COUTp + nblock * BN + t] = make_float2(mean_s[t], q);
                }
            }
        }
        if (SK) __syncthreads();         // LDS scratch of the epilogue vs the next segment's staging
    }
}

constexpr int SK_GRID = 512;            // stream-K grid: 2 blocks on each of the 256 CUs, all co-resident

template <int BN, int WM, int WN, bool SK = false>
int launch_conv_t(const ConvArgs& a, hipStream_t st)
{
    const int M = a.OH * a.OW;
    const size_t lds = (size_t)(2 * (BM + BN) * LDSS + 4 * a.CIN) * sizeof(float);
    const int dv = cur_dev();
    static bool attr_done[MAX_DEVICES] = {};   // per instantiation and device
    if (!attr_done[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<BN, WM, WN, SK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dv] = true;
    }
    dim3 grid((M + BM - 1) / BM, a.COUTp / BN);
    if (SK) {
        // every stream-K block must be resident (owners wait for later blocks): size the grid from the occupancy
        // the runtime reports for this instantiation, capped at the 2 blocks per CU the hand-off buffers are sized for
        static int sk_per_cu[MAX_DEVICES] = {}, sk_cus[MAX_DEVICES] = {};
        if (!sk_per_cu[dv]) {
            int occ = 0; int prop_cus = 0;
            FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
            FAV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_mfma_kernel<BN, WM, WN, SK>, 64 * WM * WN, lds));
            if (occ < 1) { set_error("stream-K conv: kernel does not fit on a CU"); return FAV_EHIP; }
            sk_per_cu[dv] = occ >= 2 ? 2 : 1; sk_cus[dv] = prop_cus;
        }
        int sk_blocks = sk_per_cu[dv] * std::max(1, sk_cus[dv] - a.reserve_cus);       // leave the reserved CUs to the side queues
        if (sk_blocks > SK_GRID) sk_blocks = SK_GRID;
        grid = dim3(sk_blocks, 1);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<BN, WM, WN, SK>), grid, dim3(64 * WM * WN), lds, st, a);
    FAV_LAUNCH_CHECK("conv_mfma_kernel");
    return FAV_OK;
}

}  // namespace

size_t conv_streamk_workspace_bytes() { return (size_t)SK_GRID * BM * 128 * sizeof(float); }
int conv_streamk_grid() { return SK_GRID; }

// ------------------------------------------------------------------------------------------------
// First layer (c9s1-32: 7(+1) -> 32 channels, 9x9, stride 1): LDS-resident halo + LDS-resident weights.
// With 8 input channels a filter tap is exactly one k=8 MFMA quad, and the generic kernel would re-gather
// the operand 81 times from global memory.  Here a persistent block (8 waves, one per CU) keeps all
// 32 x 648 weights in LDS, stages the (8+8) x (32+8) pixel halo of an 8x32 output tile once (as two planes
// of 4 channels so that the 16-byte fragment reads are conflict-free), and runs the 81 taps straight out of
// LDS with immediate-offset ds_read_b128: no global loads, LDS stores or barriers inside the tap loop.  The
// next tile's halo is prefetched into registers during the tap loop.  Epilogue: bias, NHWC store, per-tile
// InstanceNorm partials (mean, M2, count).
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int C8_TH = 8, C8_TW = 32;      // output tile: 8 rows x 32 columns, one row of 32 pixels per wave

struct C8Args {
    const float* in; const float* wgt; const float* bias;
    float* out; float2* partials; int* counts;
    int IH, IW, IWp, COUT, pad, OH, OW, Kpad, tiles_x, tiles_y;
};

template <int KS>
__global__ __launch_bounds__(512, 2) void conv_c8_kernel(const C8Args p)
{
    constexpr int HW = C8_TW + KS - 1;            // halo width (40)
    constexpr int HP = (C8_TH + KS - 1) * HW;     // halo pixels (16 x 40 = 640)
    constexpr int NTAP = KS * KS;
    constexpr int WS = NTAP * 8 + 4;              // weight row stride (floats): odd multiple of 16 B
    constexpr int NH = (HP * 2 + 511) / 512;      // float4 per thread per halo
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                             // [32][WS]
    float* Hs = Ws + 32 * WS;                     // [2 buffers][2 planes][HP][4]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // weights -> LDS once per block ([n][tap*8 + ci], rows >= COUT are zero in the repacked tensor)
    for (int e = t; e < 32 * NTAP * 2; e += 512) {
        const int n = e / (NTAP * 2), c = e - n * (NTAP * 2);
        *reinterpret_cast<v4f*>(Ws + n * WS + c * 4) = *reinterpret_cast<const v4f*>(p.wgt + (size_t)n * p.Kpad + c * 4);
    }

    const int ntiles = p.tiles_x * p.tiles_y;
    float4 hreg[NH];
#define C8_LOAD_HALO(tile_)                                                                         \
    {                                                                                               \
        const int ty_ = (tile_) / p.tiles_x, tx_ = (tile_) - ty_ * p.tiles_x;                       \
        _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                            \
            const int e_ = t + 512 * i;                                                             \
            const int pix_ = e_ >> 1, hy_ = pix_ / HW, hx_ = pix_ - hy_ * HW;                       \
            const int iy_ = ty_ * C8_TH - p.pad + hy_, ix_ = tx_ * C8_TW - p.pad + hx_;             \
            const bool v_ = (e_ < HP * 2) & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW); \
            const int off_ = v_ ? (iy_ * p.IWp + ix_) * 8 + (e_ & 1) * 4 : 0;                       \
            const float4 x_ = *reinterpret_cast<const float4*>(p.in + off_);                        \
            hreg[i] = v_ ? x_ : make_float4(0.f, 0.f, 0.f, 0.f);                                    \
        }                                                                                           \
    }
#define C8_STORE_HALO(buf_)                                                                         \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                            \
            const int e_ = t + 512 * i;                                                             \
            if (e_ < HP * 2) *reinterpret_cast<float4*>(Hs + (((buf_) * 2 + (e_ & 1)) * HP + (e_ >> 1)) * 4) = hreg[i]; \
        }                                                                                           \
    }

    int tile = blockIdx.x;
    if (tile < ntiles) C8_LOAD_HALO(tile);
    C8_STORE_HALO(0);
    __syncthreads();

    const int m = lane & 31, half = lane >> 5;
    const int py = wave, px = m;                  // fragment rows = 32 consecutive halo pixels: conflict-free b128 reads
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float* red = Hs + 4 * HP * 4;                 // [8 waves][32] + [32] scratch after the halo buffers
    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) C8_LOAD_HALO(nxt);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* a_base = Hs + ((cur * 2 + half) * HP + py * HW + px) * 4;
        const float* b_base = Ws + m * WS + half * 4;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const float4 af = *reinterpret_cast<const float4*>(a_base + (ky * HW + kx) * 4);
            const float4 bf = *reinterpret_cast<const float4*>(b_base + tap * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
        if (nxt < ntiles) C8_STORE_HALO(cur ^ 1);

        // epilogue: rows of the MFMA tile are pixels (2 tile rows x 16 columns of this wave), columns are channels
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const float bv = p.bias[col];
        float sm = 0.f;
        int nvalid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mi = (r & 3) + 8 * (r >> 2) + rbase;
            const int oy = ty * C8_TH + wave, ox = tx * C8_TW + mi;
            const float v = acc[r] + bv;
            acc[r] = v;
            if (oy < p.OH && ox < p.OW) {
                if (col < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + col] = v;
                sm += v; ++nvalid;
            }
        }
        if (p.partials != nullptr) {
            float2* st = reinterpret_cast<float2*>(red);          // [8 waves][32]
            int* wn = reinterpret_cast<int*>(red + 8 * 64);         // [8]
            const int oyw = ty * C8_TH + wave;
            const int nw = oyw < p.OH ? min(C8_TW, p.OW - tx * C8_TW) : 0;      // valid pixels of this wave's row
            sm += __shfl_xor(sm, 32);
            const float mu = nw ? sm / (float)nw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mi = (r & 3) + 8 * (r >> 2) + rbase;
                const float d = acc[r] - mu;
                if (oyw < p.OH && tx * C8_TW + mi < p.OW) q = fmaf(d, d, q);
            }
            q += __shfl_xor(q, 32);
            if (lane < 32) st[wave * 32 + lane] = make_float2(mu, q);
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < 32) {
                int n;
                p.partials[(size_t)tile * 32 + t] = merge_wave_stats(st, wn, 8, 32, t, &n);
                if (t == 0) p.counts[tile] = n;
            }
        }
        (void)nvalid;
        __syncthreads();            // next halo buffer written by every thread; red scratch free again
        cur ^= 1;
    }
#undef C8_LOAD_HALO
#undef C8_STORE_HALO
}

}  // namespace

bool conv_c8_eligible(int cin_pitch, int coutp, int k, int stride, int stages, int ups)
{
    return cin_pitch == 8 && coutp == 32 && k == 9 && stride == 1 && stages == 0 && ups == 0;
}
int conv_c8_tiles(int OH, int OW) { return ((OH + C8_TH - 1) / C8_TH) * ((OW + C8_TW - 1) / C8_TW); }

int launch_conv_c8(const ConvLaunch& c, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv_c8_eligible(c.CIN, c.COUTp, c.KH, c.stride, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode,
                "first-layer conv: not eligible");
    FAV_REQUIRE(c.Kpad >= 81 * 8 && (long long)c.IH * c.IWp * 8 < (1ll << 31), "first-layer conv: bad shape");
    C8Args a;
    a.in = c.in; a.wgt = c.wgt; a.bias = c.bias; a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.COUT = c.COUT; a.pad = c.pad; a.OH = c.OH; a.OW = c.OW; a.Kpad = c.Kpad;
    a.tiles_x = (c.OW + C8_TW - 1) / C8_TW; a.tiles_y = (c.OH + C8_TH - 1) / C8_TH;
    constexpr int HPc = (C8_TH + 8) * (C8_TW + 8), WSc = 81 * 8 + 4;
    const size_t lds = (size_t)(32 * WSc + 4 * HPc * 4 + 8 * 32 + 32) * sizeof(float);
    const int dv = cur_dev();
    static int nblocks[MAX_DEVICES] = {};
    if (!nblocks[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c8_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        nblocks[dv] = prop_cus;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int gridc8 = std::max(1, nblocks[dv] - c.reserve_cus);
    hipLaunchKernelGGL((conv_c8_kernel<9>), dim3(tiles < gridc8 ? tiles : gridc8), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_c8_kernel");
    return FAV_OK;
}

// ------------------------------------------------------------------------------------------------
// First layer with DENSE K (round 2).  conv_c8_kernel feeds one filter tap = 8 channels = 4 MFMAs, of which the 8th channel
// (7 real ones: content BGR, prior BGR, mask; 3 for an image model) is a zero: 12.5 % (62.5 %) of the matrix work multiplies
// zeros.  A 32x32x2 MFMA takes ONE k per half-wave, and any two k may share an instruction, so the taps of one channel are
// PAIRED such that the second half-wave's operand sits at a constant offset from the first's:
//     (ky, 2q) + (ky, 2q+1)   -> +1 halo pixel      (36 pairs per channel)
//     (2p, 8)  + (2p+1, 8)    -> +1 halo row        ( 4 pairs)
//     (8, 8)   + nothing                            ( 1, zero weight in the second half)
// = 41 MFMAs per channel, 287 (123) instead of 324 per tile.  The halo lives in LDS as one PLANE per channel (lanes = 32
// consecutive pixels: conflict-free ds_read_b32), the weights as [pair][half][32 output channels]; every operand address is a
// per-lane base (pixel + the half-wave's +1 pixel / +1 row / +32 floats) plus an immediate.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int C8D_PAIRS = 41;                         // MFMAs per real input channel (9x9 taps)

template <int CR>
__global__ __launch_bounds__(512, 2) void conv_c8d_kernel(const C8Args p)
{
    constexpr int KS = 9;
    constexpr int HW = C8_TW + KS - 1;            // halo width (40)
    constexpr int HP = (C8_TH + KS - 1) * HW;     // halo pixels per plane (16 x 40 = 640)
    constexpr int NJ = CR * C8D_PAIRS;            // MFMAs per tile
    constexpr int NH = (HP + 511) / 512;          // halo pixels per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                             // [NJ][2][32]
    float* Hs = Ws + NJ * 64;                     // [2 buffers][CR planes][HP]
    float* red = Hs + 2 * CR * HP;                // [8 waves][32] float2 + [8] int
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (int e = t; e < NJ * 16; e += 512) *reinterpret_cast<v4f*>(Ws + e * 4) = *reinterpret_cast<const v4f*>(p.wgt + e * 4);

    const int ntiles = p.tiles_x * p.tiles_y;
    float4 hlo[NH], hhi[NH];
#define C8D_LOAD_HALO(tile_)                                                                        \
    {                                                                                               \
        const int ty_ = (tile_) / p.tiles_x, tx_ = (tile_) - ty_ * p.tiles_x;                       \
        _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                            \
            const int pix_ = t + 512 * i, hy_ = pix_ / HW, hx_ = pix_ - hy_ * HW;                   \
            const int iy_ = ty_ * C8_TH - p.pad + hy_, ix_ = tx_ * C8_TW - p.pad + hx_;             \
            const bool v_ = (pix_ < HP) & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW); \
            const int off_ = v_ ? (iy_ * p.IWp + ix_) * 8 : 0;                                      \
            const float4 a_ = *reinterpret_cast<const float4*>(p.in + off_);                        \
            const float4 b_ = CR > 4 ? *reinterpret_cast<const float4*>(p.in + off_ + 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
            hlo[i] = v_ ? a_ : make_float4(0.f, 0.f, 0.f, 0.f);                                     \
            hhi[i] = v_ ? b_ : make_float4(0.f, 0.f, 0.f, 0.f);                                     \
        }                                                                                           \
    }
#define C8D_STORE_HALO(buf_)                                                                        \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                            \
            const int pix_ = t + 512 * i;                                                           \
            if (pix_ < HP) {                                                                        \
                float* d_ = Hs + (buf_) * CR * HP + pix_;                                           \
                const float c_[8] = {hlo[i].x, hlo[i].y, hlo[i].z, hlo[i].w, hhi[i].x, hhi[i].y, hhi[i].z, hhi[i].w}; \
                _Pragma("unroll") for (int c = 0; c < CR; ++c) d_[c * HP] = c_[c];                  \
            }                                                                                       \
        }                                                                                           \
    }

    int tile = blockIdx.x;
    if (tile < ntiles) C8D_LOAD_HALO(tile);
    C8D_STORE_HALO(0);
    __syncthreads();

    const int m = lane & 31, half = lane >> 5;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    // per-lane bases: pixel (wave row, m); second half-wave one pixel / one row further; weights of this lane's output channel
    const float* const a_px = Hs + wave * HW + m + half;
    const float* const a_row = Hs + (wave + half) * HW + m;
    const float* const a_one = Hs + wave * HW + m;     // the unpaired tap: both half-waves read the SAME (valid) pixel; the second one's weight is 0
    const float* const b_lo = Ws + half * 32 + m;
    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) C8D_LOAD_HALO(nxt);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* apx = a_px + cur * CR * HP;
        const float* arw = a_row + cur * CR * HP;
        const float* aon = a_one + cur * CR * HP;
#pragma unroll
        for (int c = 0; c < CR; ++c) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    constexpr int dummy = 0; (void)dummy;
                    const int j = c * C8D_PAIRS + ky * 4 + q;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(apx[c * HP + ky * HW + 2 * q], b_lo[j * 64], acc, 0, 0, 0);
                }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int j = c * C8D_PAIRS + 36 + pp;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arw[c * HP + 2 * pp * HW + 8], b_lo[j * 64], acc, 0, 0, 0);
            }
            {
                // tap (8, 8) alone: the second half-wave's weight is zero -- but 0 x NaN is NaN, so its operand must still be a
                // value this kernel wrote (one pixel further would leave the plane and, for the last plane of the first tile,
                // read LDS that nobody initialised: stale NaN patterns there zeroed a whole frame through the IN statistics)
                const int j = c * C8D_PAIRS + 40;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aon[c * HP + 8 * HW + 8], b_lo[j * 64], acc, 0, 0, 0);
            }
        }
        if (nxt < ntiles) C8D_STORE_HALO(cur ^ 1);

        // epilogue: as conv_c8_kernel (MFMA rows = 32 pixels of this wave's tile row, columns = channels)
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const float bv = p.bias[col];
        float sm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mi = (r & 3) + 8 * (r >> 2) + rbase;
            const int oy = ty * C8_TH + wave, ox = tx * C8_TW + mi;
            const float v = acc[r] + bv;
            acc[r] = v;
            if (oy < p.OH && ox < p.OW) {
                if (col < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + col] = v;
                sm += v;
            }
        }
        if (p.partials != nullptr) {
            float2* st = reinterpret_cast<float2*>(red);          // [8 waves][32]
            int* wn = reinterpret_cast<int*>(red + 8 * 64);         // [8]
            const int oyw = ty * C8_TH + wave;
            const int nw = oyw < p.OH ? min(C8_TW, p.OW - tx * C8_TW) : 0;      // valid pixels of this wave's row
            sm += __shfl_xor(sm, 32);
            const float mu = nw ? sm / (float)nw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mi = (r & 3) + 8 * (r >> 2) + rbase;
                const float d = acc[r] - mu;
                if (oyw < p.OH && tx * C8_TW + mi < p.OW) q = fmaf(d, d, q);
            }
            q += __shfl_xor(q, 32);
            if (lane < 32) st[wave * 32 + lane] = make_float2(mu, q);
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < 32) {
                int n;
                p.partials[(size_t)tile * 32 + t] = merge_wave_stats(st, wn, 8, 32, t, &n);
                if (t == 0) p.counts[tile] = n;
            }
        }
        __syncthreads();            // next halo buffer written by every thread; red scratch free again
        cur ^= 1;
    }
#undef C8D_LOAD_HALO
#undef C8D_STORE_HALO
}

}  // namespace

bool conv_c8d_eligible(int cin_pitch, int cin_real, int coutp, int k, int stride, int stages, int ups)
{
    return conv_c8_eligible(cin_pitch, coutp, k, stride, stages, ups) && (cin_real == 7 || cin_real == 3);
}

// weights [cout][cin][9][9] -> [pair j][half][32]: the pairing of conv_c8d_kernel
void conv_c8d_pack(const float* w, int cin, int cout, std::vector<float>& out)
{
    out.assign((size_t)cin * C8D_PAIRS * 64, 0.f);
    auto W = [&](int n, int c, int ky, int kx) { return w[(((size_t)n * cin + c) * 9 + ky) * 9 + kx]; };
    for (int c = 0; c < cin; ++c)
        for (int n = 0; n < cout && n < 32; ++n) {
            float* o = out.data() + (size_t)c * C8D_PAIRS * 64 + n;
            for (int ky = 0; ky < 9; ++ky)
                for (int q = 0; q < 4; ++q) { o[(ky * 4 + q) * 64] = W(n, c, ky, 2 * q); o[(ky * 4 + q) * 64 + 32] = W(n, c, ky, 2 * q + 1); }
            for (int pp = 0; pp < 4; ++pp) { o[(36 + pp) * 64] = W(n, c, 2 * pp, 8); o[(36 + pp) * 64 + 32] = W(n, c, 2 * pp + 1, 8); }
            o[40 * 64] = W(n, c, 8, 8);
        }
}

template <int CR>
static int launch_c8d_t(const C8Args& a, int reserve_cus, hipStream_t st)
{
    constexpr int HPc = (C8_TH + 8) * (C8_TW + 8);
    const size_t lds = (size_t)(CR * C8D_PAIRS * 64 + 2 * CR * HPc + 8 * 64 + 8) * sizeof(float);
    const int dv = cur_dev();
    static int nblocks[MAX_DEVICES] = {};
    if (!nblocks[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c8d_kernel<CR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        nblocks[dv] = prop_cus;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int grid = std::max(1, nblocks[dv] - reserve_cus);
    hipLaunchKernelGGL((conv_c8d_kernel<CR>), dim3(tiles < grid ? tiles : grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_c8d_kernel");
    return FAV_OK;
}

int launch_conv_c8d(const ConvLaunch& c, int cin_real, const float* wc8d, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv_c8d_eligible(c.CIN, cin_real, c.COUTp, c.KH, c.stride, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && wc8d,
                "first-layer conv (dense K): not eligible");
    FAV_REQUIRE((long long)c.IH * c.IWp * 8 < (1ll << 31), "first-layer conv: bad shape");
    C8Args a;
    a.in = c.in; a.wgt = wc8d; a.bias = c.bias; a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.COUT = c.COUT; a.pad = c.pad; a.OH = c.OH; a.OW = c.OW; a.Kpad = c.Kpad;
    a.tiles_x = (c.OW + C8_TW - 1) / C8_TW; a.tiles_y = (c.OH + C8_TH - 1) / C8_TH;
    return cin_real == 7 ? launch_c8d_t<7>(a, c.reserve_cus, st) : launch_c8d_t<3>(a, c.reserve_cus, st);
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 layers (the ten 128->128 residual convolutions and c3s1-64: 71 % of the network's FLOPs):
// halo-resident implicit GEMM.  The generic kernel re-gathers (and re-transforms) its activation operand
// for every tap; measured, that global gather costs ~20 % of the kernel.  Here a block (8 waves, one per
// CU, stream-K over all (tile, K-step) units) owns an 8 x 32 pixel output tile; for each 32-channel slice
// the (8+2) x (32+2) pixel halo is gathered ONCE, transformed (producer's IN/ReLU stages, x2 nearest
// upsample, zero padding) and kept in LDS, and the 9 taps read their A fragments straight from it with
// conflict-free ds_read_b128 (a wave = one output row of 32 pixels, so the fragment rows are 32 consecutive
// halo pixels).  Only the weight slice (BN x 32 per step) streams through LDS.  Global->LDS traffic per
// MFMA drops 3x.  K order = (channel slice, tap, 32 channels): the same repacked weights as the generic
// kernel.  Epilogue as the generic kernel (bias, NHWC store, per-tile IN partials with explicit counts).
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int H3_TH = 8, H3_TW = 32;   // output tile: 8 rows (one per wave) x 32 pixels

struct H3Args {
    const float* in; const float* wgt; const float* bias;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    float* out; float2* partials; int* counts;
    float* sk_ws; unsigned* sk_flags; unsigned sk_epoch; unsigned* sk_err;
    int IH, IW, IWp, ups, CIN, COUT, COUTp, pad, OH, OW, Kpad, tiles_x, tiles_y;
    int nb;                  // number of 16 x 16 edge tiles (fp32 kernel; see conv3_halo_tiles)
    int stages, relu1, relu2;
    const unsigned short* wgt16;   // bf16 copy of the weights (fast mode) or null
    long long* dbg;          // optional in-kernel timeline (FAV_H3_DBG), 24 slots per block
};

// fp32 MFMA and the vector ALU do not overlap on a SIMD (measured: scripts/mfma_mix.hip -- every VALU instruction in the
// loop costs its issue cycles in matrix throughput), so the K loop is built to need almost none:
//   * the nine taps of a channel slice are unrolled: tap offsets, the weight ring slot (tap % 3) and the halo piece index
//     are compile-time constants, i.e. immediate offsets on per-thread base registers that are set once per tile/slice
//   * global addresses are scalar base (advanced by the scalar ALU) + a per-thread 32-bit offset fixed for the tile
//   * what is left per step: the IN/ReLU transform of one 16-byte halo piece (6 of 9 steps)
// Software pipeline of one K step (32 channels of one tap; 4 fragment groups of 8 channels):
//   start  : weights of step s+1 (in registers since step s-1) -> LDS ring slot (s+1)%3; issue the global load of step
//            s+2's weights and of one sixth of the NEXT channel slice's halo
//   groups : the A/B fragments of group g+1 are read from LDS into the other register set while group g's 16 MFMAs issue;
//            the last group prefetches group 0 of step s+1, so no LDS latency is exposed in the steady state
//   barrier: one per step, between groups 1 and 2 -- it publishes ring slot (s+1)%3 half a step before its first read
//            and is never followed by a dependent LDS read (three slots make the write-after-read side safe)
//   end    : the halo piece, transformed, -> the other halo buffer
template <int BN, bool S2>
__global__ __launch_bounds__(512, 2) void conv3_halo_kernel(const H3Args p)
{
    constexpr int NT = 512;
    constexpr int HWD = H3_TW + 2, HP = (H3_TH + 2) * HWD;        // 34, 340 halo pixels
    constexpr int TN = BN / 32;
    constexpr int NHV = (HP * 8 + NT - 1) / NT;                   // 16-byte halo pieces per thread per slice (6)
    constexpr int ALIAS = NT * NHV - HP * 8;                       // units past the end alias earlier ones (same data, same slot)
    constexpr int BROWS = BN / 64;                                 // weight rows per thread per step
    static_assert(NHV == 6, "two halo pieces per tap row");
    static_assert(ALIAS % 8 == 0 && ALIAS <= NT, "halo aliasing");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                           // [2][HP][LDSS]
    float* Bs = Hs + 2 * HP * LDSS;             // [3][BN][LDSS]
    float* aff = Bs + 3 * BN * LDSS;            // [4][CIN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int CIN = p.CIN;
    const int nchunks = CIN >> 5, nsteps = nchunks * 9;
    // Tiles.  A: 8 rows x 32 columns, wave = one row (tiles_x columns of them, tiles_y rows).  B (p.nb > 0): the ragged right
    // edge -- fewer than 17 columns wide -- is cut into 16 x 16 tiles instead, wave = TWO rows of 16 pixels: half as many edge
    // tiles, each fully used in x.  The 18 x 18 halo of a B tile (324 pixels, pitch 18) fits the same buffers.
    const int na = p.tiles_x * p.tiles_y, ntiles = na + p.nb;

    int dbi = 0;
#define DBG_T() { if (p.dbg && t == 0 && dbi < 22) p.dbg[blockIdx.x * 24 + dbi++] = wall_clock64(); }
    DBG_T();
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    for (int i = t; i < CIN; i += NT) {
        aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f;
        aff[2 * CIN + i] = p.stages >= 2 ? p.scale2[i] : 1.f; aff[3 * CIN + i] = p.stages >= 2 ? p.shift2[i] : 0.f;
    }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float lo2 = (p.stages >= 2 && p.relu2) ? 0.f : -INFINITY;
    __syncthreads();

    const int c4 = t & 7, r0 = t >> 3;                      // staging: weight row r0 (+64) / halo pixel r0 (+64 i), 16-byte chunk c4
    const int frag_k = (lane >> 5) * 4;                     // k pair {r, 4+r} by half-wave
    const int m = lane & 31;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    // per-thread bases; everything else in the K loop is an immediate or a scalar
    const unsigned wofs = (unsigned)(r0 * p.Kpad + c4 * 4) * 4u;            // byte offset of this thread's weight chunk in a step
    const unsigned wrow64 = (unsigned)(64 * p.Kpad) * 4u;
    float* const bst = Bs + r0 * LDSS + c4 * 4;                             // weight staging slot
    float* const hst = Hs + r0 * LDSS + c4 * 4;                             // halo staging slot of piece 0, buffer 0
    constexpr int HWB = 18, HPB = HWB * HWB, ALIASB = NT * NHV - HPB * 8;    // B tiles: 18 x 18 halo
    static_assert(ALIASB % 8 == 0 && ALIASB <= NT, "halo aliasing (B tiles)");
    const int hst_lastA = (t + NT * (NHV - 1) >= HP * 8) ? (NT * (NHV - 1) - ALIAS) / 8 * LDSS : 64 * (NHV - 1) * LDSS;
    const int hst_lastB = (t + NT * (NHV - 1) >= HPB * 8) ? (NT * (NHV - 1) - ALIASB) / 8 * LDSS : 64 * (NHV - 1) * LDSS;
    const float* const afrA = Hs + (wave * HWD + m) * LDSS + frag_k;        // A fragments: tap (0,0), buffer 0
    // B tiles: lanes 0-15 = row 2w, lanes 16-31 = row 2w+1 with the columns rotated by 14 -- the 16 pixels a ds_read_b128 lane
    // group touches must differ mod 16 (row stride 36 floats), and the second row starts 18 pixels after the first
    const int colB = m < 16 ? m : ((m + 14) & 15);
    const float* const afrB = Hs + ((2 * wave + (m >> 4)) * HWB + colB) * LDSS + frag_k;
    const float* const bfr = Bs + m * LDSS + frag_k;                        // B fragments: ring slot 0
    const float* const affr = aff + c4 * 4;

    // stream-K work unit: one tap row (3 K steps) of one channel slice of one tile.  Inside a unit kx, the weight ring
    // slot (= kx) and the position of the halo pieces are compile-time constants; ky and the slice are scalars.
    const int nunits = nchunks * 3;
    const int U = ntiles * nunits;
    int u = (int)((long long)U * lb / gridDim.x);
    const int u_end = (int)((long long)U * (lb + 1) / gridDim.x);

    while (u < u_end) {
        const int tile = u / nunits;
        const int k0 = u - tile * nunits;
        const int k1 = (u_end - u) < nunits - k0 ? k0 + (u_end - u) : nunits;
        u += k1 - k0;
        const bool tb = tile >= na;                                   // B tile (uniform)
        const int ty = tb ? tile - na : tile / p.tiles_x, tx = tb ? p.tiles_x : tile - ty * p.tiles_x;
        const int oy0 = ty * (tb ? 16 : H3_TH), ox0 = tx * H3_TW;
        const int hwd = tb ? HWB : HWD;
        const int hst_last = tb ? hst_lastB : hst_lastA;
        const float* const afr = tb ? afrB : afrA;
        DBG_T();   /* work item start */

        // halo piece i: unit e = t + 512*i -> halo pixel e>>3, channel chunk e&7; per tile: byte offset (chunk 0 if outside) and mask
        int hoff[NHV]; float hmask[NHV];
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            int e = t + NT * i; e -= e >= (tb ? HPB : HP) * 8 ? (tb ? ALIASB : ALIAS) : 0;
            const int pix = e >> 3, hy = tb ? (pix * 3641) >> 16 : (pix * 1928) >> 16, hx = pix - hy * hwd;      // pix / 18, pix / 34
            const int iy = oy0 - p.pad + hy, ix = ox0 - p.pad + hx;
            const bool v = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
            hoff[i] = ((v ? ((iy >> p.ups) * p.IWp + (ix >> p.ups)) * CIN : 0) + c4 * 4) * 4;
            hmask[i] = v ? 1.f : 0.f;
        }
        const int c_first = (k0 * 21846) >> 16, ky0 = k0 - c_first * 3;      // k / 3
        const int c_last = ((k1 - 1) * 21846) >> 16;

        float4 hr; float hm; v4f rb[BROWS];
        v4f sc1, sh1, sc2, sh2;             // IN/ReLU stages of the slice being staged, this thread's 4 channels
#define H3_AFF(chunk_)                                                                              \
        { sc1 = *reinterpret_cast<const v4f*>(affr + (chunk_) * 32); sh1 = *reinterpret_cast<const v4f*>(affr + CIN + (chunk_) * 32); \
          if (S2) { sc2 = *reinterpret_cast<const v4f*>(affr + 2 * CIN + (chunk_) * 32); sh2 = *reinterpret_cast<const v4f*>(affr + 3 * CIN + (chunk_) * 32); } }
#define H3_XFORM(v_, m_)                                                                            \
        { v_.x = fmaxf(fmaf(v_.x, sc1.x, sh1.x), lo1); v_.y = fmaxf(fmaf(v_.y, sc1.y, sh1.y), lo1);  \
          v_.z = fmaxf(fmaf(v_.z, sc1.z, sh1.z), lo1); v_.w = fmaxf(fmaf(v_.w, sc1.w, sh1.w), lo1);  \
          if (S2) { v_.x = fmaxf(fmaf(v_.x, sc2.x, sh2.x), lo2); v_.y = fmaxf(fmaf(v_.y, sc2.y, sh2.y), lo2); \
                    v_.z = fmaxf(fmaf(v_.z, sc2.z, sh2.z), lo2); v_.w = fmaxf(fmaf(v_.w, sc2.w, sh2.w), lo2); } \
          v_.x *= m_; v_.y *= m_; v_.z *= m_; v_.w *= m_; }
#define H3_HLDS(i_) ((i_) == NHV - 1 ? hst_last : 64 * (i_) * LDSS)
#define H3_LOAD_B(src_)                                                                             \
        { _Pragma("unroll") for (int j = 0; j < BROWS; ++j) rb[j] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(src_) + (wofs + j * wrow64)); }
#define H3_STORE_B(slot_)                                                                           \
        { _Pragma("unroll") for (int j = 0; j < BROWS; ++j) *reinterpret_cast<v4f*>(bst + ((slot_) * BN + 64 * j) * LDSS) = rb[j]; }

        {
            // prologue: this slice's whole halo -> buffer 0; the pieces of the next slice that the skipped tap rows would
            // have staged -> buffer 1; all loads in flight before the first store
            const char* in0 = reinterpret_cast<const char*>(p.in + c_first * 32);
            const char* in1 = reinterpret_cast<const char*>(p.in + min(c_first + 1, c_last) * 32);
            float4 q0[NHV], q1[NHV];
#pragma unroll
            for (int i = 0; i < NHV; ++i) q0[i] = *reinterpret_cast<const float4*>(in0 + hoff[i]);
            H3_LOAD_B(p.wgt + k0 * 3 * BK);
#pragma unroll
            for (int i = 0; i < NHV; ++i) if (i < 2 * ky0) q1[i] = *reinterpret_cast<const float4*>(in1 + hoff[i]);
            H3_AFF(c_first);
#pragma unroll
            for (int i = 0; i < NHV; ++i) { H3_XFORM(q0[i], hmask[i]); *reinterpret_cast<float4*>(hst + H3_HLDS(i)) = q0[i]; }
            H3_STORE_B(0);
            H3_LOAD_B(p.wgt + min(k0 * 3 + 1, nsteps - 1) * BK);
            H3_AFF(min(c_first + 1, c_last));
#pragma unroll
            for (int i = 0; i < NHV; ++i) if (i < 2 * ky0) { H3_XFORM(q1[i], hmask[i]); *reinterpret_cast<float4*>(hst + HP * LDSS + H3_HLDS(i)) = q1[i]; }
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __syncthreads();

        v4f fa[2], fb[2][TN];
#define H3_FRAG(set_, ap_, bp_)                                                                     \
        { fa[set_] = *reinterpret_cast<const v4f*>(ap_);                                            \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[set_][j] = *reinterpret_cast<const v4f*>((bp_) + j * 32 * LDSS); }
#define H3_MFMA(set_)                                                                               \
        { _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                          \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].x, fb[set_][j].x, acc[j], 0, 0, 0); \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].y, fb[set_][j].y, acc[j], 0, 0, 0); \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].z, fb[set_][j].z, acc[j], 0, 0, 0); \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].w, fb[set_][j].w, acc[j], 0, 0, 0); } }
#define H3_GROUP(nds_)                                                                              \
        { __builtin_amdgcn_sched_group_barrier(0x100, nds_, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4 * TN, 0); }
        // one K step, kx = KX (compile time).  a_cu = A fragments of this tap row, a_nu = of the next unit's; the halo
        // piece staged in this step (KX < 2) is piece 2*ky + KX of the next slice
#define H3_STEP(KX)                                                                                 \
        {   constexpr int NB = ((KX) + 1) % 3;                                                      \
            const float* an_ = (KX) == 2 ? a_nu : a_cu + ((KX) + 1) * LDSS;                         \
            H3_FRAG(1, a_cu + (KX) * LDSS + 8, bfr + (KX) * BN * LDSS + 8);                         \
            H3_STORE_B(NB);                                                                         \
            H3_LOAD_B(p.wgt + min(sg + (KX) + 2, nsteps - 1) * BK);                                 \
            if ((KX) < 2) { hr = *reinterpret_cast<const float4*>(in_n + ((KX) == 0 ? ho0 : ho1)); hm = (KX) == 0 ? hm0 : hm1; } \
            H3_MFMA(0); H3_GROUP(1 + TN);                                                           \
            H3_FRAG(0, a_cu + (KX) * LDSS + 16, bfr + (KX) * BN * LDSS + 16); H3_MFMA(1); H3_GROUP(1 + TN); \
            __syncthreads();                                                                        \
            H3_FRAG(1, a_cu + (KX) * LDSS + 24, bfr + (KX) * BN * LDSS + 24); H3_MFMA(0); H3_GROUP(1 + TN); \
            H3_FRAG(0, an_, bfr + NB * BN * LDSS);                                                  \
            if ((KX) < 2) { H3_XFORM(hr, hm); *reinterpret_cast<float4*>(h_nx + ((KX) == 0 ? hl0 : hl1)) = hr; } \
            H3_MFMA(1); H3_GROUP(1 + TN);                                                           \
        }

        H3_FRAG(0, afr + ky0 * hwd * LDSS, bfr);           // fragments of the first step's group 0
        DBG_T();   /* loop start */
        const long long ck0 = p.dbg ? clock64() : 0, wk0 = p.dbg ? wall_clock64() : 0;
        int c = c_first, ky = ky0, par = 0;
        for (int uu = k0; uu < k1; ++uu) {
            const float* a_cu = afr + (par * HP + ky * hwd) * LDSS;
            const float* a_nu = ky == 2 ? afr + (par ^ 1) * (HP * LDSS) : a_cu + hwd * LDSS;
            float* h_nx = hst + (par ^ 1) * (HP * LDSS);
            const int cn = min(c + 1, c_last);                              // no next slice: the pieces land in the unused buffer
            const char* in_n = reinterpret_cast<const char*>(p.in + cn * 32);
            const int sg = uu * 3;
            // the two halo pieces of this unit: 2*ky and 2*ky + 1 (uniform selects)
            const int ho0 = ky == 0 ? hoff[0] : (ky == 1 ? hoff[2] : hoff[4]), ho1 = ky == 0 ? hoff[1] : (ky == 1 ? hoff[3] : hoff[5]);
            const float hm0 = ky == 0 ? hmask[0] : (ky == 1 ? hmask[2] : hmask[4]), hm1 = ky == 0 ? hmask[1] : (ky == 1 ? hmask[3] : hmask[5]);
            const int hl0 = 128 * ky * LDSS, hl1 = ky == 2 ? hst_last : (128 * ky + 64) * LDSS;
            H3_AFF(cn);
            H3_STEP(0) H3_STEP(1) H3_STEP(2)
            if (++ky == 3) { ky = 0; ++c; par ^= 1; }
        }
        __syncthreads();                    // the epilogue reuses the staging memory
        DBG_T();   /* loop end */
        if (p.dbg && t == 0 && k1 - k0 > 6) { p.dbg[blockIdx.x * 24 + 21] = clock64() - ck0; p.dbg[blockIdx.x * 24 + 22] = wall_clock64() - wk0; p.dbg[blockIdx.x * 24 + 20] = (k1 - k0) * 3; }
#undef H3_AFF
#undef H3_XFORM
#undef H3_HLDS
#undef H3_LOAD_B
#undef H3_STORE_B
#undef H3_FRAG
#undef H3_MFMA
#undef H3_GROUP
#undef H3_STEP

        // ------------------------------------------------------------ stream-K hand-off (see conv_mfma_kernel)
        constexpr int NV4 = TN * 4;
        if (k0 > 0) {
            float4* slot = reinterpret_cast<float4*>(p.sk_ws) + (size_t)lb * NV4 * NT + t;
            // write-through payload -> drained -> sc1 flag (no L2 write-back fence)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    store16_wt(slot + (size_t)(j * 4 + q) * NT, v4f{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(p.sk_flags + lb, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            DBG_T(); DBG_T();
            continue;
        }
        if (k1 < nunits) {
            int covered = k1;
            for (int nb = lb + 1; covered < nunits && nb < (int)gridDim.x; ++nb) {
                const int nu0 = (int)((long long)U * nb / gridDim.x), nu1 = (int)((long long)U * (nb + 1) / gridDim.x);
                const int span = (nu1 - nu0) < (nunits - covered) ? (nu1 - nu0) : (nunits - covered);
                if (t == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.sk_flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) { if (p.sk_err) __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const float4* slot = reinterpret_cast<const float4*>(p.sk_ws) + (size_t)nb * NV4 * NT + t;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = slot[(size_t)(j * 4 + q) * NT];
                        acc[j][4 * q] += v.x; acc[j][4 * q + 1] += v.y; acc[j][4 * q + 2] += v.z; acc[j][4 * q + 3] += v.w;
                    }
                covered += span;
            }
        }

        DBG_T();   /* fixup end */
        // ------------------------------------------------------------ epilogue: wave = output row, MFMA rows = columns
        float* red = smem;                 // [8][BN] float2 + [8] int
        // output pixel of MFMA row mi: A tiles (oy0 + wave, ox0 + mi); B tiles (oy0 + 2 wave + mi / 16, ox0 + un-rotated column)
#define H3_OPIX(r_)                                                                                 \
        const int mi_ = ((r_) & 3) + 8 * ((r_) >> 2) + rbase;                                       \
        const int oy = tb ? oy0 + 2 * wave + (mi_ >> 4) : oy0 + wave;                               \
        const int ox = tb ? ox0 + (mi_ < 16 ? mi_ : ((mi_ + 14) & 15)) : ox0 + mi_;                 \
        const bool ok_ = tb ? (oy < p.OH) & (ox < p.OW) & ((mi_ < 16 ? mi_ : ((mi_ + 14) & 15)) < 16) : (oy < p.OH) & (ox < p.OW);
        float lsum[TN]; int lcnt = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + col;
            const float bv = p.bias[n];
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                H3_OPIX(r);
                const float v = acc[j][r] + bv;
                acc[j][r] = v;
                if (ok_) {
                    if (n < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + n] = v;
                    sm += v;
                    if (j == 0) ++lcnt;
                }
            }
            lsum[j] = sm;
        }
        if (p.partials != nullptr) {
            float2* st = reinterpret_cast<float2*>(red);          // [8 waves][BN]
            int* wn = reinterpret_cast<int*>(red + 16 * BN);        // [8]
            const int nw = lcnt + __shfl_xor(lcnt, 32);             // valid pixels of this wave's 32
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float sm = lsum[j] + __shfl_xor(lsum[j], 32);
                const float mu = nw ? sm / (float)nw : 0.f;
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    H3_OPIX(r);
                    const float d = acc[j][r] - mu;
                    if (ok_) q = fmaf(d, d, q);
                }
                q += __shfl_xor(q, 32);
                if (lane < 32) st[wave * BN + j * 32 + lane] = make_float2(mu, q);
            }
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < BN) {
                int n;
                p.partials[(size_t)tile * p.COUTp + t] = merge_wave_stats(st, wn, 8, BN, t, &n);
                if (t == 0) p.counts[tile] = n;
            }
        }
#undef H3_OPIX
        __syncthreads();
        DBG_T();   /* epilogue end */
    }
    if (p.dbg && t == 0) p.dbg[blockIdx.x * 24 + 23] = dbi;
#undef DBG_T
}

// ------------------------------------------------------------------------------------------------
// Optional fast mode (NOT the parity mode): the same halo-resident kernel with the two operands rounded to
// bf16 on their way into LDS (activations after the pending IN/ReLU transform, weights pre-rounded on the host) and
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 activations in HBM).  One K step = 2 matrix instructions per 32x32 tile
// instead of 16, so the kernel turns from MFMA-bound into staging-bound.  Selected per network with fav_net_set_precision.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BN, bool S2>
__global__ __launch_bounds__(512, 2) void conv3_halo_bf16_kernel(const H3Args p)
{
    constexpr int NT = 512;
    constexpr int HWD = H3_TW + 2, HP = (H3_TH + 2) * HWD;        // 34, 340 halo pixels
    constexpr int TN = BN / 32;
    constexpr int NHV = (HP * 8 + NT - 1) / NT;                   // 16-byte halo pieces per thread per slice (6)
    constexpr int ALIAS = NT * NHV - HP * 8;                       // units past the end alias earlier ones (same data, same slot)
    static_assert(NHV == 6, "two halo pieces per tap row");
    static_assert(ALIAS % 8 == 0 && ALIAS <= NT, "halo aliasing");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LB = 40;                      // LDS row stride in bf16 units: 32 channels + 8 pad = 80 bytes (conflict-free ds_read_b128)
    unsigned short* Hs = reinterpret_cast<unsigned short*>(smem);    // [2][HP][LB]   bf16
    unsigned short* Bs = Hs + 2 * HP * LB;                           // [3][BN][LB]   bf16
    float* aff = reinterpret_cast<float*>(Bs + 3 * BN * LB);         // [4][CIN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int CIN = p.CIN;
    const int nchunks = CIN >> 5, nsteps = nchunks * 9;
    const int ntiles = p.tiles_x * p.tiles_y;

    int dbi = 0;
#define DBG_T() { if (p.dbg && t == 0 && dbi < 22) p.dbg[blockIdx.x * 24 + dbi++] = wall_clock64(); }
    DBG_T();
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    for (int i = t; i < CIN; i += NT) {
        aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f;
        aff[2 * CIN + i] = p.stages >= 2 ? p.scale2[i] : 1.f; aff[3 * CIN + i] = p.stages >= 2 ? p.shift2[i] : 0.f;
    }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float lo2 = (p.stages >= 2 && p.relu2) ? 0.f : -INFINITY;
    __syncthreads();

    const int c4 = t & 7, r0 = t >> 3;                      // staging: weight row r0 (+64) / halo pixel r0 (+64 i), 16-byte chunk c4
    const int frag_k = (lane >> 5) * 8;                     // 8 consecutive channels per half-wave (one 32x32x16 operand)
    const int m = lane & 31;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    // per-thread bases; everything else in the K loop is an immediate or a scalar
    const int wr = t >> 2, wc = t & 3;                                      // weight staging: row wr, 16-byte chunk wc (8 bf16)
    const bool wact = wr < BN;
    const unsigned wofs = (unsigned)(wr * p.Kpad + wc * 8) * 2u;            // byte offset of this thread's weight chunk in a step
    unsigned short* const bst = Bs + wr * LB + wc * 8;                      // weight staging slot
    unsigned short* const hst = Hs + r0 * LB + c4 * 4;                      // halo staging slot of piece 0, buffer 0 (4 bf16 = 8 bytes)
    const int hst_last = (t + NT * (NHV - 1) >= HP * 8) ? (NT * (NHV - 1) - ALIAS) / 8 * LB : 64 * (NHV - 1) * LB;
    const unsigned short* const afr = Hs + (wave * HWD + m) * LB + frag_k;  // A fragments: tap (0,0), buffer 0
    const unsigned short* const bfr = Bs + m * LB + frag_k;                 // B fragments: ring slot 0
    const float* const affr = aff + c4 * 4;

    // stream-K work unit: one tap row (3 K steps) of one channel slice of one tile.  Inside a unit kx, the weight ring
    // slot (= kx) and the position of the halo pieces are compile-time constants; ky and the slice are scalars.
    const int nunits = nchunks * 3;
    const int U = ntiles * nunits;
    int u = (int)((long long)U * lb / gridDim.x);
    const int u_end = (int)((long long)U * (lb + 1) / gridDim.x);

    while (u < u_end) {
        const int tile = u / nunits;
        const int k0 = u - tile * nunits;
        const int k1 = (u_end - u) < nunits - k0 ? k0 + (u_end - u) : nunits;
        u += k1 - k0;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int oy0 = ty * H3_TH, ox0 = tx * H3_TW;
        DBG_T();   /* work item start */

        // halo piece i: unit e = t + 512*i -> halo pixel e>>3, channel chunk e&7; per tile: byte offset (chunk 0 if outside) and mask
        int hoff[NHV]; float hmask[NHV];
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            int e = t + NT * i; e -= e >= HP * 8 ? ALIAS : 0;
            const int pix = e >> 3, hy = (pix * 1928) >> 16, hx = pix - hy * HWD;
            const int iy = oy0 - p.pad + hy, ix = ox0 - p.pad + hx;
            const bool v = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
            hoff[i] = ((v ? ((iy >> p.ups) * p.IWp + (ix >> p.ups)) * CIN : 0) + c4 * 4) * 4;
            hmask[i] = v ? 1.f : 0.f;
        }
        const int c_first = (k0 * 21846) >> 16, ky0 = k0 - c_first * 3;      // k / 3
        const int c_last = ((k1 - 1) * 21846) >> 16;

        float4 hr; float hm; v4f rb;
        v4f sc1, sh1, sc2, sh2;             // IN/ReLU stages of the slice being staged, this thread's 4 channels
#define H3_AFF(chunk_)                                                                              \
        { sc1 = *reinterpret_cast<const v4f*>(affr + (chunk_) * 32); sh1 = *reinterpret_cast<const v4f*>(affr + CIN + (chunk_) * 32); \
          if (S2) { sc2 = *reinterpret_cast<const v4f*>(affr + 2 * CIN + (chunk_) * 32); sh2 = *reinterpret_cast<const v4f*>(affr + 3 * CIN + (chunk_) * 32); } }
#define H3_XFORM(v_, m_)                                                                            \
        { v_.x = fmaxf(fmaf(v_.x, sc1.x, sh1.x), lo1); v_.y = fmaxf(fmaf(v_.y, sc1.y, sh1.y), lo1);  \
          v_.z = fmaxf(fmaf(v_.z, sc1.z, sh1.z), lo1); v_.w = fmaxf(fmaf(v_.w, sc1.w, sh1.w), lo1);  \
          if (S2) { v_.x = fmaxf(fmaf(v_.x, sc2.x, sh2.x), lo2); v_.y = fmaxf(fmaf(v_.y, sc2.y, sh2.y), lo2); \
                    v_.z = fmaxf(fmaf(v_.z, sc2.z, sh2.z), lo2); v_.w = fmaxf(fmaf(v_.w, sc2.w, sh2.w), lo2); } \
          v_.x *= m_; v_.y *= m_; v_.z *= m_; v_.w *= m_; }
#define H3_HLDS(i_) ((i_) == NHV - 1 ? hst_last : 64 * (i_) * LB)
#define H3_PUT(dst_, v_) { const bf16x2 lo_ = __builtin_convertvector(f32x2{v_.x, v_.y}, bf16x2), hi_ = __builtin_convertvector(f32x2{v_.z, v_.w}, bf16x2); \
                         uint2 w_; w_.x = __builtin_bit_cast(unsigned, lo_); w_.y = __builtin_bit_cast(unsigned, hi_); *reinterpret_cast<uint2*>(dst_) = w_; }
#define H3_LOAD_B(src_)  { if (wact) rb = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(src_) + wofs); }
#define H3_STORE_B(slot_) { if (wact) *reinterpret_cast<v4f*>(bst + (slot_) * BN * LB) = rb; }

        {
            // prologue: this slice's whole halo -> buffer 0; the pieces of the next slice that the skipped tap rows would
            // have staged -> buffer 1; all loads in flight before the first store
            const char* in0 = reinterpret_cast<const char*>(p.in + c_first * 32);
            const char* in1 = reinterpret_cast<const char*>(p.in + min(c_first + 1, c_last) * 32);
            float4 q0[NHV], q1[NHV];
#pragma unroll
            for (int i = 0; i < NHV; ++i) q0[i] = *reinterpret_cast<const float4*>(in0 + hoff[i]);
            H3_LOAD_B(p.wgt16 + k0 * 3 * BK);
#pragma unroll
            for (int i = 0; i < NHV; ++i) if (i < 2 * ky0) q1[i] = *reinterpret_cast<const float4*>(in1 + hoff[i]);
            H3_AFF(c_first);
#pragma unroll
            for (int i = 0; i < NHV; ++i) { H3_XFORM(q0[i], hmask[i]); H3_PUT(hst + H3_HLDS(i), q0[i]); }
            H3_STORE_B(0);
            H3_LOAD_B(p.wgt16 + min(k0 * 3 + 1, nsteps - 1) * BK);
            H3_AFF(min(c_first + 1, c_last));
#pragma unroll
            for (int i = 0; i < NHV; ++i) if (i < 2 * ky0) { H3_XFORM(q1[i], hmask[i]); H3_PUT(hst + HP * LB + H3_HLDS(i), q1[i]); }
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        __syncthreads();

        bf16x8 fa[2], fb[2][TN];
#define H3_FRAG(set_, ap_, bp_)                                                                     \
        { fa[set_] = *reinterpret_cast<const bf16x8*>(ap_);                                         \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[set_][j] = *reinterpret_cast<const bf16x8*>((bp_) + j * 32 * LB); }
#define H3_MFMA(set_)                                                                               \
        { _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set_], fb[set_][j], acc[j], 0, 0, 0); }
#define H3_GROUP(nds_)                                                                              \
        { __builtin_amdgcn_sched_group_barrier(0x100, nds_, 0); __builtin_amdgcn_sched_group_barrier(0x008, TN, 0); }
        // one K step (32 channels of one tap = two 16-channel MFMA groups), kx = KX (compile time)
#define H3_STEP(KX)                                                                                 \
        {   constexpr int NB = ((KX) + 1) % 3;                                                      \
            const unsigned short* an_ = (KX) == 2 ? a_nu : a_cu + ((KX) + 1) * LB;                  \
            H3_FRAG(1, a_cu + (KX) * LB + 16, bfr + (KX) * BN * LB + 16);                           \
            H3_STORE_B(NB);                                                                         \
            H3_LOAD_B(p.wgt16 + min(sg + (KX) + 2, nsteps - 1) * BK);                               \
            if ((KX) < 2) { hr = *reinterpret_cast<const float4*>(in_n + ((KX) == 0 ? ho0 : ho1)); hm = (KX) == 0 ? hm0 : hm1; } \
            H3_MFMA(0); H3_GROUP(1 + TN);                                                           \
            __syncthreads();                                                                        \
            H3_FRAG(0, an_, bfr + NB * BN * LB);                                                    \
            if ((KX) < 2) { H3_XFORM(hr, hm); H3_PUT(h_nx + ((KX) == 0 ? hl0 : hl1), hr); }         \
            H3_MFMA(1); H3_GROUP(1 + TN);                                                           \
        }

        H3_FRAG(0, afr + ky0 * HWD * LB, bfr);             // fragments of the first step's first group
        DBG_T();   /* loop start */
        const long long ck0 = p.dbg ? clock64() : 0, wk0 = p.dbg ? wall_clock64() : 0;
        int c = c_first, ky = ky0, par = 0;
        for (int uu = k0; uu < k1; ++uu) {
            const unsigned short* a_cu = afr + (par * HP + ky * HWD) * LB;
            const unsigned short* a_nu = ky == 2 ? afr + (par ^ 1) * (HP * LB) : a_cu + HWD * LB;
            unsigned short* h_nx = hst + (par ^ 1) * (HP * LB);
            const int cn = min(c + 1, c_last);                              // no next slice: the pieces land in the unused buffer
            const char* in_n = reinterpret_cast<const char*>(p.in + cn * 32);
            const int sg = uu * 3;
            const int ho0 = ky == 0 ? hoff[0] : (ky == 1 ? hoff[2] : hoff[4]), ho1 = ky == 0 ? hoff[1] : (ky == 1 ? hoff[3] : hoff[5]);
            const float hm0 = ky == 0 ? hmask[0] : (ky == 1 ? hmask[2] : hmask[4]), hm1 = ky == 0 ? hmask[1] : (ky == 1 ? hmask[3] : hmask[5]);
            const int hl0 = 128 * ky * LB, hl1 = ky == 2 ? hst_last : (128 * ky + 64) * LB;
            H3_AFF(cn);
            H3_STEP(0) H3_STEP(1) H3_STEP(2)
            if (++ky == 3) { ky = 0; ++c; par ^= 1; }
        }
        __syncthreads();                    // the epilogue reuses the staging memory
        DBG_T();   /* loop end */
        if (p.dbg && t == 0 && k1 - k0 > 6) { p.dbg[blockIdx.x * 24 + 21] = clock64() - ck0; p.dbg[blockIdx.x * 24 + 22] = wall_clock64() - wk0; p.dbg[blockIdx.x * 24 + 20] = (k1 - k0) * 3; }
#undef H3_AFF
#undef H3_XFORM
#undef H3_HLDS
#undef H3_PUT
#undef H3_LOAD_B
#undef H3_STORE_B
#undef H3_FRAG
#undef H3_MFMA
#undef H3_GROUP
#undef H3_STEP

        // ------------------------------------------------------------ stream-K hand-off (see conv_mfma_kernel)
        constexpr int NV4 = TN * 4;
        if (k0 > 0) {
            float4* slot = reinterpret_cast<float4*>(p.sk_ws) + (size_t)lb * NV4 * NT + t;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    slot[(size_t)(j * 4 + q) * NT] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.sk_flags + lb, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            DBG_T(); DBG_T();
            continue;
        }
        if (k1 < nunits) {
            int covered = k1;
            for (int nb = lb + 1; covered < nunits && nb < (int)gridDim.x; ++nb) {
                const int nu0 = (int)((long long)U * nb / gridDim.x), nu1 = (int)((long long)U * (nb + 1) / gridDim.x);
                const int span = (nu1 - nu0) < (nunits - covered) ? (nu1 - nu0) : (nunits - covered);
                if (t == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.sk_flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) { if (p.sk_err) __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const float4* slot = reinterpret_cast<const float4*>(p.sk_ws) + (size_t)nb * NV4 * NT + t;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = slot[(size_t)(j * 4 + q) * NT];
                        acc[j][4 * q] += v.x; acc[j][4 * q + 1] += v.y; acc[j][4 * q + 2] += v.z; acc[j][4 * q + 3] += v.w;
                    }
                covered += span;
            }
        }

        DBG_T();   /* fixup end */
        // ------------------------------------------------------------ epilogue: wave = output row, MFMA rows = columns
        float* red = smem;                 // [8][BN] + [BN]
        const int oy = oy0 + wave;
        const int vh = min(H3_TH, p.OH - oy0), vw = min(H3_TW, p.OW - ox0);
        const int cnt = vh * vw;
        float lsum[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = j * 32 + col;
            const float bv = p.bias[n];
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + rbase;
                const float v = acc[j][r] + bv;
                acc[j][r] = v;
                if (oy < p.OH && ox < p.OW) {
                    if (n < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + n] = v;
                    sm += v;
                }
            }
            lsum[j] = sm;
        }
        if (p.partials != nullptr) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float sm = lsum[j] + __shfl_xor(lsum[j], 32);
                if (lane < 32) red[wave * BN + j * 32 + lane] = sm;
            }
            __syncthreads();
            if (t < BN) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += red[w * BN + t];
                red[8 * BN + t] = a / (float)cnt;
            }
            __syncthreads();
            float lq[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float mu = red[8 * BN + j * 32 + col];
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + rbase;
                    const float d = acc[j][r] - mu;
                    if (oy < p.OH && ox < p.OW) q = fmaf(d, d, q);
                }
                lq[j] = q + __shfl_xor(q, 32);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (lane < 32) red[wave * BN + j * 32 + lane] = lq[j];
            __syncthreads();
            if (t < BN) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += red[w * BN + t];
                p.partials[(size_t)tile * p.COUTp + t] = make_float2(red[8 * BN + t], a);
                if (t == 0) p.counts[tile] = cnt;
            }
        }
        __syncthreads();
        DBG_T();   /* epilogue end */
    }
    if (p.dbg && t == 0) p.dbg[blockIdx.x * 24 + 23] = dbi;
#undef DBG_T
}


}  // namespace

bool conv3_halo_eligible(int cin_pitch, int coutp, int k, int stride)
{
    return k == 3 && stride == 1 && cin_pitch % 32 == 0 && cin_pitch >= 32 && cin_pitch <= 256 && (coutp == 128 || coutp == 64);
}
// Tile count.  edge_b (fp32 kernel): when the ragged right edge is at most 16 columns wide it is covered by ceil(OH / 16) tiles of
// 16 x 16 instead of ceil(OH / 8) tiles of 8 x 32 -- the residual layers' widths (338 ... 320) leave 2 ... 18 columns there, i.e.
// up to 9 % of the matrix work used to be spent on columns outside the image.
static void h3_tiling(int OH, int OW, bool edge_b, int* tx, int* ty, int* nb)
{
    const int r = OW % H3_TW;
    *ty = (OH + H3_TH - 1) / H3_TH;
    if (edge_b && r > 0 && r <= 16 && OW > H3_TW) { *tx = OW / H3_TW; *nb = (OH + 15) / 16; }
    else { *tx = (OW + H3_TW - 1) / H3_TW; *nb = 0; }
}
int conv3_halo_tiles(int OH, int OW, bool edge_b) { int tx, ty, nb; h3_tiling(OH, OW, edge_b, &tx, &ty, &nb); return tx * ty + nb; }

// FAV_H3_DBG=n: print the in-kernel timeline (prologue / K loop / stream-K fix-up / epilogue, shader clock) of the n-th launch
static void h3_debug_report(const long long* h, int grid)
{
    long long t0 = h[0];
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, h[b * 24]);
    double sum[4] = {0, 0, 0, 0}, tend = 0, ck = 0, wk = 0, steps = 0; int items = 0;
    for (int b = 0; b < grid; ++b) {
        const long long* r = &h[b * 24]; const int n = (int)r[23];
        for (int i = 1; i + 4 < n + 1 && i + 4 <= 21; i += 5) {
            for (int q = 0; q < 4; ++q) sum[q] += (r[i + q + 1] - r[i + q]) * 0.01;
            ++items; tend = std::max(tend, (r[i + 4] - t0) * 0.01);
        }
        ck += r[21]; wk += r[22]; steps += r[20];
    }
    fprintf(stderr, "H3DBG grid=%d items=%d  K loop: %.0f clk/step, %.3f GHz, %.3f us/step;  per block: prologue %.2f  loop %.2f  fix-up %.2f  epilogue %.2f us;  last block ends at %.2f us\n",
            grid, items, steps ? ck / steps : 0.0, wk ? ck / (wk * 10.0) : 0.0, steps ? wk * 0.01 / steps : 0.0,
            sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[3] / grid, tend);
}

template <int BN, bool S2, bool BF>
static int launch_h3_t(const H3Args& a0, int cin, int reserve_cus, bool no_sk, hipStream_t st)
{
    const auto kern = BF ? conv3_halo_bf16_kernel<BN, S2> : conv3_halo_kernel<BN, S2>;
    const size_t lds = BF ? (size_t)(2 * (H3_TH + 2) * (H3_TW + 2) * 40 + 3 * BN * 40) * 2 + (size_t)4 * cin * sizeof(float)
                          : (size_t)(2 * (H3_TH + 2) * (H3_TW + 2) * LDSS + 3 * BN * LDSS + 4 * cin) * sizeof(float);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int occ = 0; int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        FAV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 512, lds));
        if (occ < 1) { set_error("halo conv: kernel does not fit on a CU"); return FAV_EHIP; }
        cus[dv] = prop_cus;          // one block per CU
    }
    int nres = std::max(1, cus[dv] - reserve_cus);
    if (nres > SK_GRID) nres = SK_GRID;
    const int tiles = a0.tiles_x * a0.tiles_y + a0.nb;
    // no_sk (shared device): one block per tile -- the unit range of block b is then exactly tile b, nothing is handed between blocks
    // and nothing needs to be co-resident
    // Almost exactly one tile per CU (the 16x16 edge tiles bring six of the ten residual layers to 242 / 252 tiles for 256 CUs): one
    // whole tile per block beats stream-K there -- the 12/11.8 longer K range costs less than the second prologue and the
    // hand-off of a split tile (measured: timeline in DESIGN.md section 4)
    // (threshold swept on the MI355X: 165.1 us without, 161.3 at 96 %, 159.6 at 94 %, 159.8 at 89 %)
    const bool one_per_cu = tiles <= nres && tiles * 100 >= nres * 94;
    const int grid = (no_sk || one_per_cu) ? tiles : (tiles * (cin / 32) * 3 < nres ? 1 : nres);      // (stream-K units: tap rows)
    H3Args a = a0; a.dbg = nullptr;
    static int dbg_n = diag_env("FAV_H3_DBG") ? atoi(diag_env("FAV_H3_DBG")) : 0;
    static long long* dbuf = nullptr;
    const bool dbg = dbg_n > 0 && BN == 128 && !BF && --dbg_n == 0;
    if (dbg) { FAV_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), SK_GRID * 24 * 8)); FAV_HIP(hipMemsetAsync(dbuf, 0, SK_GRID * 24 * 8, st)); a.dbg = dbuf; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv3_halo_kernel");
    if (dbg) {
        std::vector<long long> h((size_t)SK_GRID * 24);
        FAV_HIP(hipStreamSynchronize(st)); FAV_HIP(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
        h3_debug_report(h.data(), grid);
    }
    return FAV_OK;
}

int launch_conv3_halo(const ConvLaunch& c, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3_halo_eligible(c.CIN, c.COUTp, c.KH, c.stride) && c.KH == c.KW && !c.final_mode && !c.stuff && c.sk_ws && c.sk_flags,
                "halo conv: not eligible");
    FAV_REQUIRE((long long)((c.IH >> c.ups) + 1) * c.IWp * c.CIN < (1ll << 31), "halo conv: tensor too large for 32-bit offsets");
    H3Args a;
    a.in = c.in; a.wgt = c.wgt; a.bias = c.bias;
    a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.scale2 = c.pre.scale2; a.shift2 = c.pre.shift2;
    a.stages = c.pre.stages; a.relu1 = c.pre.relu1; a.relu2 = c.pre.relu2;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.sk_ws = c.sk_ws; a.sk_flags = c.sk_flags; a.sk_epoch = c.sk_epoch; a.sk_err = c.sk_err;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.ups = c.ups; a.CIN = c.CIN; a.COUT = c.COUT; a.COUTp = c.COUTp; a.pad = c.pad;
    a.OH = c.OH; a.OW = c.OW; a.Kpad = c.Kpad;
    h3_tiling(c.OH, c.OW, c.wgt16 == nullptr, &a.tiles_x, &a.tiles_y, &a.nb);      // (the bf16 fast-mode kernel keeps 8 x 32 tiles only)
    const bool s2 = c.pre.stages >= 2;
    a.wgt16 = c.wgt16;
    if (c.wgt16) {           // fast mode: bf16 operands
        if (c.COUTp == 128) return s2 ? launch_h3_t<128, true, true>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st) : launch_h3_t<128, false, true>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st);
        return s2 ? launch_h3_t<64, true, true>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st) : launch_h3_t<64, false, true>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st);
    }
    if (c.COUTp == 128) return s2 ? launch_h3_t<128, true, false>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st) : launch_h3_t<128, false, false>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st);
    return s2 ? launch_h3_t<64, true, false>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st) : launch_h3_t<64, false, false>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st);
}

// ------------------------------------------------------------------------------------------------
// 3x3 STRIDE-2 layers (d64: 32 -> 64 at 1360x800, d128: 64 -> 128 at 680x400; models_video.lua:88-92): halo-resident implicit
// GEMM with even / odd column planes.  The generic kernel re-gathers its operand per tap with 2-5 vector-ALU instructions per
// MFMA (address arithmetic + the pending transform, nine times per element) and reaches 0.44 / 0.55 of the fp32 MFMA peak on
// these two layers.  Here a block (8 waves, one per CU, stream-K over (tile, slice, tap row) units like the stride-1 kernel)
// owns a 4 x 32 pixel output tile: wave = (output row, half of the output channels).  Per 32-channel slice the
// (2*4+1) x (2*32+1) = 9 x 65 pixel halo is gathered ONCE (IN/ReLU applied, zero padding) into LDS as two planes -- even
// input columns (33 per row) and odd input columns (32 per row) -- so that for every tap the 32 lanes of a wave (32 consecutive
// OUTPUT columns = input columns 2m + kx) read 32 CONSECUTIVE pixels of one plane: conflict-free ds_read_b128, immediate tap
// offsets.  585 pixels x 144 B = 84 KB: one halo buffer only, so the next slice's halo travels through registers (10 pieces of
// 16 bytes per thread, loaded one per K step) and is written between slices.  Weights stream through the same 3-slot ring as
// in the stride-1 kernel; one barrier per K step (mid-step), two per slice change.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int S2_TH = 4, S2_TW = 32;                 // output tile
constexpr int S2_HR = 2 * S2_TH + 1;                 // 9 halo rows
constexpr int S2_EW = S2_TW + 1, S2_OW = S2_TW;      // even / odd plane widths (33, 32)
constexpr int S2_EP = S2_HR * S2_EW;                 // 297 pixels in the even plane
constexpr int S2_HP = S2_EP + S2_HR * S2_OW;         // 585 halo pixels
constexpr int S2_NHV = 10;                           // 16-byte halo pieces per thread and slice (585 * 8 / 512 = 9.14)

struct S2Args {
    const float* in; const float* wgt; const float* bias;
    const float* scale1; const float* shift1;
    float* out; float2* partials; int* counts;
    float* sk_ws; unsigned* sk_flags; unsigned sk_epoch; unsigned* sk_err;
    int IH, IW, IWp, CIN, COUT, COUTp, pad, OH, OW, Kpad, tiles_x, tiles_y;
    int stages, relu1;
};

template <int BN>
__global__ __launch_bounds__(512, 2) void conv3s2_halo_kernel(const S2Args p)
{
    constexpr int NT = 512;
    constexpr int TN = BN / 64;                       // 32-channel accumulator tiles per wave (a wave owns BN/2 channels)
    constexpr int BROWS = BN / 64;                    // weight rows per thread per step
    constexpr int ALIAS = NT * S2_NHV - S2_HP * 8;    // staging units past the end alias earlier ones (same data, same slot)
    static_assert(ALIAS % 8 == 0 && ALIAS >= 0 && ALIAS <= NT, "halo aliasing");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                 // [585][LDSS]: even plane, then odd plane
    float* Bs = Hs + S2_HP * LDSS;                    // [3][BN][LDSS]
    float* aff = Bs + 3 * BN * LDSS;                  // [2][CIN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave & 3, nh = wave >> 2;          // output row of the tile, channel half
    const int CIN = p.CIN;
    const int nchunks = CIN >> 5, nsteps = nchunks * 9;
    const int ntiles = p.tiles_x * p.tiles_y;
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    for (int i = t; i < CIN; i += NT) { aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f; }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;

    const int c4 = t & 7, r0 = t >> 3;
    const int frag_k = (lane >> 5) * 4, m = lane & 31;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    const unsigned wofs = (unsigned)(r0 * p.Kpad + c4 * 4) * 4u;
    const unsigned wrow64 = (unsigned)(64 * p.Kpad) * 4u;
    float* const bst = Bs + r0 * LDSS + c4 * 4;                                        // weight staging slot (ring slot 0)
    // A fragments: even plane (kx = 0, 2) and odd plane (kx = 1), tap row 0, this wave's output row
    const float* const afrE = Hs + ((2 * wr) * S2_EW + m) * LDSS + frag_k;
    const float* const afrO = Hs + (S2_EP + (2 * wr) * S2_OW + m) * LDSS + frag_k;
    const float* const bfr = Bs + (nh * (BN / 2) + m) * LDSS + frag_k;                 // B fragments: ring slot 0, this wave's channels
    const float* const affr = aff + c4 * 4;

    // halo piece i of this thread: staging unit e = t + 512 i -> halo pixel e >> 3 (plane-major), 16-byte chunk c4.  Its position
    // inside the halo is fixed; the tile only moves the origin.
    int hlds[S2_NHV], hyx[S2_NHV];
#pragma unroll
    for (int i = 0; i < S2_NHV; ++i) {
        int e = t + NT * i; e -= e >= S2_HP * 8 ? ALIAS : 0;
        const int pe = e >> 3;
        int hy, hx;
        if (pe < S2_EP) { hy = pe / S2_EW; hx = 2 * (pe - hy * S2_EW); }
        else { const int q = pe - S2_EP; hy = q / S2_OW; hx = 2 * (q - hy * S2_OW) + 1; }
        hlds[i] = pe * LDSS + c4 * 4;
        hyx[i] = hy << 16 | hx;
    }

    // Work of this block: a contiguous range of stream-K units (unit = one tap row = 3 K steps of one slice of one tile), walked
    // as SEGMENTS = the part of one (tile, slice) inside the range.  While a segment computes, the halo of the NEXT segment --
    // the next slice of the tile or the first slice of the next tile -- is fetched into registers (hq), so that neither a slice
    // change nor a tile change waits for memory: these layers read 1.1 x their input once per tile and are otherwise
    // bandwidth-exposed (d64: 229 MB of traffic against 72 us of matrix work).
    const int nunits = nchunks * 3;
    const int U = ntiles * nunits;
    int u = (int)((long long)U * lb / gridDim.x);
    const int u_end = (int)((long long)U * (lb + 1) / gridDim.x);

    int hoff[S2_NHV]; float hmask[S2_NHV];            // of the segment being FETCHED
    float4 hq[S2_NHV];
    v4f rb[BROWS];
#define S2_TILE_SETUP(tile_)                                                                        \
    {   const int ty_ = (tile_) / p.tiles_x, tx_ = (tile_) - ty_ * p.tiles_x;                       \
        _Pragma("unroll") for (int i = 0; i < S2_NHV; ++i) {                                        \
            const int iy = 2 * ty_ * S2_TH - p.pad + (hyx[i] >> 16), ix = 2 * tx_ * S2_TW - p.pad + (hyx[i] & 0xffff); \
            const bool v = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);       \
            hoff[i] = ((v ? (iy * p.IWp + ix) * CIN : 0) + c4 * 4) * 4;                             \
            hmask[i] = v ? 1.f : 0.f;                                                               \
        } }
#define S2_XFORM(v_, sc_, sh_, m_)                                                                  \
    { v_.x = fmaxf(fmaf(v_.x, sc_.x, sh_.x), lo1) * m_; v_.y = fmaxf(fmaf(v_.y, sc_.y, sh_.y), lo1) * m_;  \
      v_.z = fmaxf(fmaf(v_.z, sc_.z, sh_.z), lo1) * m_; v_.w = fmaxf(fmaf(v_.w, sc_.w, sh_.w), lo1) * m_; }
// parked pieces (slice cs_) -> transformed -> the halo buffer
#define S2_COMMIT(cs_)                                                                              \
    {   const v4f sc_ = *reinterpret_cast<const v4f*>(affr + (cs_) * 32), sh_ = *reinterpret_cast<const v4f*>(affr + CIN + (cs_) * 32); \
        _Pragma("unroll") for (int i = 0; i < S2_NHV; ++i) { S2_XFORM(hq[i], sc_, sh_, hmask[i]); *reinterpret_cast<float4*>(Hs + hlds[i]) = hq[i]; } }
#define S2_LOAD_B(gs_)                                                                              \
    { const float* src_ = p.wgt + min((gs_), nsteps - 1) * BK;                                      \
      _Pragma("unroll") for (int j = 0; j < BROWS; ++j) rb[j] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(src_) + (wofs + j * wrow64)); }
#define S2_STORE_B(slot_)                                                                           \
    { _Pragma("unroll") for (int j = 0; j < BROWS; ++j) *reinterpret_cast<v4f*>(bst + ((slot_) * BN + 64 * j) * LDSS) = rb[j]; }

    // first segment of the range: fetched with exposed latency, once per block
    int tile = 0, c = 0, t_lo = 0, t_hi = 0;          // current segment: slice c of `tile`, taps [t_lo, t_hi)
    int k1 = 0;                                       // end (in units) of the current work item inside its tile
    if (u < u_end) {
        tile = u / nunits;
        const int k0 = u - tile * nunits;
        k1 = (u_end - u) < nunits - k0 ? k0 + (u_end - u) : nunits;
        c = (k0 * 21846) >> 16; t_lo = 3 * (k0 - c * 3);
        t_hi = min(9, 3 * (k1 - c * 3));
        S2_TILE_SETUP(tile);
        const char* in0 = reinterpret_cast<const char*>(p.in + c * 32);
#pragma unroll
        for (int i = 0; i < S2_NHV; ++i) hq[i] = *reinterpret_cast<const float4*>(in0 + hoff[i]);
        S2_LOAD_B(c * 9 + t_lo);
        __syncthreads();                              // transform tables
        S2_COMMIT(c);
        S2_STORE_B(0);
        S2_LOAD_B(c * 9 + t_lo + 1);
    }
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bool item_start = true;                           // the current segment opens a work item (its tile's accumulators start at 0)
    int k0_item = u < u_end ? u - tile * nunits : 0;  // first unit of the current work item inside its tile
    __syncthreads();

    v4f fa[2], fb[2][TN];
#define S2_FRAG(set_, ap_, bp_)                                                                     \
    { fa[set_] = *reinterpret_cast<const v4f*>(ap_);                                                \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[set_][j] = *reinterpret_cast<const v4f*>((bp_) + j * 32 * LDSS); }
#define S2_MFMA(set_)                                                                               \
    { _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                              \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].x, fb[set_][j].x, acc[j], 0, 0, 0);  \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].y, fb[set_][j].y, acc[j], 0, 0, 0);  \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].z, fb[set_][j].z, acc[j], 0, 0, 0);  \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set_].w, fb[set_][j].w, acc[j], 0, 0, 0); } }
// A-fragment base of tap T_ (compile time): even plane for kx = 0 / 2 (shifted by one pixel), odd plane for kx = 1
#define S2_ABASE(T_) (((T_) % 3 == 1 ? afrO + ((T_) / 3) * S2_OW * LDSS : afrE + (((T_) / 3) * S2_EW + ((T_) % 3 == 2 ? 1 : 0)) * LDSS))
// one K step = tap T_ of the current slice (32 channels); ring slot = T_ % 3.  The weights two steps ahead in EXECUTION order are
// requested (the step after the segment's last one is the first step of the next segment: ngs), and piece T_ of the next
// segment's halo (piece 9 rides with tap 0).  PIN_: the group-0 fragments were read by the previous step; POUT_: read those of
// tap T_ + 1 (compile time: a run-time flag here costs dozens of v_mov per step).
#define S2_STEP(T_, PIN_, POUT_)                                                                    \
    {                                                                                               \
        const float* a_ = S2_ABASE(T_);                                                             \
        const float* b_ = bfr + ((T_) % 3) * BN * LDSS;                                             \
        if (!(PIN_)) S2_FRAG(0, a_, b_);                                                            \
        S2_FRAG(1, a_ + 8, b_ + 8);                                                                 \
        S2_STORE_B(((T_) + 1) % 3);                                                                 \
        S2_LOAD_B((T_) + 2 < t_hi ? c * 9 + (T_) + 2 : ngs + ((T_) + 2 - t_hi));                    \
        if (has_next) { hq[T_] = *reinterpret_cast<const float4*>(in_n + hoff[T_]); if ((T_) == 0) hq[9] = *reinterpret_cast<const float4*>(in_n + hoff[9]); } \
        S2_MFMA(0);                                                                                 \
        S2_FRAG(0, a_ + 16, b_ + 16); S2_MFMA(1);                                                   \
        __syncthreads();                                                                            \
        S2_FRAG(1, a_ + 24, b_ + 24); S2_MFMA(0);                                                   \
        if (POUT_) { constexpr int TNX = ((T_) + 1) % 9; S2_FRAG(0, S2_ABASE(TNX), bfr + (TNX % 3) * BN * LDSS); } \
        S2_MFMA(1);                                                                                 \
    }
#define S2_STEP_IF(T_) if ((T_) >= t_lo && (T_) < t_hi) S2_STEP(T_, false, false)

    while (u < u_end) {
        // ---- the segment after this one
        const int seg_units = (t_hi - t_lo) / 3;
        const bool item_end = (c * 3 + t_hi / 3) == k1;             // this segment closes the work item (end of the tile or of the range)
        int n_tile = tile, n_c = c + 1, n_lo = 0, n_hi = 9, n_k1 = k1;
        const int u_next = u + seg_units;
        const bool has_next = u_next < u_end;
        if (item_end) {                                             // next segment = head of the next tile
            n_tile = tile + 1; n_c = 0; n_lo = 0;
            n_k1 = (u_end - u_next) < nunits ? (u_end - u_next) : nunits;
        }
        n_hi = min(9, 3 * (n_k1 - n_c * 3));
        const int ngs = has_next ? n_c * 9 + n_lo : nsteps - 1;
        const char* in_n = reinterpret_cast<const char*>(p.in + n_c * 32);
        if (has_next) {
            if (item_end) S2_TILE_SETUP(n_tile);                    // (hoff / hmask now describe the segment being fetched)
            // pieces whose step this (partial) segment does not execute
#pragma unroll
            for (int i = 0; i < S2_NHV; ++i) if (!(i >= t_lo && i < t_hi) && !(i == 9 && t_lo == 0)) hq[i] = *reinterpret_cast<const float4*>(in_n + hoff[i]);
        }
        if (t_lo == 0 && t_hi == 9) {
            // whole slice (the common case): fragments of the next tap are read one step ahead
            S2_STEP(0, false, true) S2_STEP(1, true, true) S2_STEP(2, true, true) S2_STEP(3, true, true) S2_STEP(4, true, true)
            S2_STEP(5, true, true) S2_STEP(6, true, true) S2_STEP(7, true, true) S2_STEP(8, true, false)
        } else {
            // a split tile's partial slice: plain steps
            S2_STEP_IF(0) S2_STEP_IF(1) S2_STEP_IF(2) S2_STEP_IF(3) S2_STEP_IF(4) S2_STEP_IF(5) S2_STEP_IF(6) S2_STEP_IF(7) S2_STEP_IF(8)
        }
        u = u_next;

        if (item_end) {
            __syncthreads();                    // everybody is done with the halo: the epilogue reuses the start of the staging memory
            const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
            const int oy0 = ty * S2_TH, ox0 = tx * S2_TW;
            // ------------------------------------------------------------ stream-K hand-off (as in conv3_halo_kernel)
            constexpr int NV4 = TN * 4;
            bool owner = true;
            if (k0_item > 0) {
                owner = false;
                float4* slot = reinterpret_cast<float4*>(p.sk_ws) + (size_t)lb * NV4 * NT + t;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        store16_wt(slot + (size_t)(j * 4 + q) * NT, v4f{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) __hip_atomic_store(p.sk_flags + lb, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (k1 < nunits) {
                int covered = k1;
                for (int nb = lb + 1; covered < nunits && nb < (int)gridDim.x; ++nb) {
                    const int nu0 = (int)((long long)U * nb / gridDim.x), nu1 = (int)((long long)U * (nb + 1) / gridDim.x);
                    const int span = (nu1 - nu0) < (nunits - covered) ? (nu1 - nu0) : (nunits - covered);
                    if (t == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(p.sk_flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
                            __builtin_amdgcn_s_sleep(4);
                            if (++spins > (1u << 22)) { if (p.sk_err) __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const float4* slot = reinterpret_cast<const float4*>(p.sk_ws) + (size_t)nb * NV4 * NT + t;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = slot[(size_t)(j * 4 + q) * NT];
                            acc[j][4 * q] += v.x; acc[j][4 * q + 1] += v.y; acc[j][4 * q + 2] += v.z; acc[j][4 * q + 3] += v.w;
                        }
                    covered += span;
                }
            }
            if (owner) {
                // -------------------------------------------------------- epilogue: wave = (output row, channel half), MFMA rows = columns
                float* red = smem;                 // [8 waves][BN/2] float2 + [8] int
                const int oy = oy0 + wr;
                float lsum[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = nh * (BN / 2) + j * 32 + col;
                    const float bv = p.bias[n];
                    float sm = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ox = ox0 + (r & 3) + 8 * (r >> 2) + rbase;
                        const float v = acc[j][r] + bv;
                        acc[j][r] = v;
                        if (oy < p.OH && ox < p.OW) {
                            if (n < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + n] = v;
                            sm += v;
                        }
                    }
                    lsum[j] = sm;
                }
                if (p.partials != nullptr) {
                    float2* st = reinterpret_cast<float2*>(red);      // [4 rows][BN]: the two channel halves of a row sit side by side
                    int* wn = reinterpret_cast<int*>(red + 2 * S2_TH * BN);
                    const int nw = oy < p.OH ? min(S2_TW, p.OW - ox0) : 0;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float sm = lsum[j] + __shfl_xor(lsum[j], 32);
                        const float mu = nw ? sm / (float)nw : 0.f;
                        float q = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ox = ox0 + (r & 3) + 8 * (r >> 2) + rbase;
                            const float d = acc[j][r] - mu;
                            if (oy < p.OH && ox < p.OW) q = fmaf(d, d, q);
                        }
                        q += __shfl_xor(q, 32);
                        if (lane < 32) st[wr * BN + nh * (BN / 2) + j * 32 + lane] = make_float2(mu, q);
                    }
                    if (lane == 0 && nh == 0) wn[wr] = nw;
                    __syncthreads();
                    if (t < BN) {
                        int n;
                        p.partials[(size_t)tile * p.COUTp + t] = merge_wave_stats(st, wn, S2_TH, BN, t, &n);
                        if (t == 0) p.counts[tile] = n;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            k0_item = 0;
        }
        if (has_next) {
            // segment change: the parked pieces (transformed) replace the halo
            __syncthreads();
            S2_COMMIT(n_c);
            __syncthreads();
        }
        tile = n_tile; c = n_c; t_lo = n_lo; t_hi = n_hi; k1 = n_k1;
        (void)item_start;
    }
#undef S2_TILE_SETUP
#undef S2_XFORM
#undef S2_COMMIT
#undef S2_LOAD_B
#undef S2_STORE_B
#undef S2_FRAG
#undef S2_MFMA
#undef S2_ABASE
#undef S2_STEP
#undef S2_STEP_IF
}

}  // namespace

bool conv3s2_eligible(int cin_pitch, int coutp, int k, int stride, int stages, int ups)
{
    // 64 output channels only: the 128-wide instance (d128) measured 126 us against 115 us of the generic kernel (register
    // pressure: accumulators + the parked halo), the 64-wide one 122 us against 148 us (d64)
    return k == 3 && stride == 2 && ups == 0 && stages <= 1 && cin_pitch % 32 == 0 && cin_pitch >= 32 && cin_pitch <= 256 && coutp == 64;
}
int conv3s2_tiles(int OH, int OW) { return ((OH + S2_TH - 1) / S2_TH) * ((OW + S2_TW - 1) / S2_TW); }

template <int BN>
static int launch_s2_t(const S2Args& a, int cin, int reserve_cus, bool no_sk, hipStream_t st)
{
    const auto kern = conv3s2_halo_kernel<BN>;
    const size_t lds = (size_t)(S2_HP * LDSS + 3 * BN * LDSS + 2 * cin) * sizeof(float);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int occ = 0; int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        FAV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 512, lds));
        if (occ < 1) { set_error("stride-2 halo conv: kernel does not fit on a CU"); return FAV_EHIP; }
        cus[dv] = prop_cus;
    }
    int nres = std::max(1, cus[dv] - reserve_cus);
    if (nres > SK_GRID) nres = SK_GRID;
    const int tiles = a.tiles_x * a.tiles_y;
    const int grid = no_sk ? tiles : (tiles * (cin / 32) * 3 < nres ? 1 : nres);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv3s2_halo_kernel");
    return FAV_OK;
}

int launch_conv3s2(const ConvLaunch& c, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3s2_eligible(c.CIN, c.COUTp, c.KH, c.stride, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && !c.stuff && c.sk_ws && c.sk_flags,
                "stride-2 halo conv: not eligible");
    FAV_REQUIRE((long long)(c.IH + 1) * c.IWp * c.CIN < (1ll << 31), "stride-2 halo conv: tensor too large for 32-bit offsets");
    S2Args a;
    a.in = c.in; a.wgt = c.wgt; a.bias = c.bias; a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.stages = c.pre.stages; a.relu1 = c.pre.relu1;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.sk_ws = c.sk_ws; a.sk_flags = c.sk_flags; a.sk_epoch = c.sk_epoch; a.sk_err = c.sk_err;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.CIN = c.CIN; a.COUT = c.COUT; a.COUTp = c.COUTp; a.pad = c.pad;
    a.OH = c.OH; a.OW = c.OW; a.Kpad = c.Kpad;
    a.tiles_x = (c.OW + S2_TW - 1) / S2_TW; a.tiles_y = (c.OH + S2_TH - 1) / S2_TH;
    return launch_s2_t<64>(a, c.CIN, c.reserve_cus, c.no_sk != 0, st);
}

// ------------------------------------------------------------------------------------------------
// Last layer (c9s1-3: 64 -> 3 channels, 9x9): "row-folded" implicit GEMM.
// With only 3 output channels a pixels x channels GEMM would waste 29/32 of every MFMA.  Instead the
// kx taps are folded into the N dimension: for one output row y
//     D[x'][(c,kx)] = sum_{ky,ci} in[y+ky-p][x'][ci] * w[c][ci][ky][kx]        (M = 128 input columns x',
//                                                                               N = 3*9 = 27 -> 32,
//                                                                               K = 9*64 = 576)
//     out[y][x][c]  = sum_kx D[x+kx-p][(c,kx)]                                  (diagonal sum, done in LDS)
// MFMA utilisation = 27/32 * 120/128 = 79 % instead of 9 %.  A block (8 waves: 4 column groups x 2 row halves)
// owns R = 8 output rows x 120 output columns: every staged (transformed, nearest-upsampled) input row feeds up to 8 output rows with 8
// different ky weight slices, all 9 slices stay resident in LDS, and with x2 upsampling each physical
// input row is staged once for its two logical rows.  Epilogue: bias, Tanh, MulConstant, VGG de-process.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int FOLD_R = 16;       // output rows per tile
constexpr int FOLD_M = 128;      // input columns per tile

struct FoldArgs {
    const float* in; const float* wfold; const float* bias;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    float* out_planar; float* out_raw;
    int IH, IW, IWp, ups, COUT, KH, KW, pad, OH, OW;
    int stages, relu1, relu2;
    float tanh_mul;
    int tiles_x, tiles_y;
    long long* dbg;      // optional in-kernel timeline (FAV_FOLD_DBG): per block tile count and the time spent in staging+MFMA loop / epilogue
};

// 16 output rows per tile (8 accumulators per wave): every staged input row feeds up to 9 output rows, so a taller tile stages
// (16 + 8) / 16 = 1.5 input rows per output row instead of 2, and the per-tile costs (the first row's latency, the ramp of
// half-used rows at the top and bottom, the diagonal-sum epilogue) are paid 495 instead of 990 times per 1280x720 frame.
// Persistent blocks: the nine ky weight slices (78 KB) are loaded into LDS once per block, not once per tile.
template <int CIN>
__global__ __launch_bounds__(512, 2) void conv_rowfold_kernel(const FoldArgs p)
{
    constexpr int NT = 512;                    // 8 waves: waves 0-3 own output rows 0-7, waves 4-7 rows 8-15 (same columns)
    constexpr int RW = FOLD_R / 2;             // output rows per wave
    constexpr int S = CIN + 4;                 // LDS row stride (floats): odd multiple of 16 B -> conflict-free b128
    constexpr int NV = CIN / 16;               // float4 per thread per staged row (4 threads per column)
    constexpr int KK = CIN / 8;                // fragment steps per row (8 k values each)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                          // [KH][32][S]
    float* aff = Bs + p.KH * 32 * S;           // [4][CIN]
    float* As = aff + 4 * CIN;                 // [2][FOLD_M][S]; the epilogue's D tile reuses it

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wcol = wave & 3, wrow = wave >> 2;
    const int XO = FOLD_M - (p.KW - 1);        // output columns per tile

    for (int i = t; i < CIN; i += NT) {
        aff[i] = p.stages >= 1 ? p.scale1[i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[i] : 0.f;
        aff[2 * CIN + i] = p.stages >= 2 ? p.scale2[i] : 1.f; aff[3 * CIN + i] = p.stages >= 2 ? p.shift2[i] : 0.f;
    }
    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float lo2 = (p.stages >= 2 && p.relu2) ? 0.f : -INFINITY;
    // all ky weight slices -> LDS, once per block (wfold is [KH][32][CIN], zero rows for n >= COUT*KW)
    for (int e = t; e < p.KH * 32 * (CIN / 4); e += NT) {
        const int row = e / (CIN / 4), c4 = e - row * (CIN / 4);
        *reinterpret_cast<v4f*>(Bs + row * S + c4 * 4) = *reinterpret_cast<const v4f*>(p.wfold + (size_t)row * CIN + c4 * 4);
    }
    // staging assignment: column xl = t>>2 of the tile, channel quarter (t&3)
    const int xl = t >> 2, ch0 = (t & 3) * (CIN / 4);
    const int frag = (lane & 31) * S + (lane >> 5) * 4;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float4 ra[NV];

    for (int tile = blockIdx.x; tile < p.tiles_x * p.tiles_y; tile += gridDim.x) {
        const int by = tile / p.tiles_x, bx = tile - by * p.tiles_x;
        const int ox0 = bx * XO, oy0 = by * FOLD_R;
        const int xs = ox0 - p.pad;                // first input column of the tile (may be negative)
        const int iy_lo = max(0, oy0 - p.pad), iy_hi = min(p.IH - 1, oy0 + FOLD_R - 1 + p.KH - 1 - p.pad);
        const int pr_lo = iy_lo >> p.ups, pr_hi = iy_hi >> p.ups;
        const int ix = xs + xl;
        const bool colv = ix >= 0 && ix < p.IW;
        const float colm = colv ? 1.f : 0.f;
        const int coloff = colv ? (ix >> p.ups) * CIN + ch0 : 0;

#define FOLD_LOAD(pr_)                                                                              \
        {                                                                                           \
            const float* src_ = p.in + (size_t)(pr_) * p.IWp * CIN + coloff;                        \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) ra[i] = *reinterpret_cast<const float4*>(src_ + 4 * i); \
        }
#define FOLD_STORE(buf_)                                                                            \
        {                                                                                           \
            float* dst_ = As + (buf_) * FOLD_M * S + xl * S + ch0;                                  \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                        \
                float4 v_ = affine4_lo(ra[i], aff + ch0 + 4 * i, aff + CIN + ch0 + 4 * i, lo1);     \
                v_ = affine4_lo(v_, aff + 2 * CIN + ch0 + 4 * i, aff + 3 * CIN + ch0 + 4 * i, lo2); \
                v_.x *= colm; v_.y *= colm; v_.z *= colm; v_.w *= colm;                             \
                *reinterpret_cast<float4*>(dst_ + 4 * i) = v_;                                      \
            }                                                                                       \
        }

        f32x16 acc[RW];
#pragma unroll
        for (int y = 0; y < RW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;

        FOLD_LOAD(pr_lo);
        __syncthreads();               // affine tables + weights visible; the previous tile's epilogue is done with the staging memory
        FOLD_STORE(0);
        __syncthreads();

        int cur = 0;
        for (int pr = pr_lo; pr <= pr_hi; ++pr) {
            const bool more = pr < pr_hi;
            if (more) FOLD_LOAD(pr + 1);
            const float* a_base = As + cur * FOLD_M * S + wcol * 32 * S + frag;
            const int iy_first = max(iy_lo, pr << p.ups), iy_last = min(iy_hi, ((pr + 1) << p.ups) - 1);
            for (int iy = iy_first; iy <= iy_last; ++iy) {
                const int kyb = iy - oy0 + p.pad - wrow * RW;      // ky for this wave's output row yy is kyb - yy
                // the row's A fragments are read once and serve every output row it feeds; per output row one wave-uniform test,
                // then a straight-line block of KK weight-fragment reads and 4 KK MFMAs (LDS latency hides inside it)
                float4 af[KK];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) af[kk] = *reinterpret_cast<const float4*>(a_base + kk * 8);
#pragma unroll
                for (int yy = 0; yy < RW; ++yy) {
                    const int ky = kyb - yy;
                    if (ky >= 0 && ky < p.KH) {            // wave-uniform
                        const float* b_base = Bs + ky * 32 * S + frag;
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk) {
                            const float4 bf = *reinterpret_cast<const float4*>(b_base + kk * 8);
                            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].x, bf.x, acc[yy], 0, 0, 0);
                            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].y, bf.y, acc[yy], 0, 0, 0);
                            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].z, bf.z, acc[yy], 0, 0, 0);
                            acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].w, bf.w, acc[yy], 0, 0, 0);
                        }
                    }
                }
            }
            if (more) FOLD_STORE(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
#undef FOLD_LOAD
#undef FOLD_STORE

        // ---- epilogue in four passes of 4 output rows (2 of each row half): D tiles -> LDS [4][128][33] in the staging area (the
        // weights stay resident), then the diagonal sum over kx
        float* D = As;
        const size_t MO = (size_t)p.OH * p.OW;
        const int per_row = XO * p.COUT;
        constexpr int PR = 2;              // rows of each half per pass
#pragma unroll
        for (int h = 0; h < RW / PR; ++h) {
#pragma unroll
            for (int y2 = 0; y2 < PR; ++y2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xr = wcol * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    D[((wrow * PR + y2) * FOLD_M + xr) * 33 + col] = acc[h * PR + y2][r];
                }
            __syncthreads();
            for (int e = t; e < 2 * PR * per_row; e += NT) {
                const int yl = e / per_row, rem = e - yl * per_row;            // D row: half = yl / PR, y2 = yl % PR
                const int c = rem / XO, xo = rem - c * XO;
                const int oy = oy0 + (yl / PR) * RW + h * PR + (yl % PR), ox = ox0 + xo;
                if (oy >= p.OH || ox >= p.OW) continue;
                float v = p.bias[c];
                const float* d = D + (yl * FOLD_M + xo) * 33 + c * p.KW;
                for (int kx = 0; kx < p.KW; ++kx) v += d[kx * 33 + kx];
                v = tanhf(v) * p.tanh_mul;                                              // models_video.lua:135-136
                const size_t o = (size_t)oy * p.OW + ox;
                if (p.out_raw) p.out_raw[(size_t)c * MO + o] = v;
                if (p.out_planar) {
                    const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
                    p.out_planar[(size_t)(2 - c) * MO + o] = (v + mean) / 255.f;          // preprocess.lua:66-71
                }
            }
            __syncthreads();
        }
    }
}

// The same layer when its input is a x2 nearest-upsampled tensor (U2 before c9s1-3, models_video.lua:129-133): the upsampled
// image holds every physical pixel four times, so three quarters of the products above are repeats.
//   * columns: D[x'][(c,kx)] is identical for the logical columns 2v and 2v+1 -- it is computed once per PHYSICAL column and the
//     diagonal sum reads it at (x + kx - p) >> 1: half the GEMM rows, no change to the weights;
//   * rows: the logical input rows 2r and 2r+1 are the same data and reach output row y through ky0 = 2r - y + p and ky0 + 1, so the
//     physical row is multiplied ONCE by the merged slice  Wm[ky0 + 1] = W[ky0] + W[ky0 + 1]  (W[-1] = W[KH] = 0; KH + 1 merged
//     slices, summed on the host in double): five merged slices per output row instead of nine.
// 3.6x fewer MFMAs than on the upsampled image, same operands otherwise (the merged weights are the only re-association).
// Tile = 16 output rows x 120 output columns = 64 physical input columns: waves = 2 column groups x 4 row groups, a row group
// owning the output rows g, g+4, g+8, g+12 -- a physical row feeds ten CONSECUTIVE output rows, so the interleave gives every
// wave two or three 32-MFMA blocks per staged row (consecutive rows per wave would leave half the waves idle at each barrier).
constexpr int FOLD2_M = 64;      // physical input columns per tile

// NH > 1 (input pitch CT = NH * CIN channels, e.g. 128 behind a c3s1-128 of a checkpoint with more filters): the merged slices of all
// channels do not fit the LDS next to the staging buffers, so a tile is computed in NH passes over its rows, one per block of CIN
// channels, into the same accumulators; the pass's slices (87 KB for CIN = 64) are reloaded from L2 at its start -- ~1 us against the
// ~50 us a pass takes
template <int CIN, int NH = 1>
__global__ __launch_bounds__(512, 2) void conv_rowfold_up2_kernel(const FoldArgs p)
{
    constexpr int NT = 512;
    constexpr int CT = CIN * NH;               // channel pitch of the input tensor
    constexpr int RW = FOLD_R / 4;             // output rows per wave (rows g + 4 yy)
    constexpr int S = CIN + 4;
    constexpr int NV = CIN / 32;               // float4 per thread per staged row (8 threads per column)
    constexpr int KK = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                          // [KH + 1][32][S] merged slices
    float* aff = Bs + (p.KH + 1) * 32 * S;     // [4][CIN]
    float* As = aff + 4 * CIN;                 // [2][FOLD2_M][S]; the epilogue's D tile [4][FOLD2_M][33] reuses it

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wcol = wave & 1, wrow = wave >> 1;
    const int XO = 2 * FOLD2_M - (p.KW - 1);   // output columns per tile (120)
    const int PH = p.IH >> 1, PW = p.IW >> 1;  // physical input size

    const float lo1 = (p.stages >= 1 && p.relu1) ? 0.f : -INFINITY;
    const float lo2 = (p.stages >= 2 && p.relu2) ? 0.f : -INFINITY;
    const float* wm = p.wfold + (size_t)p.KH * 32 * CT;            // merged slices follow the plain ones
    // the transform table and the merged slices of channels hoff .. hoff + CIN - 1 (NH == 1: once per block; else once per pass)
#define FOLD_RESIDENT(hoff_)                                                                        \
    {                                                                                               \
        for (int i = t; i < CIN; i += NT) {                                                         \
            aff[i] = p.stages >= 1 ? p.scale1[(hoff_) + i] : 1.f; aff[CIN + i] = p.stages >= 1 ? p.shift1[(hoff_) + i] : 0.f; \
            aff[2 * CIN + i] = p.stages >= 2 ? p.scale2[(hoff_) + i] : 1.f; aff[3 * CIN + i] = p.stages >= 2 ? p.shift2[(hoff_) + i] : 0.f; \
        }                                                                                           \
        for (int e = t; e < (p.KH + 1) * 32 * (CIN / 4); e += NT) {                                 \
            const int row = e / (CIN / 4), c4 = e - row * (CIN / 4);                                \
            *reinterpret_cast<v4f*>(Bs + row * S + c4 * 4) = *reinterpret_cast<const v4f*>(wm + (size_t)row * CT + (hoff_) + c4 * 4); \
        }                                                                                           \
    }
    if (NH == 1) FOLD_RESIDENT(0);
    const int xl = t >> 3, ch0 = (t & 7) * (CIN / 8);
    const int frag = (lane & 31) * S + (lane >> 5) * 4;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float4 ra[NV];

    for (int tile = blockIdx.x; tile < p.tiles_x * p.tiles_y; tile += gridDim.x) {
        const int by = tile / p.tiles_x, bx = tile - by * p.tiles_x;
        const int ox0 = bx * XO, oy0 = by * FOLD_R;
        const int pxs = (ox0 - p.pad) >> 1;        // first physical column of the tile (ox0 - pad is even; may be negative)
        const int iy_lo = max(0, oy0 - p.pad), iy_hi = min(p.IH - 1, oy0 + FOLD_R - 1 + p.KH - 1 - p.pad);
        const int pr_lo = iy_lo >> 1, pr_hi = min(iy_hi >> 1, PH - 1);
        const int pc = pxs + xl;
        const bool colv = pc >= 0 && pc < PW;
        const float colm = colv ? 1.f : 0.f;
        const int coloff = colv ? pc * CT + ch0 : 0;
        int hoff = 0;                              // first channel of the current pass

#define FOLD_LOAD(pr_)                                                                              \
        {                                                                                           \
            const float* src_ = p.in + (size_t)(pr_) * p.IWp * CT + coloff + hoff;                  \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) ra[i] = *reinterpret_cast<const float4*>(src_ + 4 * i); \
        }
#define FOLD_STORE(buf_)                                                                            \
        {                                                                                           \
            float* dst_ = As + (buf_) * FOLD2_M * S + xl * S + ch0;                                 \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                        \
                float4 v_ = affine4_lo(ra[i], aff + ch0 + 4 * i, aff + CIN + ch0 + 4 * i, lo1);     \
                v_ = affine4_lo(v_, aff + 2 * CIN + ch0 + 4 * i, aff + 3 * CIN + ch0 + 4 * i, lo2); \
                v_.x *= colm; v_.y *= colm; v_.z *= colm; v_.w *= colm;                             \
                *reinterpret_cast<float4*>(dst_ + 4 * i) = v_;                                      \
            }                                                                                       \
        }

        f32x16 acc[RW];
#pragma unroll
        for (int y = 0; y < RW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;

        const long long w0 = p.dbg ? wall_clock64() : 0;
        long long w1 = 0;
#pragma unroll 1
        for (int half = 0; half < NH; ++half) {
        hoff = half * CIN;
        FOLD_LOAD(pr_lo);
        __syncthreads();               // affine tables + weights visible; the previous tile's epilogue (the previous pass's last row) is done with the LDS
        if (NH > 1) { FOLD_RESIDENT(hoff); __syncthreads(); }
        FOLD_STORE(0);
        __syncthreads();
        if (half == 0) w1 = p.dbg ? wall_clock64() : 0;

        int cur = 0;
        for (int pr = pr_lo; pr <= pr_hi; ++pr) {
            const bool more = pr < pr_hi;
            if (more) FOLD_LOAD(pr + 1);
            const float* a_base = As + cur * FOLD2_M * S + wcol * 32 * S + frag;
            float4 af[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) af[kk] = *reinterpret_cast<const float4*>(a_base + kk * 8);
            const int msb = 2 * pr - (oy0 + wrow) + p.pad + 1;      // merged slice of this wave's output row yy: msb - 4 yy
#pragma unroll
            for (int yy = 0; yy < RW; ++yy) {
                const int ms = msb - 4 * yy;
                if (ms >= 0 && ms <= p.KH) {               // wave-uniform
                    const float* b_base = Bs + ms * 32 * S + frag;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const float4 bf = *reinterpret_cast<const float4*>(b_base + kk * 8);
                        acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].x, bf.x, acc[yy], 0, 0, 0);
                        acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].y, bf.y, acc[yy], 0, 0, 0);
                        acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].z, bf.z, acc[yy], 0, 0, 0);
                        acc[yy] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk].w, bf.w, acc[yy], 0, 0, 0);
                    }
                }
            }
            if (more) FOLD_STORE(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        }
#undef FOLD_LOAD
#undef FOLD_STORE
#undef FOLD_RESIDENT

        const long long w2 = p.dbg ? wall_clock64() : 0;
        // ---- epilogue in four passes (pass hh: output rows oy0 + g + 4 hh of the four row groups): D tiles -> LDS [4][64][33],
        // then the diagonal sum over kx with the logical -> physical column map
        float* D = As;
        const size_t MO = (size_t)p.OH * p.OW;
        const int per_row = XO * p.COUT;
        // an output (row group g, channel c, column xo) of a pass is the same for all four passes: its index arithmetic (two divisions
        // by run-time values), bias and mean are formed once per tile instead of once per output
        constexpr int NE = 3;                      // 4 * per_row = 1440 outputs per pass on 512 threads (COUT = 3, XO = 120)
        int eg[NE], ec[NE], exo[NE]; float ebias[NE], emean[NE]; const float* ed[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = t + NT * k;
            const bool ok = e < 4 * per_row;
            const int g = ok ? e / per_row : 0, rem = ok ? e - g * per_row : 0;
            const int c = rem / XO, xo = rem - c * XO;
            eg[k] = ok && ox0 + xo < p.OW ? g : -1; ec[k] = c; exo[k] = xo;
            ebias[k] = p.bias[c]; emean[k] = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
            ed[k] = D + g * FOLD2_M * 33 + c * p.KW;
        }
#pragma unroll
        for (int hh = 0; hh < RW; ++hh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = wcol * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                D[(wrow * FOLD2_M + xr) * 33 + col] = acc[hh][r];
            }
            __syncthreads();
            if (4 * per_row <= NE * NT) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const int oy = oy0 + eg[k] + 4 * hh;
                    if (eg[k] < 0 || oy >= p.OH) continue;
                    float v = ebias[k];
                    for (int kx = 0; kx < p.KW; ++kx) v += ed[k][((exo[k] + kx) >> 1) * 33 + kx];
                    v = tanhf(v) * p.tanh_mul;                                          // models_video.lua:135-136
                    const size_t o = (size_t)oy * p.OW + ox0 + exo[k];
                    if (p.out_raw) p.out_raw[(size_t)ec[k] * MO + o] = v;
                    if (p.out_planar) p.out_planar[(size_t)(2 - ec[k]) * MO + o] = (v + emean[k]) / 255.f;      // preprocess.lua:66-71
                }
            } else {
            for (int e = t; e < 4 * per_row; e += NT) {
                const int g = e / per_row, rem = e - g * per_row;
                const int c = rem / XO, xo = rem - c * XO;
                const int oy = oy0 + g + 4 * hh, ox = ox0 + xo;
                if (oy >= p.OH || ox >= p.OW) continue;
                float v = p.bias[c];
                const float* d = D + g * FOLD2_M * 33 + c * p.KW;
                for (int kx = 0; kx < p.KW; ++kx) v += d[((xo + kx) >> 1) * 33 + kx];
                v = tanhf(v) * p.tanh_mul;                                              // models_video.lua:135-136
                const size_t o = (size_t)oy * p.OW + ox;
                if (p.out_raw) p.out_raw[(size_t)c * MO + o] = v;
                if (p.out_planar) {
                    const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
                    p.out_planar[(size_t)(2 - c) * MO + o] = (v + mean) / 255.f;          // preprocess.lua:66-71
                }
            }
            }
            __syncthreads();
        }
        if (p.dbg && t == 0) {
            long long* d = p.dbg + blockIdx.x * 8;
            d[0] += 1; d[1] += w1 - w0; d[2] += w2 - w1; d[3] += wall_clock64() - w2; d[4] += pr_hi - pr_lo + 1;
        }
    }
}

template <int CIN, int NH = 1>
int launch_fold_up2_t(FoldArgs a, int reserve_cus, hipStream_t st)
{
    const int S = CIN + 4;
    const size_t wbytes = (size_t)((a.KH + 1) * 32 * S + 4 * CIN) * sizeof(float);
    size_t stage = (size_t)(2 * FOLD2_M * S) * sizeof(float);
    const size_t epi = (size_t)4 * FOLD2_M * 33 * sizeof(float);
    if (epi > stage) stage = epi;
    const size_t lds = wbytes + stage;
    if (lds > 160 * 1024) { set_error("row-folded conv: %zu bytes of LDS needed", lds); return FAV_EUNSUPPORTED; }
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rowfold_up2_kernel<CIN, NH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        cus[dv] = prop_cus;
    }
    const int XO = 2 * FOLD2_M - (a.KW - 1);
    a.tiles_x = (a.OW + XO - 1) / XO; a.tiles_y = (a.OH + FOLD_R - 1) / FOLD_R;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nres = std::max(1, cus[dv] - reserve_cus);
    static int dbg_n = diag_env("FAV_FOLD_DBG") ? atoi(diag_env("FAV_FOLD_DBG")) : 0;      // print the in-kernel timeline of the n-th launch
    const bool dbg = dbg_n > 0 && --dbg_n == 0;
    static long long* dbuf = nullptr;
    a.dbg = nullptr;
    if (dbg) { FAV_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 512 * 8 * 8)); FAV_HIP(hipMemsetAsync(dbuf, 0, 512 * 8 * 8, st)); a.dbg = dbuf; }
    hipLaunchKernelGGL((conv_rowfold_up2_kernel<CIN, NH>), dim3(tiles < nres ? tiles : nres), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_rowfold_up2_kernel");
    if (dbg) {
        std::vector<long long> hb((size_t)512 * 8);
        FAV_HIP(hipStreamSynchronize(st)); FAV_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
        double n = 0, a0 = 0, a1 = 0, a2 = 0, rows = 0;
        for (int b = 0; b < 512; ++b) { n += hb[b * 8]; a0 += hb[b * 8 + 1]; a1 += hb[b * 8 + 2]; a2 += hb[b * 8 + 3]; rows += hb[b * 8 + 4]; }
        if (n > 0) fprintf(stderr, "FOLDDBG tiles=%.0f  per tile: first row %.2f  loop %.2f (%.1f staged rows)  epilogue %.2f us\n", n, a0 / n * 0.01, a1 / n * 0.01, rows / n, a2 / n * 0.01);
    }
    return FAV_OK;
}

template <int CIN>
int launch_fold_t(FoldArgs a, int reserve_cus, hipStream_t st)
{
    const int S = CIN + 4;
    const size_t wbytes = (size_t)(a.KH * 32 * S + 4 * CIN) * sizeof(float);      // resident: weights + transform table
    size_t stage = (size_t)(2 * FOLD_M * S) * sizeof(float);
    const size_t epi = (size_t)4 * FOLD_M * 33 * sizeof(float);                   // D tile of one epilogue pass
    if (epi > stage) stage = epi;
    const size_t lds = wbytes + stage;
    if (lds > 160 * 1024) { set_error("row-folded conv: %zu bytes of LDS needed", lds); return FAV_EUNSUPPORTED; }
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rowfold_kernel<CIN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        cus[dv] = prop_cus;
    }
    const int XO = FOLD_M - (a.KW - 1);
    a.tiles_x = (a.OW + XO - 1) / XO; a.tiles_y = (a.OH + FOLD_R - 1) / FOLD_R;
    const int tiles = a.tiles_x * a.tiles_y;
    const int nres = std::max(1, cus[dv] - reserve_cus);       // persistent blocks: leave the side queues their CUs
    hipLaunchKernelGGL((conv_rowfold_kernel<CIN>), dim3(tiles < nres ? tiles : nres), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_rowfold_kernel");
    return FAV_OK;
}

}  // namespace

bool conv_fold_eligible(int cin_pitch, int cout, int k, int stride)
{
    return stride == 1 && cout * k <= 32 && k <= 9 && (cin_pitch == 16 || cin_pitch == 32 || cin_pitch == 64 || cin_pitch == 128 || cin_pitch == 256);
}
// 128 / 256 input channels (checkpoints with more filters, README.md:141): only the form for a x2-upsampled input exists (U2 + c9s1-3,
// every architecture string of the reference ends that way); anything else with that many channels takes the generic kernel
bool conv_fold_launchable(int cin_pitch, int k, int pad, int ups, int IH, int IW)
{
    static const bool no_up2 = diag_env("FAV_NO_FOLD_UP2") != nullptr;
    if (cin_pitch <= 64) return true;
    return ups == 1 && !no_up2 && (pad & 1) == 0 && (k & 1) == 1 && (IH & 1) == 0 && (IW & 1) == 0 && k + 1 <= 10;
}

int launch_conv_fold(const ConvLaunch& c, const float* wfold, hipStream_t st)
{
    FAV_REQUIRE(conv_fold_eligible(c.CIN, c.COUT, c.KW, c.stride) && c.KH == c.KW && c.final_mode, "row-folded conv: not eligible");
    FoldArgs a;
    a.dbg = nullptr;
    a.in = c.in; a.wfold = wfold; a.bias = c.bias;
    a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.scale2 = c.pre.scale2; a.shift2 = c.pre.shift2;
    a.stages = c.pre.stages; a.relu1 = c.pre.relu1; a.relu2 = c.pre.relu2;
    a.out_planar = c.out_planar; a.out_raw = c.out_raw_nchw;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.ups = c.ups; a.COUT = c.COUT; a.KH = c.KH; a.KW = c.KW; a.pad = c.pad;
    a.OH = c.OH; a.OW = c.OW; a.tanh_mul = c.tanh_mul;
    // x2 nearest-upsampled input: physical columns, merged ky slices (wfold carries them after the plain slices)
    static const bool no_up2 = diag_env("FAV_NO_FOLD_UP2") != nullptr;
    if (c.ups == 1 && !no_up2 && (c.pad & 1) == 0 && (c.KW & 1) == 1 && (c.IH & 1) == 0 && (c.IW & 1) == 0 && c.KH + 1 <= 10) {
        if (c.CIN == 256) return launch_fold_up2_t<64, 4>(a, c.reserve_cus, st);
        if (c.CIN == 128) return launch_fold_up2_t<64, 2>(a, c.reserve_cus, st);
        if (c.CIN == 64) return launch_fold_up2_t<64>(a, c.reserve_cus, st);
        if (c.CIN == 32) return launch_fold_up2_t<32>(a, c.reserve_cus, st);
    }
    FAV_REQUIRE(c.CIN <= 64, "row-folded conv: %d input channels are supported on a x2-upsampled input only", c.CIN);
    if (c.CIN == 64) return launch_fold_t<64>(a, c.reserve_cus, st);
    if (c.CIN == 32) return launch_fold_t<32>(a, c.reserve_cus, st);
    return launch_fold_t<16>(a, c.reserve_cus, st);
}

int launch_conv(const ConvLaunch& c, hipStream_t st)
{
    FAV_REQUIRE(c.CIN % 4 == 0 && c.CIN <= 1024, "conv: CIN=%d must be a multiple of 4 and <= 1024", c.CIN);
    FAV_REQUIRE(c.COUTp % 32 == 0 && c.Kpad % BK == 0, "conv: COUTp=%d / Kpad=%d not tile aligned", c.COUTp, c.Kpad);
    FAV_REQUIRE(c.Kpad >= c.KH * c.KW * c.CIN, "conv: Kpad too small");
    FAV_REQUIRE(c.ups == 0 || c.ups == 1, "conv: upsample factor must be 1 or 2");
    FAV_REQUIRE(!c.stuff || c.ups == 1, "conv: zero-stuffing needs the x2 index map");
    FAV_REQUIRE((c.CIN & (c.CIN - 1)) == 0, "conv: the channel pitch %d must be a power of two", c.CIN);
    FAV_REQUIRE(c.KH * c.KW < 4096, "conv: kernel too large");
    FAV_REQUIRE((long long)((c.IH >> c.ups) + 1) * c.IWp * c.CIN < (1ll << 31) && (long long)c.COUTp * c.Kpad < (1ll << 31),
                "conv: tensor too large for 32-bit element offsets");
    ConvArgs a;
    a.in = c.in; a.wgt = c.wgt; a.bias = c.bias;
    a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.scale2 = c.pre.scale2; a.shift2 = c.pre.shift2;
    a.stages = c.pre.stages; a.relu1 = c.pre.relu1; a.relu2 = c.pre.relu2;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials);
    a.out_planar = c.out_planar; a.out_raw = c.out_raw_nchw;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.ups = c.ups; a.stuff = c.stuff ? 1 : 0; a.CIN = c.CIN;
    a.COUT = c.COUT; a.COUTp = c.COUTp; a.KH = c.KH; a.KW = c.KW; a.stride = c.stride; a.pad = c.pad;
    a.Kpad = c.Kpad; a.OH = c.OH; a.OW = c.OW; a.final_mode = c.final_mode; a.tanh_mul = c.tanh_mul;
    a.cin_shift = __builtin_ctz((unsigned)c.CIN); a.kw_magic = (65536 + c.KW - 1) / c.KW;
    a.ntaps_magic = (65536 + c.KH * c.KW - 1) / (c.KH * c.KW);
    a.sk_ws = c.sk_ws; a.sk_flags = c.sk_flags; a.sk_epoch = c.sk_epoch; a.sk_err = c.sk_err; a.reserve_cus = c.reserve_cus;
    // stream-K when the tile count is within a few waves of the 512 resident blocks (imbalance matters there)
    const long long tiles = (long long)((c.OH * c.OW + BM - 1) / BM) * (c.COUTp / (c.COUTp % 128 == 0 ? 128 : (c.COUTp % 64 == 0 ? 64 : 32)));
    const bool sk = c.sk_ws != nullptr && c.sk_flags != nullptr && !c.no_sk && tiles >= SK_GRID / 2 && tiles <= 6 * SK_GRID && c.Kpad / BK >= 4;
    // 128-wide layers: stream-K with 4-wave blocks (64x64 per wave, two blocks per CU; measured 177.8 us against 179.8 us for the
    // 8-wave stream-K instance and 182.2 us for the data-parallel 8-wave instance on the residual layers), else 8 waves data-parallel
    if (c.COUTp % 128 == 0) return sk ? launch_conv_t<128, 2, 2, true>(a, st) : launch_conv_t<128, 4, 2>(a, st);
    if (c.COUTp % 64 == 0) return sk ? launch_conv_t<64, 2, 2, true>(a, st) : launch_conv_t<64, 2, 2>(a, st);
    return launch_conv_t<32, 4, 1>(a, st);
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm finalize: merge per-tile (mean, M2) in fp64 (Chan et al.), emit scale/shift.
// InstanceNormalization.lua:33-53: biased variance, eps inside the sqrt.
// ------------------------------------------------------------------------------------------------
namespace {

// One pass over the per-tile (mean, M2, count) partials in fp64:  mean = sum n_b mean_b / M,  var = (sum M2_b + sum n_b mean_b^2) / M
// - mean^2 (biased).  The cancellation in the last step costs (mean^2 / var) ulps of fp64 -- far below the fp32 result's own
// rounding -- and saves the second dependent sweep + block reduction of the textbook two-pass merge: this kernel is pure
// latency (16 launches per frame), not bandwidth.  (Measured and dropped, profiles/r02p_*: blocks of 16 channels x 64 rows with
// line-coalesced reads and four loads in flight per thread -- 6.2 us against 5.2 us for this form: the launch plus ONE round trip
// to memory for data another XCD's L2 has just written back is what the 5 us are made of, not the read pattern.)
__global__ __launch_bounds__(256) void in_finalize_kernel(const float2* partials, const int* counts, int mblocks, int M, int bp,
                                                          int Cpitch, const float* gamma, const float* beta,
                                                          float eps, float* scale, float* shift)
{
    __shared__ double sh[8];
    const int c = blockIdx.x, t = threadIdx.x;
    double s1 = 0, s2 = 0;
    float gq = 1.f, bq = 0.f;
    if (t == 0) { gq = gamma ? gamma[c] : 1.f; bq = beta ? beta[c] : 0.f; }      // requested before the sweep, used after it
    // four independent rows per thread in flight (one batch covers 1024 partial rows: a single round trip to memory for every layer
    // of the 1280x720 network; a rolled loop waits for each row before it asks for the next)
    for (int b0 = t; b0 < mblocks; b0 += 1024) {
        float2 pr[4]; int nb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + 256 * k, bc = min(b, mblocks - 1);
            pr[k] = partials[(size_t)bc * Cpitch + c];
            nb[k] = b < mblocks ? (counts ? counts[bc] : min(bp, M - bc * bp)) : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double n = (double)nb[k], mu = (double)pr[k].x;
            s1 += n * mu;
            s2 += (nb[k] ? (double)pr[k].y : 0.0) + n * mu * mu;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((t & 63) == 0) { sh[2 * (t >> 6)] = s1; sh[2 * (t >> 6) + 1] = s2; }
    __syncthreads();
    if (t == 0) {
        const double a = ((sh[0] + sh[2]) + sh[4]) + sh[6], q = ((sh[1] + sh[3]) + sh[5]) + sh[7];
        const double mean = a / (double)M;
        double var = q / (double)M - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double g = (double)gq, bt = (double)bq;
        const double sc = g / sqrt(var + (double)eps);
        scale[c] = (float)sc;
        shift[c] = (float)(bt - mean * sc);
    }
}

__device__ __forceinline__ float4 apply_affine_g(float4 v, const Affine& a, int c)
{
    if (a.stages >= 1) {
        v = affine4(v, a.scale1 + c, a.shift1 + c, a.relu1);
        if (a.stages >= 2) v = affine4(v, a.scale2 + c, a.shift2 + c, a.relu2);
    }
    return v;
}

// statistics of t(x) over [M][C]: one block per 128 pixels.  NPT > 0: the block's elements stay in registers between the mean
// and the M2 sweep (NPT = 128 / (256 / (C / 4)) float4 per thread: 8 for C = 64, 16 for C = 128) -- one read of the tensor
// instead of two; NPT = 0: any channel count, the tile is read twice.
template <int NPT>
__global__ __launch_bounds__(256) void stats_kernel(const float* x, int M, int C, const Affine a, float2* partials)
{
    __shared__ float red[1024];
    __shared__ float mean_s[1024];
    const int t = threadIdx.x;
    const int groups = C >> 2;                 // float4 groups per pixel
    const int nl = 256 / groups;               // pixel lanes
    const int g = t % groups, pl = t / groups;
    const int m0 = blockIdx.x * 128;
    const int cnt = min(128, M - m0);
    const bool active = pl < nl;
    float4 s = make_float4(0, 0, 0, 0);
    float4 keep[NPT > 0 ? NPT : 1];
    if (NPT > 0) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int pix = pl + i * nl;
            float4 v = make_float4(0, 0, 0, 0);
            if (pix < cnt) { v = *reinterpret_cast<const float4*>(x + (size_t)(m0 + pix) * C + 4 * g); v = apply_affine_g(v, a, 4 * g); }
            keep[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) { s.x += keep[i].x; s.y += keep[i].y; s.z += keep[i].z; s.w += keep[i].w; }      // (elements past cnt are zeros)
    } else if (active)
        for (int pix = pl; pix < cnt; pix += nl) {
            float4 v = *reinterpret_cast<const float4*>(x + (size_t)(m0 + pix) * C + 4 * g);
            v = apply_affine_g(v, a, 4 * g);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    if (active) *reinterpret_cast<float4*>(red + pl * C + 4 * g) = s;
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float r = 0;
        for (int i = 0; i < nl; ++i) r += red[i * C + c];
        mean_s[c] = r / (float)cnt;
    }
    __syncthreads();
    float4 q = make_float4(0, 0, 0, 0);
    if (active) {
        const float4 mu = *reinterpret_cast<const float4*>(mean_s + 4 * g);
        if (NPT > 0) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const float4 v = keep[i];
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                if (pl + i * nl < cnt) { q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w); }
            }
        } else
            for (int pix = pl; pix < cnt; pix += nl) {
                float4 v = *reinterpret_cast<const float4*>(x + (size_t)(m0 + pix) * C + 4 * g);
                v = apply_affine_g(v, a, 4 * g);
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
            }
        *reinterpret_cast<float4*>(red + pl * C + 4 * g) = q;
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float r = 0;
        for (int i = 0; i < nl; ++i) r += red[i * C + c];
        partials[(size_t)blockIdx.x * C + c] = make_float2(mean_s[c], r);
    }
}

// residual join: nn.CAddTable of (IN(conv_b) , ShaveImage(skip))  -- models_video.lua:41-53.
// res_add_stats_kernel: the join feeds an InstanceNorm (directly or through a nearest upsample, which leaves mean and biased
// variance unchanged: the R128 -> U2 -> IN tail of models_video.lua:94-98): one block = one row segment of up to 128 pixels, and
// the same pass yields that norm's per-segment (mean, M2, count) partials instead of a second read-only pass over the joined tensor.
// ACC (round 5): the branch's InstanceNorm arrives as accumulators (Affine::acc1, fav_internal.h) -- every block forms scale / shift
// for all C channels in its prologue (the arithmetic of in_finalize_kernel on exact integer sums) and keeps them in LDS; the launch
// uses at most 1024 blocks then (32 KB of accumulator words per block)
template <bool ACC>
__global__ __launch_bounds__(256) void res_add_kernel(const float* y, const float* scale, const float* shift,
                                                      const float* skip, int SW, int shave, const Affine sa,
                                                      int OH, int OW, int C, float* z, const Affine br)
{
    __shared__ float ss[ACC ? 2048 : 4];       // [scale C | shift C], C <= 1024
    if (ACC) {
        for (int i = threadIdx.x; i < C; i += 256) {
            long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
            for (int cp = 0; cp < STAT_COPIES; ++cp) {
                const longlong2* a = reinterpret_cast<const longlong2*>(br.acc1 + ((size_t)cp * C + i) * 4);
                const longlong2 lo = a[0], hi = a[1];
                w0 += lo.x; w1 += lo.y; w2 += hi.x; w3 += hi.y;
            }
            const double s1 = ((double)w1 * 4294967296.0 + (double)w0) * (1.0 / 1099511627776.0);
            const double s2 = ((double)w3 * 4294967296.0 + (double)w2) * (1.0 / 1099511627776.0);
            const double mean = s1 / (double)br.count1;
            double var = s2 / (double)br.count1 - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const double sc = stat_acc_poisoned(w1, w3) ? (double)NAN : (double)br.gamma1[i] / sqrt(var + (double)br.eps1);
            ss[i] = (float)sc; ss[C + i] = (float)((double)br.beta1[i] - mean * sc);
            if (blockIdx.x == 0) {      // the other parity's accumulators: zero for the next frame
#pragma unroll
                for (int cp = 0; cp < STAT_COPIES; ++cp) {
                    longlong2* zz = reinterpret_cast<longlong2*>(br.acc1_zero + ((size_t)cp * C + i) * 4);
                    zz[0] = longlong2{0, 0}; zz[1] = longlong2{0, 0};
                }
            }
        }
        __syncthreads();
        scale = ss; shift = ss + C;
    }
    const int groups = C >> 2;
    const size_t total = (size_t)OH * OW * groups;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int g = (int)(idx % groups);
        const size_t pix = idx / groups;
        const int oy = (int)(pix / OW), ox = (int)(pix - (size_t)oy * OW);
        float4 v = *reinterpret_cast<const float4*>(y + pix * C + 4 * g);
        v = affine4(v, scale + 4 * g, shift + 4 * g, 0);
        float4 k = *reinterpret_cast<const float4*>(skip + ((size_t)(oy + shave) * SW + ox + shave) * C + 4 * g);
        k = apply_affine_g(k, sa, 4 * g);
        v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
        *reinterpret_cast<float4*>(z + pix * C + 4 * g) = v;
    }
}

template <int NPT>      // > 0: the joined values of the segment stay in registers for the M2 sweep (128 / (256 / (C / 4)) float4 per thread)
__global__ __launch_bounds__(256) void res_add_stats_kernel(const float* y, const float* scale, const float* shift,
                                                            const float* skip, int SW, int shave, const Affine sa,
                                                            int OW, int C, float* z, float2* partials, int* counts)
{
    __shared__ float red[1024];
    __shared__ float mean_s[1024];
    const int t = threadIdx.x;
    const int groups = C >> 2, nl = 256 / groups;         // float4 groups per pixel, pixel lanes
    const int g = t % groups, pl = t / groups;
    const int segs = (OW + 127) / 128;
    const int oy = blockIdx.x / segs, x0 = (blockIdx.x - oy * segs) * 128;
    const int cnt = min(128, OW - x0);
    const bool active = pl < nl;
    const float* yr = y + ((size_t)oy * OW + x0) * C + 4 * g;
    const float* kr = skip + ((size_t)(oy + shave) * SW + x0 + shave) * C + 4 * g;
    float* zr = z + ((size_t)oy * OW + x0) * C + 4 * g;
    float4 sm = make_float4(0, 0, 0, 0);
    float4 keep[NPT > 0 ? NPT : 1];
    if (NPT > 0) {
        float4 kk[NPT > 0 ? NPT : 1];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int px = pl + i * nl;
            keep[i] = make_float4(0, 0, 0, 0); kk[i] = make_float4(0, 0, 0, 0);
            if (px < cnt) { keep[i] = *reinterpret_cast<const float4*>(yr + (size_t)px * C); kk[i] = *reinterpret_cast<const float4*>(kr + (size_t)px * C); }
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int px = pl + i * nl;
            float4 v = affine4(keep[i], scale + 4 * g, shift + 4 * g, 0);
            const float4 k = apply_affine_g(kk[i], sa, 4 * g);
            v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
            if (px < cnt) { *reinterpret_cast<float4*>(zr + (size_t)px * C) = v; sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w; }
            keep[i] = v;
        }
    } else if (active)
        for (int px = pl; px < cnt; px += nl) {
            float4 v = *reinterpret_cast<const float4*>(yr + (size_t)px * C);
            v = affine4(v, scale + 4 * g, shift + 4 * g, 0);
            float4 k = *reinterpret_cast<const float4*>(kr + (size_t)px * C);
            k = apply_affine_g(k, sa, 4 * g);
            v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
            *reinterpret_cast<float4*>(zr + (size_t)px * C) = v;
            sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
        }
    if (active) *reinterpret_cast<float4*>(red + pl * C + 4 * g) = sm;
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float r = 0;
        for (int i = 0; i < nl; ++i) r += red[i * C + c];
        mean_s[c] = r / (float)cnt;
    }
    __syncthreads();
    float4 q = make_float4(0, 0, 0, 0);
    if (active) {
        const float4 mu = *reinterpret_cast<const float4*>(mean_s + 4 * g);
        if (NPT > 0) {
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const float4 v = keep[i];
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                if (pl + i * nl < cnt) { q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w); }
            }
        } else
        for (int px = pl; px < cnt; px += nl) {
            const float4 v = *reinterpret_cast<const float4*>(zr + (size_t)px * C);      // this thread's own stores
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
        *reinterpret_cast<float4*>(red + pl * C + 4 * g) = q;
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float r = 0;
        for (int i = 0; i < nl; ++i) r += red[i * C + c];
        partials[(size_t)blockIdx.x * C + c] = make_float2(mean_s[c], r);
    }
    if (t == 0) counts[blockIdx.x] = cnt;
}

__device__ __forceinline__ int reflect(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// NCHW -> reflection-padded NHWC with channel padding (nn.SpatialReflectionPadding, train_video.lua:319-325)
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* in, int C, int H, int W, int pad, int Cp,
                                                               float* out)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const size_t total = (size_t)Hp * Wp * Cp;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % Cp);
        const size_t pix = idx / Cp;
        const int y = (int)(pix / Wp), x = (int)(pix - (size_t)y * Wp);
        float v = 0.f;
        if (c < C) v = in[((size_t)c * H + reflect(y - pad, H)) * W + reflect(x - pad, W)];
        out[idx] = v;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* in, int M, int C, const Affine a, float* out)
{
    const size_t total = (size_t)M * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx / M);
        const size_t m = idx - (size_t)c * M;
        float v = in[m * C + c];
        if (a.stages >= 1) {
            v = fmaf(v, a.scale1[c], a.shift1[c]); if (a.relu1) v = fmaxf(v, 0.f);
            if (a.stages >= 2) { v = fmaf(v, a.scale2[c], a.shift2[c]); if (a.relu2) v = fmaxf(v, 0.f); }
        }
        out[idx] = v;
    }
}

inline int grid_for(size_t total) { size_t b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b ? b : 1)); }

}  // namespace

int launch_in_finalize(const float* partials, const int* counts, int mblocks, int M, int block_pixels, int C, int Cpitch,
                       const float* gamma, const float* beta, float eps, float* scale, float* shift, hipStream_t st)
{
    hipLaunchKernelGGL(in_finalize_kernel, dim3(C), dim3(256), 0, st, reinterpret_cast<const float2*>(partials), counts,
                       mblocks, M, block_pixels, Cpitch, gamma, beta, eps, scale, shift);
    FAV_LAUNCH_CHECK("in_finalize_kernel");
    return FAV_OK;
}

int launch_stats(const float* x, int M, int C, const Affine& t, float* partials, hipStream_t st)
{
    FAV_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "stats: unsupported channel count %d", C);
    const dim3 grid((M + 127) / 128);
    float2* pp = reinterpret_cast<float2*>(partials);
    if (C == 64) hipLaunchKernelGGL(stats_kernel<8>, grid, dim3(256), 0, st, x, M, C, t, pp);
    else if (C == 128) hipLaunchKernelGGL(stats_kernel<16>, grid, dim3(256), 0, st, x, M, C, t, pp);
    else hipLaunchKernelGGL(stats_kernel<0>, grid, dim3(256), 0, st, x, M, C, t, pp);
    FAV_LAUNCH_CHECK("stats_kernel");
    return FAV_OK;
}

int launch_res_add(const float* y, const float* scale, const float* shift, const float* skip, int SH, int SW,
                   int shave, const Affine& skip_t, int C, float* z, float* partials, int* counts, hipStream_t st, int skip_pitch, const Affine* branch_acc)
{
    const int OH = SH - 2 * shave, OW = SW - 2 * shave;
    if (skip_pitch > 0) SW = skip_pitch;              // the kernels use SW as the skip's row pitch only
    FAV_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && OH > 0 && OW > 0, "res_add: bad shape (C=%d)", C);
    FAV_REQUIRE(!(partials && branch_acc && branch_acc->acc1), "res_add: the statistics-taking join does not take an accumulator-form InstanceNorm");
    if (partials) {
        const dim3 grid(res_add_stat_blocks(OH, OW));
        float2* pp = reinterpret_cast<float2*>(partials);
        if (C == 128) hipLaunchKernelGGL(res_add_stats_kernel<16>, grid, dim3(256), 0, st, y, scale, shift, skip, SW, shave, skip_t, OW, C, z, pp, counts);
        else if (C == 64) hipLaunchKernelGGL(res_add_stats_kernel<8>, grid, dim3(256), 0, st, y, scale, shift, skip, SW, shave, skip_t, OW, C, z, pp, counts);
        else hipLaunchKernelGGL(res_add_stats_kernel<0>, grid, dim3(256), 0, st, y, scale, shift, skip, SW, shave, skip_t, OW, C, z, pp, counts);
    }
    else if (branch_acc != nullptr && branch_acc->acc1 != nullptr)
        hipLaunchKernelGGL(res_add_kernel<true>, dim3(std::min(1024, grid_for((size_t)OH * OW * (C / 4)))), dim3(256), 0, st, y, scale, shift, skip, SW, shave, skip_t,
                           OH, OW, C, z, *branch_acc);
    else
        hipLaunchKernelGGL(res_add_kernel<false>, dim3(grid_for((size_t)OH * OW * (C / 4))), dim3(256), 0, st, y, scale, shift, skip, SW, shave, skip_t,
                           OH, OW, C, z, Affine());
    FAV_LAUNCH_CHECK("res_add_kernel");
    return FAV_OK;
}

int launch_nchw_to_nhwc_pad(const float* in, int C, int H, int W, int pad, int Cp, float* out, hipStream_t st)
{
    FAV_REQUIRE(pad < H && pad < W, "reflection pad %d must be smaller than the image (%dx%d)", pad, W, H);
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for((size_t)(H + 2 * pad) * (W + 2 * pad) * Cp)), dim3(256), 0,
                       st, in, C, H, W, pad, Cp, out);
    FAV_LAUNCH_CHECK("nchw_to_nhwc_pad_kernel");
    return FAV_OK;
}

int launch_nhwc_to_nchw(const float* in, int M, int C, const Affine& t, float* out, hipStream_t st)
{
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)M * C)), dim3(256), 0, st, in, M, C, t, out);
    FAV_LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return FAV_OK;
}

}  // namespace fav
