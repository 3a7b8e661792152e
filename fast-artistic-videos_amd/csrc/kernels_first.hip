// kernels_first.hip -- the first layer (c9s1-32: 9x9, 7 real input channels of the NHWC8 network input -> 32 channels, full
// resolution; models_video.lua:57-80; 3 channels for first-frame image models) with 1-D minimal filtering along x.
//
// The direct form (conv_c8d_kernel, kernels_conv.hip) is a dense-K implicit GEMM at 0.78 of the fp32 MFMA peak: K = 7 x 81,
// N = 32 -- too few output channels for a 2-D Winograd transform to pay (its vector-ALU work is per input element).  Along ONE
// axis it does: the 9 taps of a filter row are three blocks of three, each block a 3-tap correlation computed as F(2,3) -- 4
// multiplies per output pair instead of 6 (first_pack.h) -- so a tile needs 4 x 95 instead of 2 x 287 matrix instructions
// per 64 pixels and 32 channels, for ONE extra vector-ALU instruction per MFMA (the input difference / sum), and the output
// transform stays inside a lane (all four positions of a tile are accumulators of the same wave).
//
// Block = 8 waves, persistent, tile = 8 rows x 64 columns: wave = one output row = 32 tiles of two pixels.  LDS: the four
// transformed weight sets [position][pair][half][32] (97 KB, resident) + ONE halo buffer, 16 x 72 pixels per channel with every row
// stored as [even columns | odd columns] (32 KB): tile t reads columns 2t + e, i.e. consecutive words of one half -- conflict-free
// ds_read_b32, every address = one of five per-lane bases (the pairing types of first_pack.h) + an immediate.  The next tile's
// halo travels through registers while this one is computed.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "fav_internal.h"
#include "first_pack.h"
#include "first2d_pack.h"

namespace fav {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

constexpr int F_TH = 8, F_TW = 64;                 // output tile
constexpr int F_HR = F_TH + 8, F_HC = F_TW + 8;    // halo 16 x 72
constexpr int F_HALF = F_HC / 2;                   // 36 words: the even (then the odd) columns of a halo row
constexpr int F_RP = F_HC;                         // 72 words per halo row: [even columns | odd columns] -- the ten reads of a group sit within 42 words
constexpr int F_CPL = F_HR * F_RP;                 // 1152 words per channel
constexpr int F_HP = F_HR * F_HC;                  // 1152 halo pixels
constexpr int F_NH = (F_HP + 511) / 512;           // 3 halo pixels per thread

struct FirstArgs {
    const float* in; const float* wpk; const float* bias;
    float* out; float2* partials; int* counts;
    int IH, IW, IWp, COUT, pad, OH, OW, tiles_x, tiles_y;
    // conv_first2d_kernel<CR, true> only (more than 32 filters): the output channels are computed in `groups` groups of 32; a block
    // keeps ONE group's transformed weights in LDS and walks the tiles (gridDim.x % groups == 0); COUTP = the channel pitch of partials;
    // wpk holds one packed block (conv_first2d_pack) per group
    int COUTP, groups;
    long long* dbg = nullptr;       // make DIAG=1 + FAV_FIRST_DBG=n: in-kernel timeline of the n-th launch (accumulated wall-clock ticks per phase and block)
};

__device__ __forceinline__ float2 merge_rows(const float2* st, const int* wn, int c, int* n_out)
{
    int n = 0; float s = 0.f;
    for (int w = 0; w < 8; ++w) { n += wn[w]; s += (float)wn[w] * st[w * 32 + c].x; }
    const float mean = n ? s / (float)n : 0.f;
    float m2 = 0.f;
    for (int w = 0; w < 8; ++w) { const float d = st[w * 32 + c].x - mean; m2 += st[w * 32 + c].y + (float)wn[w] * d * d; }
    *n_out = n;
    return make_float2(mean, m2);
}

constexpr int f_off(int e) { return (e & 1) * F_HALF + (e >> 1); }       // word offset of halo column 2t + e relative to tile t's base

template <int CR>
__global__ __launch_bounds__(512, 2) void conv_first_kernel(const FirstArgs p)
{
    constexpr int NCC = (CR / 2) * 27, NJ = NCC + 14, LAST = CR - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ws = smem;                       // [4][NJ][2][32]
    float* const Hs = Ws + 4 * NJ * 64;           // [CR][2 planes][16][36]
    float* const red = Hs + CR * F_CPL;           // [8][32] float2 + [8] int
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (int e = t; e < 4 * NJ * 16; e += 512) *reinterpret_cast<v4f*>(Ws + e * 4) = *reinterpret_cast<const v4f*>(p.wpk + e * 4);

    const int ntiles = p.tiles_x * p.tiles_y;
    float4 hlo[F_NH], hhi[F_NH];
#define FL_LOAD_HALO(tile_)                                                                         \
    {                                                                                               \
        const int ty_ = (tile_) / p.tiles_x, tx_ = (tile_) - ty_ * p.tiles_x;                       \
        _Pragma("unroll") for (int i = 0; i < F_NH; ++i) {                                          \
            const int pix_ = t + 512 * i, hy_ = pix_ / F_HC, hx_ = pix_ - hy_ * F_HC;               \
            const int iy_ = ty_ * F_TH - p.pad + hy_, ix_ = tx_ * F_TW - p.pad + hx_;               \
            const bool v_ = (pix_ < F_HP) & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW); \
            const int off_ = v_ ? (iy_ * p.IWp + ix_) * 8 : 0;                                      \
            const float4 a_ = *reinterpret_cast<const float4*>(p.in + off_);                        \
            const float4 b_ = CR > 4 ? *reinterpret_cast<const float4*>(p.in + off_ + 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
            hlo[i] = v_ ? a_ : make_float4(0.f, 0.f, 0.f, 0.f);                                     \
            hhi[i] = v_ ? b_ : make_float4(0.f, 0.f, 0.f, 0.f);                                     \
        }                                                                                           \
    }
#define FL_STORE_HALO()                                                                             \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < F_NH; ++i) {                                          \
            const int pix_ = t + 512 * i;                                                           \
            if (pix_ < F_HP) {                                                                      \
                const int hy_ = pix_ / F_HC, hx_ = pix_ - hy_ * F_HC;                               \
                float* d_ = Hs + hy_ * F_RP + (hx_ & 1) * F_HALF + (hx_ >> 1);                       \
                const float c_[8] = {hlo[i].x, hlo[i].y, hlo[i].z, hlo[i].w, hhi[i].x, hhi[i].y, hhi[i].z, hhi[i].w}; \
                _Pragma("unroll") for (int c = 0; c < CR; ++c) d_[c * F_CPL] = c_[c];               \
            }                                                                                       \
        }                                                                                           \
    }

    int tile = blockIdx.x;
    if (tile < ntiles) FL_LOAD_HALO(tile);
    FL_STORE_HALO();
    __syncthreads();

    const int m = lane & 31, half = lane >> 5;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    // per-lane bases (tile m of output row `wave`); the second half-wave's operand of a pair sits at a constant offset
    const float* const a_one = Hs + wave * F_RP + m;
    const float* const a_cc = a_one + half * F_CPL;                      // type 0: next channel
    const float* const a_ky = a_one + half * F_RP;                       // type 1: next halo row
    const float* const a_be = a_one + half * (F_HALF + 1);               // type 2, even column offsets: + 3 columns = odd half, + 1
    const float* const a_bo = a_one + half * (2 - F_HALF);               // type 2, odd column offsets: + 3 columns = even half, + 2
    const float* const b_lo = Ws + half * 32 + m;                        // positions 0, 1
    int hi_words = 2 * NJ * 64;                                          // positions 2, 3 (their offsets would not fit a DS immediate)
    asm volatile("" : "+v"(hi_words));                                   // (an opaque OFFSET keeps b_hi a register of its own AND an LDS pointer;
    const float* const b_hi = b_lo + hi_words;                           //  laundering the pointer itself turns the reads into flat loads)

    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) FL_LOAD_HALO(nxt);
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

        // A "group" = the three blocks b of one (pair, filter row): ten reads (halo columns 2t + e, e = 0..9; block b uses e = 3b .. 3b+3)
        // + twelve weight reads -> twelve operands -> twelve MFMAs.  The reads of group g + 1 are issued before the MFMAs of group
        // g (two register sets, pinned with scheduling fences): LDS latency never sits in front of a matrix instruction.
        float R[2][10], B[2][12];
#define FL_READ(set_, abase_, goff_, j0_)                                                           \
        {   int go_ = (goff_); asm volatile("" : "+s"(go_));    /* opaque scalar: ONE address add per group, then 8-bit read2 offsets */ \
            const float* gb_ = (abase_) + go_;                                                      \
            _Pragma("unroll") for (int e = 0; e < 10; ++e) R[set_][e] = gb_[f_off(e)];              \
            _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                         \
                B[set_][4 * b + 0] = b_lo[((0 * NJ) + (j0_) + b) * 64]; B[set_][4 * b + 1] = b_lo[((1 * NJ) + (j0_) + b) * 64]; \
                B[set_][4 * b + 2] = b_hi[((0 * NJ) + (j0_) + b) * 64]; B[set_][4 * b + 3] = b_hi[((1 * NJ) + (j0_) + b) * 64]; } }
#define FL_COMP(set_)                                                                               \
        {   float V[12];                                                                            \
            _Pragma("unroll") for (int b = 0; b < 3; ++b) {                                         \
                V[4 * b + 0] = R[set_][3 * b] - R[set_][3 * b + 2]; V[4 * b + 1] = R[set_][3 * b + 1] + R[set_][3 * b + 2]; \
                V[4 * b + 2] = R[set_][3 * b + 2] - R[set_][3 * b + 1]; V[4 * b + 3] = R[set_][3 * b + 1] - R[set_][3 * b + 3]; } \
            _Pragma("unroll") for (int b = 0; b < 3; ++b)                                           \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                       \
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[4 * b + q], B[set_][4 * b + q], acc[q], 0, 0, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0); __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); }
#define FL_FENCE() __builtin_amdgcn_sched_barrier(0)
        constexpr int NG = (CR / 2) * 9 + 4;           // groups: (pair, ky) of type 0, then the row pairs q of type 1
        // base pointer and first pair index of group g
#define FL_GBASE(g_) ((g_) < (CR / 2) * 9 ? a_cc : a_ky)
#define FL_GOFF(g_) ((g_) < (CR / 2) * 9 ? 2 * ((g_) / 9) * F_CPL + ((g_) % 9) * F_RP : LAST * F_CPL + 2 * ((g_) - (CR / 2) * 9) * F_RP)
        FL_READ(0, FL_GBASE(0), FL_GOFF(0), 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) { FL_READ((g + 1) & 1, FL_GBASE(g + 1), FL_GOFF(g + 1), 3 * (g + 1)); }
            FL_FENCE(); FL_COMP(g & 1); FL_FENCE();
        }
        {
            // type 2 (ky = 8, blocks 0 | 1) and type 3 (block 2 alone): one combo each
            const float* ae = a_be + LAST * F_CPL + 8 * F_RP;
            const float* ao = a_bo + LAST * F_CPL + 8 * F_RP;
            const float* a1 = a_one + LAST * F_CPL + 8 * F_RP;
            const float r0 = ae[f_off(0)], r1 = ao[f_off(1)], r2 = ae[f_off(2)], r3 = ao[f_off(3)];
            const float s0 = a1[f_off(6)], s1 = a1[f_off(7)], s2 = a1[f_off(8)], s3 = a1[f_off(9)];
            constexpr int J2 = NCC + 12, J3 = NCC + 13;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0 - r2, b_lo[((0 * NJ) + J2) * 64], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1 + r2, b_lo[((1 * NJ) + J2) * 64], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(r2 - r1, b_hi[((0 * NJ) + J2) * 64], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1 - r3, b_hi[((1 * NJ) + J2) * 64], acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0 - s2, b_lo[((0 * NJ) + J3) * 64], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1 + s2, b_lo[((1 * NJ) + J3) * 64], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(s2 - s1, b_hi[((0 * NJ) + J3) * 64], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(s1 - s3, b_hi[((1 * NJ) + J3) * 64], acc[3], 0, 0, 0);
        }
#undef FL_READ
#undef FL_COMP
#undef FL_FENCE
#undef FL_GBASE
#undef FL_GOFF
        __syncthreads();                    // every wave is done with the halo
        if (nxt < ntiles) FL_STORE_HALO();

        // ---- output transform + epilogue: tile mi of row `wave` -> output columns 2 mi, 2 mi + 1; MFMA columns = channels
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int oy = ty * F_TH + wave, ox0 = tx * F_TW;
        const float bv = p.bias[col];
        float sm = 0.f;
        float y0[16], y1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mi = (r & 3) + 8 * (r >> 2) + rbase;
            y0[r] = (acc[0][r] + acc[1][r]) + acc[2][r] + bv;
            y1[r] = (acc[1][r] - acc[2][r]) - acc[3][r] + bv;
            const int ox = ox0 + 2 * mi;
            if (oy < p.OH && ox < p.OW) {
                if (col < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + col] = y0[r];
                sm += y0[r];
            }
            if (oy < p.OH && ox + 1 < p.OW) {
                if (col < p.COUT) p.out[((size_t)oy * p.OW + ox + 1) * p.COUT + col] = y1[r];
                sm += y1[r];
            }
        }
        if (p.partials != nullptr) {
            float2* st = reinterpret_cast<float2*>(red);          // [8 waves][32]
            int* wn = reinterpret_cast<int*>(red + 8 * 64);         // [8]
            const int nw = oy < p.OH ? min(F_TW, p.OW - ox0) : 0;   // valid pixels of this wave's row
            sm += __shfl_xor(sm, 32);
            const float mu = nw ? sm / (float)nw : 0.f;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + 2 * ((r & 3) + 8 * (r >> 2) + rbase);
                const float d0 = y0[r] - mu, d1 = y1[r] - mu;
                if (oy < p.OH && ox < p.OW) q = fmaf(d0, d0, q);
                if (oy < p.OH && ox + 1 < p.OW) q = fmaf(d1, d1, q);
            }
            q += __shfl_xor(q, 32);
            if (lane < 32) st[wave * 32 + lane] = make_float2(mu, q);
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < 32) {
                int n;
                p.partials[(size_t)tile * 32 + t] = merge_rows(st, wn, t, &n);
                if (t == 0) p.counts[tile] = n;
            }
        }
        __syncthreads();            // the halo of the next tile is complete; red scratch free again
    }
#undef FL_LOAD_HALO
#undef FL_STORE_HALO
}

template <int CR>
int launch_first_t(const FirstArgs& a, int reserve_cus, hipStream_t st)
{
    const size_t lds = (size_t)(4 * conv_first_pairs(CR) * 64 + CR * F_CPL + 8 * 64 + 8) * sizeof(float);
    const int dv = cur_dev();
    static int nblocks[MAX_DEVICES] = {};
    if (!nblocks[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_kernel<CR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        nblocks[dv] = prop_cus;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int grid = std::max(1, nblocks[dv] - reserve_cus);
    hipLaunchKernelGGL((conv_first_kernel<CR>), dim3(tiles < grid ? tiles : grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_first_kernel");
    return FAV_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same layer with 2-D minimal filtering (round 3): the 9x9 correlation as nine 3x3 correlations (filter rows 3 a + r, columns
// 3 b + s), each computed with Winograd F(2x2, 3x3) and ACCUMULATED IN THE TRANSFORMED DOMAIN -- 16 multiplies per (channel, a, b) and
// 2x2 outputs against 24 for the 1-D form above (first2d_pack.h).  The 1-D kernel is bound by its matrix instructions (84 % of a
// tile's time); this form issues two thirds of them for twice the vector-ALU work per instruction (2 instead of 1).
//
// v_mfma_f32_16x16x4_f32: one wave = 16 tiles of 2x2 outputs x 32 channels, ALL sixteen positions in its own accumulators (16
// positions x 2 channel halves x 4 registers = 128): the output transform never leaves the lane.  An instruction multiplies four k =
// (channel, a, b) at once, one per group of 16 lanes; every lane group reads its own 4x4 patch -- rows 2 ty + 3 a + r, columns
// 2 tx + 3 b + d of its channel's halo plane -- at a per-lane offset computed once per kernel and quad (first2d_pack.h fixes which k
// share an instruction).  Block = 8 waves = 8 tile rows: 16 output rows x 32 columns, halo 24 x 40 pixels per channel as plain rows
// (lanes of a group read every second word: no conflicts inside a group), 131 KB of transformed weights resident in LDS, each
// B-operand read = 64 consecutive words.
constexpr int G_TH = 16, G_TW = 32;
constexpr int G_HR = G_TH + 8, G_HC = G_TW + 8;     // halo 24 x 40
constexpr int G_CPL = G_HR * G_HC + 16;             // words per channel plane (+16: consecutive channels start in different bank quarters)
constexpr int G_HP = G_HR * G_HC;                   // 960 halo pixels
constexpr int G_NH = (G_HP + 511) / 512;            // 2 halo pixels per thread

template <int CR, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void conv_first2d_kernel(const FirstArgs p)
{
    constexpr int NQ = (9 * CR + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ws = smem;                        // [16][NQ][2][4][16]
    float* const Hs = Ws + 16 * NQ * 128;          // [CR][24][40] (+16 per plane)
    float* const red = Hs + CR * G_CPL;            // [8][32] float2 + [8] int
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    const int grp = WIDE ? blockIdx.x % p.groups : 0, tstep = WIDE ? gridDim.x / p.groups : gridDim.x;
    const int PITCH = WIDE ? p.COUTP : 32;
    const float* const wsrc = p.wpk + (WIDE ? (size_t)grp * (16 * NQ * 128) : 0);
    for (int e = t; e < 16 * NQ * 32; e += 512) *reinterpret_cast<v4f*>(Ws + e * 4) = *reinterpret_cast<const v4f*>(wsrc + e * 4);

    const int ntiles = p.tiles_x * p.tiles_y;
    float4 hlo[G_NH], hhi[G_NH];
    // (round 6: the halo arrives by range-checked buffer loads -- an offset past the tensor returns zeros -- and both macros are free of
    //  branches: with the loads and the LDS stores under `if`s the compiler could not pair them up and drained EVERY outstanding memory
    //  operation, the previous tile's 33 output stores included, at the top of each tile: 1.4 us per tile in the timeline.  Threads past
    //  the halo's 960 pixels repeat its last pixel: same source, same value, same LDS word)
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.IH * p.IWp * 32, 0x00020000);
#define G_LOAD_HALO(tile_)                                                                          \
    {                                                                                               \
        const bool tv_ = (tile_) < ntiles;                                                          \
        const int ty_ = (tile_) / p.tiles_x, tx_ = (tile_) - ty_ * p.tiles_x;                       \
        _Pragma("unroll") for (int i = 0; i < G_NH; ++i) {                                          \
            const int pix_ = min(t + 512 * i, G_HP - 1), hy_ = pix_ / G_HC, hx_ = pix_ - hy_ * G_HC; \
            const int iy_ = ty_ * G_TH - p.pad + hy_, ix_ = tx_ * G_TW - p.pad + hx_;               \
            const bool v_ = tv_ & ((unsigned)iy_ < (unsigned)p.IH) & ((unsigned)ix_ < (unsigned)p.IW); \
            const int off_ = v_ ? (iy_ * p.IWp + ix_) * 32 : (int)0xFFFFFF00;                       \
            hlo[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irs, off_, 0, 0));          \
            hhi[i] = CR > 4 ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irs, off_ + 16, 0, 0)) : make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                           \
    }
#define G_STORE_HALO()                                                                              \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G_NH; ++i) {                                          \
            float* d_ = Hs + min(t + 512 * i, G_HP - 1);                                            \
            const float c_[8] = {hlo[i].x, hlo[i].y, hlo[i].z, hlo[i].w, hhi[i].x, hhi[i].y, hhi[i].z, hhi[i].w}; \
            _Pragma("unroll") for (int c = 0; c < CR; ++c) d_[c * G_CPL] = c_[c];                   \
        }                                                                                           \
    }

    int tile = WIDE ? blockIdx.x / p.groups : blockIdx.x;
    G_LOAD_HALO(tile);
    G_STORE_HALO();

    // lane = (tile column tx, lane group g): the group's k of quad q is (c, a, b) = conv_first2d_combo(CR, q, g)
    const int txl = lane & 15, g = lane >> 4;
    int qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int idx = 4 * q + g;
        const int c = idx / 9, a = (idx - 9 * c) / 3, b = idx - 9 * c - 3 * a;
        qoff[q] = idx < 9 * CR ? c * G_CPL + 3 * a * G_HC + 3 * b : 0;        // (no tap: zero weights, any address)
    }
    const float* const a_lane = Hs + 2 * wave * G_HC + 2 * txl;              // patch (0, 0) of tile (wave, txl)
    const float* const b_lo = Ws + lane;                                     // positions 0..7
    int hi_words = 8 * NQ * 128;
    asm volatile("" : "+v"(hi_words));                                       // (an opaque OFFSET: see conv_first_kernel)
    const float* const b_hi = b_lo + hi_words;                               // positions 8..15 (their offsets would not fit a DS immediate)
    __syncthreads();

#ifdef FAV_DIAG
    long long dt_[5] = {0, 0, 0, 0, 0}, tq_ = 0; int ntl_ = 0;
#define G_DBG(i_) { if (p.dbg && t == 0) { const long long n_ = wall_clock64(); dt_[i_] += n_ - tq_; tq_ = n_; } }
    if (p.dbg && t == 0) tq_ = wall_clock64();
#else
#define G_DBG(i_)
#endif
    for (; tile < ntiles; tile += tstep) {
        const int nxt = tile + tstep;
        G_LOAD_HALO(nxt);
        G_DBG(0);      /* issue of the next halo's loads (+ the previous tile's last barrier) */
        v4f acc[16][2];
#pragma unroll
        for (int ps = 0; ps < 16; ++ps)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[ps][nt] = v4f{0.f, 0.f, 0.f, 0.f};

        // per quad: the 4x4 patch (sixteen words, one quad ahead), then per transform line i: the row combination of the patch rows, the
        // four column operands, and eight MFMAs (4 positions x 2 channel halves) whose weights were read one line ahead
        float R[2][16], B[2][8];
#define G_READ_R(set_, q_)                                                                          \
        { const float* rb_ = a_lane + qoff[q_];                                                     \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) _Pragma("unroll") for (int d = 0; d < 4; ++d) R[set_][4 * r + d] = rb_[r * G_HC + d]; }
#define G_READ_B(set_, q_, i_)                                                                      \
        { _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
              B[set_][2 * j + nt] = ((i_) < 2 ? b_lo : b_hi)[(((4 * ((i_) & 1) + j) * NQ + (q_)) * 2 + nt) * 64]; }
#define G_FENCE() __builtin_amdgcn_sched_barrier(0)
        G_READ_R(0, 0);
        G_READ_B(0, 0, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) { G_READ_R((q + 1) & 1, q + 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // B^T d along the rows: l0 = r0 - r2, l1 = r1 + r2, l2 = r2 - r1, l3 = r1 - r3; then the same along the columns
                float L[4], V[4];
                const float* Rq = R[q & 1];
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    L[d] = i == 0 ? Rq[d] - Rq[8 + d] : i == 1 ? Rq[4 + d] + Rq[8 + d] : i == 2 ? Rq[8 + d] - Rq[4 + d] : Rq[4 + d] - Rq[12 + d];
                V[0] = L[0] - L[2]; V[1] = L[1] + L[2]; V[2] = L[2] - L[1]; V[3] = L[1] - L[3];
                const int nset = (4 * q + i + 1) & 1;
                if (i < 3) { G_READ_B(nset, q, i + 1); } else if (q + 1 < NQ) { G_READ_B(nset, q + 1, 0); }
                G_FENCE();
                const float* Bq = B[(4 * q + i) & 1];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[4 * i + j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[j], Bq[2 * j + nt], acc[4 * i + j][nt], 0, 0, 0);
                G_FENCE();
            }
        }
#undef G_READ_R
#undef G_READ_B
#undef G_FENCE
        G_DBG(1);      /* the quads: matrix instructions + patch transforms */
        __syncthreads();                    // every wave is done with the halo
        G_STORE_HALO();
        G_DBG(2);      /* barrier + next halo into LDS */

        // ---- output transform (A^T M A, in registers) + epilogue.  Register r of an accumulator = tile column 4 g + r, lane & 15 =
        // output channel inside the half nt
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int oy0 = ty * G_TH + 2 * __builtin_amdgcn_readfirstlane(wave), ox0 = tx * G_TW;
        // nine tiles in ten lie wholly inside the image (wave-uniform): no per-pixel test, one per-lane base pointer and uniform offsets
        // (round 6: the per-pixel form spent 64 64-bit address computations and 72 divergent branches per tile on its 32 stores)
        const bool inside = oy0 + 2 <= p.OH && ox0 + G_TW <= p.OW && grp * 32 + 32 <= p.COUT;
        float y[2][4][2][2];               // [nt][r][row a][column b]
        float sm[2] = {0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ch = grp * 32 + nt * 16 + txl;
            const float bv = p.bias[ch];
            const v2f bv2 = v2f{bv, bv};
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {      // tile columns rp, rp + 1 as one pair of values (packed adds, rounded like the single ones)
                v2f Q[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2f m0 = rp ? acc[4 * i][nt].zw : acc[4 * i][nt].xy, m1 = rp ? acc[4 * i + 1][nt].zw : acc[4 * i + 1][nt].xy;
                    const v2f m2 = rp ? acc[4 * i + 2][nt].zw : acc[4 * i + 2][nt].xy, m3 = rp ? acc[4 * i + 3][nt].zw : acc[4 * i + 3][nt].xy;
                    Q[i][0] = (m0 + m1) + m2; Q[i][1] = (m1 - m2) - m3;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const v2f y0 = (Q[0][b] + Q[1][b]) + Q[2][b] + bv2, y1 = (Q[1][b] - Q[2][b]) - Q[3][b] + bv2;
                    y[nt][rp][0][b] = y0.x; y[nt][rp + 1][0][b] = y0.y;
                    y[nt][rp][1][b] = y1.x; y[nt][rp + 1][1][b] = y1.y;
                }
            }
        }
        if (inside) {
            int gq = g, tq = txl;
            asm volatile("" : "+v"(gq), "+v"(tq));   // (the lane's part of the address is formed here, per tile: hoisted out of the tile loop it costs the WIDE form a spill)
            float* const pb = p.out + ((size_t)oy0 * p.OW + ox0 + 8 * gq) * p.COUT + grp * 32 + tq;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            pb[((size_t)a * p.OW + 2 * r + b) * p.COUT + nt * 16] = y[nt][r][a][b];
                            sm[nt] += y[nt][r][a][b];
                        }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int ch = grp * 32 + nt * 16 + txl;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int oy = oy0 + a, ox = ox0 + 2 * (4 * g + r) + b;
                            if (oy < p.OH && ox < p.OW) {
                                if (ch < p.COUT) p.out[((size_t)oy * p.OW + ox) * p.COUT + ch] = y[nt][r][a][b];
                                sm[nt] += y[nt][r][a][b];
                            }
                        }
            }
        }
        G_DBG(3);      /* output transform + stores */
        if (p.partials != nullptr) {
            float2* st = reinterpret_cast<float2*>(red);          // [8 waves][32]
            int* wn = reinterpret_cast<int*>(red + 8 * 64);         // [8]
            const int nrows = max(0, min(2, p.OH - oy0)), ncols = max(0, min(G_TW, p.OW - ox0));
            const int nw = nrows * ncols;                           // valid pixels of this wave's two rows
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float s = sm[nt];
                s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
                const float mu = nw ? s / (float)nw : 0.f;
                float qv = 0.f;
                if (inside) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) { const float d = y[nt][r][a][b] - mu; qv = fmaf(d, d, qv); }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const int oy = oy0 + a, ox = ox0 + 2 * (4 * g + r) + b;
                                const float d = y[nt][r][a][b] - mu;
                                if (oy < p.OH && ox < p.OW) qv = fmaf(d, d, qv);
                            }
                }
                qv += __shfl_xor(qv, 16); qv += __shfl_xor(qv, 32);
                if (lane < 16) st[wave * 32 + nt * 16 + lane] = make_float2(mu, qv);
            }
            if (lane == 0) wn[wave] = nw;
            __syncthreads();
            if (t < 32) {
                int n;
                p.partials[(size_t)tile * PITCH + grp * 32 + t] = merge_rows(st, wn, t, &n);
                if (t == 0 && grp == 0) p.counts[tile] = n;
            }
        }
        __syncthreads();            // the halo of the next tile is complete; red scratch free again
        G_DBG(4);      /* statistics + their two barriers */
#ifdef FAV_DIAG
        ++ntl_;
#endif
    }
#ifdef FAV_DIAG
    if (p.dbg && t == 0) { for (int i = 0; i < 5; ++i) p.dbg[blockIdx.x * 8 + i] = dt_[i]; p.dbg[blockIdx.x * 8 + 5] = ntl_; }
#endif
#undef G_DBG
#undef G_LOAD_HALO
#undef G_STORE_HALO
}

// (Measured and removed in round 5, profiles/r6p_first_layer_subblocks_ab.log, r6q_*: the block's eight waves as two independent halves
//  of four -- own 8 x 32 tiles, own halo buffers, a counter in LDS as their barrier, position 15's weights from global memory to make
//  room for the second halo -- so that one half's halo load, epilogue and statistics run under the other's matrix instructions.  Correct
//  (48 network tests), and 198-200 us against 190-192 for this kernel, whatever the start offset between the halves: the quads need BOTH
//  waves of a SIMD to keep the matrix pipe busy (a wave's operands come out of an LDS read -> row / column combination -> MFMA chain), so
//  a half that is alone in its quads runs at little more than half rate and the overlap buys nothing; the halo load it exposes costs 8 us.)
template <int CR, bool WIDE = false>
int launch_first2d_t(const FirstArgs& a, int reserve_cus, hipStream_t st)
{
    const size_t lds = (size_t)(16 * conv_first2d_quads(CR) * 128 + CR * G_CPL + 8 * 64 + 8) * sizeof(float);
    const int dv = cur_dev();
    static int nblocks[MAX_DEVICES] = {};
    if (!nblocks[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first2d_kernel<CR, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        nblocks[dv] = prop_cus;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int ngrp = WIDE ? a.groups : 1;
    int grid = std::max(1, nblocks[dv] - reserve_cus);
    grid = std::min(grid, tiles * ngrp);
    if (WIDE) grid = std::max(ngrp, grid / ngrp * ngrp);      // a block keeps one group's weights
#ifdef FAV_DIAG
    static int dbg_n = diag_env("FAV_FIRST_DBG") ? atoi(diag_env("FAV_FIRST_DBG")) : 0;
    static long long* dbuf = nullptr;
    const bool dbg = dbg_n > 0 && --dbg_n == 0;
    FirstArgs ad = a;
    if (dbg) { FAV_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 8 * 8)); FAV_HIP(hipMemsetAsync(dbuf, 0, 1024 * 8 * 8, st)); ad.dbg = dbuf; }
    hipLaunchKernelGGL((conv_first2d_kernel<CR, WIDE>), dim3(grid), dim3(512), lds, st, ad);
    FAV_LAUNCH_CHECK("conv_first2d_kernel");
    if (dbg) {
        std::vector<long long> hb((size_t)1024 * 8);
        FAV_HIP(hipStreamSynchronize(st)); FAV_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
        double sum[5] = {0, 0, 0, 0, 0}, nt = 0;
        for (int b = 0; b < grid; ++b) { for (int i = 0; i < 5; ++i) sum[i] += hb[b * 8 + i] * 0.01; nt += hb[b * 8 + 5]; }
        fprintf(stderr, "FIRSTDBG grid=%d tiles=%.0f  per tile (us): halo request %.2f  quads %.2f  barrier + halo to LDS %.2f  transform + stores %.2f  statistics %.2f\n",
                grid, nt, sum[0] / nt, sum[1] / nt, sum[2] / nt, sum[3] / nt, sum[4] / nt);
    }
    return FAV_OK;
#else
    hipLaunchKernelGGL((conv_first2d_kernel<CR, WIDE>), dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv_first2d_kernel");
    return FAV_OK;
#endif
}

}  // namespace

int conv_first_tiles(int OH, int OW) { return ((OH + F_TH - 1) / F_TH) * ((OW + F_TW - 1) / F_TW); }

// eligibility = conv_c8d_eligible (8-channel input pitch, 7 or 3 real channels, 9x9, stride 1, <= 32 output channels, no pending transform)
int launch_conv_first(const ConvLaunch& c, int cin_real, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(c.CIN == 8 && (cin_real == 7 || cin_real == 3) && c.COUTp == 32 && c.KH == 9 && c.KW == 9 && c.stride == 1 && c.pre.stages == 0 &&
                c.ups == 0 && !c.final_mode && wpk, "first-layer conv (F(2,3) along x): not eligible");
    FAV_REQUIRE((long long)c.IH * c.IWp * 8 < (1ll << 31), "first-layer conv: bad shape");
    FirstArgs a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.COUT = c.COUT; a.pad = c.pad; a.OH = c.OH; a.OW = c.OW;
    a.tiles_x = (c.OW + F_TW - 1) / F_TW; a.tiles_y = (c.OH + F_TH - 1) / F_TH;
    return cin_real == 7 ? launch_first_t<7>(a, c.reserve_cus, st) : launch_first_t<3>(a, c.reserve_cus, st);
}

// the 2-D form takes any number of 32-filter groups (c9s1-32: one; c9s1-64, the "more filters" checkpoints of README.md:141: two)
bool conv_first2d_eligible(int cin_pitch, int cin_real, int coutp, int k, int stride, int stages, int ups)
{
    return cin_pitch == 8 && (cin_real == 7 || cin_real == 3) && k == 9 && stride == 1 && stages == 0 && ups == 0 && coutp % 32 == 0 && coutp >= 32 && coutp <= 256;
}
int conv_first2d_tiles(int OH, int OW) { return ((OH + G_TH - 1) / G_TH) * ((OW + G_TW - 1) / G_TW); }

// same eligibility as launch_conv_first; wpk = conv_first2d_pack() (first2d_pack.h)
int launch_conv_first2d(const ConvLaunch& c, int cin_real, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv_first2d_eligible(c.CIN, cin_real, c.COUTp, c.KH, c.stride, c.pre.stages, c.ups) && c.KW == 9 && !c.final_mode && wpk,
                "first-layer conv (F(2x2,3x3) over the nine 3x3 blocks): not eligible");
    FAV_REQUIRE((long long)c.IH * c.IWp * 32 < (1ll << 31) - 512, "first-layer conv: input too large for 32-bit byte offsets");
    FirstArgs a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.COUT = c.COUT; a.pad = c.pad; a.OH = c.OH; a.OW = c.OW;
    a.tiles_x = (c.OW + G_TW - 1) / G_TW; a.tiles_y = (c.OH + G_TH - 1) / G_TH;
    a.COUTP = c.COUTp; a.groups = c.COUTp / 32; a.dbg = nullptr;
    if (a.groups > 1) return cin_real == 7 ? launch_first2d_t<7, true>(a, c.reserve_cus, st) : launch_first2d_t<3, true>(a, c.reserve_cus, st);
    return cin_real == 7 ? launch_first2d_t<7>(a, c.reserve_cus, st) : launch_first2d_t<3>(a, c.reserve_cus, st);
}

}  // namespace fav
