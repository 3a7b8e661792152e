// kernels_wino.hip -- the residual 3x3 convolutions (128 -> 128, stride 1, no padding; models_video.lua:10-39) as
// Winograd F(2x2, 3x3) minimal filtering on the fp32 matrix cores of gfx950.
//
// Why: the ten residual convolutions are 56 % of the network's multiply-adds, and the halo-resident implicit GEMM
// (kernels_conv.hip, conv3_halo_kernel) already runs them at 0.75 of the fp32 MFMA peak with 94 % matrix-pipe efficiency inside
// its K loop -- what is left there is clock and launch ramp.  F(2x2, 3x3) computes every 2x2 output patch from a 4x4 input
// patch with 16 multiplies per (input channel, output channel) instead of 36: 2.25x fewer matrix instructions for the same
// result (different rounding: the products are of transformed operands, fp32 throughout, errors of the order of the direct
// form's -- the parity tests bound them against the oracle).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        (matrices in wino_pack.h)
//
// Work unit = 4 x 8 Winograd tiles = 8 x 16 output pixels x 128 output channels, one block of 8 waves per CU:
//   * the 16 positions (i, j) of the transformed domain are 16 independent GEMMs  [32 tiles] x [CIN] x [128 couts];
//     wave w owns positions (i = w >> 1, j = 2 (w & 1) + {0, 1}) for ALL 32 tiles and ALL 128 output channels:
//     2 positions x 4 channel tiles x 16 registers = 128 accumulator registers, 256 KB per block -- the register file is the
//     reason a unit is 32 tiles and a CU works on one unit at a time
//   * input: per 32-channel slice the 10 x 18 pixel halo is loaded once, the pending InstanceNorm/ReLU of the producing layer is
//     applied, and the ROW half of the input transform (B^T d) is done by the staging threads: LDS holds 4 tile rows x 4
//     transformed lines x 18 columns, columns split into an even and an odd plane so that the 8 tiles of a tile row read
//     consecutive pixels.  A wave builds its two A fragments per 8-channel group from three conflict-free ds_read_b128 and 8
//     vector-ALU instructions (the COLUMN half: one add/subtract per element) -- 0.25 VALU per MFMA, as in the direct kernel
//   * weights: transformed and packed on the host in exactly the fragment order (wino_pack.h); no two waves share a weight, so
//     they go global -> registers directly (one contiguous 1 KiB load per 4 MFMAs, L2-resident: 1 MB per layer), prefetched
//     one half-step (16 MFMAs) ahead.  No LDS ring, no barrier for weights; ONE barrier per 32-channel slice for the halo
//   * output transform: each wave folds its two columns (A^T along j) in registers; the row fold (A^T along i) crosses waves
//     through LDS in two passes (147 KB each), followed by bias, the NHWC store (256 contiguous bytes per wave instruction) and
//     the per-unit InstanceNorm partials (mean, M2, count) like the other convolution kernels
// Units are independent (no stream-K hand-off, nothing co-resident assumed): persistent blocks walk the units round-robin.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "fav_internal.h"
#include "wino_pack.h"

namespace fav {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

constexpr int LDSS = 36;                               // pixel pitch in floats (32 channels + 4): conflict-free 16-byte fragment reads
constexpr int WG_TROW = 18 * LDSS;                     // 648: one row-transformed line = even plane (9 pixels) | odd plane (9 pixels)
constexpr int WG_TTYP = 4 * WG_TROW;                   // 2592: the four lines i = 0..3 of one tile row; 648 sixteen-byte slots = 8 (mod 16)
constexpr int WG_TBUF = 4 * WG_TTYP;                   // 10368 floats per halo buffer (4 tile rows)
constexpr int WG_PS = 8 * 128 * LDSS;                  // epilogue exchange [wave][cout][36] (147 456 B); the halo buffers live inside it
static_assert(2 * WG_TBUF <= WG_PS, "halo buffers overlay the exchange area");

struct WinoArgs {
    const float* in; const float* wpk; const float* bias; const float* scale1; const float* shift1;
    float* out; float2* partials; int* counts;
    const float* skip; float* zout;      // MODE 2 (pending residual join): the skip tensor's pixel under input pixel (0, 0) -- same pitch IWp --, the joined tensor
    int OWp;                             // row pitch of `out` in pixels (OW, or the pitch of the tensor a later join adds it to)
    int IH, IW, IWp, CIN, OH, OW, units_x, units_y, relu1;
    int nfull;               // units 0 .. nfull-1 are computed whole, the rest in four quarters (32 output channels each)
    long long* dbg;          // optional in-kernel timeline (FAV_WINO_DBG), 24 slots per block
};

// exact merge of NW groups' (mean, M2) with per-group counts (Chan et al.); same as kernels_conv.hip
__device__ __forceinline__ float2 merge_group_stats(const float2* st, const int* wn, int NW, int pitch, int c, int* n_out)
{
    int n = 0; float s = 0.f;
    for (int w = 0; w < NW; ++w) { n += wn[w]; s += (float)wn[w] * st[w * pitch + c].x; }
    const float mean = n ? s / (float)n : 0.f;
    float m2 = 0.f;
    for (int w = 0; w < NW; ++w) { const float d = st[w * pitch + c].x - mean; m2 += st[w * pitch + c].y + (float)wn[w] * d * d; }
    *n_out = n;
    return make_float2(mean, m2);
}

// MODE 1: the input carries a pending per-channel scale/shift (+ReLU) -- the InstanceNorm of the producing convolution.
// MODE 2: the input is a pending RESIDUAL JOIN (models_video.lua:41-53): z = skip + scale * y + shift, formed while the halo is staged
//   (same operations in the same order as res_add_kernel, kernels_conv.hip) and written out once -- every input pixel by the unit whose
//   8 x 16 outputs start at it -- as the next block's skip.  y and skip share pitch and offsets (net.cpp lays y out under the skip),
//   the skip rows bypass the register file: buffer_load ... lds into the part of the exchange area the halo buffers leave free, read
//   back by the thread that requested them when it commits the item (requested BEFORE y's rows: loads return in order, so y's arrival
//   implies theirs).
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv3_wino_kernel(const WinoArgs p)
{
    constexpr bool AFF = MODE != 0, JOIN = MODE == 2;
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ts = smem;                       // [2][WG_TBUF] in the K loop, [8][128][LDSS] in the epilogue
    float* const aff = smem + WG_PS;              // [2][CIN]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool wodd = (wave & 1) != 0;
    const int CIN = p.CIN, nslices = CIN >> 5, nkg = CIN >> 3;

    int dbi = 0;
#define DBG_T() { if (p.dbg && t == 0 && dbi < 21) p.dbg[blockIdx.x * 24 + dbi++] = wall_clock64(); }
    DBG_T();
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (AFF) for (int i = t; i < CIN; i += NT) { aff[i] = p.scale1[i]; aff[CIN + i] = p.shift1[i]; }
    const float lo1 = (MODE == 1 && p.relu1) ? 0.f : -INFINITY;
    // landing area of the skip rows (MODE 2): [row a][thread] sixteen bytes each behind the two halo buffers, then wave 0's item B
    float* const land = Ts + 2 * WG_TBUF;
    static_assert(2 * WG_TBUF + 4 * 512 * 4 + 4 * 64 * 4 <= WG_PS, "skip landing area fits behind the halo buffers");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // staging items: (tile row ty, raw column x, 16-byte channel chunk c4) -> raw rows 2 ty .. 2 ty + 3 of column x, four transformed
    // lines out.  4 x 18 x 8 = 576 items: item A = t (pixel t >> 3 = 0..63), item B = 512 + t for wave 0 (pixels 64..71: ty 3, x 10..17)
    const int c4 = t & 7;
    const int pixA = t >> 3, tyA = (pixA * 3641) >> 16, xA = pixA - tyA * 18;        // pixA / 18
    const int xB = 10 + (lane >> 3);
    float* const tstA = Ts + tyA * WG_TTYP + ((xA & 1) * 9 + (xA >> 1)) * LDSS + c4 * 4;
    float* const tstB = Ts + 3 * WG_TTYP + ((xB & 1) * 9 + (xB >> 1)) * LDSS + c4 * 4;
    const float* const affr = aff + c4 * 4;

    // fragments: lane = (tile m = lane & 31 -> ty = m >> 3, tx = m & 7; channel half h).  Line i = wave >> 1.  Raw columns
    // 2 tx + d of the line sit at pixel ((d & 1) * 9 + tx + (d >> 1)): d = 0 -> +0, 1 -> +9, 2 -> +1, 3 -> +10.
    // One form for both wave parities (no selects in the loop):  A0 = R0 - R2,  A1 = R1 + sg * R2
    //   even wave (j = 0, 1): R0 = c0, R1 = c1, R2 = c2, sg = +1:  V0 = c0 - c2,  V1 = c1 + c2
    //   odd wave  (j = 2, 3): R0 = c2, R1 = c3, R2 = c1, sg = -1:  V2 = c2 - c1, -V3 = c3 - c1 (the packed weights of j = 3 are negated: (-V)(-U) = VU)
    const int m = lane & 31, h = lane >> 5;
    const float* const ab = Ts + (m >> 3) * WG_TTYP + (wave >> 1) * WG_TROW + (m & 7) * LDSS + 4 * h;
    const float* const ap0 = ab + (wodd ? 1 : 0) * LDSS;
    const float* const ap1 = ab + (wodd ? 10 : 9) * LDSS;
    const float* const ap2 = ab + (wodd ? 9 : 1) * LDSS;
    const float sg = wodd ? -1.f : 1.f;
    // global reads are raw buffer loads: scalar descriptor + one 32-bit lane offset + a scalar offset + an immediate -- no 64-bit
    // vector address arithmetic in the loop.  Weights: lane * 16 + [wave * 8192 + kg * 65536 + q * 4096] + nt * 1024
    const int wlo = lane * 16, wso = wave * 8192;

    if (AFF) __syncthreads();
    // One work item: NTW = 4: a whole unit (128 output channels); NTW = 1: a quarter of one (output channels 32 nq .. 32 nq + 31) --
    // the units of a thin last round are cut in four so that the round takes a quarter of the time (see launch_wino_t)
    auto work = [&](auto ntw_c, const int u, const int nq) {
        constexpr int NTW = decltype(ntw_c)::value;
        constexpr int NC = 32 * NTW;               // output channels of the item
        // (descriptors are built here from kernel arguments: captured ones are not provably wave-uniform and get a waterfall loop.
        //  Measured and dropped: requesting the NEXT item's first halo rows from inside this epilogue -- the 32 registers it keeps
        //  alive between items cost 6 % in the K loop, against 0.4 us of prologue it hides)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (p.CIN >> 3) * 65536, 0x00020000);
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(JOIN ? p.skip : p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(JOIN ? p.zout : const_cast<float*>(p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const int uy = u / p.units_x, ux = u - uy * p.units_x;
        const int oy0 = uy * 8, ox0 = ux * 16;
        DBG_T();   /* unit start */
        // no padding: input pixel (oy0 + r, ox0 + c) for halo (r, c); coordinates past the image only feed outputs past the image
        // (never stored), so they are clamped instead of masked
        int hoA[4], hoB[4];
        {
            const int ixa = min(ox0 + xA, p.IW - 1), ixb = min(ox0 + xB, p.IW - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                hoA[a] = ((min(oy0 + 2 * tyA + a, p.IH - 1) * p.IWp + ixa) * CIN + c4 * 4) * 4;
                hoB[a] = ((min(oy0 + 6 + a, p.IH - 1) * p.IWp + ixb) * CIN + c4 * 4) * 4;
            }
        }
        // MODE 2: which of an item's four rows this thread writes to the joined tensor (bits 0-3 item A, 4-7 item B): rows 2 ty, 2 ty + 1
        // of columns 0..15 -- the unit's own 8 x 16 pixels -- plus the halo fringe (rows 8, 9 / columns 16, 17) where no other unit follows
        int zm = 0;
        if (JOIN && nq == 0) {
            const bool lastx = ux == p.units_x - 1, lasty = uy == p.units_y - 1;
            const bool ca = ox0 + xA < p.IW && (xA < 16 || lastx), cb = ox0 + xB < p.IW && (xB < 16 || lastx);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool ra = oy0 + 2 * tyA + a < p.IH && (a < 2 || (tyA == 3 && lasty));
                const bool rb = oy0 + 6 + a < p.IH && (a < 2 || lasty);
                zm |= (ca && ra ? 1 : 0) << a | (cb && rb ? 16 : 0) << a;
            }
        }

        v4f qr[4];
        v4f sc, sh;
        const bool lasty_ = uy == p.units_y - 1;
        float* const landA = land + wave * 256;            // (wave-uniform: M0 of the LDS loads)
        float* const landB = land + 8192;
        const float* const readA = land + t * 4;
        const float* const readB = land + 8192 + lane * 4;
#define WG_LOAD_RAW(qr, slice_, ho_)                                                                \
        { _Pragma("unroll") for (int a = 0; a < 4; ++a) qr[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho_[a], (slice_) * 128, 0)); }
#define WG_AFF(slice_)                                                                              \
        { if (AFF) { sc = *reinterpret_cast<const v4f*>(affr + (slice_) * 32); sh = *reinterpret_cast<const v4f*>(affr + CIN + (slice_) * 32); } }
#define WG_XF(v_)                                                                                   \
        { if (MODE == 1) { v_.x = fmaxf(fmaf(v_.x, sc.x, sh.x), lo1); v_.y = fmaxf(fmaf(v_.y, sc.y, sh.y), lo1); \
                           v_.z = fmaxf(fmaf(v_.z, sc.z, sh.z), lo1); v_.w = fmaxf(fmaf(v_.w, sc.w, sh.w), lo1); } }
        // MODE 2: skip rows straight into LDS (one instruction = one row of all 64 lanes, 1 KiB at the wave's place in the landing area)
#define WG_LOAD_SKIP(slice_, ho_, lbase_, lrow_)                                                    \
        { if (JOIN) { asm volatile("" ::: "memory");      /* (never above the reads of the rows these loads replace) */ \
              _Pragma("unroll") for (int a = 0; a < 4; ++a)                                         \
              __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)((lbase_) + a * (lrow_)), 16, ho_[a], (slice_) * 128, 0, 0); } }
        // MODE 2: z = fma(y, scale, shift) + skip (res_add_kernel's order), written where the mask says so (elsewhere the offset is out of
        // the buffer's range and the hardware drops the store), then transformed like any other input
        // (the compiler does not see that the LDS reads depend on the LDS loads: an explicit wait -- at most `vm_` vector-memory
        //  operations younger than the skip rows may still be in flight: y's four rows are needed here anyway, the eight weight
        //  fragments requested since are not -- and a compiler barrier keep the reads behind the data)
#define WG_JOIN(qr, slice_, ho_, lread_, lrow_, zbit_, vm_)                                         \
        { if (JOIN) { asm volatile("s_waitcnt vmcnt(" #vm_ ")" ::: "memory");                       \
              _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                       \
              const v4f xs_ = *reinterpret_cast<const v4f*>((lread_) + a * (lrow_));                \
              qr[a].x = fmaf(qr[a].x, sc.x, sh.x) + xs_.x; qr[a].y = fmaf(qr[a].y, sc.y, sh.y) + xs_.y; \
              qr[a].z = fmaf(qr[a].z, sc.z, sh.z) + xs_.z; qr[a].w = fmaf(qr[a].w, sc.w, sh.w) + xs_.w; \
              if (a < 2 || lasty_) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, qr[a]), zrs, (zm & ((zbit_) << a)) ? ho_[a] : (int)0xFFFFFFF0, (slice_) * 128, 0); } } }
        // B^T along the rows: l0 = r0 - r2, l1 = r1 + r2, l2 = r2 - r1, l3 = r1 - r3
#define WG_COMMIT(qr, dst_)                                                                         \
        { WG_XF(qr[0]); WG_XF(qr[1]); WG_XF(qr[2]); WG_XF(qr[3]);                                    \
          *reinterpret_cast<v4f*>(dst_) = qr[0] - qr[2];                                             \
          *reinterpret_cast<v4f*>((dst_) + WG_TROW) = qr[1] + qr[2];                                 \
          *reinterpret_cast<v4f*>((dst_) + 2 * WG_TROW) = qr[2] - qr[1];                             \
          *reinterpret_cast<v4f*>((dst_) + 3 * WG_TROW) = qr[1] - qr[3]; }

        v4f R0, R1, R2, A0, A1, fb[2][NTW];
        const int wlq = wlo + nq * 1024;
#define WG_READ_T(par_, kg_)                                                                        \
        { R0 = *reinterpret_cast<const v4f*>(ap0 + (par_) * WG_TBUF + (kg_) * 8);                   \
          R1 = *reinterpret_cast<const v4f*>(ap1 + (par_) * WG_TBUF + (kg_) * 8);                   \
          R2 = *reinterpret_cast<const v4f*>(ap2 + (par_) * WG_TBUF + (kg_) * 8); }
#define WG_MAKE_A()                                                                                 \
        { A0 = R0 - R2; A1.x = fmaf(sg, R2.x, R1.x); A1.y = fmaf(sg, R2.y, R1.y); A1.z = fmaf(sg, R2.z, R1.z); A1.w = fmaf(sg, R2.w, R1.w); }
#define WG_LOAD_B(set_, kgg_, q_)                                                                   \
        { const int so_ = wso + (kgg_) * 65536 + (q_) * 4096;                                      \
          _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                        \
              fb[set_][nt] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlq + nt * 1024, so_, 0)); }
#define WG_MFMA(set_, a_, q_)                                                                       \
        { _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[q_][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.x, fb[set_][nt].x, acc[q_][nt], 0, 0, 0); \
          _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[q_][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.y, fb[set_][nt].y, acc[q_][nt], 0, 0, 0); \
          _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[q_][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.z, fb[set_][nt].z, acc[q_][nt], 0, 0, 0); \
          _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[q_][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_.w, fb[set_][nt].w, acc[q_][nt], 0, 0, 0); }

        // ---- prologue: slice 0 -> halo buffer 0
        {
            v4f qb[4];
            // (MODE 2: the first slice's skip rows through registers -- there are plenty before the accumulators exist.  The skip rows
            //  arrive 2.4 us behind y's either way (in-kernel timeline, profiles/r04_lazy_join_ab.log) and that is exposed here, once per
            //  unit; warming the L2 for the next unit's rows from inside the output transform did not move it)
            v4f xa[4], xb[4];
            if (JOIN) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    xa[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(srs, hoA[a], 0, 0));
                    if (wave == 0) xb[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(srs, hoB[a], 0, 0));
                }
            }
            WG_LOAD_RAW(qr, 0, hoA);
            if (wave == 0) { WG_LOAD_RAW(qb, 0, hoB); }
            WG_LOAD_B(0, 0, 0);
            WG_AFF(0);
#define WG_JOIN_REG(qr, xs, ho_, zbit_)                                                             \
            { if (JOIN) { _Pragma("unroll") for (int a = 0; a < 4; ++a) {                           \
                  qr[a].x = fmaf(qr[a].x, sc.x, sh.x) + xs[a].x; qr[a].y = fmaf(qr[a].y, sc.y, sh.y) + xs[a].y; \
                  qr[a].z = fmaf(qr[a].z, sc.z, sh.z) + xs[a].z; qr[a].w = fmaf(qr[a].w, sc.w, sh.w) + xs[a].w; \
                  if (a < 2 || lasty_) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, qr[a]), zrs, (zm & ((zbit_) << a)) ? ho_[a] : (int)0xFFFFFFF0, 0, 0); } } }
            WG_JOIN_REG(qr, xa, hoA, 1); WG_COMMIT(qr, tstA);
            if (wave == 0) { WG_JOIN_REG(qb, xb, hoB, 16); WG_COMMIT(qb, tstB); }
#undef WG_JOIN_REG
            // the skip rows come from further away than y's (the tensor was written two launches ago): they are requested a whole
            // slice before their use -- landing in LDS they cost no registers -- i.e. right behind the reads of the slice before
            WG_LOAD_SKIP(min(1, nslices - 1), hoA, landA, 2048);
            if (wave == 0) { WG_LOAD_SKIP(min(1, nslices - 1), hoB, landB, 256); }
        }
        f32x16 acc[2][NTW];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][nt][r] = 0.f;
        __syncthreads();
        WG_READ_T(0, 0);
        DBG_T();   /* loop start */
        const long long ck0 = p.dbg ? clock64() : 0, wk0 = p.dbg ? wall_clock64() : 0;

        // ---- K loop: per 32-channel slice 4 groups of 8 channels, per group two half-steps (position q = 0, 1) of 16 MFMAs
        //   (g, 0): A fragments of group g from the raw reads; weights of (g, 1) requested; MFMAs of position 0
        //   (g, 1): raw reads of group g + 1; weights of (g + 1, 0) requested; MFMAs of position 1
        //   halo of the next slice: item A loaded in group 0, committed in group 1; item B (wave 0) loaded in 1, committed in 2;
        //   the slice's barrier sits at (3, 0): after it the next slice's buffer is complete and this slice's is no longer read
        // (the last slice stages a copy of itself into the idle buffer: no branches inside the loop body)
#define WG_FENCE() __builtin_amdgcn_sched_barrier(0)
        for (int s = 0; s < nslices; ++s) {
            const int par = s & 1;
            const int sn = min(s + 1, nslices - 1), sn2 = min(s + 2, nslices - 1);
            float* const tA = tstA + (par ^ 1) * WG_TBUF;
            float* const tB = tstB + (par ^ 1) * WG_TBUF;
            const int kgg = s * 4;
            // group 0
            WG_MAKE_A(); WG_LOAD_B(1, kgg, 1); WG_LOAD_RAW(qr, sn, hoA);
            WG_FENCE(); WG_MFMA(0, A0, 0); WG_FENCE();
            WG_READ_T(par, 1); WG_LOAD_B(0, kgg + 1, 0);
            WG_FENCE(); WG_MFMA(1, A1, 1); WG_FENCE();
            // group 1
            WG_MAKE_A(); WG_LOAD_B(1, kgg + 1, 1);
            WG_AFF(sn); WG_JOIN(qr, sn, hoA, readA, 2048, 1, 8); WG_COMMIT(qr, tA); WG_LOAD_SKIP(sn2, hoA, landA, 2048); if (wave == 0) { WG_LOAD_RAW(qr, sn, hoB); }
            WG_FENCE(); WG_MFMA(0, A0, 0); WG_FENCE();
            WG_READ_T(par, 2); WG_LOAD_B(0, kgg + 2, 0);
            WG_FENCE(); WG_MFMA(1, A1, 1); WG_FENCE();
            // group 2
            WG_MAKE_A(); WG_LOAD_B(1, kgg + 2, 1);
            if (wave == 0) { WG_JOIN(qr, sn, hoB, readB, 256, 16, 8); WG_COMMIT(qr, tB); WG_LOAD_SKIP(sn2, hoB, landB, 256); }
            WG_FENCE(); WG_MFMA(0, A0, 0); WG_FENCE();
            WG_READ_T(par, 3); WG_LOAD_B(0, kgg + 3, 0);
            WG_FENCE(); WG_MFMA(1, A1, 1); WG_FENCE();
            // group 3
            WG_MAKE_A();
            __syncthreads();
            WG_LOAD_B(1, kgg + 3, 1);
            WG_FENCE(); WG_MFMA(0, A0, 0); WG_FENCE();
            WG_READ_T(par ^ 1, 0); WG_LOAD_B(0, min(kgg + 4, nkg - 1), 0);
            WG_FENCE(); WG_MFMA(1, A1, 1); WG_FENCE();
        }
#undef WG_FENCE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                    // every wave is done with the halo buffers: the exchange area takes their place
        DBG_T();   /* loop end */
        if (p.dbg && t == 0) { p.dbg[blockIdx.x * 24 + 21] += clock64() - ck0; p.dbg[blockIdx.x * 24 + 22] += wall_clock64() - wk0; }
        DBG_T();   /* (no fix-up) */
#undef WG_LOAD_RAW
#undef WG_AFF
#undef WG_XF
#undef WG_COMMIT
#undef WG_LOAD_SKIP
#undef WG_JOIN
#undef WG_READ_T
#undef WG_MAKE_A
#undef WG_LOAD_B
#undef WG_MFMA

        // ---- output transform.  acc[q][nt][r] = M[i][j0 + q] of tile mi = (r & 3) + 8 (r >> 2) + 4 h, output channel nt * 32 + n.
        // Column fold in registers (A^T = |1 1 1 0; 0 1 -1 -1| along j): even wave  P0 = M0 + M1, P1 = M1;
        //                                                                odd wave   P0 = M2,      P1 = -(M2 + M3).
        // Row fold across the waves through LDS, one pass per output column parity b:
        //   Z_i = P(wave 2i) + P(wave 2i+1);   Y[0][b] = Z0 + Z1 + Z2,   Y[1][b] = Z1 - Z2 - Z3
        // Reduction thread (4 NC of them): output channel c = t % NC, tile row qq = t / NC (tiles 8 qq .. 8 qq + 7 = tx 0..7).
        const int n = lane & 31;
        const int c = t & (NC - 1), qq = (t / NC) & 3;
        const bool red = t < 4 * NC;
        const int co = nq * 32 + c;
        const float bv = p.bias[co];
        float yk[2][2][8];                 // [b][a][tx]
        float* const pw = Ts + (wave * NC + n) * LDSS + 4 * h;
        const float* const pr = Ts + c * LDSS + 8 * qq;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    v4f v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m0 = acc[0][nt][4 * g + e], m1 = acc[1][nt][4 * g + e];
                        v[e] = wodd ? (b == 0 ? m0 : -(m0 + m1)) : (b == 0 ? m0 + m1 : m1);
                    }
                    *reinterpret_cast<v4f*>(pw + nt * 32 * LDSS + 8 * g) = v;
                }
            __syncthreads();
            if (red) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v4f z[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    z[i] = *reinterpret_cast<const v4f*>(pr + (2 * i) * NC * LDSS + 4 * hf) + *reinterpret_cast<const v4f*>(pr + (2 * i + 1) * NC * LDSS + 4 * hf);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yk[b][0][4 * hf + e] = (z[0][e] + z[1][e]) + z[2][e] + bv;
                    yk[b][1][4 * hf + e] = (z[1][e] - z[2][e]) - z[3][e] + bv;
                }
            }
            }
            __syncthreads();
        }
        // store + per-thread statistics of the 32 outputs (2 rows x 16 columns) of channel c
        int nv = 0; float sm = 0.f;
        if (red) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = oy0 + 2 * qq + a;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ox = ox0 + 2 * k + b;
                    if (oy < p.OH && ox < p.OW) { p.out[((size_t)oy * p.OWp + ox) * 128 + co] = yk[b][a][k]; sm += yk[b][a][k]; ++nv; }
                }
        }
        }
        if (p.partials != nullptr) {
            const float mu = nv ? sm / (float)nv : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int oy = oy0 + 2 * qq + a;
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int ox = ox0 + 2 * k + b;
                        const float d = yk[b][a][k] - mu;
                        if (oy < p.OH && ox < p.OW) m2 = fmaf(d, d, m2);
                    }
            }
            float2* st = reinterpret_cast<float2*>(Ts);           // [4][NC]
            int* wn = reinterpret_cast<int*>(Ts + 2 * 4 * 128);     // [4]
            if (red) { st[qq * NC + c] = make_float2(mu, m2); if (c == 0) wn[qq] = nv; }
            __syncthreads();
            if (t < NC) {
                int nn;
                p.partials[(size_t)u * 128 + nq * 32 + t] = merge_group_stats(st, wn, 4, NC, t, &nn);
                if (t == 0) p.counts[u] = nn;          // (the four quarters of a unit write the same count)
            }
            __syncthreads();
        }
        DBG_T();   /* epilogue end */
    };
    const int nitems = p.nfull + 4 * (p.units_x * p.units_y - p.nfull);
    for (int it = lb; it < nitems; it += gridDim.x) {
        if (it < p.nfull) work(std::integral_constant<int, 4>{}, it, 0);
        else work(std::integral_constant<int, 1>{}, p.nfull + ((it - p.nfull) >> 2), (it - p.nfull) & 3);
    }
    if (p.dbg && t == 0) p.dbg[blockIdx.x * 24 + 23] = dbi;
#undef DBG_T
}


// FAV_WINO_DBG=n: print the in-kernel timeline of the n-th launch
void wino_debug_report(const long long* hbuf, int grid)
{
    long long t0 = hbuf[0];
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, hbuf[b * 24]);
    double sum[4] = {0, 0, 0, 0}, tend = 0, ck = 0, wk = 0; int items = 0;
    for (int b = 0; b < grid; ++b) {
        const long long* r = &hbuf[b * 24]; const int n = (int)r[23];
        for (int i = 1; i + 4 < n + 1 && i + 4 <= 20; i += 5) {
            for (int q = 0; q < 4; ++q) sum[q] += (r[i + q + 1] - r[i + q]) * 0.01;
            ++items; tend = std::max(tend, (r[i + 4] - t0) * 0.01);
        }
        ck += r[21]; wk += r[22];
    }
    fprintf(stderr, "WINODBG grid=%d units=%d  K loop: %.3f GHz;  per unit: prologue %.2f  loop %.2f  epilogue %.2f us;  last block ends at %.2f us\n",
            grid, items, wk ? ck / (wk * 10.0) : 0.0, items ? sum[0] / items : 0.0, items ? sum[1] / items : 0.0, items ? sum[3] / items : 0.0, tend);
}

template <int MODE>
int launch_wino_t(const WinoArgs& a0, int reserve_cus, hipStream_t st)
{
    const auto kern = conv3_wino_kernel<MODE>;
    const size_t lds = (size_t)(WG_PS + 2 * a0.CIN) * sizeof(float);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int occ = 0; int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));      // (hipGetDeviceProperties costs a millisecond or two per call)
        FAV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 512, lds));
        if (occ < 1) { set_error("winograd conv: kernel does not fit on a CU"); return FAV_EHIP; }
        cus[dv] = prop_cus;          // one block per CU
    }
    const int units = a0.units_x * a0.units_y;
    const int grid = std::min(units, std::max(1, cus[dv] - reserve_cus));
    WinoArgs a = a0; a.dbg = nullptr;
    // A CU holds one unit at a time (the accumulators fill the register file), so a launch takes ceil(units / grid) rounds of one
    // unit time.  When the last round has at most grid / 4 units (the first three residual layers at 1280x720: 550, 525, 525 units
    // on 256 CUs) its units are cut into quarters of 32 output channels: four times as many CUs work on that round and it takes
    // roughly a third of a unit time instead of a whole one.
    static const bool no_quarters = diag_env("FAV_WINO_NO_QUARTERS") != nullptr;
    const int rounds = (units + grid - 1) / grid, rem = units - (rounds - 1) * grid;
    a.nfull = (rounds >= 2 && rem * 4 <= grid && !no_quarters) ? (rounds - 1) * grid : units;
    static int dbg_n = diag_env("FAV_WINO_DBG") ? atoi(diag_env("FAV_WINO_DBG")) : 0;
    static long long* dbuf = nullptr;
    const bool dbg = dbg_n > 0 && --dbg_n == 0;
    if (dbg) { FAV_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 512 * 24 * 8)); FAV_HIP(hipMemsetAsync(dbuf, 0, 512 * 24 * 8, st)); a.dbg = dbuf; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv3_wino_kernel");
    if (dbg) {
        std::vector<long long> hb((size_t)512 * 24);
        FAV_HIP(hipStreamSynchronize(st)); FAV_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
        wino_debug_report(hb.data(), grid);
    }
    return FAV_OK;
}


}  // namespace

bool conv3_wino_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups)
{
    return k == 3 && stride == 1 && pad == 0 && ups == 0 && stages <= 1 && cin_pitch % 32 == 0 && cin_pitch >= 32 && cin_pitch <= 256 &&
           cout == 128 && coutp == 128;
}
int conv3_wino_tiles(int OH, int OW) { return ((OH + 7) / 8) * ((OW + 15) / 16); }

int launch_conv3_wino(const ConvLaunch& c, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3_wino_eligible(c.CIN, c.COUT, c.COUTp, c.KH, c.stride, c.pad, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && !c.stuff && wpk,
                "winograd conv: not eligible");
    FAV_REQUIRE((long long)(c.IH + 1) * c.IWp * c.CIN < (1ll << 29), "winograd conv: tensor too large for 32-bit byte offsets");
    FAV_REQUIRE(c.OH == c.IH - 2 && c.OW == c.IW - 2, "winograd conv: bad geometry");
    WinoArgs a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.relu1 = c.pre.relu1;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.CIN = c.CIN; a.OH = c.OH; a.OW = c.OW;
    a.units_x = (c.OW + 15) / 16; a.units_y = (c.OH + 7) / 8;
    a.dbg = nullptr;
    a.skip = c.join_skip; a.zout = c.join_out; a.OWp = c.OWp > 0 ? c.OWp : c.OW;
    if (c.join_skip != nullptr) {
        FAV_REQUIRE(c.join_out != nullptr && c.pre.stages == 1 && c.pre.relu1 == 0, "winograd conv: a pending residual join needs its output tensor and exactly one pending normalisation");
        return launch_wino_t<2>(a, c.reserve_cus, st);
    }
    return c.pre.stages >= 1 ? launch_wino_t<1>(a, c.reserve_cus, st) : launch_wino_t<0>(a, c.reserve_cus, st);
}

}  // namespace fav
