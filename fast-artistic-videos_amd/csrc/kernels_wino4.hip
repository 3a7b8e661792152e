// kernels_wino4.hip -- the residual 3x3 convolutions (128 -> 128, stride 1, no padding; models_video.lua:10-39) as Winograd
// F(4x4, 3x3) minimal filtering on the fp32 matrix cores of gfx950 (round 4).
//
// Why: kernels_wino.hip (F(2x2, 3x3)) sits at 0.58 of the fp32 MFMA peak with 94 % matrix-pipe efficiency inside its K loop; what is
// left there is clock, prologue and a cross-wave output transform.  The remaining lever is the NUMBER of matrix instructions:
// F(4x4, 3x3) computes a 4x4 output patch from a 6x6 input patch with 36 multiplies per (input channel, output channel) -- 2.25 per
// output against 4 (F(2x2)) and 9 (direct).  Its price is accuracy (larger transform constants): with the interpolation points
// 0, +-3/4, +-3/2 (wino4_pack.h) a layer's fp32 error is ~3x the F(2x2) form's, still 2x a plain fp32 accumulation chain's.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        (matrices in wino4_pack.h)
//
// Work unit = 4 x 4 tiles = 16 x 16 output pixels x 128 output channels, one block of EIGHT waves (512 threads) per CU, two waves per
// SIMD with 256 registers each (amdgpu_waves_per_eu(2, 2)); layers with more than 128 filters run as several such units per pixel unit
// (WIDE instantiations, one group of 128 output channels each):
//   * v_mfma_f32_16x16x4_f32: M = the unit's 16 tiles, N = 16 output channels, K = 4 input channels.  Wave w owns output channels
//     16 w .. 16 w + 15 (wave = 2 w' + nt in wino4_pack.h's (w', nt) numbering) for ALL 36 transform positions and all 16 tiles:
//     acc[36] x 4 = 144 accumulator registers.  Every (tile, output channel) has its 36 position values in ONE lane, so the output
//     transform A^T M A never leaves the lane -- no exchange between waves, no LDS pass (the F(2x2) kernel spends 4.9 of 39 us per unit there)
//   * input: per 16-channel slice the 18 x 18 pixel halo arrives as raw rows by `buffer_load ... lds` (requested a slice or two ahead: two
//     slices per request since round 6, 128 contiguous bytes per eight adjacent lanes), the producing layer's pending InstanceNorm / ReLU (or pending residual join) is applied, and
//     B^T d B is formed in TWO passes through LDS between the matrix instructions: rows (288 items of 6 -> 6: threads 0..287, i.e.
//     waves 0-3 and half of wave 4) into L, columns (384 items: waves 2..7, one transform line each) into V[position][chunk][tile].
//     An A fragment is then ONE conflict-free ds_read_b128 per position and 16 channels
//   * weights: transformed in double and packed on the host in exactly the fragment order (wino4_pack.h); no two waves share a
//     weight, so they go global -> registers directly (one contiguous 1 KiB load per position), four to five positions ahead through a
//     ring of SIX register quads (fb[6]) that lives across slices and units; 2.36 MB per layer and group, L2-resident
//   * two barriers per slice (L complete, V complete); the slice's 144 matrix instructions per wave run between them
//   * epilogue: since round 5 the transformed input is the M operand, so a lane holds ONE channel of the four tiles of a tile row and
//     the 16 adjacent lanes of a group store 64 contiguous bytes of a pixel (2 048 store requests per unit instead of 8 192)
// Units are independent (nothing handed over, nothing co-resident assumed): persistent blocks walk the units round-robin; a launch with a
// thin last round is dealt out as one sequence of slices instead ("stream-K": a unit cut by a share boundary is computed in two parts).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "fav_internal.h"
#include "wino4_pack.h"

namespace fav {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_DEVICES = 64;
constexpr int W4_MEET_MAX = 256;             // meeting places of ks_ws / ks_cnt (conv3_wino4_ksplit_bytes): one per share boundary of a stream-K launch
constexpr int W4_SHARES = 252;               // shares of a stream-K launch: a constant, so that the cuts (and with them the output's bits) do not depend on the device
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

// LDS layouts, in 16-byte SLOTS (4 words).  A ds_read_b128 is served in four NON-contiguous groups of 16 lanes -- {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} and the same + 32 -- over 16 slots (64 banks); a ds_write_b128 in eight contiguous groups of 8 lanes over 8
// slots (32 banks) (MI355X_MICROARCH.md, LDS).  Every layout below makes the lanes of a group hit different slots:
//   V[position][kq][tile]            one slot per (tile, 4-channel chunk kq): the A-fragment read of lane (kq = lane >> 4, tile = lane & 15)
//                                    is slot `lane` of the position's 64 -- dense, conflict-free for any grouping
//   L[kq][tile row ty][line i][x]    one slot per (pixel, chunk); tile-row pitch 113 = 1 (mod 16), chunk-plane pitch 464 = 0 (mod 16):
//                                    the column pass reads slot ty + 4 tx + ... for the 16 tiles of a group (all four ty in every group)
// (the first version -- pixel-major with a 20-word pitch, checked against CONTIGUOUS 16-lane groups -- measured 38 % of its LDS cycles
//  as bank conflicts: profiles/r04g_wino4_pmc.json)
constexpr int W4_VPOS = 256;                   // words per transform position: 4 chunks x 16 tiles x 4
constexpr int W4_VBUF = 36 * W4_VPOS;          // 9216 words per V buffer
constexpr int W4_LLINE = 18 * 4;               // 72: one row-transformed line (18 columns) of a tile row and chunk
constexpr int W4_LTY = 113 * 4;                // 452: the six lines of a tile row (108 slots) padded to 113
constexpr int W4_LKQ = 464 * 4;                // 1856: the four tile rows of a chunk plane (452 slots) padded to 464
constexpr int W4_LBUF = 4 * W4_LKQ;            // 7424 words
constexpr int W4_RBUF = 6 * 288 * 4;           // landing area of the raw rows (buffer_load ... lds): [row][thread < 288] 16 bytes each: 6912 words
constexpr int W4_SMEM = 2 * W4_VBUF + W4_LBUF + W4_RBUF;      // 32768 words = 131 072 B (+ a second landing area -- the second slice of a
                                                              //  paired request, or the skip rows of a pending join --, + 2 CIN words of
                                                              //  pending scale / shift: 159 744 B)
static_assert((W4_LTY / 4) % 16 == 1 && (W4_LKQ / 4) % 16 == 0 && W4_LTY >= 6 * W4_LLINE && W4_LKQ >= 4 * W4_LTY, "pitches of L");

struct Wino4Args {
    const float* in; const float* wpk; const float* bias; const float* scale1; const float* shift1;
    float* out; float2* partials; int* counts;
    const float* skip; float* zout;      // MODE 2 (pending residual join), as in kernels_wino.hip
    int OWp;
    int IH, IW, IWp, CIN, OH, OW, units_x, units_y, relu1;
    // stream != 0: the launch's units x slices are dealt out as ONE sequence of 16-channel slices, an equal share per block (launch_wino4_t);
    // a unit cut by a share boundary is computed in two parts -- the slices before the cut by one block, those behind it by the next --
    // whose partial outputs meet in ks_ws: whichever part is finished second adds the other's to its own (+ bias) and stores the unit
    // (shares: how many -- the constant W4_SHARES, NOT a function of the device or of how many blocks this launch may use: every device
    //  and every side-queue setting must cut the same units at the same slices, i.e. produce the same bits)
    int stream, shares; float* ks_ws; int* ks_cnt;
    long long* dbg;
    // InstanceNorm statistics as accumulators (fav_internal.h, Affine::acc1): stat_acc = where THIS launch adds its units' (sum, sum of
    // squares) -- instead of partials --; acc1 ... count1 = the pending InstanceNorm of the INPUT, formed in the prologue (MODE 1)
    long long* stat_acc; const long long* acc1; long long* acc1_zero; const float* gamma1; const float* beta1; float eps1; int count1;
    // WIDE instantiations only (layers with more than 128 filters: the VR checkpoints "have more filters", README.md:141): the output
    // channels are computed in `groups` groups of 128, a work unit = (pixel unit, group); COUT = 128 * groups is the channel pitch of
    // out / partials, wpk holds one packed block (conv_wino4_pack of 128 filters) per group
    int COUT, groups;
};

// 16-byte write-through store (sc1), as in kernels_conv.hip: the partial outputs of a K-split unit are published with these +
// `s_waitcnt vmcnt(0)` + a relaxed agent-scope counter instead of plain stores + a release fence (which would write back the whole
// XCD L2's dirty lines, i.e. the output tiles of the round before)
__device__ __forceinline__ void w4_store16_wt(void* ptr, v4f v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}

// B^T d and A^T m (wino4_pack.h) with every multiply-add WRITTEN as one: the three instantiations of the kernel (plain input, pending
// normalisation, pending join) must round alike -- a network computes the same bits whether a residual join is launched or left pending
// (test_pending_residual_joins_give_the_bits_of_the_launched_ones) -- and the compiler's own choice of contractions differs with context
// (round 6: the same operations on PAIRS of values -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, each component rounded exactly like
//  the single-value instruction; four-wide vectors are legalised to single-value instructions, two-wide ones to the packed forms)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f w4_fma(float a, v2f b, v2f c) { return __builtin_elementwise_fma(v2f{a, a}, b, c); }
__device__ __forceinline__ void w4_bt2(const v2f d[6], v2f v[6])
{
    const v2f e1 = w4_fma(-2.25f, d[2], d[4]), o1 = w4_fma(-1.6875f, d[1], 0.75f * d[3]);
    const v2f e2 = w4_fma(-0.5625f, d[2], d[4]), o2 = w4_fma(-0.84375f, d[1], 1.5f * d[3]);
    v[0] = w4_fma(1.265625f, d[0], w4_fma(-2.8125f, d[2], d[4]));
    v[1] = e1 + o1; v[2] = e1 - o1; v[3] = e2 + o2; v[4] = e2 - o2;
    v[5] = w4_fma(1.265625f, d[1], w4_fma(-2.8125f, d[3], d[5]));
}
__device__ __forceinline__ void w4_bt(const v4f d[6], v4f v[6])
{
    v2f dl[6], dh[6], vl[6], vh[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { dl[i] = d[i].xy; dh[i] = d[i].zw; }
    w4_bt2(dl, vl); w4_bt2(dh, vh);
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = v4f{vl[i].x, vl[i].y, vh[i].x, vh[i].y};
}
__device__ __forceinline__ void w4_at(const v2f m[6], v2f y[4])
{
    const v2f s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    y[0] = (m[0] + s1) + s2;
    y[1] = w4_fma(1.5f, d2, 0.75f * d1);
    y[2] = w4_fma(2.25f, s2, 0.5625f * s1);
    y[3] = w4_fma(3.375f, d2, w4_fma(0.421875f, d1, m[5]));
}

// MODE 0: plain input; MODE 1: pending per-channel scale / shift (+ ReLU) of the producing convolution's InstanceNorm; MODE 2: pending
// residual join z = skip + scale * y + shift (res_add_kernel's operations in its order), written out once as the next block's skip
// (VAR: timing experiments only -- FAV_W4_VAR: bit 0 the K loop requests no weights, bit 1 stages nothing, bit 2 no barriers inside the slice;
//  16 (MODE 2): the joined tensor is not stored -- K loop 55.5 us instead of 65.5: what its stores cost.  Tried against that in round 4
//  (profiles/r4p_*, r4q_*, r4r_*): the stores in an epilogue pass over the unit (re-read, add, store: 80 -> 92 us per launch, the traffic of
//  all CUs in one burst), behind the slice's first barrier out of LDS (-> 88 us), with the non-temporal hint (-1.5 us);
//  the results are garbage, the timeline of FAV_WINO_DBG says what each part costs)
template <int MODE, int VAR = 0, bool WIDE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_wino4_kernel(const Wino4Args p)
{
    constexpr bool AFF = MODE != 0, JOIN = MODE == 2;
    // PAIR (round 6; not with a pending join, whose skip rows have the room): the raw rows are requested for TWO slices at a time -- the
    // 128 contiguous bytes of a pixel's slices (a, a + 1), a even, by eight adjacent lanes, every other slice -- into a landing area of
    // twice the size.  70.8 -> 69.3 us per launch (profiles/r9a_w4_two_slices_per_request_ab.log); VAR bit 3 = one slice per request
    constexpr bool PAIR = (VAR & 8) == 0 && MODE != 2;
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Vs = smem;                        // [2][W4_VBUF]
    float* const Ls = smem + 2 * W4_VBUF;          // [W4_LBUF]
    float* const Rs = Ls + W4_LBUF;                // landing area of the raw rows
    float* const Ss = Rs + W4_RBUF;                // MODE 2: landing area of the skip rows (same shape)
    float* const aff = Ss + ((MODE == 2 || PAIR) ? W4_RBUF : 0);      // [2][CIN]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int CIN = p.CIN, nslices = CIN >> 4;

    int dbi = 0;
#define DBG_T() { if (p.dbg && t == 0 && dbi < 21) p.dbg[blockIdx.x * 24 + dbi++] = wall_clock64(); }
    DBG_T();
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (MODE == 1 && p.acc1 != nullptr) {
        // the input's InstanceNorm from its accumulators (what in_finalize_kernel computes from partials: mean = S1 / M, biased variance =
        // S2 / M - mean^2 in double): channel i sums the copies' words as integers -- exact, whatever order the atomics arrived in
        for (int i = t; i < CIN; i += NT) {
            long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
            for (int cp = 0; cp < STAT_COPIES; ++cp) {
                const longlong2* a = reinterpret_cast<const longlong2*>(p.acc1 + ((size_t)cp * CIN + i) * 4);
                const longlong2 lo = a[0], hi = a[1];
                w0 += lo.x; w1 += lo.y; w2 += hi.x; w3 += hi.y;
            }
            const double s1 = ((double)w1 * 4294967296.0 + (double)w0) * (1.0 / 1099511627776.0);
            const double s2 = ((double)w3 * 4294967296.0 + (double)w2) * (1.0 / 1099511627776.0);
            const double mean = s1 / (double)p.count1;
            double var = s2 / (double)p.count1 - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const double sc = stat_acc_poisoned(w1, w3) ? (double)NAN : (double)p.gamma1[i] / sqrt(var + (double)p.eps1);
            aff[i] = (float)sc; aff[CIN + i] = (float)((double)p.beta1[i] - mean * sc);
            if (blockIdx.x == 0) {      // the other parity's accumulators were last read a frame ago: zero for the next frame
#pragma unroll
                for (int cp = 0; cp < STAT_COPIES; ++cp) {
                    longlong2* z = reinterpret_cast<longlong2*>(p.acc1_zero + ((size_t)cp * CIN + i) * 4);
                    z[0] = longlong2{0, 0}; z[1] = longlong2{0, 0};
                }
            }
        }
    } else if (AFF) for (int i = t; i < CIN; i += NT) { aff[i] = p.scale1[i]; aff[CIN + i] = p.shift1[i]; }
    const float lo1 = (MODE == 1 && p.relu1) ? 0.f : -INFINITY;

    // stage 1 (rows): item = (pixel pix = 18 ty + x of the 4 x 18 (tile row, raw column) grid, 16-byte channel chunk cq) -> raw rows
    // 4 ty .. 4 ty + 5 of column x, six transformed lines out.  288 items = threads 0..287 (waves 0-3 and half of wave 4), numbered
    // t = (pix >> 3) * 32 + cq * 8 + (pix & 7): eight consecutive lanes write eight consecutive pixels of one chunk plane
    const bool has1 = t < 288;
    const int cq = (t >> 3) & 3;
    const int pix1 = min((t >> 5) * 8 + (t & 7), 71), ty1 = (pix1 * 3641) >> 16, x1 = pix1 - ty1 * 18;
    float* const l1 = Ls + cq * W4_LKQ + ty1 * W4_LTY + x1 * 4;
    const float* const affr = aff + cq * 4;
    // stage 2 (columns): item = (tile m, line i, channel chunk kq) -> columns 4 tx .. 4 tx + 5 of line i, six positions out.
    // 16 x 6 x 4 = 384 items = waves 2..7 (line i = wave - 2; waves 0, 1 carry the row pass only, waves 5..7 this pass only);
    // lane = (kq = lane >> 4, m = lane & 15), as the matrix operand
    const bool has2 = wave >= 2;
    const int i2 = max(wave - 2, 0);
    const int m2 = lane & 15, kq2 = lane >> 4;
    const float* const l2 = Ls + kq2 * W4_LKQ + (m2 >> 2) * W4_LTY + 4 * (m2 & 3) * 4 + i2 * W4_LLINE;
    float* const v2 = Vs + lane * 4 + 6 * i2 * W4_VPOS;
    // matrix operands: lane = (k quarter kq = lane >> 4, tile m = lane & 15): V[p][kq][m] -- step j of a slice multiplies channel
    // 16 s + 4 kq + j; weights likewise (wino4_pack.h: wave = 2 w + nt there): lane * 16 + [wave * 1024 + (s * 36 + p) * 8192]
    const float* const aA = Vs + lane * 4;
    const int wlo = lane * 16, wso = wave * 1024;

    // the weight ring lives across units: the last slice of a unit requests the first positions of slice 0 -- the next unit's
    v4f fb[6];
    bool ring_primed = false;
    int ring_grp = 0;                  // WIDE: the output group whose weights the ring holds
    const int upix = p.units_x * p.units_y;
    if (AFF) __syncthreads();
    // one work item: slices s0 .. s1 - 1 of unit u -- all of them (meet < 0), or one of the two parts of a unit that a share boundary
    // cuts (meet = the boundary's meeting place in ks_ws / ks_cnt, part = 0: the slices before the cut, 1: behind it)
    auto work = [&](const int ug, const int s0, const int s1, const int meet, const int part) {
        const int nsl = s1 - s0;
        const bool whole = meet < 0;
        if (s0 != 0) ring_primed = false;                   // (the ring holds the first positions of a slice 0)
        // WIDE: ug = (pixel unit u, output group grp) -- group-minor when whole units are walked (the groups of a pixel unit run on
        // neighbouring blocks of one XCD and share its input in L2), group-major in a stream-K sequence (a share stays inside a group)
        int u = ug, grp = 0;
        if (WIDE) {
            if (p.stream) { grp = ug / upix; u = ug - grp * upix; } else { u = ug / p.groups; grp = ug - u * p.groups; }
            if (grp != ring_grp) { ring_primed = false; ring_grp = grp; }
        }
        const int wgb = (p.CIN >> 4) * 36 * 8192;           // bytes of one group's packed weights
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk) + (WIDE ? (size_t)grp * (wgb >> 2) : 0), 0, wgb, 0x00020000);
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(JOIN ? p.skip : p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(JOIN ? p.zout : const_cast<float*>(p.in), 0, p.IH * p.IWp * p.CIN * 4, 0x00020000);
        const int uy = u / p.units_x, ux = u - uy * p.units_x;
        const int oy0 = uy * 16, ox0 = ux * 16;
        DBG_T();   /* unit start */
        // no padding: input pixel (oy0 + r, ox0 + c) for halo (r, c); coordinates past the image only feed outputs past the image
        // (never stored), so they are clamped instead of masked
        // (two registers instead of six offsets -- row a = row 0 + min(a, rows before the image ends) * row pitch: the accumulators leave
        //  no room, and a spilled value's reload inside the K loop is a `s_waitcnt vmcnt(0)` that drains the weight prefetch)
        const int ho0 = ((min(oy0 + 4 * ty1, p.IH - 1) * p.IWp + min(ox0 + x1, p.IW - 1)) * CIN + cq * 4) * 4;
        const int hrmax = max(p.IH - 1 - (oy0 + 4 * ty1), 0);
        const int hrow = p.IWp * CIN * 4;
#define ho_(a_) (ho0 + min((a_), hrmax) * hrow)
        // the row REQUESTS use another item numbering than the row pass: request lane d asks for pixel d >> 2, chunk d & 3 -- four
        // adjacent lanes = the 64 contiguous bytes of one pixel's slice, one request instead of four 16-byte ones -- and lands it in slot d;
        // the row pass's thread reads slot 4 pix + cq (slots 64 w .. 64 w + 63 are requested AND read by wave w, but by different lanes: the
        // request's `s_waitcnt vmcnt` precedes the read in the same wave; the barrier in between serves the readers of L)
        const int dpix = min(t >> 2, 71), dty = (dpix * 3641) >> 16, dx = dpix - dty * 18;
        const int hd0 = ((min(oy0 + 4 * dty, p.IH - 1) * p.IWp + min(ox0 + dx, p.IW - 1)) * CIN + (t & 3) * 4) * 4;
        const int hdmax = max(p.IH - 1 - (oy0 + 4 * dty), 0);
#define hd_(a_) (hd0 + min((a_), hdmax) * hrow)
        // MODE 2: which of the item's six rows this thread writes to the joined tensor: rows 4 ty .. 4 ty + 3 of columns 0..15 -- the
        // unit's own 16 x 16 pixels -- plus the halo fringe (rows 16, 17 / columns 16, 17) where no other unit follows
        int zm = 0;
        const bool lastx = ux == p.units_x - 1, lasty = uy == p.units_y - 1;
        if (JOIN) {
            const bool cv = ox0 + x1 < p.IW && (x1 < 16 || lastx);
#pragma unroll
            for (int a = 0; a < 6; ++a) zm |= (cv && oy0 + 4 * ty1 + a < p.IH && (a < 4 || (ty1 == 3 && lasty)) ? 1 : 0) << a;
        }

        v4f sc, sh;
        float* const land = Rs + wave * 256;               // (wave-uniform: M0 of the LDS loads; lane i lands 16 i bytes further)
        const float* const rread = Rs + (pix1 * 4 + cq) * 4;
        constexpr int RROW = 288 * 4;                       // words between the landing places of two rows
        constexpr int SKO = W4_RBUF;                        // skip rows: the same places one landing area further on
#define W4_LOAD_RAW(q_, slice_)                                                                     \
        { _Pragma("unroll") for (int a = 0; a < 6; ++a) q_[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(irs, ho_(a), (slice_) * 64, 0)); }
#define W4_LOAD_SKIP(x_, slice_)                                                                    \
        { if (JOIN) { _Pragma("unroll") for (int a = 0; a < 6; ++a) x_[a] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(srs, ho_(a), (slice_) * 64, 0)); } }
        // the NEXT slices' raw rows bypass the register file: buffer_load ... lds into the landing area, requested a whole slice before
        // their use, read back by the requesting thread behind an explicit s_waitcnt (the compiler does not see that dependency;
        // loads return in order and the ring's younger weight loads are in flight by then)
#define W4_REQ_RAW(slice_)                                                                          \
        { asm volatile("" ::: "memory");      /* (never above the reads of the rows these loads replace) */ \
          if (JOIN) { _Pragma("unroll") for (int a = 0; a < 6; ++a)     /* the skip rows first: the rows' arrival implies theirs */ \
              __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(land + SKO + a * RROW), 16, hd_(a), (slice_) * 64, 0, 0); } \
          _Pragma("unroll") for (int a = 0; a < 6; ++a)                                             \
              __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (lds_ptr_t)(land + a * RROW), 16, hd_(a), (slice_) * 64, 0, 0); }
#define W4_TAKE_RAW(q_)                                                                             \
        { _Pragma("unroll") for (int a = 0; a < 6; ++a) q_[a] = *reinterpret_cast<const v4f*>(rread + a * RROW); }
        // PAIR: slot d of the landing area = (pixel d >> 3, 16-byte piece d & 7 of the pair's 128 bytes: slice a for pieces 0..3, a + 1 for
        // 4..7); wave w < 4 requests slots 128 w .. 128 w + 127 (two instructions per row), wave 4 slots 512 .. 575 -- the slots its OWN row-pass
        // threads read (pixels 16 w .. 16 w + 15), so that, as above, the wave's own s_waitcnt orders request and read.  The per-lane offsets are
        // formed where they are used (every other slice; the opaque copy of the lane keeps them out of the loop's live registers)
        constexpr int RROW2 = 576 * 4;
        const float* const rread2 = Rs + (pix1 * 8 + cq) * 4;
        float* const land2 = Rs + wave * 512;
#define W4_REQ_PAIR(a_)                                                                             \
        { asm volatile("" ::: "memory");                                                            \
          int ln_ = lane; asm volatile("" : "+v"(ln_));                                             \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) { if (j == 0 || wave < 4) {                 \
              const int dp_ = 16 * wave + 8 * j + (ln_ >> 3), dy_ = (dp_ * 3641) >> 16, dx_ = dp_ - 18 * dy_;                     \
              const int b_ = ((min(oy0 + 4 * dy_, p.IH - 1) * p.IWp + min(ox0 + dx_, p.IW - 1)) * CIN + (ln_ & 7) * 4) * 4;       \
              const int mx_ = max(p.IH - 1 - (oy0 + 4 * dy_), 0);                                   \
              _Pragma("unroll") for (int a = 0; a < 6; ++a)                                         \
                  __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (lds_ptr_t)(land2 + j * 256 + a * RROW2), 16, b_ + min(a, mx_) * hrow, (a_) * 64, 0, 0); } } }
        // (tried, profiles/r9b_*: the pair's 54 request instructions dealt out over ALL eight waves behind the slice's first barrier -- seven
        //  per wave instead of twelve on four -- 634 against 637 frames/s: the requests' place in the waves' queues is not what costs)
#define W4_TAKE_PAIR(q_, slice_)                                                                    \
        { const float* const rr_ = rread2 + ((slice_) & 1) * 16;                                     \
          _Pragma("unroll") for (int a = 0; a < 6; ++a) q_[a] = *reinterpret_cast<const v4f*>(rr_ + a * RROW2); }
        // (the skip rows of a pending join are read one at a time where they are added: six more live rows would not fit the registers)
#define W4_SKIP_LDS(a_) (*reinterpret_cast<const v4f*>(rread + SKO + (a_) * RROW))
#define W4_AFF(slice_)                                                                              \
        { if (AFF) { sc = *reinterpret_cast<const v4f*>(affr + (slice_) * 16); sh = *reinterpret_cast<const v4f*>(affr + CIN + (slice_) * 16); } }
        // pending transform of the raw rows; MODE 2: z = fma(y, scale, shift) + skip, stored where the mask says so (elsewhere the offset is
        // out of the buffer's range and the hardware drops the store)
#define W4_PEND(q_, xs_, slice_)                                                                    \
        { _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                           \
            if (MODE == 1) { const v2f fl_ = __builtin_elementwise_fma(q_[a].xy, sc.xy, sh.xy), fh_ = __builtin_elementwise_fma(q_[a].zw, sc.zw, sh.zw); \
                             q_[a].x = fmaxf(fl_.x, lo1); q_[a].y = fmaxf(fl_.y, lo1); q_[a].z = fmaxf(fh_.x, lo1); q_[a].w = fmaxf(fh_.y, lo1); } \
            if (JOIN) { const v4f x1_ = xs_(a);                                                     \
                        q_[a].x = fmaf(q_[a].x, sc.x, sh.x) + x1_.x; q_[a].y = fmaf(q_[a].y, sc.y, sh.y) + x1_.y;                \
                        q_[a].z = fmaf(q_[a].z, sc.z, sh.z) + x1_.z; q_[a].w = fmaf(q_[a].w, sc.w, sh.w) + x1_.w;                \
                        if (!(VAR & 16) && (a < 4 || lasty)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q_[a]), zrs, (zm & (1 << a)) ? ho_(a) : (int)0xFFFFFFF0, (slice_) * 64, 0); } } }
#define W4_COMMIT1(q_)                                                                              \
        { v4f l_[6]; w4_bt(q_, l_);                                                                 \
          _Pragma("unroll") for (int i = 0; i < 6; ++i) *reinterpret_cast<v4f*>(l1 + i * W4_LLINE) = l_[i]; }
        // column pass in two halves: the six reads, and -- a segment of matrix instructions later -- transform and writes
#define W4_S2_READ(c_)                                                                              \
        { _Pragma("unroll") for (int k = 0; k < 6; ++k) c_[k] = *reinterpret_cast<const v4f*>(l2 + k * 4); }
#define W4_S2_DONE(c_, nb_)                                                                         \
        { v4f o_[6]; w4_bt(c_, o_);                                                                 \
          _Pragma("unroll") for (int j = 0; j < 6; ++j) *reinterpret_cast<v4f*>(v2 + (nb_) * W4_VBUF + j * W4_VPOS) = o_[j]; }

        v4f fa[4];
#define W4_READ_A(slot_, par_, pos_) { fa[slot_] = *reinterpret_cast<const v4f*>(aA + (par_) * W4_VBUF + (pos_) * W4_VPOS); }
#define W4_LOAD_B(slot_, sl_, pos_)                                                                 \
        { fb[slot_] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, wso + ((sl_) * 36 + (pos_)) * 8192, 0)); }

        // ---- prologue: the item's first slice through both transform passes into V[0]
        {
            v4f qa[6], xa[6];
            if (has1) { W4_LOAD_SKIP(xa, s0); W4_LOAD_RAW(qa, s0); }
            if (!ring_primed) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { W4_LOAD_B(q, s0, q); }
                ring_primed = true;
            }
            if (has1) {
                W4_AFF(s0);
#define W4_SKIP_REG(a_) xa[a_]
                W4_PEND(qa, W4_SKIP_REG, s0); W4_COMMIT1(qa);
#undef W4_SKIP_REG
                if (!PAIR) W4_REQ_RAW(min(s0 + 1, s1 - 1));
            }
            if (PAIR && wave < 5) W4_REQ_PAIR((s0 + 1) & ~1);        // (the pair that holds slice s0 + 1)
            __syncthreads();
            if (has2) { v4f c2[6]; W4_S2_READ(c2); W4_S2_DONE(c2, 0); }
        }
        v4f acc[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        W4_READ_A(0, 0, 0); W4_READ_A(1, 0, 1);
        DBG_T();   /* loop start */
        const long long ck0 = p.dbg ? clock64() : 0, wk0 = p.dbg ? wall_clock64() : 0;

        // ---- K loop: per 16-channel slice 36 positions of 4 matrix instructions per wave; between them the NEXT slice is staged:
        //   position 6      the raw rows (requested a slice ago) read back from the landing area, committed to L (pending transform, rows
        //                   of B^T d), the rows of the slice AFTER the next requested                  (threads 0..287)
        //   position 12     barrier: L complete
        //   position 14     column pass into the other V buffer (waves 2..7)        after position 35   barrier: V complete
        // (the last slice stages a copy of itself into the idle buffer: no branches inside the loop body)
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
        // positions [from_, to_) of the slice, two at a time: A fragments of the pair after, weights seven / eight positions ahead, 8 matrix
        // instructions alternating between the pair's two accumulators (the SIMD's other wave fills the gaps a dependent pair leaves)
        // (segments of a few positions each, so that every loop is small enough to be unrolled completely: the accumulators are
        //  indexed by the position)
#define W4_POSITIONS(from_, to_)                                                                    \
        { _Pragma("unroll") for (int pos = (from_); pos < (to_); pos += 2) {                       \
            if (pos + 2 < 36) { W4_READ_A((pos + 2) % 4, par, pos + 2); W4_READ_A((pos + 3) % 4, par, pos + 3); }    \
            if (!(VAR & 1)) {                                                                       \
            if (pos + 4 < 36) { W4_LOAD_B((pos + 4) % 6, s, pos + 4); } else { W4_LOAD_B((pos + 4) % 6, sw, pos + 4 - 36); } \
            if (pos + 5 < 36) { W4_LOAD_B((pos + 5) % 6, s, pos + 5); } else { W4_LOAD_B((pos + 5) % 6, sw, pos + 5 - 36); } } \
            W4_FENCE();                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[pos % 4][j], fb[pos % 6][j], acc[pos], 0, 0, 0); \
                acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(pos + 1) % 4][j], fb[(pos + 1) % 6][j], acc[pos + 1], 0, 0, 0); \
            }                                                                                       \
            W4_FENCE(); } }
        for (int sl = 0; sl < nsl; ++sl) {
            const int s = s0 + sl, par = sl & 1;
            const int sn = min(s + 1, s1 - 1);
            const int sw = s + 1 < s1 ? s + 1 : 0;            // (weights: whatever this block computes next starts at a slice 0)
            const int sn2 = min(s + 2, s1 - 1);
            W4_POSITIONS(0, 6);
            if (sl == 0) {      // the rows requested in the prologue (by other waves): landed, and everybody knows
                if (PAIR ? wave < 5 : has1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                __syncthreads();
            }
            if (PAIR) {
                // slice sn = s + 1 is one half of a pair that landed a slice or two ago; behind an even slice's take both halves of the
                // area's pair are used up and the next pair (s + 2, s + 3) is requested into it
                if (has1) { v4f qa[6]; W4_TAKE_PAIR(qa, sn); W4_AFF(sn); W4_PEND(qa, W4_SKIP_LDS, sn); W4_COMMIT1(qa); }
                if (wave < 5 && !(s & 1) && s + 2 < s1) W4_REQ_PAIR(s + 2);
            } else
            if (!(VAR & 2) && has1) {
                // slice sn's rows were requested a slice ago (in the prologue for s = 0: six weight loads have followed): read back, pending
                // transform, rows of B^T d, into L -- in one piece (nothing is held across matrix instructions: the registers are the
                // accumulators'; the SIMD's other wave covers the LDS round trip)
                v4f qa[6];
                W4_TAKE_RAW(qa);
                W4_AFF(sn); W4_PEND(qa, W4_SKIP_LDS, sn); W4_COMMIT1(qa);
                W4_REQ_RAW(sn2);
            }
            W4_POSITIONS(6, 12);
            if (!(VAR & 4)) __syncthreads();
            W4_POSITIONS(12, 14);
            if (!(VAR & 2) && has2) { v4f c2[6]; W4_S2_READ(c2); W4_S2_DONE(c2, par ^ 1); }
            W4_POSITIONS(14, 36);
            if (PAIR ? wave < 5 : has1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this slice's row requests (30 weight loads ago) have landed: the barrier tells the readers
            if (!(VAR & 4)) __syncthreads();
            W4_READ_A(0, par ^ 1, 0); W4_READ_A(1, par ^ 1, 1);
        }
#undef W4_POSITIONS
#undef W4_FENCE
        DBG_T();   /* loop end */
        if (p.dbg && t == 0) { p.dbg[blockIdx.x * 24 + 21] += clock64() - ck0; p.dbg[blockIdx.x * 24 + 22] += wall_clock64() - wk0; }
        DBG_T();
#undef ho_
#undef hd_
#undef W4_LOAD_RAW
#undef W4_REQ_RAW
#undef W4_TAKE_RAW
#undef W4_REQ_PAIR
#undef W4_TAKE_PAIR
#undef W4_SKIP_LDS
#undef W4_S2_READ
#undef W4_S2_DONE
#undef W4_LOAD_SKIP
#undef W4_AFF
#undef W4_PEND
#undef W4_COMMIT1
#undef W4_READ_A
#undef W4_LOAD_B

        // ---- output transform, in the lane.  The matrix instructions are issued with the transformed INPUT as the M operand (round 5;
        // rounds 4's form had the weights there and a lane held four consecutive channels of one tile -- its 16-byte stores were 8 192
        // separate requests per unit, four NON-adjacent lanes per 64 contiguous bytes): acc[6 i + j][r] = M[i][j] of tile (row g = lane >> 4,
        // column r), output channel 16 wave + (lane & 15) -- a lane holds ONE channel of the four tiles of a tile row, and the 16 adjacent
        // lanes of a group store 16 consecutive channels of a pixel, 64 contiguous bytes, as one request (2 048 per unit).
        // Y = A^T M A (4 x 4), bias, NHWC store, per-unit InstanceNorm partials (mean, M2, count) like the other kernels
        const int cn = lane & 15, g = lane >> 4;
        const int nrows = max(0, min(16, p.OH - oy0)), ncols = max(0, min(16, p.OW - ox0));
        const int nv = nrows * ncols;
        const int cpb = WIDE ? p.COUT * 4 : 512;             // bytes per output pixel
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.OH * p.OWp * cpb, 0x00020000);
        const bool inside = nv == 256;         // (wave-uniform: nine units in ten lie wholly inside the image and skip every per-pixel test)
        const int oyb = oy0 + 4 * g;           // first output row of this lane's tile row
        const int cb = (WIDE ? grp * 128 : 0) + wave * 16 + cn;      // this lane's channel
        float y[4][4][4];                      // [row a][column b][tile column r]
        {
            const float bv = whole ? p.bias[cb] : 0.f;
            const v2f bv2 = v2f{bv, bv};
#pragma unroll
            for (int r = 0; r < 4; r += 2) {       // tile columns r, r + 1 as one pair of values
                v2f Q[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const v2f mcol[6] = {r ? acc[j].zw : acc[j].xy, r ? acc[6 + j].zw : acc[6 + j].xy, r ? acc[12 + j].zw : acc[12 + j].xy,
                                         r ? acc[18 + j].zw : acc[18 + j].xy, r ? acc[24 + j].zw : acc[24 + j].xy, r ? acc[30 + j].zw : acc[30 + j].xy};
                    v2f o[4]; w4_at(mcol, o);
                    Q[0][j] = o[0]; Q[1][j] = o[1]; Q[2][j] = o[2]; Q[3][j] = o[3];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    v2f o[4]; w4_at(Q[a], o);
#pragma unroll
                    for (int b = 0; b < 4; ++b) { const v2f ob = o[b] + bv2; y[a][b][r] = ob.x; y[a][b][r + 1] = ob.y; }
                }
            }
        }
        if (!whole) {
            // a part of the unit's input channels: the output transform is linear, so this is a PARTIAL output.  If the other part is
            // there already, add it; otherwise publish this one (write-through, drained) and count -- and add the other's after all
            // if it arrived in the meantime.  a + b = b + a: the sum does not depend on who adds
            // (the meeting place is laid out in the order the threads hold the values -- [pixel (a, b) of the tile][wave][lane] x 16 bytes: every
            //  store / load instruction of a wave is one contiguous KiB; the other part's thread of the same index holds the same outputs)
            float* const mine = p.ks_ws + ((size_t)meet * 2 + part) * 32768 + (wave * 64 + lane) * 4;
            const float* const other = p.ks_ws + ((size_t)meet * 2 + (part ^ 1)) * 32768 + (wave * 64 + lane) * 4;
            int* const flag = reinterpret_cast<int*>(aff + 2 * CIN);
            if (t == 0) *flag = __hip_atomic_load(p.ks_cnt + meet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            bool second = *flag == 1;
            __syncthreads();                   // (the flag word is free again)
            if (!second) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        w4_store16_wt(mine + (a * 4 + b) * 2048, v4f{y[a][b][0], y[a][b][1], y[a][b][2], y[a][b][3]});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (t == 0) *flag = __hip_atomic_fetch_add(p.ks_cnt + meet, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                second = *flag == 1;
                __syncthreads();
                if (!second) { DBG_T(); return; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (t == 0) __hip_atomic_store(p.ks_cnt + meet, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            const float bv = p.bias[cb];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const v4f o = *reinterpret_cast<const v4f*>(other + (a * 4 + b) * 2048);
                    y[a][b][0] = (y[a][b][0] + o.x) + bv; y[a][b][1] = (y[a][b][1] + o.y) + bv;
                    y[a][b][2] = (y[a][b][2] + o.z) + bv; y[a][b][3] = (y[a][b][3] + o.w) + bv;
                }
        }
        // pixel (a, 4 r + b) of the lane's tile row.  Nine units in ten lie wholly inside the image (wave-uniform) and take the first form:
        // no per-pixel test.  In a ragged unit the COLUMNS past the image are skipped by wave-uniform branches (every lane of a wave has
        // the same columns) and the ROWS past it -- they differ between the lane groups -- get an offset past the buffer (the hardware
        // drops the store) and a zero weight in the statistics: four per-lane conditions, not sixty-four
        float sm = 0.f, mu = 0.f, m2 = 0.f;
        if (inside) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ro = ((oyb + a) * p.OWp + ox0) * cpb + cb * 4;      // byte offset of column 0 of the unit; columns follow one pixel (512 B) apart
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, y[a][b][r]), ors, ro + r * 4 * cpb + b * cpb, 0, 0);
                        sm += y[a][b][r];
                    }
            }
            if (p.partials != nullptr) {
                // per channel: a lane holds 64 of the unit's 256 pixels, the lanes cn + 16, cn + 32, cn + 48 the rest
                sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
                mu = sm * (1.f / 256.f);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int b = 0; b < 4; ++b) { const float d = y[a][b][r] - mu; m2 = fmaf(d, d, m2); }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool rv = oyb + a < p.OH;
                const int ro = rv ? ((oyb + a) * p.OWp + ox0) * cpb + cb * 4 : (int)0xFFFF0000;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c < ncols) {                                           // (wave-uniform)
                        const float v = y[a][c & 3][c >> 2];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ors, ro + c * cpb, 0, 0);
                        sm += rv ? v : 0.f;
                    }
                }
            }
            if (p.partials != nullptr) {
                sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
                mu = nv ? sm / (float)nv : 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const bool rv = oyb + a < p.OH;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        if (c < ncols) {
                            const float d = y[a][c & 3][c >> 2] - mu;
                            m2 = rv ? fmaf(d, d, m2) : m2;
                        }
                    }
                }
            }
        }
        if (p.partials != nullptr) {
            m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
            if (p.stat_acc != nullptr) {
                // the unit's (n mean, M2 + n mean^2) -- the terms in_finalize_kernel sums in double -- as 2^-40 fixed point in two words each
                // (value * 2^40 = hi * 2^32 + lo, 0 <= lo < 2^32), added with integer atomics: the sum does not depend on the order.
                // All four lanes of a channel hold (mu, m2): lane group g adds word g (0: sum lo, 1: sum hi, 2: squares lo, 3: squares hi)
                const double nd = (double)nv, mud = (double)mu;
                const double v = (g & 2) ? (double)m2 + nd * mud * mud : nd * mud;
                const double tt = v * 1099511627776.0;
                const double hi = floor(tt * (1.0 / 4294967296.0));
                const bool fin = fabs(tt) < 9.0e18;                          // (false for NaN / Inf too: STAT_NONFINITE, fav_internal.h)
                const long long word = !fin ? ((g & 1) ? STAT_NONFINITE : 0ll) : (g & 1) ? (long long)hi : (long long)(tt - hi * 4294967296.0);
                long long* const dst = p.stat_acc + ((size_t)(blockIdx.x & (STAT_COPIES - 1)) * (WIDE ? p.COUT : 128) + cb) * 4 + g;
                __hip_atomic_fetch_add(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (g == 0) p.partials[(size_t)u * (WIDE ? p.COUT : 128) + cb] = make_float2(mu, m2);
        }
        if (p.partials != nullptr && p.stat_acc == nullptr && t == 0 && grp == 0) p.counts[u] = nv;
        // (no barrier here: the last slice ended with one, the epilogue touches no LDS, the next prologue has its own)
        DBG_T();   /* epilogue end */
    };
    const int units = WIDE ? upix * p.groups : upix;
    if (!p.stream) {
        for (int u = lb; u < units; u += gridDim.x) work(u, 0, nslices, -1, 0);
    } else {
        // share sh of the units x nslices slices: [g, g1); every share is at least a unit long, so a unit has at most two parts: the
        // slices before the cut close share sh (meeting place sh), those behind it open share sh + 1
        const long long tot = (long long)units * nslices;
        for (int sh = lb; sh < p.shares; sh += gridDim.x) {
            int g = (int)(tot * sh / p.shares);
            const int g1 = (int)(tot * (sh + 1) / p.shares);
            while (g < g1) {
                const int u = g / nslices, s0 = g - u * nslices, s1 = min(nslices, s0 + (g1 - g));
                const bool whole = s0 == 0 && s1 == nslices;
                work(u, s0, s1, whole ? -1 : (s0 == 0 ? sh : sh - 1), s0 == 0 ? 0 : 1);
                g += s1 - s0;
            }
        }
    }
    if (p.dbg && t == 0) p.dbg[blockIdx.x * 24 + 23] = dbi;
#undef DBG_T
}

// FAV_WINO_DBG=n: in-kernel timeline of the n-th launch (same report as the F(2x2) kernel's)
void wino4_debug_report(const long long* hbuf, int grid, int mode)
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
    fprintf(stderr, "WINO4DBG mode=%d grid=%d units=%d  K loop: %.3f GHz;  per unit: prologue %.2f  loop %.2f  epilogue %.2f us;  last block ends at %.2f us\n",
            mode, grid, items, wk ? ck / (wk * 10.0) : 0.0, items ? sum[0] / items : 0.0, items ? sum[1] / items : 0.0, items ? sum[3] / items : 0.0, tend);
}

template <int MODE, bool WIDE>
int launch_wino4_t(const Wino4Args& a0, int reserve_cus, hipStream_t st)
{
#ifdef FAV_DIAG
    // timing experiments (make DIAG=1 only): these instantiations skip parts of the kernel and produce garbage
    static const int var = diag_env("FAV_W4_VAR") ? atoi(diag_env("FAV_W4_VAR")) : 0;
    const auto kern = (!WIDE && MODE == 1 && var == 1) ? conv3_wino4_kernel<1, 1> : (!WIDE && MODE == 1 && var == 2) ? conv3_wino4_kernel<1, 2> : (!WIDE && MODE == 1 && var == 3) ? conv3_wino4_kernel<1, 3> :
                      (!WIDE && MODE == 1 && var == 7) ? conv3_wino4_kernel<1, 7> : (!WIDE && MODE == 1 && var == 6) ? conv3_wino4_kernel<1, 6> :
                      (!WIDE && MODE == 2 && var == 16) ? conv3_wino4_kernel<2, 16> :
                      (MODE != 2 && var == 8) ? conv3_wino4_kernel<MODE, MODE != 2 ? 8 : 0, WIDE> : conv3_wino4_kernel<MODE, 0, WIDE>;
    const bool pair = MODE != 2 && !(var & 8);
#else
    const auto kern = conv3_wino4_kernel<MODE, 0, WIDE>;
    const bool pair = MODE != 2;
#endif
    const size_t lds = (size_t)(W4_SMEM + ((MODE == 2 || pair) ? W4_RBUF : 0) + 2 * a0.CIN + 4) * sizeof(float);
    const int dv = cur_dev();
    static int cus[MAX_DEVICES] = {};
    if (!cus[dv]) {
        FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int occ = 0; int prop_cus = 0;
        FAV_HIP(hipDeviceGetAttribute(&prop_cus, hipDeviceAttributeMultiprocessorCount, dv));
        FAV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 512, lds));
        if (occ < 1) { set_error("winograd F(4x4) conv: kernel does not fit on a CU"); return FAV_EHIP; }
        cus[dv] = prop_cus;          // one block per CU
    }
    const int units = a0.units_x * a0.units_y * (WIDE ? a0.groups : 1);
    int grid = std::min(units, std::max(1, cus[dv] - reserve_cus));
    Wino4Args a = a0; a.dbg = nullptr;
    // A CU holds one unit at a time, so a launch takes ceil(units / grid) rounds of one unit time -- and three of the ten layers at
    // 1280x720 have 273-286 units for 256 CUs (a unit is 256 pixels, the layers 64.8-66.9 k): two rounds for 1.07-1.12 rounds of work.
    // When the last round is that thin the units x slices are dealt out as one sequence instead, an equal share of 16-channel
    // slices per block ("stream-K"): a share boundary inside a unit makes two parts of it, computed by two neighbouring blocks, the
    // second to finish adds them (ks_ws / ks_cnt: one meeting place per boundary).  Costs a second prologue and output transform per
    // block and the hand-over, i.e. ~0.3 unit times: worth it when the last round is less than ~0.6 full
    // (FAV_W4_GRID: fewer blocks than CUs -- the tests reach the many-shares case with small images through it)
    // The number of shares is a CONSTANT (W4_SHARES = 252: the 256 CUs of an MI355X minus the four the look-ahead mode leaves the side
    // queues) -- not a function of the device's CU count or of reserve_cus: which units are cut, and at which slice, decides the order
    // the channels of a cut unit are summed in, i.e. the fp32 bits of the output.  Every device, partition mode and side-queue
    // setting (FAV_SIDE_CUS) therefore produces the same bits; with fewer blocks than shares a block simply takes several shares
    static const bool no_stream = diag_env("FAV_W4_NO_STREAM") != nullptr;
    static const int grid_cap = diag_env("FAV_W4_GRID") ? atoi(diag_env("FAV_W4_GRID")) : 0;
    if (grid_cap > 0) grid = std::min(grid, std::max(1, grid_cap));
    const int shares = grid_cap > 0 ? std::max(1, std::min(W4_SHARES, grid_cap)) : W4_SHARES;
    const int rounds = (units + shares - 1) / shares, rem = units - (rounds - 1) * shares;
    a.stream = (rounds >= 2 && rem * 5 <= shares * 3 && shares <= W4_MEET_MAX && a0.ks_ws && a0.ks_cnt && !no_stream) ? 1 : 0;
    a.shares = shares;
    if (a.stream) grid = std::min(grid, shares);
    static int dbg_n = diag_env("FAV_WINO_DBG") ? atoi(diag_env("FAV_WINO_DBG")) : 0;
    static long long* dbuf = nullptr;
    const bool dbg = dbg_n > 0 && --dbg_n == 0;
    if (dbg) { FAV_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 512 * 24 * 8)); FAV_HIP(hipMemsetAsync(dbuf, 0, 512 * 24 * 8, st)); a.dbg = dbuf; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    FAV_LAUNCH_CHECK("conv3_wino4_kernel");
    if (dbg) {
        std::vector<long long> hb((size_t)512 * 24);
        FAV_HIP(hipStreamSynchronize(st)); FAV_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
        wino4_debug_report(hb.data(), grid, MODE);
    }
    return FAV_OK;
}

}  // namespace

// any number of 128-filter groups (the canonical network has one; checkpoints with more filters -- README.md:141 -- two or more)
bool conv3_wino4_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups)
{
    return k == 3 && stride == 1 && pad == 0 && ups == 0 && stages <= 1 && cin_pitch % 16 == 0 && cin_pitch >= 16 && cin_pitch <= 512 &&
           cout == coutp && cout % 128 == 0 && cout >= 128 && cout <= 1024;
}
int conv3_wino4_tiles(int OH, int OW) { return ((OH + 15) / 16) * ((OW + 15) / 16); }

int launch_conv3_wino4(const ConvLaunch& c, const float* wpk, int* counts, hipStream_t st)
{
    FAV_REQUIRE(conv3_wino4_eligible(c.CIN, c.COUT, c.COUTp, c.KH, c.stride, c.pad, c.pre.stages, c.ups) && c.KH == c.KW && !c.final_mode && !c.stuff && wpk,
                "winograd F(4x4) conv: not eligible");
    FAV_REQUIRE((long long)(c.IH + 1) * c.IWp * c.CIN < (1ll << 29), "winograd F(4x4) conv: tensor too large for 32-bit byte offsets");
    FAV_REQUIRE(c.OH == c.IH - 2 && c.OW == c.IW - 2, "winograd F(4x4) conv: bad geometry");
    FAV_REQUIRE((long long)(c.OH + 1) * (c.OWp > 0 ? c.OWp : c.OW) * c.COUT < (1ll << 29), "winograd F(4x4) conv: output too large for 32-bit byte offsets");
    Wino4Args a;
    a.in = c.in; a.wpk = wpk; a.bias = c.bias; a.scale1 = c.pre.scale1; a.shift1 = c.pre.shift1; a.relu1 = c.pre.relu1;
    a.out = c.out; a.partials = reinterpret_cast<float2*>(c.partials); a.counts = counts;
    a.IH = c.IH; a.IW = c.IW; a.IWp = c.IWp; a.CIN = c.CIN; a.OH = c.OH; a.OW = c.OW;
    a.units_x = (c.OW + 15) / 16; a.units_y = (c.OH + 15) / 16;
    a.dbg = nullptr;
    a.skip = c.join_skip; a.zout = c.join_out; a.OWp = c.OWp > 0 ? c.OWp : c.OW;
    a.ks_ws = c.ks_ws; a.ks_cnt = c.ks_cnt; a.stream = 0; a.shares = 1;
    a.COUT = c.COUT; a.groups = c.COUT / 128;
    a.stat_acc = c.stat_acc; a.acc1 = c.pre.acc1; a.acc1_zero = c.pre.acc1_zero; a.gamma1 = c.pre.gamma1; a.beta1 = c.pre.beta1; a.eps1 = c.pre.eps1; a.count1 = c.pre.count1;
    // (accumulator mode reuses the `partials != null` test of the epilogue: any non-null value selects the statistics, stat_acc the form)
    if (c.stat_acc != nullptr && a.partials == nullptr) a.partials = reinterpret_cast<float2*>(c.stat_acc);
    FAV_REQUIRE(c.pre.acc1 == nullptr || (c.join_skip == nullptr && c.pre.stages == 1 && c.pre.count1 > 0), "winograd F(4x4) conv: accumulator-form InstanceNorm on an unsupported input");
    if (c.join_skip != nullptr) {
        FAV_REQUIRE(c.join_out != nullptr && c.pre.stages == 1 && c.pre.relu1 == 0, "winograd F(4x4) conv: a pending residual join needs its output tensor and exactly one pending normalisation");
        FAV_REQUIRE(a.groups == 1, "winograd F(4x4) conv: a pending residual join is formed by 128-filter layers only");
        return launch_wino4_t<2, false>(a, c.reserve_cus, st);
    }
    if (a.groups > 1) return c.pre.stages >= 1 ? launch_wino4_t<1, true>(a, c.reserve_cus, st) : launch_wino4_t<0, true>(a, c.reserve_cus, st);
    return c.pre.stages >= 1 ? launch_wino4_t<1, false>(a, c.reserve_cus, st) : launch_wino4_t<0, false>(a, c.reserve_cus, st);
}

}  // namespace fav
