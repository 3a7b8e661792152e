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
// the line exactly like CFilter.h:1416-1464; iir_rows_kernel), eigenvalue, CMatrix::normalize with its order
// dependent min/max quirk (CMatrix.h:721-737, reproduced with an exact parallel formulation) and
// CMatrix::avg, an order-dependent fp32 running sum (CMatrix.h:1245-1251) evaluated exactly AND in parallel: inside one binade
// the running sum is an integer multiple of its ulp, so every addend is a two-state (parity) transducer and the chain is a scan of
// transducer compositions (avg_scan_kernel below; pinned against the scalar loop by test_sequential_sum_bit_exact).
#include <mutex>

#include "fav_internal.h"
#include "consistency_pixel.h"

namespace fav {
namespace {

// WHERE a long-lived block of the look-ahead mask may run (round 6; measured with scripts/cu_probe2.hip, profiles/r8_xcd_dispatch_probe.log).
// The hardware deals the blocks of a launch to the eight XCDs round-robin by block index, starting -- for every launch of a queue -- at an
// XCD that belongs to the QUEUE (block i -> XCD (q0 + i) mod 8; q0 differs between queues and may change when the runtime re-maps them).
// The network's persistent grids are 252 blocks for 256 CUs (fav_net::reserve_cus = 4): the XCDs q0_main .. q0_main + 3 are FULL (32 blocks
// on 32 CUs), the four free CUs sit in the XCDs q0_main + 4 .. q0_main + 7, one each.  A block of a side-queue kernel that lands on a full
// XCD waits for a CU, takes the first that frees up and holds it against the block the next network kernel has dealt there (one CU
// missing = that kernel's slowest block starts a whole round late: first layer 190 -> 298 us, d128 91 -> 155 us, F(4x4) launches 70 -> 78 us
// depending on which kernel the pass happened to overlap).  So the mask's long-lived kernels (the two recursive passes, the one-block
// scans) are launched as EIGHT blocks -- one per XCD whatever the queue's q0 -- and every block looks up where it is (XCC_ID) and
// where the network's queue starts (a word its first kernel of the frame writes, prep_input_kernel): the blocks on the four free XCDs
// share the work, the others leave at once.  At most ONE mask block sits on a free XCD at any time -- next to 31 network blocks on 32 CUs.
__device__ __forceinline__ int xcd_share(const int* q0_main, int shares)
{
    // this block's share of the work in [0, shares), or -1: not on one of the XCDs the network leaves a CU free on (shares <= 4)
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int rel = ((int)(xcc & 7u) - *q0_main) & 7;
    return rel >= 4 && rel - 4 < shares ? rel - 4 : -1;
}

// one block per XCD?  (what xcd_share relies on: eight consecutive blocks of a launch land on eight different XCDs)
__global__ void xcd_probe_kernel(int* out)
{
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 15u);
}

constexpr int MAX_DEVICES = 64;
inline int cur_dev() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEVICES) ? d : 0; }

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
// (the three planes are written with a row pitch `pw` that is a multiple of 4 floats: the smoothing passes read 16 bytes per lane)
// (the three planes leave TRANSPOSED -- [W][ph], ph a multiple of 4 floats -- through a 32x32 LDS tile: the X smoothing pass wants the
//  image's rows as lines with the line index fastest, see iir_col.  Wide kernels walk their items with a grid stride.)
__global__ __launch_bounds__(256) void moments_t_kernel(const uint8_t* rgb_hwc, float* dxxT, float* dyyT, float* dxyT, int W, int H, int ph)
{
    __shared__ float tl[3][32][33];
    const int ntx = (W + 31) / 32, nty = (H + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int it = blockIdx.x; it < ntx * nty; it += gridDim.x) {
        const int x0 = (it % ntx) * 32, y0 = (it / ntx) * 32;
        for (int j = ty; j < 32; j += 8) {
            const int y = y0 + j, x = x0 + tx;
            if (y >= H || x >= W) continue;
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
            tl[0][j][tx] = sxx; tl[1][j][tx] = syy; tl[2][j][tx] = sxy;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int x = x0 + j, y = y0 + tx;
            if (x < W && y < H) { const size_t o = (size_t)x * ph + y; dxxT[o] = tl[0][tx][j]; dyyT[o] = tl[1][tx][j]; dxyT[o] = tl[2][tx][j]; }
        }
        __syncthreads();
    }
}

// recursiveSmoothX / recursiveSmoothY (CFilter.h:1416-1464): one LANE per line, the recurrences of :1426-1437 / :1451-1462 in their
// exact order (they round at every step: nothing along a line can be re-associated); lines are independent.
//
// Round 6 form.  The 64 lanes of a wave are 64 NEIGHBOURING lines and the array is stored with the line index fastest: sample x of line
// l lives at a[x * pitch + l], so every step of a wave is one 256-byte load (and store) -- the X pass therefore runs on the TRANSPOSED
// planes (moments_t_kernel writes them that way), a tile transpose follows, and the Y pass runs on the row-major planes.  Loads run D steps
// ahead through a register ring; the result overwrites the input (the anti-causal sweep needs m(x+1), m(x+2) of the ORIGINAL line: two
// registers), the causal half v1 goes to `scratch`.
// Why not a contiguous line per lane (rounds 4-5: 16-byte pieces through a ring of register groups): a wave's load then touches 64 cache
// lines for 16 bytes each and the pass is bound by the CU's address / tag pipe (~70 cycles per step where the arithmetic needs ~30), which
// is tolerable only with ONE wave per CU -- and that footprint is what the look-ahead mode cannot afford:
// what a pass costs the network next to it is not its arithmetic but the CUs it touches.  The network's persistent kernels (252 blocks
// that need a CU's whole register file: 8 waves x 250 VGPRs for the first layer, 8 x 256 for the F(4x4) stage) cannot start a block on a
// CU that holds even ONE foreign wave, and a pass lives ~100 us: 36 / 60 one-wave blocks spread over as many CUs kept that many of the
// first layer's blocks waiting (first layer 190 -> 233 us, d64 100 -> 140 us; attribution: profiles/r8e_4arg_attribution.log -- without
// the two passes the 4-argument mode runs at 632 frames/s instead of 595, without the mask's WIDE kernels at all only at 636).
// Work unit = one wave = 64 lines of one plane (a "task"); a wave walks tasks blockIdx.x * waves-per-block + wave, + gridDim.x *
// waves-per-block, ...: launched as `pack_cus` blocks of sixteen waves a pass occupies at most that many CUs (the ones the network
// leaves free, fav_net::reserve_cus) and hides its memory latency behind its own waves; launched as one block per task it is the widest
// form (the stand-alone mask), with a deeper ring instead.
// The ring is fed by hand and lives in LDS: `global_load_lds` (the DMA form of a load: wave-uniform LDS base + lane x 4 bytes) lands the
// sample D - 1 steps before its step reads it back, and every read waits with an EXACT `s_waitcnt vmcnt(N)`.  Left to the compiler
// the same loop with a register ring waits for (nearly) everything in flight at the top of every round (`s_waitcnt vmcnt(2)`,
// `vmcnt(1)`: its bookkeeping does not carry a partly drained counter across the loop's back edge), i.e. the ring reaches one round
// ahead whatever its depth, and a step costs ~350 cycles instead of the ~40 its arithmetic needs (372 us per pass packed on four CUs,
// profiles/r8h_*); a register ring fed through inline assembly is not an option either -- the register allocator copies a slot
// between the peeled round and the loop while its load is still in flight (seen in the ISA; the copy reads stale bits).
// Vector memory operations of a wave complete in issue order, so "the sample has landed" is a count of what was issued since: the
// sample of step s is requested in step s - D + 1 (the slot read in step s - D is refilled one step LATER, when its value has long been
// consumed), behind it that step's store and D - 2 steps of one request + one store each (causal sweep: 2 D - 3 younger operations) or
// two requests + one store (anti-causal sweep, which streams m and v1: 3 D - 5).  The first round waits for the prologue's requests.
typedef __attribute__((address_space(3))) void* lds_vptr_t;

template <int N>
__device__ __forceinline__ void vm_wait_mem()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// A LANE carries FOUR neighbouring lines (one 16-byte piece per step: a wave = 256 lines) as two packed pairs: v_pk_mul_f32 / v_pk_add_f32
// round each component exactly like the scalar instruction, so a line's arithmetic is unchanged while a wave-instruction does the work
// of two.  Why four: a CU's address pipe takes ~16 cycles per wave-wide memory instruction whatever its width -- with one dword per
// lane the packed pass was bound by exactly that (9 waves x (1 request + 1 store) per step = 288 cycles per step, 360 us per pass,
// profiles/r8j_*), with 16 bytes per lane the same traffic is a quarter of the instructions and the pass is bound by its arithmetic.
typedef float f2 __attribute__((ext_vector_type(2)));
struct F4 { f2 lo, hi; };
__device__ __forceinline__ F4 f4_from(const float4& q) { F4 r; r.lo = f2{q.x, q.y}; r.hi = f2{q.z, q.w}; return r; }
__device__ __forceinline__ F4 operator*(float s, const F4& v) { F4 r; r.lo = f2{s, s} * v.lo; r.hi = f2{s, s} * v.hi; return r; }
__device__ __forceinline__ F4 operator+(const F4& a, const F4& b) { F4 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi; return r; }
__device__ __forceinline__ F4 operator-(const F4& a, const F4& b) { F4 r; r.lo = a.lo - b.lo; r.hi = a.hi - b.hi; return r; }

// The ring is READ by hand as well (ds_read_b128 through inline assembly, one step ahead, with its own lgkmcnt wait): the compiler does
// not know which DMA a read of LDS depends on and puts `s_waitcnt vmcnt(0)` in front of every LDS read it emits itself -- a full drain
// of the ring per step (seen in the ISA).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ F4 f4_from(const v4f& q) { F4 r; r.lo = f2{q.x, q.y}; r.hi = f2{q.z, q.w}; return r; }
template <int OFF>
__device__ __forceinline__ void lds_read_issue(v4f& q, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"(OFF) : "memory"); }
__device__ __forceinline__ void lds_read_wait(v4f& q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q) : : "memory"); }

template <int D>
__device__ __forceinline__ void iir_col(float* mb, float* vb, int lane, size_t pitch, int n, const IIR& c, float4* ring /* [2][D][64], this wave's */)
{
    // mb, vb: sample 0 of the wave's first line in the plane / in the scratch plane (wave-uniform); this lane's four lines are 4 * lane floats further on
    static_assert(D >= 4 && 3 * D - 8 <= 63 && 2 * D * 1024 <= 65536, "ring depth against the 6-bit vmcnt / the 16-bit ds offset");
    const float c0f = 0.5f - c.k * c.pm, cd = c.a2 - c.e2;
    float4* ringm = ring; float4* ringv = ring + D * 64;
    const unsigned la = (unsigned)(unsigned long)(lds_vptr_t)ring + 16u * (unsigned)lane;      // LDS byte address of this lane's piece of slot 0 (m ring; the v ring is D KB further on)
    constexpr int VO = D * 1024;
    const F4 zero = {f2{0.f, 0.f}, f2{0.f, 0.f}};
    auto request = [&](const float* src, float4* slot) {         // one DMA: 64 lanes x 16 bytes -> slot[0..63]
        __builtin_amdgcn_global_load_lds(src + 4 * lane, (lds_vptr_t)slot, 16, 0, 0);
    };
    auto put = [&](float* dst, const F4& v) {
        v4f t = {v.lo.x, v.lo.y, v.hi.x, v.hi.y};
        *reinterpret_cast<v4f*>(dst) = t;
    };
    // Step s: [sample s + 1 has landed?] [read it from LDS, asynchronously] [request sample s - 1 + D into the slot step s - 1 read] [the
    // arithmetic of step s] [store] [the LDS read has arrived].  The sample of step s + 1 was requested in step s - D + 2: behind it that
    // step's store and D - 3 steps of (one request + one store) or (two requests + one store): 2 D - 5 / 3 D - 8 younger operations.
    // ---------------------------------------------------------------- causal sweep, x = 0 .. n-1
    {
#pragma unroll
        for (int k = 0; k < D - 1; ++k) request(mb + (size_t)min(k, n - 1) * pitch, ringm + k * 64);
        F4 a0 = zero, a1 = zero, mp = zero;
        // x = 0: v1 = (0.5 - k pm) m0;  x = 1: v1 = k (m1 + pm m0) + (a2 - e2) v1(0);  then the three-term recurrence
        auto step = [&](const F4& mx, int x, bool may_be_first) {
            F4 a = c.k * (mx + c.pm * mp) + c.a2 * a1 - c.e2 * a0;
            if (may_be_first && x < 2) a = x == 0 ? c0f * mx : c.k * (mx + c.pm * mp) + cd * a1;
            a0 = a1; a1 = a; mp = mx;
            put(vb + (size_t)x * pitch + 4 * lane, a);
        };
        v4f q, qn;
        vm_wait_mem<D - 2>(); lds_read_issue<0>(q, la); lds_read_wait(q);                 // sample 0
        int x = 0;
        auto round = [&](bool first) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (first) vm_wait_mem<D - 3>(); else vm_wait_mem<2 * D - 5>();
                switch ((k + 1) % D) {                                                       // the slot of sample x + k + 1 (compile-time after unrolling)
#define FAV_RD(j) case j: lds_read_issue<(j) * 1024>(qn, la); break;
                    FAV_RD(0) FAV_RD(1) FAV_RD(2) FAV_RD(3) FAV_RD(4) FAV_RD(5) FAV_RD(6) FAV_RD(7) FAV_RD(8) FAV_RD(9) FAV_RD(10) FAV_RD(11) FAV_RD(12) FAV_RD(13) FAV_RD(14) FAV_RD(15)
                    FAV_RD(16) FAV_RD(17) FAV_RD(18) FAV_RD(19) FAV_RD(20) FAV_RD(21) FAV_RD(22) FAV_RD(23)
#undef FAV_RD
                }
                request(mb + (size_t)min(x + k - 1 + D, n - 1) * pitch, ringm + ((k + D - 1) % D) * 64);      // (clamped: always issued, the counts stay exact)
                step(f4_from(q), x + k, first);
                lds_read_wait(qn); q = qn;
            }
            x += D;
        };
        if (n >= D) round(true);
        while (x + D <= n) round(false);
        // the tail: fewer than D samples; sample x is in q already, the others are read in order (everything has landed)
        vm_wait_mem<0>();
#pragma unroll
        for (int k = 0; k < D; ++k) if (x + k < n) { if (k) { const float4 t = ringm[k * 64 + lane]; step(f4_from(t), x + k, true); } else step(f4_from(q), x + k, true); }
    }
    // ---------------------------------------------------------------- anti-causal sweep, x = n-1 .. 0; m is overwritten with v1 + v2
    {
        vm_wait_mem<0>();                                      // every v1 of the causal sweep has left, every slot has been read
#pragma unroll
        for (int k = 0; k < D - 1; ++k) { const size_t o = (size_t)max(n - 1 - k, 0) * pitch; request(mb + o, ringm + k * 64); request(vb + o, ringv + k * 64); }
        // v2(n-1) = (0.5 + k pm) m(n-1);  v2(n-2) = k ((pp - e2) m(n-1)) + (a2 - e2) v2(n-1);  then the recurrence on the ORIGINAL m
        F4 b0 = zero, b1 = zero, mo1 = zero, mo2 = zero;         // b0 = v2(x+1), b1 = v2(x+2), mo1 = m(x+1), mo2 = m(x+2)
        const float c1f = 0.5f + c.k * c.pm, cpe = c.pp - c.e2;
        auto step = [&](const F4& mx, const F4& vx, int x, bool may_be_last) {
            F4 bv = c.k * (c.pp * mo1 - c.e2 * mo2) + c.a2 * b0 - c.e2 * b1;
            if (may_be_last && x == n - 1) bv = c1f * mx;
            if (may_be_last && x == n - 2) bv = c.k * (cpe * mo1) + cd * b0;
            put(mb + (size_t)x * pitch + 4 * lane, vx + bv);
            b1 = b0; b0 = bv; mo2 = mo1; mo1 = mx;
        };
        v4f qm, qv, qmn, qvn;
        vm_wait_mem<2 * (D - 2)>(); lds_read_issue<0>(qm, la); lds_read_issue<VO>(qv, la); lds_read_wait(qm); lds_read_wait(qv);      // sample n - 1
        int x = n - 1;
        auto round = [&](bool first) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (first) vm_wait_mem<2 * (D - 3)>(); else vm_wait_mem<3 * D - 8>();
                switch ((k + 1) % D) {
#define FAV_RD(j) case j: lds_read_issue<(j) * 1024>(qmn, la); lds_read_issue<VO + (j) * 1024>(qvn, la); break;
                    FAV_RD(0) FAV_RD(1) FAV_RD(2) FAV_RD(3) FAV_RD(4) FAV_RD(5) FAV_RD(6) FAV_RD(7) FAV_RD(8) FAV_RD(9) FAV_RD(10) FAV_RD(11) FAV_RD(12) FAV_RD(13) FAV_RD(14) FAV_RD(15)
                    FAV_RD(16) FAV_RD(17) FAV_RD(18) FAV_RD(19) FAV_RD(20) FAV_RD(21) FAV_RD(22) FAV_RD(23)
#undef FAV_RD
                }
                const size_t o = (size_t)max(x - k + 1 - D, 0) * pitch;
                request(mb + o, ringm + ((k + D - 1) % D) * 64); request(vb + o, ringv + ((k + D - 1) % D) * 64);
                step(f4_from(qm), f4_from(qv), x - k, first);
                lds_read_wait(qmn); lds_read_wait(qvn); qm = qmn; qv = qvn;
            }
            x -= D;
        };
        if (n >= D) round(true);
        while (x - (D - 1) >= 0) round(false);
        vm_wait_mem<0>();
#pragma unroll
        for (int k = 0; k < D; ++k) if (x - k >= 0) {
            if (k) { const float4 tm = ringm[k * 64 + lane], tv = ringv[k * 64 + lane]; step(f4_from(tm), f4_from(tv), x - k, true); }
            else step(f4_from(qm), f4_from(qv), x - k, true);
        }
        vm_wait_mem<0>();
    }
}

// (lines beyond nlines inside a lane's group of four are the row pitch's padding -- a multiple of 4 floats: computed and stored like the
//  others, read by nobody)
// Packed form (look-ahead path): eight blocks, of which the `nwork` (<= 4) on the free XCDs work (xcd_share), each with `nw` working waves.
// The blocks are 1024 threads: the waves beyond nw wait at the closing barrier (a wave at a barrier holds its registers and issues
// nothing) -- a block that fills a CU's register file can only be placed on a CU that is completely idle, and nothing joins it there.
// With four-wave blocks the same placement measured first layer 190 -> 270-310 us (profiles/r8p_4arg_packed_xcd_ab.log, FAV_IIR_FILL=0):
// a small block is placed at once on ANY CU of its XCD that has room, next to whatever light kernel of the network runs there.
// Wide form: one block (one wave) per task.
template <int D, int MAXT>
__global__ __launch_bounds__(MAXT) void iir_cols_kernel(float* plane0, size_t plane_stride, float* scratch0, int nlines, int n, int pitch, int groups, int nwork, int nw, const int* q0_main, IIR c)
{
    extern __shared__ __attribute__((aligned(16))) float4 iir_lds[];      // [nw][2][D][64]
    if (n < 2) return;
    const int wb = q0_main ? xcd_share(q0_main, nwork) : (int)blockIdx.x;      // (look-ahead path: eight blocks, those on the free XCDs work -- see xcd_share)
    if (wb < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform by construction; says so to the compiler: bases stay in scalar registers)
    if (wave < nw) {
        float4* ring = iir_lds + (size_t)wave * 2 * D * 64;
        for (int task = wb * nw + wave; task < 3 * groups; task += nwork * nw) {      // groups: of 256 lines
            const int plane = task / groups, line0 = (task - plane * groups) * 256;
            if (line0 + 4 * lane >= nlines) continue;                 // (the wave's other lanes go on; a DMA of an inactive lane writes nothing)
            iir_col<D>(plane0 + (size_t)plane * plane_stride + line0, scratch0 + (size_t)plane * plane_stride + line0, lane, (size_t)pitch, n, c, ring);
        }
    }
    if ((int)blockDim.x > 64 * nw) __syncthreads();      // (a CU-filling launch: the idle waves keep their registers until the working ones are through; a wave at a barrier issues nothing)
}

// [R][C] (row pitch pin) -> [C][R] (row pitch pout) per plane, 32x32 LDS tiles, both sides coalesced (between the two smoothing passes)
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, size_t plane_stride, int R, int C, int pin, int pout, int planes)
{
    __shared__ float tl[32][33];
    const int ntx = (C + 31) / 32, nty = (R + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int it = blockIdx.x; it < ntx * nty * planes; it += gridDim.x) {
        const int pl = it / (ntx * nty), rem = it - pl * ntx * nty;
        const float* ip = in + (size_t)pl * plane_stride;
        float* op = out + (size_t)pl * plane_stride;
        const int c0 = (rem % ntx) * 32, r0 = (rem / ntx) * 32;
        for (int j = ty; j < 32; j += 8)
            if (r0 + j < R && c0 + tx < C) tl[j][tx] = ip[(size_t)(r0 + j) * pin + c0 + tx];
        __syncthreads();
        for (int j = ty; j < 32; j += 8)
            if (c0 + j < C && r0 + tx < R) op[(size_t)(c0 + j) * pout + r0 + tx] = tl[tx][j];
        __syncthreads();
    }
}

// eigenvalue (consistencyChecker.cpp:69-77) from the three smoothed planes (row-major again, row pitch pw), written in image order
// [H][W] -- the order CMatrix::normalize's scan and CMatrix::avg depend on -- together with the maximum of every NB consecutive elements
// (pass 1 of the normalize scan below: one launch and one read of the map less)
constexpr int NB = 1024;   // elements per block in the normalize scans

__device__ __forceinline__ float eigen_min(float a, float c, float b)
{
    const float temp = (float)(0.5 * (double)(a + c));                      // consistencyChecker.cpp:73
    const float temp2 = temp * temp + b * b - a * c;
    return temp2 < 0.0f ? 0.0f : temp - sqrtf(temp2);
}

__global__ __launch_bounds__(256) void eigen_blockmax_kernel(const float* p3, size_t ps, int pw, float* corners, int H, int W, float* bmax, int nb)
{
    __shared__ float sh[4];
    const size_t n = (size_t)H * W;
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const size_t i0 = (size_t)b * NB + (size_t)threadIdx.x * 4;
        float mx = -INFINITY;
        if (i0 < n) {
            const int y = (int)(i0 / W), x = (int)(i0 - (size_t)y * W);
            if ((W & 3) == 0) {                          // (then i0 .. i0 + 3 lie in one row and every address is 16-byte aligned)
                const size_t o = (size_t)y * pw + x;
                const float4 a = *reinterpret_cast<const float4*>(p3 + o), c = *reinterpret_cast<const float4*>(p3 + ps + o), bb = *reinterpret_cast<const float4*>(p3 + 2 * ps + o);
                float4 r;
                r.x = eigen_min(a.x, c.x, bb.x); r.y = eigen_min(a.y, c.y, bb.y); r.z = eigen_min(a.z, c.z, bb.z); r.w = eigen_min(a.w, c.w, bb.w);
                *reinterpret_cast<float4*>(corners + i0) = r;
                mx = fmaxf(fmaxf(r.x, r.y), fmaxf(r.z, r.w));
            } else {
                int yy = y, xx = x;
                for (int j = 0; j < 4 && i0 + j < n; ++j) {
                    const size_t o = (size_t)yy * pw + xx;
                    const float r = eigen_min(p3[o], p3[ps + o], p3[2 * ps + o]);
                    corners[i0 + j] = r; mx = fmaxf(mx, r);
                    if (++xx == W) { xx = 0; ++yy; }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) bmax[b] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        __syncthreads();
    }
}

// CMatrix::normalize's scan (CMatrix.h:727-729):
//     if (v > cmax) cmax = v; else if (v < cmin) cmin = v;        cmax0 = -30000, cmin0 = +30000
// cmax = max(-30000, max_i v_i).  An element updates cmin only when it is NOT a strict running
// maximum, i.e. v_i <= max(-30000, v_0..v_{i-1}).  Both are computed exactly in parallel:
// pass 1: per-block maxima (eigen_blockmax_kernel above); pass 2: exclusive prefix max over blocks (single block) then, per block,
// an in-block exclusive running max and the min over non-record elements.
// single block: exclusive prefix max of bmax (seeded with -30000) -> bpre; total max -> mm[0]
__global__ __launch_bounds__(1024) void prefixmax_kernel(const float* bmax, int nb, float* bpre, float* mm, const int* q0_main)
{
    if (q0_main && xcd_share(q0_main, 1) != 0) return;      // (look-ahead path: eight blocks, the one on the first free XCD works -- see xcd_share)
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

__global__ __launch_bounds__(256) void quirkmin_kernel(const float* v, size_t n, const float* bpre, float* bmin, int nb)
{
    __shared__ float sh[NB];
    __shared__ float red[4];
    __shared__ float wmax[4];
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const size_t base = (size_t)b * NB;
        for (int j = threadIdx.x; j < NB; j += 256) sh[j] = base + j < n ? v[base + j] : -INFINITY;
        __syncthreads();
        // each thread owns 4 consecutive elements; running max of everything before them: exclusive prefix max over the 256 threads'
        // own maxima (max is exact and associative: wave scan by shuffles, then the waves before this one) -- the first version re-read
        // up to 1020 LDS words per thread (0.15 ms per mask)
        const int j0 = threadIdx.x * 4;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float inc = fmaxf(fmaxf(sh[j0], sh[j0 + 1]), fmaxf(sh[j0 + 2], sh[j0 + 3]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float up = __shfl_up(inc, o); if (lane >= o) inc = fmaxf(inc, up); }
        if (lane == 63) wmax[wv] = inc;
        __syncthreads();
        float pre = __shfl_up(inc, 1);
        if (lane == 0) pre = -INFINITY;
        for (int k = 0; k < wv; ++k) pre = fmaxf(pre, wmax[k]);
        pre = fmaxf(pre, bpre[b]);
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
        if (threadIdx.x == 0) bmin[b] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void minreduce_kernel(const float* bmin, int nb, float* mm, const int* q0_main)
{
    if (q0_main && xcd_share(q0_main, 1) != 0) return;      // (look-ahead path: eight blocks, the one on the first free XCD works -- see xcd_share)
    __shared__ float red[4];
    float mn = 30000.0f;
    for (int j = threadIdx.x; j < nb; j += 256) mn = fminf(mn, bmin[j]);
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) mm[1] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));   // cmin
}

// normalize(0, 1) (CMatrix.h:731-736) fused with step 1 of the parallel CMatrix::avg below (the fp64 sum and the "has an element the
// transducer cannot take" flag of every chunk of 256 NORMALISED elements -- same lanes, same reduction order as avg_chunk_sum_kernel):
// one read and one launch less per mask.  csum == nullptr: normalisation only.
__device__ __forceinline__ float normalize_one(float x, float cmin, float t) { x -= cmin; x *= t; x += 0.0f; return x; }

__global__ __launch_bounds__(256) void normalize_sum_kernel(float* v, size_t n, const float* mm, double* csum, int* cbad)
{
    const float cmax = mm[0], cmin = mm[1];
    float t = cmax - cmin;
    if (t == 0.f) t = 1.f; else t = (1.0f - 0.0f) / t;
    const int lane = threadIdx.x & 63;
    const size_t nblk = (n + 1023) / 1024;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t chunk = blk * 4 + (threadIdx.x >> 6);
        const size_t base = chunk * 256 + (size_t)lane * 4;
        if (chunk * 256 >= n) continue;
        float x[4];
        if (base + 4 <= n) {
            const float4 q = *reinterpret_cast<const float4*>(v + base);
            x[0] = normalize_one(q.x, cmin, t); x[1] = normalize_one(q.y, cmin, t); x[2] = normalize_one(q.z, cmin, t); x[3] = normalize_one(q.w, cmin, t);
            *reinterpret_cast<float4*>(v + base) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
            for (int j = 0; j < 4; ++j) { x[j] = 0.f; if (base + j < n) { x[j] = normalize_one(v[base + j], cmin, t); v[base + j] = x[j]; } }
        }
        if (!csum) continue;
        double s = ((double)x[0] + (double)x[1]) + ((double)x[2] + (double)x[3]);
        int bad = 0;
        for (int j = 0; j < 4; ++j) { const unsigned b = __float_as_uint(x[j]); bad |= (int)(b >> 31) | (int)(((b >> 23) & 255u) == 255u); }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); bad |= __shfl_xor(bad, o); }
        if (lane == 0) { csum[chunk] = s; cbad[chunk] = bad; }
    }
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

// (`start`: optional {index, sum bits} left by avg_chunk_scan_kernel below -- this kernel then finishes from there; with the whole
//  array behind it only the final store is left)
__global__ __launch_bounds__(1024) void avg_scan_kernel(const float* v, int n, float* avg_out, float* sum_out, const int* start, const int* q0_main)
{
    if (q0_main && xcd_share(q0_main, 1) != 0) return;      // (look-ahead path: eight blocks, the one on the first free XCD works -- see xcd_share)
    constexpr int NT = 1024, E = 16, WIN = NT * E, NW = NT / 64;
    __shared__ int wD[2][NW], wPP[NW];                       // per-wave totals, then their exclusive scan
    __shared__ __attribute__((aligned(16))) float sX[WIN];
    __shared__ int s_cross, s_i, s_stretch;
    __shared__ float s_sum;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const Xd ident = {0, 0, 2};                              // parity 0 -> 0, 1 -> 1
    if (t == 0) { s_sum = start ? __int_as_float(start[1]) : 0.f; s_i = start ? start[0] : 0; s_stretch = 256; }
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

// ---- the same sum on MANY blocks (round 4) ---------------------------------------------------------------------------------------
// The one-block scan above walks the array window by window: 57 dependent windows of ~14 us at 1280x720 = 0.8 ms on ONE CU, more
// than half of a 4-argument mask.  What is sequential about the sum is only WHICH BINADE the running sum is in when an element
// arrives (the element's transducer depends on the sum's exponent e) -- and that can be predicted: the exact prefix sums (fp64, per
// chunk of 256 elements) tell the exponent at every chunk to within the running sum's own rounding drift.  So:
//   1. avg_chunk_sum_kernel   (n/256 waves, all CUs): fp64 sum of every chunk + "has an element the transducer cannot take" flag;
//   2. avg_chunk_class_kernel (n/256 waves): exclusive fp64 prefix P_c of the chunk sums (every block re-adds the sums before it:
//      28 KB from L2), candidate exponents e_lo = exponent(0.97 P_c) and e_lo + 1 (if (P_c + T_c) * 1.03 reaches it), and the chunk's
//      256 elements composed into ONE transducer per candidate (4 elements per lane, wave scan by shuffles);
//   3. avg_chunk_scan_kernel  (one block): state = the exact running sum S.  1024 chunk transducers at a time for S's actual
//      exponent, block-wide scan, first chunk whose end could leave the binade (or that has no transducer for this exponent): every
//      chunk before it is absorbed in one step (S = ldexp(M + D)), that chunk is added natively (256 real fp32 additions by one
//      lane), on to the next 1024.  ~15 rounds instead of 57 windows, each a scan of 1024 three-word items instead of 16384 elements;
//   4. avg_scan_kernel (above) as the finisher from the state step 3 leaves: if step 3 met something outside its premise (negative /
//      non-finite elements, a running sum that stays zero or stagnates far below the prediction) it stops THERE and the one-block
//      kernel, whose every path is pinned by test_sequential_sum_bit_exact, takes over; otherwise it only stores the result.
// A wrong prediction costs time, never bits: step 3 uses a chunk's transducer only for the exponent it was built for.  The scheme
// (margins, crossing rule, hand-over) is pinned on the CPU by a numpy restatement: tests/test_cpu_oracle.py.
constexpr int ACH = 256;                                      // elements per chunk (one wave, a float4 per lane)
struct ChunkXd { int e_lo, flags, d0[2], d1[2], pp[2]; };     // flags: bit 0 slot 0 valid, bit 1 slot 1 valid, bit 8 bad element

__global__ __launch_bounds__(256) void avg_chunk_sum_kernel(const float* v, int n, double* csum, int* cbad)
{
    const int lane = threadIdx.x & 63;
    for (int blk = blockIdx.x; blk * 4 * ACH < n; blk += gridDim.x) {
        const int chunk = blk * 4 + (threadIdx.x >> 6);
        const int base = chunk * ACH + lane * 4;
        if (chunk * ACH >= n) continue;
        float x[4];
        if (base + 4 <= n) { const float4 q = *reinterpret_cast<const float4*>(v + base); x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w; }
        else { for (int j = 0; j < 4; ++j) x[j] = base + j < n ? v[base + j] : 0.f; }
        double t = ((double)x[0] + (double)x[1]) + ((double)x[2] + (double)x[3]);
        int bad = 0;
        for (int j = 0; j < 4; ++j) { const unsigned b = __float_as_uint(x[j]); bad |= (int)(b >> 31) | (int)(((b >> 23) & 255u) == 255u); }
        for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); bad |= __shfl_xor(bad, o); }
        if (lane == 0) { csum[chunk] = t; cbad[chunk] = bad; }
    }
}

__device__ __forceinline__ int f32_exponent_of(double p)     // exponent field of p rounded to fp32 (0 if zero / denormal / not finite)
{
    const float f = (float)p;
    const int e = (int)(__float_as_uint(f) >> 23) & 255;
    return (f > 0.f && e >= 1 && e <= 254) ? e : 0;
}

__global__ __launch_bounds__(256) void avg_chunk_class_kernel(const float* v, int n, const double* csum, const int* cbad, int nchunks, ChunkXd* out)
{
    __shared__ double red[4];
    __shared__ double s_pre;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int blk = blockIdx.x; blk * 4 < nchunks; blk += gridDim.x) {
        const int chunk0 = blk * 4;
        // exclusive prefix of the chunk sums at this block's first chunk (the estimate only steers the choice of exponents)
        double acc = 0.0;
        for (int j = t; j < chunk0; j += 256) acc += csum[j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        __syncthreads();                                     // (the previous item's s_pre / red have been read by everyone)
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (t == 0) s_pre = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        const int chunk = chunk0 + wave;
        if (chunk >= nchunks) continue;
        double P = s_pre;
        for (int j = 0; j < wave; ++j) P += csum[chunk0 + j];
        const double T = csum[chunk];
        const int e_lo = f32_exponent_of(P * 0.96875), e_hi = f32_exponent_of((P + T) * 1.03125);
        const int base = chunk * ACH + lane * 4;
        float x[4];
        if (base + 4 <= n) { const float4 q = *reinterpret_cast<const float4*>(v + base); x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w; }
        else { for (int j = 0; j < 4; ++j) x[j] = base + j < n ? v[base + j] : 0.f; }      // (padding zeros are identity transducers)
        ChunkXd r; r.e_lo = e_lo; r.flags = cbad[chunk] ? 256 : 0;
        const int nslots = (e_lo == 0 || cbad[chunk]) ? 0 : (e_hi > e_lo ? 2 : 1);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            r.d0[sl] = 0; r.d1[sl] = 0; r.pp[sl] = 2;
            if (sl >= nslots) continue;                          // (wave-uniform)
            const int e = e_lo + sl;
            int d0 = 0, d1 = 0, p0 = 0, p1 = 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int f, g, tie; elem_class(x[j], e, f, g, tie);
                const int fo = f & 1;
                const int dl0 = f + g + (tie & (p0 ^ fo)), dl1 = f + g + (tie & (p1 ^ fo));
                d0 = sat_add(d0, dl0); p0 = (p0 + dl0) & 1;
                d1 = sat_add(d1, dl1); p1 = (p1 + dl1) & 1;
            }
            Xd inc = {d0, d1, p0 | (p1 << 1)};
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const Xd a = xd_shfl_up(inc, off); if (lane >= off) inc = xd_then(a, inc); }
            r.d0[sl] = __shfl(inc.d0, 63); r.d1[sl] = __shfl(inc.d1, 63); r.pp[sl] = __shfl(inc.pp, 63);
            r.flags |= 1 << sl;
        }
        if (lane == 0) out[chunk] = r;
    }
}

__global__ __launch_bounds__(1024) void avg_chunk_scan_kernel(const float* v, int n, const ChunkXd* cx, int nchunks, int* state_out, const int* q0_main)
{
    if (q0_main && xcd_share(q0_main, 1) != 0) return;      // (look-ahead path: eight blocks, the one on the first free XCD works -- see xcd_share)
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ int wD[2][NW], wPP[NW];
    __shared__ __attribute__((aligned(16))) float sX[ACH];
    __shared__ int s_first, s_ci, s_stop, s_miss;
    __shared__ float s_sum;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const Xd ident = {0, 0, 2};
    if (t == 0) { s_sum = 0.f; s_ci = 0; s_stop = 0; s_miss = 0; }
    __syncthreads();
    for (;;) {
        const float s = s_sum; const int ci = s_ci; const int stop = s_stop;
        if (ci >= nchunks || stop) break;
        __syncthreads();                                   // everyone has read the state
        if (t == 0) s_first = NT;
        const unsigned bits = __float_as_uint(s);
        const int e = (int)(bits >> 23) & 255;
        const bool ok = s > 0.f && e >= 1 && e <= 254;
        const int M = (int)((bits & 0x7FFFFFu) | 0x800000u);
        const int limit = (1 << 24) - M;
        const int c = ci + t;
        Xd me = ident; bool usable = false, bad = false;
        if (c < nchunks) {
            const ChunkXd q = cx[c];
            const int sl = e - q.e_lo;
            bad = (q.flags & 256) != 0;
            usable = ok && !bad && sl >= 0 && sl < 2 && ((q.flags >> sl) & 1);
            if (usable) { me.d0 = q.d0[sl]; me.d1 = q.d1[sl]; me.pp = q.pp[sl]; }
        }
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
            Xd ex = xd_shfl_up(wi, 1);
            if (lane == 0) ex = ident;
            if (in) { wD[0][lane] = ex.d0; wD[1][lane] = ex.d1; wPP[lane] = ex.pp; }
        }
        __syncthreads();
        const int par0 = M & 1;
        Xd pre = {wD[0][wave], wD[1][wave], wPP[wave]};
        { Xd exl = xd_shfl_up(inc, 1); if (lane == 0) exl = ident; pre = xd_then(pre, exl); }
        const int dbefore = par0 ? pre.d1 : pre.d0;          // ulps added by the chunks ci .. c-1 (true start parity)
        const int pin = (pre.pp >> par0) & 1;
        const int dmine = pin ? me.d1 : me.d0;
        // the chunk cannot be absorbed: no transducer for this exponent, or one of its elements might leave the binade
        // (every prefix inside the chunk is <= dbefore + dmine: increments are non-negative)
        const bool hit = c < nchunks && (!usable || sat_add(sat_add(dbefore, dmine), 1) >= limit);
        if (hit) atomicMin(&s_first, t);
        __syncthreads();
        const int first = s_first;                         // window-relative index of the first chunk that needs native treatment
        const int cnt = nchunks - ci < NT ? nchunks - ci : NT;
        const int owner = first < cnt ? first : cnt - 1;
        // the native chunk's elements into LDS (coalesced), while the owner forms the sum in front of it
        const int nc = ci + first;
        if (first < cnt && t < ACH) { const int idx = nc * ACH + t; sX[t] = idx < n ? v[idx] : 0.f; }
        __syncthreads();
        if (t == owner) {
            const int dabs = first < cnt ? dbefore : sat_add(dbefore, dmine);
            float sn = (ok && dabs > 0) ? ldexpf((float)(M + dabs), e - 150) : s;      // exact: M + dabs < 2^24
            int next = ci + cnt, miss = 0, halt = 0;
            if (first < cnt) {
                // premise broken (negative / non-finite element) or no transducer round after round (a sum that stays zero -- a black
                // frame's structure map: 3600 one-chunk rounds would be slower than the chain -- or stagnates below the prediction):
                // leave the rest to the one-block kernel, from exactly here
                miss = (!usable && !bad) ? s_miss + 1 : 0;
                if (bad || miss > 8) halt = 1;
                else {
                    const int m = n - nc * ACH < ACH ? n - nc * ACH : ACH;
                    int q = 0;
#pragma unroll 4
                    for (; q + 4 <= m; q += 4) { const float4 w = *reinterpret_cast<const float4*>(&sX[q]); sn += w.x; sn += w.y; sn += w.z; sn += w.w; }
                    for (; q < m; ++q) sn += sX[q];
                }
                next = halt ? nc : nc + 1;
            }
            s_sum = sn; s_ci = next; s_stop = halt; s_miss = miss;
        }
        __syncthreads();
    }
    if (t == 0) {
        const long long i = (long long)s_ci * ACH;
        state_out[0] = i < n ? (int)i : n; state_out[1] = __float_as_int(s_sum);
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// scratch of the multi-block sequential sum: [state: 2 ints][chunk sums: double][bad flags: int][transducers: ChunkXd], 256-aligned parts
size_t seqsum_workspace_bytes(size_t n)
{
    const size_t nc = (n + ACH - 1) / ACH;
    return 256 + align_up(nc * 8, 256) + align_up(nc * 4, 256) + align_up(nc * sizeof(ChunkXd), 256);
}

// the four launches of the sum (see avg_chunk_sum_kernel): short arrays go straight to the one-block kernel
struct SeqsumPlan { bool multi = false; int nc = 0; int* state = nullptr; double* csum = nullptr; int* cbad = nullptr; ChunkXd* cx = nullptr; };

static SeqsumPlan seqsum_plan(size_t n, void* ws)
{
    static const bool one_block = diag_env("FAV_AVG_ONE_BLOCK") != nullptr;       // (A/B: the round-3 form)
    SeqsumPlan pl;
    // (very long arrays too: avg_chunk_class_kernel re-adds the chunk sums before its chunk in every block -- nc^2 / 8 reads of 8 bytes:
    //  nothing at 1280x720 (3 600 chunks) or a 1504^2 VR face (8 836), 1 GB of L2 reads (~0.1 ms) for a 3840x2160 frame (32 400), 4 GB at
    //  the 65 536 chunks below -- against 1 024 dependent windows of ~14 us for the one-block kernel at that length -- and minutes at the
    //  2^31 elements fav_sequential_sum_f32 accepts, where the one-block kernel is linear.  Both produce the same bits: the sequential fp32
    //  sum is what they compute)
    if (n < 65536 || one_block || !ws || n > (size_t)65536 * ACH) return pl;
    pl.multi = true;
    pl.nc = (int)((n + ACH - 1) / ACH);
    char* w = static_cast<char*>(ws);
    pl.state = reinterpret_cast<int*>(w);
    pl.csum = reinterpret_cast<double*>(w + 256);
    pl.cbad = reinterpret_cast<int*>(w + 256 + align_up((size_t)pl.nc * 8, 256));
    pl.cx = reinterpret_cast<ChunkXd*>(w + 256 + align_up((size_t)pl.nc * 8, 256) + align_up((size_t)pl.nc * 4, 256));
    return pl;
}

// presummed: the chunk sums / flags of step 1 are already in the workspace (normalize_sum_kernel wrote them)
static void launch_seqsum(const float* v, size_t n, void* ws, float* avg_out, float* sum_out, hipStream_t st, const int* q0_main = nullptr, bool presummed = false)
{
    auto cap = [&](int g) { return g; };
    const dim3 one(q0_main ? 8 : 1);
    const SeqsumPlan pl = seqsum_plan(n, ws);
    if (!pl.multi) {
        hipLaunchKernelGGL(avg_scan_kernel, one, dim3(1024), 0, st, v, (int)n, avg_out, sum_out, static_cast<const int*>(nullptr), q0_main);
        return;
    }
    const int nc = pl.nc;
    if (!presummed) hipLaunchKernelGGL(avg_chunk_sum_kernel, dim3(cap((nc + 3) / 4)), dim3(256), 0, st, v, (int)n, pl.csum, pl.cbad);
    hipLaunchKernelGGL(avg_chunk_class_kernel, dim3(cap((nc + 3) / 4)), dim3(256), 0, st, v, (int)n, pl.csum, pl.cbad, nc, pl.cx);
    hipLaunchKernelGGL(avg_chunk_scan_kernel, one, dim3(1024), 0, st, v, (int)n, pl.cx, nc, pl.state, q0_main);
    hipLaunchKernelGGL(avg_scan_kernel, one, dim3(1024), 0, st, v, (int)n, avg_out, sum_out, static_cast<const int*>(pl.state), q0_main);
}

// floats per plane: the larger of the [H][pw] and the transposed [W][ph] layout (row pitches rounded up to 4 floats)
static size_t structure_plane_floats(int W, int H)
{
    const size_t pw = ((size_t)W + 3) & ~(size_t)3, ph = ((size_t)H + 3) & ~(size_t)3;
    return align_up(std::max((size_t)H * pw, (size_t)W * ph) * 4, 256) / 4;
}

size_t structure_workspace_bytes(int W, int H)
{
    const size_t n = (size_t)W * H;
    const size_t nb = (n + NB - 1) / NB;
    // 3 planes + 3 scratch planes (v1) + corners + 3 transposed planes + bmax + bpre + bmin + mm(2) + avg(1) + the sum's scratch
    return structure_plane_floats(W, H) * 4 * 10 + align_up(nb * 4, 256) * 3 + 256 + seqsum_workspace_bytes(n);
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

// pack_cus > 0 (the look-ahead path, fav_stream_prefetch_mask: the mask runs next to the network of an earlier frame): the two recursive
// passes run as pack_cus blocks of eight waves instead of one block per wave (see iir_rows_kernel) -- together with the one-block scans
// they are the mask's only long-lived blocks, and they then hold at most pack_cus CUs at any time.  The wide kernels stay wide: their
// blocks live for microseconds (measured: all of them together cost the network 4 us per frame).  0: the widest form (stand-alone mask).
int launch_structure(const uint8_t* rgb_hwc, int W, int H, void* ws, size_t ws_bytes, const float** structure_out,
                     const float** avg_out, hipStream_t st, int pack_cus, const int* q0_main)
{
    if (!q0_main) pack_cus = 0;
    if (pack_cus > 0) {
        // The placement scheme is written for a device that presents EIGHT XCDs and deals consecutive blocks to consecutive XCDs (MI355X /
        // MI300X in SPX mode).  On anything else -- a partitioned device, another XCD count -- no block would find itself on a "free" XCD and
        // the mask would never be computed: asked ONCE per device (eight blocks report their XCC_ID; one stream synchronisation), and the
        // wide form is used when the answer is not a permutation of 0..7.
        static int xcd8[MAX_DEVICES] = {};      // 0: not asked yet, 1: yes, -1: no
        const int dv = cur_dev();
        if (!xcd8[dv]) {
            int* d = nullptr; int h[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
            bool ok = hipMalloc(reinterpret_cast<void**>(&d), sizeof h) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(xcd_probe_kernel, dim3(8), dim3(64), 0, st, d);
                ok = hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
                (void)hipFree(d);
            }
            unsigned seen = 0;
            for (int i = 0; i < 8; ++i) if (h[i] >= 0 && h[i] < 8) seen |= 1u << h[i];
            xcd8[dv] = ok && seen == 0xffu ? 1 : -1;
            if (!ok) (void)hipGetLastError();
        }
        if (xcd8[dv] < 0) { pack_cus = 0; q0_main = nullptr; }
    }
    FAV_REQUIRE(ws != nullptr && ws_bytes >= structure_workspace_bytes(W, H), "consistency: workspace too small");
    FAV_REQUIRE(W >= 2 && H >= 2, "consistency: structure mode needs W,H >= 2");
    const size_t n = (size_t)W * H, ps = structure_plane_floats(W, H);
    const int pw = (W + 3) & ~3, ph = (H + 3) & ~3;
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
    const unsigned tiles = (unsigned)(((W + 31) / 32) * ((H + 31) / 32));
    hipLaunchKernelGGL(moments_t_kernel, dim3(tiles), dim3(256), 0, st, rgb_hwc, tmp3, tmp3 + ps, tmp3 + 2 * ps, W, H, ph);
    // recursiveSmoothX then Y on dxx, dyy, dxy (:62-67): X pass in place on the transposed planes (lines = image rows, [W][ph]), tile
    // transpose, Y pass in place on the row-major planes (lines = image columns, [H][pw]); see iir_col for the packed / wide launch forms
    constexpr int DP = 14, DW = 22;      // ring depths of the packed form (1024-thread blocks: 128 registers per lane) / the wide form (3 D - 8 <= 63, D <= 24: the read switch)
    const int gx = (H + 255) / 256, gy = (W + 255) / 256;      // tasks (waves) per plane: 256 lines each
    if (pack_cus > 0) {
        // packed: eight CU-filling blocks (1024 threads), the pack_cus (<= 4) on the free XCDs work (xcd_share) with as many WORKING waves
        // each as the tasks need (<= 4: one per SIMD); a working wave's ring = 2 x DP x 1 KB
        const int shares = std::min(pack_cus, 4);
        auto launch = [&](float* pl, int nlines, int n, int pitch, int g) {
            const int nwv = std::min(4, (3 * g + shares - 1) / shares);
            hipLaunchKernelGGL((iir_cols_kernel<DP, 1024>), dim3(8), dim3(1024), (size_t)nwv * 2 * DP * 1024, st, pl, ps, scratch, nlines, n, pitch, g, shares, nwv, q0_main, c);
        };
        static bool attr_done[MAX_DEVICES] = {};
        { const int dv = cur_dev();
          if (!attr_done[dv]) { FAV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(iir_cols_kernel<DP, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * DP * 1024)); attr_done[dv] = true; } }
        launch(tmp3, H, W, ph, gx);
        hipLaunchKernelGGL(transpose_kernel, dim3(tiles * 3), dim3(256), 0, st, tmp3, planes, ps, W, H, ph, pw, 3);
        launch(planes, W, H, pw, gy);
    } else {
        const size_t lds = (size_t)2 * DW * 1024;
        hipLaunchKernelGGL((iir_cols_kernel<DW, 64>), dim3(3 * gx), dim3(64), lds, st, tmp3, ps, scratch, H, W, ph, gx, 3 * gx, 1, static_cast<const int*>(nullptr), c);
        hipLaunchKernelGGL(transpose_kernel, dim3(tiles * 3), dim3(256), 0, st, tmp3, planes, ps, W, H, ph, pw, 3);
        hipLaunchKernelGGL((iir_cols_kernel<DW, 64>), dim3(3 * gy), dim3(64), lds, st, planes, ps, scratch, W, H, pw, gy, 3 * gy, 1, static_cast<const int*>(nullptr), c);
    }
    hipLaunchKernelGGL(eigen_blockmax_kernel, dim3(nb), dim3(256), 0, st, planes, ps, pw, corners, H, W, bmax, nb);
    hipLaunchKernelGGL(prefixmax_kernel, dim3(q0_main ? 8 : 1), dim3(1024), 0, st, bmax, nb, bpre, mm, q0_main);
    hipLaunchKernelGGL(quirkmin_kernel, dim3(nb), dim3(256), 0, st, corners, n, bpre, bmin, nb);
    hipLaunchKernelGGL(minreduce_kernel, dim3(q0_main ? 8 : 1), dim3(256), 0, st, bmin, nb, mm, q0_main);
    void* sum_ws = reinterpret_cast<char*>(mm) + 256;
    const SeqsumPlan pl = seqsum_plan(n, sum_ws);
    hipLaunchKernelGGL(normalize_sum_kernel, dim3(nb), dim3(256), 0, st, corners, n, mm, pl.multi ? pl.csum : nullptr, pl.multi ? pl.cbad : nullptr);
    launch_seqsum(corners, n, sum_ws, mm + 2, nullptr, st, q0_main, true);
    FAV_LAUNCH_CHECK("structure kernels");
    *structure_out = corners;
    *avg_out = mm + 2;
    return FAV_OK;
}

int launch_sequential_sum(const float* x, size_t n, float* sum_out, hipStream_t st)
{
    FAV_REQUIRE(n > 0 && n < (1ull << 31) - 1024, "sequential sum: bad length");
    // operator-level entry (no workspace in its signature): one scratch buffer per device, grown on demand; calls on different
    // streams are serialised by draining the previous one first (a test / tool entry point, not on the frame path)
    struct Scratch { void* p = nullptr; size_t bytes = 0; hipStream_t last = nullptr; bool used = false; };
    static std::mutex mu;
    static Scratch scratch[64];
    int device = 0; FAV_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(mu);
    Scratch& sc = scratch[(device >= 0 && device < 64) ? device : 0];
    const size_t need = seqsum_workspace_bytes(n);
    if (sc.used && (sc.last != st || need > sc.bytes)) { if (hipStreamSynchronize(sc.last) != hipSuccess) (void)hipGetLastError(); }
    if (need > sc.bytes) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr; sc.bytes = 0;
        FAV_HIP(hipMalloc(&sc.p, need));
        sc.bytes = need;
    }
    sc.last = st; sc.used = true;
    launch_seqsum(x, n, sc.p, nullptr, sum_out, st);
    FAV_LAUNCH_CHECK("sequential sum kernels");
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
