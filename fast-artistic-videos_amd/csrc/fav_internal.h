// Internal declarations shared by the libfav translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/fav.h"

// Environment variables.  A RELEASE build of the library (the default `make`: libfav.so) reads only the documented ones -- FAV_SIDE_CUS,
// FAV_SIDE_QUEUES, FAV_ROCTX in the library; FAV_GPU, FAV_CC_*, FAV_RCCL_TIMEOUT_S, FAV_TEST_WORKER_FAIL in the executables.  Every
// kernel-selection, tuning and debug switch (FAV_NO_*, FAV_WINO_F2, FAV_W4_*, FAV_*_DBG, ...) goes through diag_env() and exists in the
// DIAGNOSTIC build only (`make diag`: libfav_diag.so, -DFAV_DIAG; what the cross-check tests and the A/B scripts load): a stray FAV_NO_WINO
// in a user's environment cannot silently halve the rate of the library that ships.
#include <cstdlib>
inline const char* diag_env(const char* name)
{
#ifdef FAV_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace fav {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define FAV_HIP(expr)                                              \
    do {                                                           \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) return ::fav::hip_fail(e__, #expr); \
    } while (0)
#define FAV_LAUNCH_CHECK(name)                                            \
    do {                                                                  \
        hipError_t e__ = hipGetLastError();                               \
        if (e__ != hipSuccess) return ::fav::hip_fail(e__, "launch " name); \
    } while (0)
#define FAV_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::fav::set_error(__VA_ARGS__);     \
            return FAV_EINVAL;                 \
        }                                      \
    } while (0)

// No exception may cross the C ABI (include/fav.h): entry points that parse untrusted files run their body inside this pair.
#define FAV_ABI_TRY try {
#define FAV_ABI_CATCH(where_)                                                                                            \
    } catch (const std::bad_alloc&) {                                                                                   \
        ::fav::set_error(where_ ": out of memory while parsing (damaged size field?)"); return FAV_EFORMAT;                \
    } catch (const std::exception& e__) {                                                                               \
        ::fav::set_error(where_ ": %s", e__.what()); return FAV_EFORMAT;                                                  \
    } catch (...) {                                                                                                     \
        ::fav::set_error(where_ ": unexpected exception"); return FAV_EFORMAT;                                            \
    }

int ensure_device();   // FAV_OK or FAV_ENODEVICE

// roctx ranges around the stages of a frame (mask / certainty + input assembly / network, and every convolution inside it) for
// rocprofv3 --marker-trace.  Off unless FAV_ROCTX=1: the marker library is then dlopen'ed (libfav itself does not link it).
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    static bool enabled();
    bool on;
};

// ------------------------------------------------------------------------------------------------
// host-side description of a parsed checkpoint (t7_reader.cpp)
// ------------------------------------------------------------------------------------------------
enum LayerType { L_PAD, L_CONV, L_IN, L_RELU, L_RES, L_UP, L_TANH, L_MUL, L_IDENTITY, L_BN };

struct Layer {
    LayerType type = L_IDENTITY;
    // pad: nn.SpatialReflectionPadding (pad_mode 0) / nn.SpatialReplicationPadding (pad_mode 1), models_video.lua:12-16,27-31,70-77
    int pl = 0, pr = 0, pt = 0, pb = 0, pad_mode = 0;
    // conv
    int cin = 0, cout = 0, k = 0, stride = 1, pad = 0;
    int transposed = 0, adj = 0;    // nn.SpatialFullConvolution: w is [cin][cout][k][k]
    std::vector<float> w, b;        // [cout][cin][k][k], [cout] (may be empty)
    // instance norm / batch norm (evaluate mode: running mean / var)
    std::vector<float> gamma, beta, mean, var;
    float eps = 1e-5f;
    // upsample / mul / shave
    int scale = 1;
    float mul = 1.f;
    int shave = 0;
    std::vector<Layer> block;       // residual branch
};

int t7_parse_model(const char* path, std::vector<Layer>& out);          // .t7 -> layer list
int blob_pack(const std::vector<Layer>& layers, std::vector<uint8_t>& blob);
int blob_unpack(const void* blob, size_t bytes, std::vector<Layer>& out);
std::string describe_layers(const std::vector<Layer>& layers, int indent = 0);

// ------------------------------------------------------------------------------------------------
// device kernels: launch wrappers (each enqueues on `st` and does not synchronise)
// ------------------------------------------------------------------------------------------------
struct Affine {            // pending per-channel transform t(x) = relu?(x*scale+shift), up to two stages
    const float* scale1 = nullptr; const float* shift1 = nullptr; int relu1 = 0;
    const float* scale2 = nullptr; const float* shift2 = nullptr; int relu2 = 0;
    int stages = 0;
    // Round 5: stage 1 as an InstanceNorm whose statistics are still ACCUMULATORS (stat_acc.h): the producing convolution added every
    // work unit's exact fixed-point (sum, sum of squares) to them with integer atomics, the CONSUMER forms scale / shift in its
    // prologue (no in_finalize launch in between).  acc1 != null => scale1 / shift1 are null; only conv3_wino4_kernel<1> takes it
    const long long* acc1 = nullptr; long long* acc1_zero = nullptr;      // this frame's accumulators | the other parity's (zeroed by the consumer for the next frame)
    const float* gamma1 = nullptr; const float* beta1 = nullptr; float eps1 = 0.f; int count1 = 0;
};
// accumulator layout of one InstanceNorm: [parity 2][copy STAT_COPIES][channel C][4 words: sum lo, sum hi, squares lo, squares hi]
constexpr int STAT_COPIES = 8;       // one copy per XCD (blockIdx & 7): 8x shallower chains of atomics on one address
inline size_t stat_acc_words(int C) { return (size_t)STAT_COPIES * C * 4; }      // per parity
// A unit whose statistics are not finite (a diverged clip, a damaged checkpoint: NaN / Inf activations) cannot be written as fixed point
// (the double -> integer conversion would be undefined): its producer adds STAT_NONFINITE to the HIGH words instead, and a consumer that
// finds a high word beyond STAT_NONFINITE / 2 in magnitude forms NaN scale / shift -- what in_finalize_kernel gives for the same input.
// (legitimate high words stay below 2^45 for any activation range the fp32 network can produce; 300 units x 2^53 still fits 63 bits)
constexpr long long STAT_NONFINITE = 1ll << 53;
__host__ __device__ inline bool stat_acc_poisoned(long long hi_sum, long long hi_squares)
{
    return hi_sum >= STAT_NONFINITE / 2 || hi_sum <= -(STAT_NONFINITE / 2) || hi_squares >= STAT_NONFINITE / 2 || hi_squares <= -(STAT_NONFINITE / 2);
}

struct ConvLaunch {
    const float* in = nullptr;      // NHWC physical [IHp][IWp][CIN]
    int IH = 0, IW = 0;             // logical input size (after `ups` nearest upsampling)
    int IWp = 0;                    // physical row pitch in pixels
    int ups = 0;                    // log2 of the nearest-upsample factor applied on load (0|1)
    int stuff = 0;                  // with ups=1: zero-stuffed instead of replicated (transposed convolution as a convolution)
    int CIN = 0;                    // physical channels (multiple of 4)
    Affine pre;                     // applied on load (zero padding is applied AFTER it)
    const float* wgt = nullptr;     // [COUTp][Kpad], k = tap*CIN + ci
    const unsigned short* wgt16 = nullptr;   // bf16 copy of wgt: selects the bf16-operand halo kernel (fast mode) when set
    const float* bias = nullptr;    // [COUTp]
    int COUT = 0, COUTp = 0;        // valid / padded (multiple of the N tile) output channels
    int KH = 0, KW = 0, stride = 1, pad = 0;
    int Kpad = 0;                   // multiple of 32
    int OH = 0, OW = 0;
    float* out = nullptr;           // NHWC [OH][OW][COUT]   (mode 0)
    float* partials = nullptr;      // [mblocks][COUTp] float2 (mean, M2) or null
    // final-layer epilogue (mode 1): out_planar[2-c][m] = (tanh(v)*tanh_mul + mean[c]) / 255
    int final_mode = 0;
    float tanh_mul = 1.f;
    float* out_planar = nullptr;    // [3][OH][OW] RGB  (deprocessed)   or null
    float* out_raw_nchw = nullptr;  // [3][OH][OW] BGR  (150*tanh)      or null
    // stream-K hand-off state (conv_streamk_workspace_bytes() floats, conv_streamk_grid() flags, a launch-unique
    // epoch > 0); null => plain data-parallel launch
    float* sk_ws = nullptr; unsigned* sk_flags = nullptr; unsigned sk_epoch = 0;
    unsigned* sk_err = nullptr;     // host-mapped error word (device pointer): set by a hand-off wait that timed out
    int reserve_cus = 0;            // CUs left to concurrent side-queue work: persistent / stream-K grids shrink by this many
    int no_sk = 0;                  // shared device: data-parallel grids only (no hand-off between blocks, no co-residency assumption)
    // Winograd kernel only.  A pending residual join as the input: in = the branch's raw output y (one pending InstanceNorm, no ReLU),
    // join_skip = the skip tensor's pixel under y's pixel (0, 0) at the SAME row pitch IWp, join_out = where the joined tensor is
    // written (pitch IWp as well).  OWp > 0: row pitch of `out` in pixels (the output is laid out under a later join's skip tensor).
    const float* join_skip = nullptr; float* join_out = nullptr; int OWp = 0;
    // F(4x4) Winograd kernel only: accumulators of the InstanceNorm that follows (this frame's parity) instead of `partials`, or null
    long long* stat_acc = nullptr;
    // F(4x4) Winograd kernel only: meeting places of the two parts of a unit that a stream-K share boundary cuts (conv3_wino4_ksplit_bytes()
    // bytes: 256 boundaries x 2 parts x 16 x 16 pixels x 128 channels) and their arrival counters (256 ints, zero between launches);
    // null => whole units only
    float* ks_ws = nullptr; int* ks_cnt = nullptr;
};
inline size_t conv3_wino4_ksplit_bytes() { return (size_t)256 * 2 * 256 * 128 * sizeof(float); }
size_t conv_streamk_workspace_bytes();
int conv_streamk_grid();
constexpr int CONV_BM = 128;
inline int conv_mblocks(int OH, int OW) { return (OH * OW + CONV_BM - 1) / CONV_BM; }
int launch_conv(const ConvLaunch& p, hipStream_t st);
// last layer with few output channels: kx taps folded into N (see kernels_conv.hip); wfold = [KH][32][CIN]
bool conv_fold_eligible(int cin_pitch, int cout, int k, int stride);
bool conv_fold_launchable(int cin_pitch, int k, int pad, int ups, int IH, int IW);      // 128 / 256 input channels: on a x2-upsampled input only
int launch_conv_fold(const ConvLaunch& p, const float* wfold, hipStream_t st);

// per-channel finalize of (mean, M2) partials -> scale/shift:  scale = gamma/sqrt(var+eps)
// first layer (8-channel input pitch, 9x9): LDS-resident halo + weights, persistent blocks; partial statistics are
// per 16x16 tile with explicit counts
bool conv_c8_eligible(int cin_pitch, int coutp, int k, int stride, int stages, int ups);
int conv_c8_tiles(int OH, int OW);
int launch_conv_c8(const ConvLaunch& p, int* counts, hipStream_t st);
// the same layer with dense K for 7 (video model) or 3 (image model) real input channels: taps paired so that no zero channel is
// multiplied; wc8d = conv_c8d_pack() of the [cout][cin][9][9] weights
bool conv_c8d_eligible(int cin_pitch, int cin_real, int coutp, int k, int stride, int stages, int ups);
void conv_c8d_pack(const float* w, int cin, int cout, std::vector<float>& out);
int launch_conv_c8d(const ConvLaunch& p, int cin_real, const float* wc8d, int* counts, hipStream_t st);
// the same first layer with 1-D minimal filtering F(2,3) along x (kernels_first.hip); wpk = conv_first_pack() (first_pack.h);
// eligibility as conv_c8d_eligible; partials per 8x64 tile with explicit counts
int conv_first_tiles(int OH, int OW);
int launch_conv_first(const ConvLaunch& p, int cin_real, const float* wpk, int* counts, hipStream_t st);
// ... and with 2-D minimal filtering F(2x2,3x3) over its nine 3x3 blocks (conv_first2d_kernel); wpk = conv_first2d_pack() (first2d_pack.h);
// partials per 16x32-pixel tile
bool conv_first2d_eligible(int cin_pitch, int cin_real, int coutp, int k, int stride, int stages, int ups);      // any number of 32-filter groups
int conv_first2d_tiles(int OH, int OW);
int launch_conv_first2d(const ConvLaunch& p, int cin_real, const float* wpk, int* counts, hipStream_t st);
// 3x3 stride-1 UNPADDED 128-channel layers (the residual blocks): Winograd F(2x2,3x3), kernels_wino.hip; wpk = conv_wino_pack()
// of the [cout][cin][3][3] weights (wino_pack.h); partials per 8x16-pixel unit with explicit counts
bool conv3_wino_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups);
int conv3_wino_tiles(int OH, int OW);
int launch_conv3_wino(const ConvLaunch& p, const float* wpk, int* counts, hipStream_t st);
// the same layers as Winograd F(4x4,3x3), kernels_wino4.hip (round 4); wpk = conv_wino4_pack() (wino4_pack.h); partials per 16x16-pixel unit
bool conv3_wino4_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups);
int conv3_wino4_tiles(int OH, int OW);
int launch_conv3_wino4(const ConvLaunch& p, const float* wpk, int* counts, hipStream_t st);
// 3x3 stride-1 pad-1 64-channel layer on a x2 nearest-upsampled materialised input (U2 + c3s1-64): four 2x2 convolutions on the
// physical pixels with merged weights, kernels_up2.hip; wpk = conv_up2_pack() (up2_pack.h); partials per 8x32 physical-pixel tile
bool conv3_up2_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups);
int conv3_up2_tiles(int OH, int OW);
int launch_conv3_up2(const ConvLaunch& p, const float* wpk, int* counts, hipStream_t st);
// the 3x3 stride-2 layers (d64 / d128) with fragment-order weights read global -> registers, halo chunks of 16 channels double-buffered
// in LDS, whole tiles per block (kernels_s2.hip); wpk = conv_s2w_pack() (s2_pack.h); partials per TR x 32 output-pixel tile (TR = 4 for
// 64 output channels, 2 for 128)
bool conv3s2w_eligible(int cin_pitch, int cout, int coutp, int k, int stride, int pad, int stages, int ups);
int conv3s2w_tiles(int OH, int OW, int coutp);
int launch_conv3s2w(const ConvLaunch& p, const float* wpk, int* counts, hipStream_t st);
// 3x3 stride-1 layers: halo-resident implicit GEMM (stream-K, needs the ConvLaunch sk_* fields); partials per 8x32 tile
bool conv3_halo_eligible(int cin_pitch, int coutp, int k, int stride);
int conv3_halo_tiles(int OH, int OW, bool edge_b);      // edge_b: fp32 kernel (16 x 16 tiles on a narrow ragged right edge)
int launch_conv3_halo(const ConvLaunch& p, int* counts, hipStream_t st);
// 3x3 stride-2 layers: halo-resident implicit GEMM with even / odd column planes (stream-K); partials per 4x32 tile
bool conv3s2_eligible(int cin_pitch, int coutp, int k, int stride, int stages, int ups);
int conv3s2_tiles(int OH, int OW);
int launch_conv3s2(const ConvLaunch& p, int* counts, hipStream_t st);
// counts: per-partial pixel counts or null (then block b holds min(block_pixels, M - b*block_pixels) pixels)
int launch_in_finalize(const float* partials, const int* counts, int mblocks, int M, int block_pixels, int C, int Cpitch,
                       const float* gamma, const float* beta, float eps,
                       float* scale, float* shift, hipStream_t st);
// statistics of t(x) over an NHWC tensor [M][C] -> partials [ceil(M/128)][C] float2
int launch_stats(const float* x, int M, int C, const Affine& t, float* partials, hipStream_t st);
// a padding layer INSIDE the network (padding_type reflect / replicate): out[y][x][:] = in[map(y - pt)][map(x - pl)][:] of the (optionally
// x2 nearest-upsampled) NHWC tensor `in` with row pitch `in_pitch` pixels; mode 0 mirrors without repeating the edge, 1 repeats the edge
int launch_pad_nhwc(const float* in, int Hp, int Wp, int in_pitch, int C, int ups, float* out, int pl, int pr, int pt, int pb, int mode, hipStream_t st);
// z[oy][ox][c] = y[oy][ox][c]*scale[c]+shift[c] + t(skip[oy+s][ox+s][c]).  partials != null: also the per-row-segment
// (mean, M2, count) statistics of z ([res_add_stat_blocks(OH, OW)][C] float2 + counts) for an InstanceNorm that follows the join
int launch_res_add(const float* y, const float* scale, const float* shift,
                   const float* skip, int SH, int SW, int shave, const Affine& skip_t,
                   int C, float* z, float* partials, int* counts, hipStream_t st, int skip_pitch = 0, const Affine* branch_acc = nullptr);
// (branch_acc: the branch's InstanceNorm as accumulators -- Affine::acc1 -- instead of scale / shift; plain joins only)
inline int res_add_stat_blocks(int OH, int OW) { return OH * ((OW + 127) / 128); }
// NCHW [C][H][W] -> NHWC [H+2p][W+2p][Cp] with reflection padding p and zero channels >= C
int launch_nchw_to_nhwc_pad(const float* in, int C, int H, int W, int pad, int Cp, float* out, hipStream_t st);
// NHWC [M][C] with transform -> NCHW [C][M]
int launch_nhwc_to_nchw(const float* in, int M, int C, const Affine& t, float* out, hipStream_t st);

#ifdef __HIPCC__
// counter RNG standing in for the reference's unseeded torch.rand (core.lua:109); oracle: vr_oracle.fill_uniform
__device__ __forceinline__ float fill_uniform(unsigned seed, unsigned index, unsigned c, unsigned y, unsigned x)
{
    unsigned k = seed * 0x9E3779B1u + index * 0x85EBCA77u + c * 0xC2B2AE3Du + y * 0x27D4EB2Fu + x * 0x165667B1u;
    k ^= k >> 15; k *= 0x2C1B3C6Du;
    k ^= k >> 12; k *= 0x297A2D39u;
    k ^= k >> 15;
    return (float)(k >> 8) * (1.0f / 16777216.0f);
}
#endif

// frame-level kernels (kernels_frame.hip / kernels_consistency.hip)
int launch_warp(const float* img, const float* flow, float* out, int B, int C, int H, int W, int Ho, int Wo,
                int border, hipStream_t st);
int launch_min_filter_f32(const float* cert, float* out, int H, int W, int r, hipStream_t st);
int launch_assemble(const float* frame, const float* warped, const float* cert, float* in7, int H, int W,
                    hipStream_t st);
// certainty preparation for the fused path: u8 mask (optionally inverted / multiplied by the
// fix_occlusions term) -> min-filtered float certainty [H][W]
int launch_cert_prepare(const uint8_t* mask, const float* backward_flo, int invert, int fix_occ, int border,
                        int r, float* cert_tmp, float* cert, int H, int W, hipStream_t st);
// fused A2+A6+A7+reflection pad: writes the padded NHWC8 network input
int launch_prep_input(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float* backward_flo,
                      const float* cert, int border, int H, int W, int pad, float* in8, hipStream_t st,
                      int fill_random = 0, unsigned seed = 0, unsigned index = 0, int* q0_out = nullptr);
int launch_quantize_rgb8(const float* rgb_planar, uint8_t* out_hwc, int H, int W, hipStream_t st);
// Huffman codes of the PNG encoder (png_tables.cpp): PNG_NTABLES model codes with their dynamic-block headers + the fixed code
constexpr int PNG_NSYM = 277;            // literals 0..255, end of block 256, length symbols 257..276 (runs of 3..66)
constexpr int PNG_NTABLES = 12;
constexpr int PNG_HDR_WORDS = 40;
struct PngTable {
    uint32_t sym[PNG_NSYM];              // (code length << 16) | bit-reversed code
    uint32_t hdr[PNG_HDR_WORDS];         // the block header behind BFINAL / BTYPE, LSB first
    uint32_t hdr_bits, btype, dist_len, dist_code;
};
const std::vector<PngTable>& png_tables();
size_t png_capacity(int W, int H);
size_t png_workspace_bytes(int W, int H);
uint32_t png_crc32_combine_host(uint32_t crc_a, uint32_t crc_b, uint32_t len_b);
int launch_png_encode(const uint8_t* rgb_hwc, const float* rgb_planar, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes,
                      void* workspace, size_t ws_bytes, hipStream_t st);
int launch_check_cert(const float* backward_flo, const float* forward_flo, const float* structure, const float* avg, uint8_t* mask_out,
                      int invert, int fix_occ, int border, int r, float* cert, int H, int W, hipStream_t st);
// ... and the input assembly on top of it (launch_check_cert + launch_prep_input in one launch; round 5)
int launch_check_prep(const uint8_t* frame_hwc, const float* prev_rgb, int Hs, int Ws, const float* backward_flo, const float* forward_flo,
                      const float* structure, const float* avg, uint8_t* mask_out, float* cert, int invert, int fix_occ, int border, int r,
                      int H, int W, int pad, float* in8, hipStream_t st, int fill_random, unsigned seed, unsigned index);
int launch_store_flag(uint32_t* flag_host, uint32_t value, hipStream_t st);
int launch_unpad_input(const float* in8, int H, int W, int pad, float* in7, hipStream_t st);
int launch_temporal_loss(const float* prev_rgb, const float* cur_rgb, const float* backward_flo, const uint8_t* cert_u8, int border,
                         int H, int W, double* partial256, hipStream_t st);

size_t structure_workspace_bytes(int W, int H);
int launch_structure(const uint8_t* rgb_hwc, int W, int H, void* ws, size_t ws_bytes,
                     const float** structure_out, const float** avg_out, hipStream_t st, int pack_cus = 0, const int* q0_main = nullptr);
int launch_sequential_sum(const float* x, size_t n, float* sum_out, hipStream_t st);
int launch_consistency(const float* f1_flo, const float* f2_flo, const float* structure, const float* avg,
                       uint8_t* out, int W, int H, hipStream_t st);

// cube-map orchestration kernels (kernels_vr.hip)
int launch_vr_rotate(const float* src, float* dst, int H, int W, int mode, hipStream_t st);           // 1: +90, 2: -90, 3: 180
int launch_vr_accum(float* acc, const float* w, const float* div, size_t n, int first, hipStream_t st);
int launch_vr_cert(const uint8_t* cert_u8, const float* m0, const float* m1, const float* m2, const float* m3, float* out, size_t n,
                   hipStream_t st);
int launch_vr_prior(const float* lfw, const float* border, const float* grad, const float* cert, const float* m, const float* m2,
                    float* out, size_t n, hipStream_t st);
int launch_vr_blend(const float* seg, const float* borders, const float* g, const float* anti, float* out, size_t n, hipStream_t st);
int launch_vr_median(const float* src, float* dst, int H, int W, int r, hipStream_t st);
int launch_vr_strip(const float* face, int FH, int FW, int cy, int cx, int CH, int CW, int mode, float* strip, int SH, int SW,
                    int x0, hipStream_t st);
int launch_vr_prep(const uint8_t* frame_hwc, const float* prior, const float* cert, int fill_random, unsigned seed, unsigned index,
                   int H, int W, int pad, float* in8, hipStream_t st);
int launch_vr_flo_to_lua(const float* flo_uv, float* lua_dydx, size_t n, hipStream_t st);

}  // namespace fav

struct fav_net;
namespace fav {
int net_device(const fav_net* n);
int net_pad(const fav_net* n);
int net_in_channels(const fav_net* n);
void net_out_size(const fav_net* n, int H, int W, int* Ho, int* Wo);
int net_forward_padded(fav_net* n, const float* in8, int H, int W, float* out_planar, hipStream_t st);
}  // namespace fav

