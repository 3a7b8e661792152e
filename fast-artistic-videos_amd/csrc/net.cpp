// net.cpp -- host side of the transformer network (A8) and of the fused per-frame pipeline.
//
// fav_net  : replaces `checkpoint.model` + `model:forward(input)` (fast_artistic_video_core.lua:38-57,172).
// fav_stream: replaces one iteration of run_fast_neural_video's loop (core.lua:194-211) with the video
//             CLI's callbacks (fast_artistic_video.lua:93-172), keeping last_frame_stylized on the device.
//
// Execution model ("lazy activations"): every activation lives in HBM exactly once, as the RAW output
// of the kernel that produced it (NHWC fp32).  nn.InstanceNormalization / nn.ReLU /
// nn.SpatialUpSamplingNearest never run as kernels of their own: they become a pending per-channel
// transform + index map attached to the tensor and are applied by the NEXT convolution while it
// stages its operand into LDS.  The InstanceNorm statistics come from the producing convolution's
// epilogue (per-tile mean/M2, merged in fp64), or -- when the normalised tensor is itself a pending
// transform of another tensor (the IN that follows an upsample of an already normalised+rectified
// tensor, models_video.lua:94-98,121-130) -- from one read-only reduction pass.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>

#include <mutex>
#include <unistd.h>

#include "fav_internal.h"
#include "wino_pack.h"
#include "wino4_pack.h"
#include "up2_pack.h"
#include "s2_pack.h"
#include "first_pack.h"
#include "first2d_pack.h"

using namespace fav;

namespace {

struct DevBuf { void* p = nullptr; size_t bytes = 0; };
struct DevConvW { float* wgt = nullptr; float* bias = nullptr; float* wfold = nullptr; float* wc8d = nullptr; float* wwino = nullptr; float* wwino4 = nullptr; float* wup2 = nullptr; float* ws2w = nullptr; float* wfirst = nullptr; float* wfirst2d = nullptr; unsigned short* wgt16 = nullptr; int cinp = 0, coutp = 0, kpad = 0; };
struct DevIN { float* gamma = nullptr; float* beta = nullptr; float* scale = nullptr; float* shift = nullptr;
               long long* acc = nullptr; size_t acc_bytes = 0; };      // accumulator form of the statistics (fav_internal.h, Affine::acc1): [2 parities][stat_acc_words(C)], zero between frames

struct Act {
    float* data = nullptr;
    int Hp = 0, Wp = 0;          // physical size
    int C = 0;                   // channel pitch
    int ups = 0;                 // pending nearest upsample (log2)
    Affine pre;                  // pending per-channel transform
    float* partials = nullptr;   // (mean, M2) tiles of the raw tensor from the producing conv, or null
    int mblocks = 0, ppitch = 0;
    int* counts = nullptr;       // per-partial pixel counts (first-layer kernel) or null
    int pitch = 0;               // row pitch in pixels when the tensor sits inside a wider allocation (0: Wp)
    // pending residual join (conv3_wino_kernel MODE 2): data = the branch's raw output, pre = its InstanceNorm, join_skip = the skip
    // tensor's pixel under data's pixel (0, 0) at the same pitch; join_out = where the consuming convolution writes the joined tensor
    const float* join_skip = nullptr; float* join_out = nullptr;
    long long* acc = nullptr; long long* acc_other = nullptr;      // the producing convolution added its statistics to accumulators (this frame's parity | the other one)
    int H() const { return Hp << ups; }
    int W() const { return Wp << ups; }
    int P() const { return pitch ? pitch : Wp; }
};

int dev_upload(const std::vector<float>& h, size_t pad_to, float** out)
{
    const size_t n = std::max(pad_to, h.size());
    std::vector<float> tmp(n, 0.f);
    std::copy(h.begin(), h.end(), tmp.begin());
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(out), (n ? n : 1) * sizeof(float)));
    if (n) FAV_HIP(hipMemcpy(*out, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return FAV_OK;
}

// repack [cout][cin][k][k] -> [coutp][kpad].  K order of the implicit GEMM (must match FAV_TAP_SETUP):
//   cinp >= 32: k = ((ci/32)*taps + tap)*32 + ci%32   (channel slice outermost: consecutive K-steps shift by one tap)
//   cinp <  32: k = tap*cinp + ci                     (several taps per 32-wide K slice)
// nn.SpatialFullConvolution (weight [cin][cout][k][k]) runs as a stride-1 convolution over the zero-stuffed input with the
// taps flipped: out[oy] = sum_ky' stuffed[oy + ky' - (k-1-p)] * w[k-1-ky'].
void repack_weights(const Layer& L, int cinp, int coutp, int kpad, std::vector<float>& out)
{
    out.assign((size_t)coutp * kpad, 0.f);
    const int taps = L.k * L.k;
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci)
            for (int ky = 0; ky < L.k; ++ky)
                for (int kx = 0; kx < L.k; ++kx) {
                    const int tap = ky * L.k + kx;
                    const size_t k = cinp >= 32 ? ((size_t)(ci / 32) * taps + tap) * 32 + ci % 32 : (size_t)tap * cinp + ci;
                    out[(size_t)co * kpad + k] = L.transposed
                        ? L.w[(((size_t)ci * L.cout + co) * L.k + (L.k - 1 - ky)) * L.k + (L.k - 1 - kx)]
                        : L.w[(((size_t)co * L.cin + ci) * L.k + ky) * L.k + kx];
                }
}

// Tuning / ablation switches (not part of the product contract): read ONCE per process, never on the launch path.
struct Tuning { bool no_fold, no_c8, no_h3, no_s2, no_c8d, no_wino, no_up2, no_first, no_s2w, wino_f2, no_acc_stats; };
const Tuning& tuning()
{
    static const Tuning t = {diag_env("FAV_NO_FOLD") != nullptr, diag_env("FAV_NO_C8") != nullptr, diag_env("FAV_NO_H3") != nullptr, diag_env("FAV_NO_S2") != nullptr, diag_env("FAV_NO_C8D") != nullptr, diag_env("FAV_NO_WINO") != nullptr, diag_env("FAV_NO_UP2") != nullptr, diag_env("FAV_NO_FIRST") != nullptr, diag_env("FAV_NO_S2W") != nullptr,
                             diag_env("FAV_WINO_F2") != nullptr,       // FAV_WINO_F2: the residual convolutions as F(2x2,3x3) (rounds 2-3) instead of F(4x4,3x3)
                             diag_env("FAV_NO_ACC_STATS") != nullptr}; // FAV_NO_ACC_STATS: every InstanceNorm through partials + an in_finalize launch (rounds 1-4)
    return t;
}

bool only_tail(const std::vector<Layer>& ls, size_t from, bool& has_tanh, float& mul);
int pow2_ceil(int c) { int p = 4; while (p < c) p <<= 1; return p; }

// Channel pitches of the kernels are powers of two (the generic kernel locates (tap, channel) with shifts, the elementwise kernels
// split 256 threads over the channels).  Architectures with other filter counts -- models_video.lua:55-140 builds any `c9s1-48,d96,...`,
// the VR checkpoints have "more filters" (README.md:141) -- are EXECUTED as the next power of two with zero filters: a padded output
// channel has zero weights and zero bias (identically 0), its InstanceNorm / BatchNorm gets scale 0 and shift 0 (stays 0 through ReLU,
// residual joins and upsampling), and the next convolution has zero weights for it -- every real channel sees exactly the sums it saw
// before (x + 0 * 0 = x).  `chan`: channels of the tensor flowing in (already padded).  The 3-channel last layer keeps its size.
// `real`: the channel count the checkpoint itself gives that tensor (what a following InstanceNorm / BatchNorm / conv must match:
// a malformed file -- IN(64) behind a real 128-channel conv -- is rejected, not run with its upper channels forced to zero);
// `seen_conv`: false until the network's first convolution, whose 7 (3) inputs ride in an 8-channel pixel.
int pad_channel_counts(std::vector<Layer>& ls, int& chan, int& real, bool& seen_conv, bool top)
{
    for (size_t li = 0; li < ls.size(); ++li) {
        Layer& L = ls[li];
        if (L.type == L_CONV) {
            const bool first_input = !seen_conv && L.cin <= 8;           // the 7 (3) network inputs ride in an 8-channel pixel
            seen_conv = true;
            const int cin_new = first_input ? L.cin : chan;
            bool has_tanh = false; float mul = 1.f;
            const bool is_final = top && L.cout == 3 && only_tail(ls, li + 1, has_tanh, mul) && has_tanh;
            const int cout_new = is_final ? L.cout : pow2_ceil(L.cout);
            if (!first_input && L.cin != real) { set_error("network: conv expects %d input channels, its producer has %d", L.cin, real); return FAV_EFORMAT; }
            const int cout_real = L.cout;
            if (cin_new != L.cin || cout_new != L.cout) {
                std::vector<float> w((size_t)cout_new * cin_new * L.k * L.k, 0.f);
                const size_t kk = (size_t)L.k * L.k;
                for (int co = 0; co < L.cout; ++co)
                    for (int ci = 0; ci < L.cin; ++ci) {
                        const float* src = L.transposed ? &L.w[((size_t)ci * L.cout + co) * kk] : &L.w[((size_t)co * L.cin + ci) * kk];
                        float* dst = L.transposed ? &w[((size_t)ci * cout_new + co) * kk] : &w[((size_t)co * cin_new + ci) * kk];
                        std::copy(src, src + kk, dst);
                    }
                L.w.swap(w);
                if (!L.b.empty()) L.b.resize((size_t)cout_new, 0.f);
                L.cin = cin_new; L.cout = cout_new;
            }
            chan = L.cout; real = cout_real;
        } else if (L.type == L_IN) {
            if ((int)L.gamma.size() != real) { set_error("network: InstanceNormalization(%zu) after %d channels", L.gamma.size(), real); return FAV_EFORMAT; }
            L.gamma.resize((size_t)chan, 0.f); L.beta.resize((size_t)chan, 0.f);
        } else if (L.type == L_BN) {
            if ((int)L.mean.size() != real) { set_error("network: SpatialBatchNormalization(%zu) after %d channels", L.mean.size(), real); return FAV_EFORMAT; }
            L.gamma.resize((size_t)chan, 0.f); L.beta.resize((size_t)chan, 0.f); L.mean.resize((size_t)chan, 0.f); L.var.resize((size_t)chan, 1.f);
        } else if (L.type == L_RES) {
            int c = chan, r = real;
            int rc = pad_channel_counts(L.block, c, r, seen_conv, false); if (rc) return rc;
            if (c != chan || r != real) { set_error("network: residual branch changes the channel count"); return FAV_EUNSUPPORTED; }
        }
    }
    return FAV_OK;
}

long long count_params(const std::vector<Layer>& ls)
{
    long long n = 0;
    for (const Layer& L : ls) {
        if (L.type == L_CONV) n += (long long)L.w.size() + (long long)L.b.size();
        else if (L.type == L_IN) n += 2 * (long long)L.gamma.size();
        else if (L.type == L_BN) n += 4 * (long long)L.mean.size();
        else if (L.type == L_RES) n += count_params(L.block);
    }
    return n;
}

bool only_tail(const std::vector<Layer>& ls, size_t from, bool& has_tanh, float& mul)
{
    has_tanh = false; mul = 1.f;
    for (size_t i = from; i < ls.size(); ++i) {
        if (ls[i].type == L_TANH) { if (has_tanh) return false; has_tanh = true; }
        else if (ls[i].type == L_MUL) { if (!has_tanh) return false; mul *= ls[i].mul; }
        else if (ls[i].type != L_IDENTITY) return false;
    }
    return true;
}

}  // namespace

struct fav_net {
    int device = 0;
    std::vector<Layer> layers;    // as parsed from the checkpoint (describe / output size / parameter count)
    std::vector<Layer> exec;      // what runs: the same network with channel counts padded to powers of two (pad_channel_counts)
    int pad = 0;                  // leading nn.SpatialReflectionPadding (train_video.lua:319-325)
    bool pad_folded = false;      // ... folded into the input assembly (layers[0] is then skipped by the executor)
    int in_channels = 0;
    long long params = 0;
    std::vector<DevConvW> convs;  // traversal order
    std::vector<DevIN> ins;
    float* ones = nullptr; float* zeros = nullptr;
    float* sk_ws = nullptr; unsigned* sk_flags = nullptr; unsigned sk_epoch = 0;   // stream-K hand-off state
    float* ks_ws = nullptr; int* ks_cnt = nullptr;                                    // meeting places of the F(4x4) Winograd kernel's stream-K launches
    // a residual block whose join stays pending (run(), L_RES): its last convolution lays its output out under the skip tensor
    struct LazyOut { bool active = false; int pitch = 0, rows = 0, shave = 0, conv_index = -1; } lazy;
    unsigned* sk_err_host = nullptr; unsigned* sk_err_dev = nullptr;               // host-mapped: a hand-off wait timed out
    bool shared_device = false;     // data-parallel grids only: set by the caller (fav_net_set_shared_device) or by a timed-out hand-off
    int precision = 0;              // 0 = fp32 (parity mode), 1 = bf16 operands in the halo-resident 3x3 convolutions (fast mode)
    int acc_parity = 0;             // which half of the InstanceNorm accumulators this forward adds to (the consumers zero the other half)
    bool acc_dirty = false;         // a forward failed half-way (or the precision changed): every accumulator is zeroed before the next forward
    bool branch_tail_acc_ok = false; // set by L_RES around its branch: the join is a plain res_add launch, which takes the last InstanceNorm as accumulators
    int reserve_cus = 0;            // set when a stream uses the look-ahead side queues (they are CU-masked to this many CUs)
    bool use_c8 = false, use_h3 = false, use_s2 = false, use_wino = false, use_wino4 = false, use_up2 = false, use_first = false, use_first2d = false, use_s2w = false; int* c8_counts = nullptr;                                    // first-layer kernel selection for the next launch
    // activation arena: buffers are created on the first forward for a given (H, W) and reused after
    int curH = 0, curW = 0;
    std::vector<DevBuf> bufs;
    // ... carved out of a few large slabs: sixty hipMalloc calls cost the first frame of a run 10 ms (profiles/e2e_r04s_startup.log)
    std::vector<void*> slabs; char* slab_cur = nullptr; size_t slab_left = 0;
    size_t cursor = 0, conv_cursor = 0, in_cursor = 0;
    hipStream_t st = nullptr;
    float* stage = nullptr; size_t stage_bytes = 0;   // NCHW-boundary staging (fav_net_forward)
    // optional per-convolution event timing (bench.py roofline)
    bool profiling = false;
    struct ProfRec { hipEvent_t a, b; int conv; };
    std::vector<ProfRec> prof_pending;
    std::vector<double> prof_ms, prof_macs; std::vector<int> prof_n, prof_tile;

    ~fav_net()
    {
        (void)hipSetDevice(device);
        (void)hipFree(stage);
        for (auto& c : convs) { (void)hipFree(c.wgt); (void)hipFree(c.bias); (void)hipFree(c.wfold); (void)hipFree(c.wc8d); (void)hipFree(c.wwino); (void)hipFree(c.wwino4); (void)hipFree(c.wup2); (void)hipFree(c.ws2w); (void)hipFree(c.wfirst); (void)hipFree(c.wfirst2d); (void)hipFree(c.wgt16); }
        for (auto& i : ins) { (void)hipFree(i.gamma); (void)hipFree(i.beta); (void)hipFree(i.scale); (void)hipFree(i.shift); (void)hipFree(i.acc); }
        for (void* sp : slabs) (void)hipFree(sp);
        (void)hipFree(ones); (void)hipFree(zeros); (void)hipFree(sk_ws); (void)hipFree(sk_flags); (void)hipFree(ks_ws); (void)hipFree(ks_cnt); if (sk_err_host) (void)hipHostFree(sk_err_host);
    }
    int upload_layers(std::vector<Layer>& ls, int& chan_pitch, int& maxc);
    int upload();
    int alloc(size_t bytes, float** out);
    int timed_conv(const ConvLaunch& c, int conv_index, const Layer& L);
    int run(std::vector<Layer>& ls, Act& cur, bool top, float* out_planar, float* out_raw);
    bool res_block_is_winograd(const Layer& R, size_t first_conv) const;
    bool acc_stats_ok(const std::vector<Layer>& ls, size_t li) const;
    int forward_padded(const float* in8, int H, int W, float* out_planar, float* out_raw, hipStream_t stream);
    int forward_padded_unordered(const float* in8, int H, int W, float* out_planar, float* out_raw, hipStream_t stream);
    void out_size(int H, int W, int* Ho, int* Wo) const;
};

int fav_net::upload_layers(std::vector<Layer>& ls, int& chan_pitch, int& maxc)
{
    for (Layer& L : ls) {
        if (L.type == L_CONV) {
            convs.emplace_back();                 // registered first: the destructor frees whatever a failed upload leaves behind
            DevConvW& d = convs.back();
            d.cinp = chan_pitch;
            if (L.cin > chan_pitch || (chan_pitch != 8 && L.cin != chan_pitch)) {
                set_error("network: conv expects %d input channels, producer has %d", L.cin, chan_pitch); return FAV_EFORMAT; }
            if (L.k < 1 || L.stride < 1 || L.pad < 0) { set_error("network: bad convolution geometry"); return FAV_EFORMAT; }
            // channel pitches are powers of two (the generic kernel locates (tap, channel) with shifts, the elementwise kernels split 256
            // threads over the channels): 4 ... 1024 -- every architecture string of models_video.lua / train_video.lua:21-23.  Refused
            // here, at load time: a 48-channel model used to overrun the repacked weight matrix before any launch could refuse it
            if ((d.cinp & (d.cinp - 1)) != 0) {
                set_error("network: a convolution with %d input channels is unsupported (channel counts must be powers of two)", d.cinp);
                return FAV_EUNSUPPORTED; }
            d.coutp = (L.cout + 31) / 32 * 32;
            d.kpad = (L.k * L.k * d.cinp + 31) / 32 * 32;
            std::vector<float> w;
            repack_weights(L, d.cinp, d.coutp, d.kpad, w);
            int rc = dev_upload(w, 0, &d.wgt); if (rc) return rc;
            rc = dev_upload(L.b, (size_t)d.coutp, &d.bias); if (rc) return rc;
            if (!L.transposed && conv3_halo_eligible(d.cinp, d.coutp, L.k, L.stride)) {      // bf16 copy for the fast mode (round to nearest even)
                std::vector<unsigned short> w16(w.size());
                for (size_t i = 0; i < w.size(); ++i) { unsigned b; memcpy(&b, &w[i], 4); w16[i] = (unsigned short)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16); }
                FAV_HIP(hipMalloc(reinterpret_cast<void**>(&d.wgt16), w16.size() * 2));
                FAV_HIP(hipMemcpy(d.wgt16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
            }
            if (L.transposed && (L.stride != 2 || L.adj != 1 || L.pad > L.k - 1)) {
                set_error("network: SpatialFullConvolution is supported for stride 2, adj 1 (models_video.lua:99-102), got s=%d adj=%d", L.stride, L.adj);
                return FAV_EUNSUPPORTED; }
            if (!L.transposed && conv_c8d_eligible(d.cinp, L.cin, d.coutp, L.k, L.stride, 0, 0)) {      // first layer: dense-K pairing
                std::vector<float> wd;
                conv_c8d_pack(L.w.data(), L.cin, L.cout, wd);
                rc = dev_upload(wd, 0, &d.wc8d); if (rc) return rc;
                std::vector<float> wf1;                                            // the same layer with F(2,3) along x (kernels_first.hip)
                conv_first_pack(L.w.data(), L.cin, L.cout, wf1);
                rc = dev_upload(wf1, 0, &d.wfirst); if (rc) return rc;
                std::vector<float> wf2;                                            // ... and with F(2x2,3x3) over its nine 3x3 blocks
                conv_first2d_pack(L.w.data(), L.cin, L.cout, wf2);
                rc = dev_upload(wf2, 0, &d.wfirst2d); if (rc) return rc;
            } else if (!L.transposed && conv_first2d_eligible(d.cinp, L.cin, d.coutp, L.k, L.stride, 0, 0)) {      // first layer with more than 32 filters: groups of 32 on the 2-D form
                std::vector<float> wf2;
                conv_first2d_pack_groups(L.w.data(), L.cin, L.cout, d.coutp, wf2);
                rc = dev_upload(wf2, 0, &d.wfirst2d); if (rc) return rc;
            }
            if (!L.transposed && L.cin == d.cinp && conv3_wino_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, 0, 0)) {      // residual 3x3: Winograd
                std::vector<float> ww;
                conv_wino_pack(L.w.data(), L.cin, L.cout, ww);
                rc = dev_upload(ww, 0, &d.wwino); if (rc) return rc;
            }
            if (!L.transposed && L.cin == d.cinp && !tuning().wino_f2 && conv3_wino4_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, 0, 0)) {      // ... as F(4x4,3x3) (round 4), any number of 128-filter groups (round 5)
                std::vector<float> w4;
                conv_wino4_pack_groups(L.w.data(), L.cin, L.cout, w4);
                rc = dev_upload(w4, 0, &d.wwino4); if (rc) return rc;
            }
            if (!L.transposed && L.cin == d.cinp && conv3_up2_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, 1, 1)) {      // 3x3 after a x2 upsampling: merged 2x2 taps
                std::vector<float> wu, wu9;
                if (L.cout == 64) {
                    conv_up2_pack(L.w.data(), L.cin, wu);
                    conv_up2w_pack(L.w.data(), L.cin, wu9);                        // the nine-position form follows the phase-merged one
                    wu.insert(wu.end(), wu9.begin(), wu9.end());
                } else conv_up2w_pack_groups(L.w.data(), L.cin, L.cout, wu);      // more than 64 filters: the nine-position form only, one block per group of 64
                rc = dev_upload(wu, 0, &d.wup2); if (rc) return rc;
            }
            if (!L.transposed && L.cin == d.cinp && conv3s2w_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, 1, 0)) {      // 3x3 stride 2: fragment order
                std::vector<float> ws;
                conv_s2w_pack_groups(L.w.data(), L.cin, L.cout, ws);
                rc = dev_upload(ws, 0, &d.ws2w); if (rc) return rc;
            }
            if (!L.transposed && conv_fold_eligible(d.cinp, L.cout, L.k, L.stride)) {
                // [ky][n = c*k + kx][ci] for the row-folded last-layer kernel
                // followed by the K+1 merged slices Wm[m] = W[m-1] + W[m] (W[-1] = W[K] = 0) for a x2-upsampled input
                // (conv_rowfold_up2_kernel: the logical rows 2r and 2r+1 are the same physical row)
                std::vector<float> wf((size_t)(2 * L.k + 1) * 32 * d.cinp, 0.f);
                for (int co = 0; co < L.cout; ++co)
                    for (int ci = 0; ci < L.cin; ++ci)
                        for (int ky = 0; ky < L.k; ++ky)
                            for (int kx = 0; kx < L.k; ++kx)
                                wf[((size_t)ky * 32 + co * L.k + kx) * d.cinp + ci] = L.w[(((size_t)co * L.cin + ci) * L.k + ky) * L.k + kx];
                for (int co = 0; co < L.cout; ++co)
                    for (int ci = 0; ci < L.cin; ++ci)
                        for (int m = 0; m <= L.k; ++m)
                            for (int kx = 0; kx < L.k; ++kx) {
                                const float* wr = &L.w[((size_t)co * L.cin + ci) * L.k * L.k + kx];
                                const double a = m >= 1 ? (double)wr[(size_t)(m - 1) * L.k] : 0.0, b = m < L.k ? (double)wr[(size_t)m * L.k] : 0.0;
                                wf[((size_t)(L.k + m) * 32 + co * L.k + kx) * d.cinp + ci] = (float)(a + b);
                            }
                rc = dev_upload(wf, 0, &d.wfold); if (rc) return rc;
            }
            chan_pitch = L.cout;
            maxc = std::max(maxc, std::max(d.coutp, d.cinp));
        } else if (L.type == L_IN) {
            if ((int)L.gamma.size() != chan_pitch) { set_error("network: InstanceNormalization(%zu) after %d channels", L.gamma.size(), chan_pitch); return FAV_EFORMAT; }
            ins.emplace_back();
            DevIN& d = ins.back();
            int rc = dev_upload(L.gamma, 0, &d.gamma); if (rc) return rc;
            rc = dev_upload(L.beta, 0, &d.beta); if (rc) return rc;
            FAV_HIP(hipMalloc(reinterpret_cast<void**>(&d.scale), L.gamma.size() * sizeof(float)));
            FAV_HIP(hipMalloc(reinterpret_cast<void**>(&d.shift), L.gamma.size() * sizeof(float)));
            d.acc_bytes = 2 * stat_acc_words((int)L.gamma.size()) * sizeof(long long);
            FAV_HIP(hipMalloc(reinterpret_cast<void**>(&d.acc), d.acc_bytes));
            FAV_HIP(hipMemset(d.acc, 0, d.acc_bytes));
        } else if (L.type == L_BN) {
            if ((int)L.mean.size() != chan_pitch) { set_error("network: SpatialBatchNormalization(%zu) after %d channels", L.mean.size(), chan_pitch); return FAV_EFORMAT; }
            // evaluate mode: a fixed per-channel affine, folded into the consumer's load like InstanceNorm's
            std::vector<float> sc(L.mean.size()), sh(L.mean.size());
            for (size_t i = 0; i < sc.size(); ++i) {
                const double s = (double)L.gamma[i] / std::sqrt((double)L.var[i] + (double)L.eps);
                sc[i] = (float)s; sh[i] = (float)((double)L.beta[i] - (double)L.mean[i] * s);
            }
            ins.emplace_back();
            DevIN& d = ins.back();
            int rc = dev_upload(sc, 0, &d.scale); if (rc) return rc;
            rc = dev_upload(sh, 0, &d.shift); if (rc) return rc;
        } else if (L.type == L_RES) {
            int cp = chan_pitch;
            int rc = upload_layers(L.block, cp, maxc); if (rc) return rc;
            if (cp != chan_pitch) { set_error("network: residual branch changes the channel count"); return FAV_EUNSUPPORTED; }
        }
    }
    return FAV_OK;
}

int fav_net::upload()
{
    FAV_HIP(hipSetDevice(device));
    if (layers.empty()) { set_error("network: empty model"); return FAV_EFORMAT; }
    // A LEADING symmetric reflection padding (padding_type reflect-start: train_video.lua:319-325; also the pad in front of the first
    // convolution with padding_type reflect, models_video.lua:70-72) is folded into the input assembly (prep_input / check_prep write the
    // reflected copies).  Every other padding layer -- replication, asymmetric, or further inside the network (models_video.lua:12-16,27-31:
    // padding_type reflect / replicate pads in front of EVERY convolution) -- runs as a gather launch (launch_pad_nhwc).
    pad = 0; pad_folded = false;
    if (layers[0].type == L_PAD) {
        const Layer& P = layers[0];
        if (P.pad_mode == 0 && P.pl == P.pr && P.pl == P.pt && P.pl == P.pb) { pad = P.pl; pad_folded = true; }
    }
    in_channels = 0;
    for (const Layer& L : layers) if (L.type == L_CONV) { in_channels = L.cin; break; }
    if (in_channels != 7 && in_channels != 3) {
        set_error("network: first convolution has %d input channels; video models take 7 (models_video.lua:57), image models 3", in_channels);
        return FAV_EUNSUPPORTED; }
    exec = layers;
    int chan0 = 8;
    int real0 = in_channels; bool seen_conv = false;
    int rc = pad_channel_counts(exec, chan0, real0, seen_conv, true); if (rc) return rc;
    int chan = 8, maxc = 8;
    rc = upload_layers(exec, chan, maxc); if (rc) return rc;
    params = count_params(layers);
    std::vector<float> o((size_t)maxc, 1.f), z((size_t)maxc, 0.f);
    rc = dev_upload(o, 0, &ones); if (rc) return rc;
    rc = dev_upload(z, 0, &zeros); if (rc) return rc;
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(&sk_ws), conv_streamk_workspace_bytes()));
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(&sk_flags), conv_streamk_grid() * sizeof(unsigned)));
    FAV_HIP(hipMemset(sk_flags, 0, conv_streamk_grid() * sizeof(unsigned)));
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(&ks_ws), conv3_wino4_ksplit_bytes()));
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(&ks_cnt), 256 * sizeof(int)));
    FAV_HIP(hipMemset(ks_cnt, 0, 256 * sizeof(int)));
    FAV_HIP(hipHostMalloc(reinterpret_cast<void**>(&sk_err_host), sizeof(unsigned), hipHostMallocMapped));
    *sk_err_host = 0;
    FAV_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&sk_err_dev), sk_err_host, 0));
    return FAV_OK;
}

int fav_net::alloc(size_t bytes, float** out)
{
    bytes = (bytes + 255) / 256 * 256;
    if (cursor < bufs.size()) {
        if (bufs[cursor].bytes < bytes) { set_error("internal: activation arena mismatch"); return FAV_EINVAL; }
        *out = static_cast<float*>(bufs[cursor++].p);
        return FAV_OK;
    }
    if (slab_left < bytes) {
        const size_t sz = std::max(bytes, (size_t)256 << 20);
        void* sp = nullptr;
        FAV_HIP(hipMalloc(&sp, sz));
        slabs.push_back(sp); slab_cur = static_cast<char*>(sp); slab_left = sz;
    }
    DevBuf b; b.bytes = bytes; b.p = slab_cur;
    slab_cur += bytes; slab_left -= bytes;
    bufs.push_back(b); ++cursor;
    *out = static_cast<float*>(b.p);
    return FAV_OK;
}

int fav_net::timed_conv(const ConvLaunch& c, int conv_index, const Layer& L)
{
    const float* wfold = (c.final_mode && !tuning().no_fold && conv_fold_launchable(c.CIN, c.KH, c.pad, c.ups, c.IH, c.IW)) ? convs[conv_index].wfold : nullptr;
    const float* c8d_w = (use_c8 && !tuning().no_c8d && conv_c8d_eligible(c.CIN, L.cin, c.COUTp, L.k, c.stride, c.pre.stages, c.ups)) ? convs[conv_index].wc8d : nullptr;
    if (c.pre.acc1 != nullptr && !(use_wino && use_wino4)) { set_error("internal: accumulator-form InstanceNorm in front of a kernel that cannot take it"); return FAV_EINVAL; }
    ConvLaunch cs = c;
    cs.reserve_cus = reserve_cus;
    cs.no_sk = shared_device ? 1 : 0;
    cs.sk_ws = sk_ws; cs.sk_flags = sk_flags; cs.sk_epoch = ++sk_epoch;      // launches of one net are stream-ordered
    cs.sk_err = sk_err_dev;
    cs.ks_ws = ks_ws; cs.ks_cnt = ks_cnt;
    if (sk_epoch == 0xffffffffu) sk_epoch = 0;
    // generic kernel while look-ahead masks are in flight: its stream-K hand-off assumes that all blocks are resident at once, and the
    // side queues' kernels land on any CU -- owners then wait for blocks that have not started (d128: 186 us against 115 us alone,
    // profiles/r02p_4arg_kernel_stats.csv).  Data-parallel grids do not wait for anybody.  The halo-resident 3x3 kernel (bf16 fast
    // mode, FAV_NO_WINO) hands tiles over the same way and takes the same descriptor.
    ConvLaunch cg = cs;
    static const int side_sk_mode = diag_env("FAV_SIDE_SK") ? atoi(diag_env("FAV_SIDE_SK")) : 0;      // (tuning: read once) 1: keep stream-K next to the side queues, 2: for the stride-2 halo kernel only
    if (reserve_cus > 0 && side_sk_mode != 1) cg.no_sk = 1;
    auto go = [&]() { return use_first ? (use_first2d ? launch_conv_first2d(cs, L.cin, convs[conv_index].wfirst2d, c8_counts, st) : launch_conv_first(cs, L.cin, convs[conv_index].wfirst, c8_counts, st)) : use_s2w ? launch_conv3s2w(cs, convs[conv_index].ws2w, c8_counts, st) : use_up2 ? launch_conv3_up2(cs, convs[conv_index].wup2, c8_counts, st) : use_wino ? (use_wino4 ? launch_conv3_wino4(cs, convs[conv_index].wwino4, c8_counts, st) : launch_conv3_wino(cs, convs[conv_index].wwino, c8_counts, st)) : wfold ? launch_conv_fold(cs, wfold, st) : (use_c8 ? (c8d_w ? launch_conv_c8d(cs, L.cin, c8d_w, c8_counts, st) : launch_conv_c8(cs, c8_counts, st)) : (use_h3 ? launch_conv3_halo(cg, c8_counts, st) : (use_s2 ? launch_conv3s2(side_sk_mode == 2 ? cs : cg, c8_counts, st) : launch_conv(cg, st)))); };
    char tag[96] = "";
    if (TraceRange::enabled()) snprintf(tag, sizeof tag, "fav:conv%d k%d s%d %d->%d %dx%d", conv_index, L.k, L.stride, L.cin, L.cout, c.OW, c.OH);
    TraceRange tr(tag);
    if (!profiling) return go();
    ProfRec r; r.conv = conv_index;
    // (no system-scope fence at the event: the default one flushes the caches around every timed kernel -- 6 us on either side of each
    // convolution in the rocprofv3 trace, 0.18 ms per 1280x720 frame)
    FAV_HIP(hipEventCreateWithFlags(&r.a, hipEventDisableSystemFence)); FAV_HIP(hipEventCreateWithFlags(&r.b, hipEventDisableSystemFence));
    FAV_HIP(hipEventRecord(r.a, st));
    int rc = go();
    FAV_HIP(hipEventRecord(r.b, st));
    prof_pending.push_back(r);
    if ((int)prof_ms.size() <= conv_index) { prof_ms.resize(conv_index + 1, 0.0); prof_macs.resize(conv_index + 1, 0.0); prof_n.resize(conv_index + 1, 0); prof_tile.resize(conv_index + 1, 0); }
    prof_macs[conv_index] = (double)c.OH * c.OW * L.cout * L.cin * L.k * L.k;      // useful MACs only
    // kernel id: 16 first layer with F(2x2,3x3) over its nine 3x3 blocks, 6 first layer with F(2,3) along x, 500+N 3x3 on a x2-upsampled input (merged taps), 700+N stride-2 3x3 (fragment-order weights), 400+N Winograd 3x3 (+1: with a pending residual join as its input), 1 row-folded last layer, 8 first layer, 300+N halo 3x3 (N = 64|128), 200+N stride-2 halo 3x3, else the generic kernel's N tile
    prof_tile[conv_index] = use_first ? (use_first2d ? 16 : 6) : use_s2w ? 700 + c.COUTp : use_up2 ? 500 + c.COUTp : use_wino ? (use_wino4 ? 600 : 400) + c.COUTp + (c.join_skip ? 1 : 0) : wfold ? 1 : (use_c8 ? (c8d_w ? 7 : 8) : (use_h3 ? 300 + c.COUTp : (use_s2 ? 200 + c.COUTp : (c.COUTp % 128 == 0 ? 128 : (c.COUTp % 64 == 0 ? 64 : 32)))));
    return rc;
}

static int count_convs(const std::vector<Layer>& ls)
{
    int n = 0;
    for (const Layer& l : ls) { if (l.type == L_CONV) ++n; else if (l.type == L_RES) n += count_convs(l.block); }
    return n;
}

// conv - InstanceNorm - ReLU - conv with both convolutions on the F(4x4) kernel (the first half of a residual branch,
// models_video.lua:10-39): the first convolution adds its units' statistics to the InstanceNorm's accumulators and the second forms
// scale / shift from them in its prologue -- no in_finalize launch between the two (round 5).  ls[li] = the first convolution;
// conv_cursor already points at the second one's weights
bool fav_net::acc_stats_ok(const std::vector<Layer>& ls, size_t li) const
{
    if (tuning().no_acc_stats || tuning().no_wino || precision != 0) return false;
    if (li + 3 >= ls.size() || ls[li + 1].type != L_IN || ls[li + 2].type != L_RELU || ls[li + 3].type != L_CONV) return false;
    if (conv_cursor >= convs.size()) return false;
    const Layer& c2 = ls[li + 3]; const DevConvW& d2 = convs[conv_cursor];
    return !c2.transposed && d2.wwino4 != nullptr && conv3_wino4_eligible(d2.cinp, c2.cout, d2.coutp, c2.k, c2.stride, c2.pad, 1, 0);
}

// conv - InstanceNorm - ReLU - conv - InstanceNorm (models_video.lua:10-39) with both convolutions on the Winograd kernel
bool fav_net::res_block_is_winograd(const Layer& R, size_t first_conv) const
{
    const std::vector<Layer>& b = R.block;
    if (tuning().no_wino || b.size() != 5 || b[0].type != L_CONV || b[1].type != L_IN || b[2].type != L_RELU || b[3].type != L_CONV || b[4].type != L_IN) return false;
    if (first_conv + 1 >= convs.size()) return false;
    for (int k = 0; k < 2; ++k) {
        const Layer& c = b[k ? 3 : 0]; const DevConvW& d = convs[first_conv + (size_t)k];
        if (c.transposed || d.wwino == nullptr || !conv3_wino_eligible(d.cinp, c.cout, d.coutp, c.k, c.stride, c.pad, 1, 0)) return false;
    }
    return true;
}

int fav_net::run(std::vector<Layer>& ls, Act& cur, bool top, float* out_planar, float* out_raw)
{
    for (size_t li = 0; li < ls.size(); ++li) {
        Layer& L = ls[li];
        switch (L.type) {
        case L_PAD: {
            if (top && li == 0 && pad_folded) break;        // folded into the input assembly (upload())
            if (cur.data == nullptr || cur.join_skip != nullptr) { set_error("network: misplaced padding layer"); return FAV_EUNSUPPORTED; }
            // reflection needs pad < size on each axis (nn.SpatialReflectionPadding asserts the same)
            if (L.pad_mode == 0 && (std::max(L.pl, L.pr) >= cur.W() || std::max(L.pt, L.pb) >= cur.H())) { set_error("network: reflection padding %d %d %d %d of a %dx%d tensor", L.pl, L.pr, L.pt, L.pb, cur.W(), cur.H()); return FAV_EINVAL; }
            // index map only: a pending per-channel transform (InstanceNorm / ReLU) commutes with it and stays pending
            Act nxt;
            nxt.Hp = cur.H() + L.pt + L.pb; nxt.Wp = cur.W() + L.pl + L.pr; nxt.C = cur.C; nxt.pre = cur.pre;
            if (nxt.pre.acc1 != nullptr) { set_error("internal: accumulator-form InstanceNorm in front of a padding layer"); return FAV_EINVAL; }
            int rc = alloc((size_t)nxt.Hp * nxt.Wp * nxt.C * sizeof(float), &nxt.data); if (rc) return rc;
            rc = launch_pad_nhwc(cur.data, cur.Hp, cur.Wp, cur.P(), cur.C, cur.ups, nxt.data, L.pl, L.pr, L.pt, L.pb, L.pad_mode, st); if (rc) return rc;
            cur = nxt;
            break;
        }
        case L_CONV: {
            const DevConvW& d = convs[conv_cursor++];
            if (cur.C != d.cinp) { set_error("internal: channel pitch mismatch (%d vs %d)", cur.C, d.cinp); return FAV_EINVAL; }
            ConvLaunch c;
            c.in = cur.data; c.IH = cur.H(); c.IW = cur.W(); c.IWp = cur.P(); c.ups = cur.ups; c.CIN = d.cinp;
            c.pre = cur.pre;
            c.wgt = d.wgt; c.bias = d.bias; c.COUT = L.cout; c.COUTp = d.coutp; c.KH = c.KW = L.k; c.stride = L.stride;
            c.pad = L.pad; c.Kpad = d.kpad;
            if (L.transposed) {
                // stride-2 transposed convolution = stride-1 convolution over the zero-stuffed input (size 2*in with adj 1)
                if (cur.ups != 0) { set_error("network: SpatialFullConvolution directly after an upsampling is unsupported"); return FAV_EUNSUPPORTED; }
                c.ups = 1; c.stuff = 1; c.IH = 2 * cur.Hp; c.IW = 2 * cur.Wp; c.stride = 1; c.pad = L.k - 1 - L.pad;
                c.OH = c.IH + 2 * c.pad - L.k + 1; c.OW = c.IW + 2 * c.pad - L.k + 1;
            } else {
                c.OH = (c.IH + 2 * L.pad - L.k) / L.stride + 1;
                c.OW = (c.IW + 2 * L.pad - L.k) / L.stride + 1;
            }
            if (c.IH + 2 * c.pad < L.k || c.IW + 2 * c.pad < L.k) { set_error("network: input too small for the architecture"); return FAV_EINVAL; }
            bool has_tanh = false; float mul = 1.f;
            const bool is_final = top && only_tail(ls, li + 1, has_tanh, mul) && has_tanh && L.cout == 3;
            Act nxt;
            nxt.Hp = c.OH; nxt.Wp = c.OW; nxt.C = L.cout;
            if (is_final && L.transposed) { set_error("network: a transposed convolution as the last layer is unsupported"); return FAV_EUNSUPPORTED; }
            if (is_final) {
                c.final_mode = 1; c.tanh_mul = mul; c.out_planar = out_planar; c.out_raw_nchw = out_raw;
                int rc = timed_conv(c, (int)conv_cursor - 1, L); if (rc) return rc;
                cur = nxt;
                return FAV_OK;        // Tanh / MulConstant / TotalVariation are folded into the epilogue
            }
            if (L.cout % 4 != 0) { set_error("network: %d output channels (must be a multiple of 4 except for the last layer)", L.cout); return FAV_EUNSUPPORTED; }
            const bool pitched_out = lazy.active && (int)conv_cursor - 1 == lazy.conv_index;
            int rc;
            if (pitched_out) {
                // the output of a block whose join stays pending: pixel (i, j) at the linear index of the skip's pixel (i + shave, j + shave)
                if (c.OH != lazy.rows - 2 * lazy.shave || c.OW > lazy.pitch - 2 * lazy.shave) { set_error("internal: pending residual join of mismatching shapes"); return FAV_EINVAL; }
                float* base = nullptr;
                rc = alloc((size_t)lazy.rows * lazy.pitch * L.cout * sizeof(float), &base); if (rc) return rc;
                nxt.data = base + ((size_t)lazy.shave * lazy.pitch + lazy.shave) * L.cout; nxt.pitch = lazy.pitch; c.OWp = lazy.pitch;
            } else { rc = alloc((size_t)c.OH * c.OW * L.cout * sizeof(float), &nxt.data); if (rc) return rc; }
            const bool want_stats = li + 1 < ls.size() && ls[li + 1].type == L_IN;
            const bool c8 = !L.transposed && conv_c8_eligible(d.cinp, d.coutp, L.k, L.stride, cur.pre.stages, cur.ups) && !tuning().no_c8;
            const bool wino4 = !L.transposed && d.wwino4 != nullptr && precision == 0 && !tuning().no_wino &&
                               conv3_wino4_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, cur.pre.stages, cur.ups);
            const bool wino = wino4 || (!L.transposed && d.wwino != nullptr && precision == 0 && !tuning().no_wino &&
                                        conv3_wino_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, cur.pre.stages, cur.ups));
            // (a first layer with more than 32 filters: the 2-D minimal-filtering kernel in groups of 32; no other special kernel takes it)
            const bool first_wide = !c8 && !L.transposed && d.wfirst2d != nullptr && !tuning().no_first && !tuning().no_c8d &&
                                    conv_first2d_eligible(d.cinp, L.cin, d.coutp, L.k, L.stride, cur.pre.stages, cur.ups);
            const bool first = first_wide || (c8 && d.wfirst != nullptr && !tuning().no_first && !tuning().no_c8d && conv_c8d_eligible(d.cinp, L.cin, d.coutp, L.k, L.stride, cur.pre.stages, cur.ups));
            const bool up2 = !L.transposed && d.wup2 != nullptr && precision == 0 && !tuning().no_up2 &&
                             conv3_up2_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, cur.pre.stages, cur.ups);
            const bool s2w = !L.transposed && d.ws2w != nullptr && !tuning().no_s2w &&
                             conv3s2w_eligible(d.cinp, L.cout, d.coutp, L.k, L.stride, L.pad, cur.pre.stages, cur.ups);
            const bool h3 = !wino && !up2 && !s2w && !L.transposed && conv3_halo_eligible(d.cinp, d.coutp, L.k, L.stride) && !tuning().no_h3;
            const bool s2 = !L.transposed && !s2w && !h3 && !c8 && conv3s2_eligible(d.cinp, d.coutp, L.k, L.stride, cur.pre.stages, cur.ups) && !tuning().no_s2;
            static const bool first_1d = diag_env("FAV_FIRST_1D") != nullptr;      // (tuning: read once) the 1-D form of the first layer
            const bool first2d = first_wide || (first && d.wfirst2d != nullptr && !first_1d);
            nxt.mblocks = first ? (first2d ? conv_first2d_tiles(c.OH, c.OW) : conv_first_tiles(c.OH, c.OW)) : s2w ? conv3s2w_tiles(c.OH, c.OW, d.coutp) : up2 ? conv3_up2_tiles(c.OH, c.OW) : wino ? (wino4 ? conv3_wino4_tiles(c.OH, c.OW) : conv3_wino_tiles(c.OH, c.OW)) : c8 ? conv_c8_tiles(c.OH, c.OW) : (h3 ? conv3_halo_tiles(c.OH, c.OW, precision == 0) : (s2 ? conv3s2_tiles(c.OH, c.OW) : conv_mblocks(c.OH, c.OW))); nxt.ppitch = d.coutp;
            const bool acc = wino4 && want_stats && !pitched_out && cur.join_skip == nullptr &&
                             (acc_stats_ok(ls, li) || (branch_tail_acc_ok && !top && li + 2 == ls.size() && !tuning().no_acc_stats && precision == 0));
            if (acc) {
                DevIN& din = ins[in_cursor];                   // the InstanceNorm that follows
                nxt.acc = din.acc + (size_t)acc_parity * stat_acc_words(L.cout); nxt.acc_other = din.acc + (size_t)(acc_parity ^ 1) * stat_acc_words(L.cout);
                c.stat_acc = nxt.acc;
            }
            if (want_stats && !acc) { rc = alloc((size_t)nxt.mblocks * d.coutp * 2 * sizeof(float), &nxt.partials); if (rc) return rc; }
            if (want_stats && !acc && (c8 || first || h3 || s2 || wino || up2 || s2w)) { float* cp = nullptr; rc = alloc((size_t)nxt.mblocks * sizeof(int), &cp); if (rc) return rc; nxt.counts = reinterpret_cast<int*>(cp); }
            c.out = nxt.data; c.partials = nxt.partials;
            if (cur.join_skip != nullptr || pitched_out) {
                if (!wino) { set_error("internal: a pending residual join next to a convolution that is not the Winograd kernel's"); return FAV_EINVAL; }
                c.join_skip = cur.join_skip; c.join_out = cur.join_out;
            }
            if (h3 && precision == 1) c.wgt16 = d.wgt16;
            c8_counts = (c8 || first || h3 || s2 || wino || up2 || s2w) ? (nxt.counts ? nxt.counts : reinterpret_cast<int*>(zeros)) : nullptr; use_c8 = c8; use_h3 = h3; use_s2 = s2; use_wino = wino; use_wino4 = wino4; use_up2 = up2; use_first = first; use_first2d = first2d; use_s2w = s2w;
            rc = timed_conv(c, (int)conv_cursor - 1, L); if (rc) return rc;
            use_c8 = false; use_h3 = false; use_s2 = false; use_wino = false; use_wino4 = false; use_up2 = false; use_first = false; use_first2d = false; use_s2w = false;
            cur = nxt;
            break;
        }
        case L_IN: {
            const DevIN& d = ins[in_cursor++];
            const int C = (int)L.gamma.size();
            const int M = cur.Hp * cur.Wp;
            if (cur.data == nullptr || C != cur.C) { set_error("network: misplaced InstanceNormalization"); return FAV_EUNSUPPORTED; }
            if (cur.acc != nullptr && cur.pre.stages == 0) {
                // nothing is launched: the consuming convolution forms scale / shift from the accumulators (acc_stats_ok)
                cur.pre = Affine();
                cur.pre.acc1 = cur.acc; cur.pre.acc1_zero = cur.acc_other; cur.pre.gamma1 = d.gamma; cur.pre.beta1 = d.beta; cur.pre.eps1 = L.eps; cur.pre.count1 = M;
                cur.pre.relu1 = 0; cur.pre.stages = 1;
            } else if (cur.partials != nullptr && cur.pre.stages == 0) {
                int rc = launch_in_finalize(cur.partials, cur.counts, cur.mblocks, M, CONV_BM, C, cur.ppitch, d.gamma, d.beta, L.eps,
                                            d.scale, d.shift, st);
                if (rc) return rc;
                cur.pre.scale1 = d.scale; cur.pre.shift1 = d.shift; cur.pre.relu1 = 0; cur.pre.stages = 1;
            } else {
                if (cur.pre.stages >= 2) { set_error("network: more than two stacked normalisations on one tensor are unsupported"); return FAV_EUNSUPPORTED; }
                // statistics of the pending-transformed tensor (nearest upsampling replicates every
                // element s*s times and leaves mean and biased variance unchanged)
                float* part = nullptr;
                const int mb = (M + 127) / 128;
                int rc = alloc((size_t)mb * C * 2 * sizeof(float), &part); if (rc) return rc;
                rc = launch_stats(cur.data, M, C, cur.pre, part, st); if (rc) return rc;
                rc = launch_in_finalize(part, nullptr, mb, M, 128, C, C, d.gamma, d.beta, L.eps, d.scale, d.shift, st); if (rc) return rc;
                if (cur.pre.stages == 0) { cur.pre.scale1 = d.scale; cur.pre.shift1 = d.shift; cur.pre.relu1 = 0; cur.pre.stages = 1; }
                else { cur.pre.scale2 = d.scale; cur.pre.shift2 = d.shift; cur.pre.relu2 = 0; cur.pre.stages = 2; }
            }
            cur.partials = nullptr; cur.counts = nullptr; cur.acc = nullptr; cur.acc_other = nullptr;
            break;
        }
        case L_BN: {
            const DevIN& d = ins[in_cursor++];
            if (cur.data == nullptr || (int)L.mean.size() != cur.C) { set_error("network: misplaced SpatialBatchNormalization"); return FAV_EUNSUPPORTED; }
            if (cur.pre.stages >= 2) { set_error("network: more than two stacked normalisations on one tensor are unsupported"); return FAV_EUNSUPPORTED; }
            if (cur.pre.stages == 0) { cur.pre.scale1 = d.scale; cur.pre.shift1 = d.shift; cur.pre.relu1 = 0; cur.pre.stages = 1; }
            else { cur.pre.scale2 = d.scale; cur.pre.shift2 = d.shift; cur.pre.relu2 = 0; cur.pre.stages = 2; }
            cur.partials = nullptr; cur.counts = nullptr;
            break;
        }
        case L_RELU:
            if (cur.pre.stages == 0) { cur.pre.scale1 = ones; cur.pre.shift1 = zeros; cur.pre.relu1 = 1; cur.pre.stages = 1; }
            else if (cur.pre.stages == 1) cur.pre.relu1 = 1;
            else cur.pre.relu2 = 1;
            cur.partials = nullptr;
            break;
        case L_UP:
            if (L.scale != 2 || cur.ups != 0) { set_error("network: only a single x2 nearest upsampling per convolution is supported"); return FAV_EUNSUPPORTED; }
            cur.ups = 1;
            break;
        case L_RES: {
            if (cur.ups != 0) { set_error("network: residual block directly after an upsampling is unsupported"); return FAV_EUNSUPPORTED; }
            Act skip = cur;
            Act br = cur;
            br.partials = nullptr;
            int rc;
            if (cur.join_skip != nullptr) {
                // the previous block's join is pending: this block's first convolution forms it while staging its input and writes it
                // out -- that tensor is this block's skip
                float* zb = nullptr;
                rc = alloc((size_t)cur.Hp * cur.P() * cur.C * sizeof(float), &zb); if (rc) return rc;
                br.join_out = zb;
                skip = Act(); skip.data = zb; skip.Hp = cur.Hp; skip.Wp = cur.Wp; skip.C = cur.C; skip.pitch = cur.P();
            }
            // Leave THIS block's join pending when the next layer is another residual block that starts with a Winograd convolution
            // (models_video.lua:41-53, R128 x 5): the 15 us res_add launch (88 MB at the HBM roofline) becomes 33 MB of extra reads and
            // 33 MB of writes inside a kernel that is bound by its matrix instructions.  The skip must be a plain tensor (the first
            // block's skip still carries d128's InstanceNorm + ReLU: its join stays a launch).
            // Since round 4 only with the F(2x2) kernels (FAV_WINO_F2), or on request (FAV_LAZY_JOIN): inside the F(4x4) kernel the joined
            // rows' stores cost 10 us of its K loop (the weight ring runs dry behind them) on top of 8 us of staging -- more than the
            // 16 us launch they replace (639 against 634 frames/s, profiles/r4s_stream_k_and_joins_ab.log).  (The F(4x4) kernel's pending-join
            // instantiation is kept for FAV_LAZY_JOIN and the tests that pin its bits against the launched joins; since the row requests carry
            // their own offsets it spills inside its K loop -- nobody tuned it further)
            static const bool no_lazy = diag_env("FAV_NO_LAZY_JOIN") != nullptr;      // (tuning: read once)
            static const bool want_lazy = diag_env("FAV_LAZY_JOIN") != nullptr;
            const int nconv = count_convs(L.block);
            const bool lazy_out = !no_lazy && (tuning().wino_f2 || want_lazy) && precision == 0 && li + 1 < ls.size() && ls[li + 1].type == L_RES && skip.pre.stages == 0 && skip.ups == 0 &&
                                  res_block_is_winograd(L, conv_cursor) && res_block_is_winograd(ls[li + 1], conv_cursor + (size_t)nconv);
            if (lazy_out) { lazy.active = true; lazy.pitch = skip.P(); lazy.rows = skip.Hp; lazy.shave = L.shave; lazy.conv_index = (int)conv_cursor + nconv - 1; }
            // the join below is a plain res_add launch (no statistics of its own, not left pending): it can take the branch's last
            // InstanceNorm as accumulators (round 5) -- no in_finalize launch between the branch's last convolution and the join
            {
                size_t nx0 = li + 1;
                if (nx0 < ls.size() && ls[nx0].type == L_UP && ls[nx0].scale == 2) ++nx0;
                const bool join_stats = nx0 < ls.size() && ls[nx0].type == L_IN;
                branch_tail_acc_ok = !lazy_out && !join_stats;
            }
            rc = run(L.block, br, false, nullptr, nullptr);
            lazy.active = false; branch_tail_acc_ok = false;
            if (rc) return rc;
            if (br.pre.stages != 1 || br.pre.relu1 || br.ups != 0) { set_error("network: residual branch must end in conv + InstanceNormalization"); return FAV_EUNSUPPORTED; }
            if (br.Hp != skip.Hp - 2 * L.shave || br.Wp != skip.Wp - 2 * L.shave || br.C != skip.C) {
                set_error("network: residual branch output %dx%d does not match the shaved skip %dx%d", br.Wp, br.Hp,
                          skip.Wp - 2 * L.shave, skip.Hp - 2 * L.shave);
                return FAV_EUNSUPPORTED; }
            if (lazy_out) {
                // nothing is launched: data = the branch's output (laid out under the skip), pre = its InstanceNorm, plus the skip
                cur = br;
                cur.partials = nullptr; cur.counts = nullptr;
                cur.join_skip = skip.data + ((size_t)L.shave * skip.P() + L.shave) * skip.C; cur.join_out = nullptr;
                break;
            }
            Act z; z.Hp = br.Hp; z.Wp = br.Wp; z.C = br.C;
            rc = alloc((size_t)z.Hp * z.Wp * z.C * sizeof(float), &z.data); if (rc) return rc;
            // the join feeds an InstanceNorm (directly, or through the x2 nearest upsample of models_video.lua:94-98, which leaves
            // mean and biased variance unchanged): take that norm's statistics in the same pass
            size_t nx = li + 1;
            if (nx < ls.size() && ls[nx].type == L_UP && ls[nx].scale == 2) ++nx;
            if (nx < ls.size() && ls[nx].type == L_IN) {
                z.mblocks = res_add_stat_blocks(z.Hp, z.Wp); z.ppitch = z.C;
                rc = alloc((size_t)z.mblocks * z.C * 2 * sizeof(float), &z.partials); if (rc) return rc;
                float* cp = nullptr; rc = alloc((size_t)z.mblocks * sizeof(int), &cp); if (rc) return rc; z.counts = reinterpret_cast<int*>(cp);
            }
            rc = launch_res_add(br.data, br.pre.scale1, br.pre.shift1, skip.data, skip.Hp, skip.Wp, L.shave, skip.pre, z.C,
                                z.data, z.partials, z.counts, st, skip.P(), &br.pre);
            if (rc) return rc;
            cur = z;
            break;
        }
        case L_TANH: case L_MUL:
            set_error("network: Tanh/MulConstant are only supported after the last convolution"); return FAV_EUNSUPPORTED;
        case L_IDENTITY: break;
        }
    }
    if (top) { set_error("network: the model does not end in a 3-channel convolution followed by Tanh"); return FAV_EUNSUPPORTED; }
    return FAV_OK;
}

void fav_net::out_size(int H, int W, int* Ho, int* Wo) const
{
    // shape walk (models_video.lua:55-140): convs floor((in+2p-k)/s)+1, residual blocks shave, upsampling doubles
    int h = H + 2 * pad, w = W + 2 * pad;
    std::function<void(const std::vector<Layer>&)> walk = [&](const std::vector<Layer>& ls) {
        for (const Layer& L : ls) {
            if (L.type == L_PAD) { if (&L == &layers[0] && pad_folded) continue; h += L.pt + L.pb; w += L.pl + L.pr; }
            else if (L.type == L_CONV && L.transposed) { h = (h - 1) * L.stride - 2 * L.pad + L.k + L.adj; w = (w - 1) * L.stride - 2 * L.pad + L.k + L.adj; }
            else if (L.type == L_CONV) { h = (h + 2 * L.pad - L.k) / L.stride + 1; w = (w + 2 * L.pad - L.k) / L.stride + 1; }
            else if (L.type == L_UP) { h *= L.scale; w *= L.scale; }
            else if (L.type == L_RES) walk(L.block);
        }
    };
    walk(layers);
    *Ho = h; *Wo = w;
}

// The persistent / stream-K convolution grids assume that the forwards of ONE process on a device do not overlap (fav.h, concurrency
// note).  Forwards enqueued on different HIP streams -- two networks, or one network driven from two streams -- used to be a
// documented foot-gun; now the library orders them itself: when a forward arrives on another stream than the previous forward on
// that device, the HOST first waits for the previous stream to drain (hipStreamSynchronize: rare, a misuse made safe).  No marker
// is left in the compute queue in the common single-stream case: on this runtime an event record behind long-running kernels keeps
// a runtime thread spinning until it fires (scripts/runtime_thread_bench.hip).  Other PROCESSES on the device remain the caller's
// business (fav_net_set_shared_device).
namespace {
struct DeviceOrder { std::mutex mu; hipStream_t last = nullptr; bool valid = false; };
DeviceOrder& device_order(int device)
{
    static DeviceOrder order[64];
    return order[(device >= 0 && device < 64) ? device : 0];
}
}  // namespace

// the owner of `stream` is about to destroy it: drain it while the handle is still valid and forget it, so that the next forward
// never synchronises a dead (or recycled) handle
static void forget_stream(int device, hipStream_t stream)
{
    DeviceOrder& ord = device_order(device);
    std::lock_guard<std::mutex> order_lock(ord.mu);
    if (ord.valid && ord.last == stream) { (void)hipStreamSynchronize(stream); ord.valid = false; ord.last = nullptr; }
}

extern "C" int fav_net_forget_stream(fav_net* net, fav_hipstream_t stream)
{
    FAV_REQUIRE(net, "fav_net_forget_stream: null network");
    FAV_HIP(hipSetDevice(net->device));
    forget_stream(net->device, static_cast<hipStream_t>(stream));
    return FAV_OK;
}

int fav_net::forward_padded(const float* in8, int H, int W, float* out_planar, float* out_raw, hipStream_t stream)
{
    FAV_HIP(hipSetDevice(device));
    DeviceOrder& ord = device_order(device);
    std::lock_guard<std::mutex> order_lock(ord.mu);       // (handles are not thread-safe; this only keeps the bookkeeping consistent)
    if (ord.valid && ord.last != stream && hipStreamSynchronize(ord.last) != hipSuccess)
        (void)hipGetLastError();       // (a handle destroyed without fav_net_forget_stream / fav_stream_destroy first: fav.h asks for either)
    ord.valid = true; ord.last = stream;
    return forward_padded_unordered(in8, H, W, out_planar, out_raw, stream);
}

int fav_net::forward_padded_unordered(const float* in8, int H, int W, float* out_planar, float* out_raw, hipStream_t stream)
{
    if (sk_err_host && *reinterpret_cast<volatile unsigned*>(sk_err_host)) {      // reported by an earlier launch of this net
        *sk_err_host = 0;
        shared_device = true;
        set_error("a stream-K hand-off between convolution blocks timed out in an earlier launch of this network: its result was "
                  "wrong (two networks running concurrently on one device? see the concurrency note in fav.h)");
        return FAV_EHIP;
    }
    if (H != curH || W != curW) {
        if (!bufs.empty()) {
            FAV_HIP(hipDeviceSynchronize());
            for (void* sp : slabs) (void)hipFree(sp);
            slabs.clear(); slab_cur = nullptr; slab_left = 0;
            bufs.clear();
        }
        curH = H; curW = W;
    }
    st = stream; cursor = 0; conv_cursor = 0; in_cursor = 0;
    // InstanceNorm accumulators (Affine::acc1): a forward adds to one half and its consumers zero the other for the next one.  The halves
    // alternate over the forwards that USE them (fp32 mode; which layers do depends on the network and the process-wide switches only), so a
    // half is zero whenever it is added to -- unless a forward stopped between a producer and its consumer: then everything is zeroed here
    if (acc_dirty) {
        for (DevIN& i : ins) if (i.acc) FAV_HIP(hipMemsetAsync(i.acc, 0, i.acc_bytes, stream));
        acc_dirty = false;
    }
    if (precision == 0 && !tuning().no_acc_stats && !tuning().no_wino) acc_parity ^= 1;
    Act cur;
    cur.data = const_cast<float*>(in8); cur.Hp = H + 2 * pad; cur.Wp = W + 2 * pad; cur.C = 8;
    const int rc = run(exec, cur, true, out_planar, out_raw);
    if (rc) acc_dirty = true;
    return rc;
}

// ================================================================================================
// C ABI: network
// ================================================================================================
static int finish_create(fav_net* net, fav_net** out)
{
    int rc = net->upload();
    if (rc) { delete net; return rc; }
    *out = net;
    return FAV_OK;
}

extern "C" int fav_net_create(const char* t7_path_host, int device, fav_net** out)
{
    FAV_REQUIRE(t7_path_host && out, "fav_net_create: null argument");
    int rc = ensure_device(); if (rc) return rc;
    FAV_ABI_TRY
    std::unique_ptr<fav_net> net(new fav_net());
    net->device = device;
    rc = t7_parse_model(t7_path_host, net->layers);
    if (rc) return rc;
    return finish_create(net.release(), out);
    FAV_ABI_CATCH("fav_net_create")
}

extern "C" int fav_net_pack_host(const char* t7_path_host, void* blob_host, size_t capacity, size_t* bytes)
{
    FAV_REQUIRE(t7_path_host && bytes, "fav_net_pack_host: null argument");
    FAV_ABI_TRY
    std::vector<Layer> layers;
    int rc = t7_parse_model(t7_path_host, layers); if (rc) return rc;
    std::vector<uint8_t> blob;
    rc = blob_pack(layers, blob); if (rc) return rc;
    *bytes = blob.size();
    if (blob_host) {
        FAV_REQUIRE(capacity >= blob.size(), "fav_net_pack_host: capacity %zu < %zu", capacity, blob.size());
        memcpy(blob_host, blob.data(), blob.size());
    }
    return FAV_OK;
    FAV_ABI_CATCH("fav_net_pack_host")
}

extern "C" int fav_net_create_from_blob(const void* blob_host, size_t bytes, int device, fav_net** out)
{
    FAV_REQUIRE(blob_host && out, "fav_net_create_from_blob: null argument");
    int rc = ensure_device(); if (rc) return rc;
    FAV_ABI_TRY
    std::unique_ptr<fav_net> net(new fav_net());
    net->device = device;
    rc = blob_unpack(blob_host, bytes, net->layers);
    if (rc) return rc;
    return finish_create(net.release(), out);
    FAV_ABI_CATCH("fav_net_create_from_blob")
}

extern "C" void fav_net_destroy(fav_net* net) { delete net; }

extern "C" int fav_net_check(fav_net* net)
{
    FAV_REQUIRE(net, "fav_net_check: null net");
    if (net->sk_err_host && *reinterpret_cast<volatile unsigned*>(net->sk_err_host)) {
        *net->sk_err_host = 0;
        net->shared_device = true;      // from now on: data-parallel grids, which need no co-resident blocks
        set_error("a stream-K hand-off between convolution blocks timed out: the frame(s) computed since the last check are wrong "
                  "(another context holding compute units of this device? see the concurrency note in fav.h); this network now runs "
                  "with data-parallel grids (fav_net_set_shared_device)");
        return FAV_EHIP;
    }
    return FAV_OK;
}

extern "C" int fav_net_set_shared_device(fav_net* net, int shared)
{
    FAV_REQUIRE(net, "fav_net_set_shared_device: null net");
    net->shared_device = shared != 0;
    return FAV_OK;
}

extern "C" int fav_net_describe_host(const fav_net* net, char* buf_host, size_t capacity)
{
    FAV_REQUIRE(net && buf_host && capacity > 0, "fav_net_describe_host: null argument");
    const std::string s = describe_layers(net->layers);
    FAV_REQUIRE(s.size() + 1 <= capacity, "fav_net_describe_host: need %zu bytes", s.size() + 1);
    memcpy(buf_host, s.c_str(), s.size() + 1);
    return FAV_OK;
}

extern "C" int fav_net_profile_enable(fav_net* net, int on)
{
    FAV_REQUIRE(net, "fav_net_profile_enable: null net");
    net->profiling = on != 0;
    return FAV_OK;
}

extern "C" int fav_net_profile_read_host(fav_net* net, int capacity, int* count, double* ms_sum, int* launches,
                                         double* macs_per_launch, int* ntile)
{
    FAV_REQUIRE(net && count && ms_sum && launches && macs_per_launch && ntile, "fav_net_profile_read_host: null argument");
    FAV_HIP(hipSetDevice(net->device));
    for (auto& r : net->prof_pending) {
        FAV_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        FAV_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        net->prof_ms[r.conv] += ms; net->prof_n[r.conv] += 1;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    net->prof_pending.clear();
    const int n = (int)net->prof_ms.size();
    FAV_REQUIRE(n <= capacity, "fav_net_profile_read_host: need capacity %d", n);
    for (int i = 0; i < n; ++i) { ms_sum[i] = net->prof_ms[i]; launches[i] = net->prof_n[i]; macs_per_launch[i] = net->prof_macs[i]; ntile[i] = net->prof_tile[i]; }
    *count = n;
    std::fill(net->prof_ms.begin(), net->prof_ms.end(), 0.0); std::fill(net->prof_n.begin(), net->prof_n.end(), 0);
    return FAV_OK;
}

extern "C" int fav_t7_describe_host(const char* t7_path_host, char* buf_host, size_t capacity)
{
    FAV_REQUIRE(t7_path_host && buf_host && capacity > 0, "fav_t7_describe_host: null argument");
    FAV_ABI_TRY
    std::vector<Layer> layers;
    int rc = t7_parse_model(t7_path_host, layers); if (rc) return rc;
    const std::string s = describe_layers(layers);
    FAV_REQUIRE(s.size() + 1 <= capacity, "fav_t7_describe_host: need %zu bytes", s.size() + 1);
    memcpy(buf_host, s.c_str(), s.size() + 1);
    return FAV_OK;
    FAV_ABI_CATCH("fav_t7_describe_host")
}

extern "C" long long fav_net_param_count(const fav_net* net) { return net ? net->params : 0; }

extern "C" int fav_net_output_size(const fav_net* net, int H, int W, int* Ho, int* Wo)
{
    FAV_REQUIRE(net && Ho && Wo && H > 0 && W > 0, "fav_net_output_size: bad argument");
    net->out_size(H, W, Ho, Wo);
    return FAV_OK;
}

extern "C" int fav_net_forward(fav_net* net, const float* in7, float* out3, int H, int W, fav_hipstream_t stream)
{
    FAV_REQUIRE(net && in7 && out3 && H > 0 && W > 0, "fav_net_forward: bad argument");
    FAV_REQUIRE(net->pad < H && net->pad < W, "fav_net_forward: %dx%d is smaller than the reflection padding %d", W, H, net->pad);
    FAV_HIP(hipSetDevice(net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // boundary conversion: NCHW [7][H][W] -> reflection-padded NHWC8 (nn.SpatialReflectionPadding folded in)
    const size_t bytes = (size_t)(H + 2 * net->pad) * (W + 2 * net->pad) * 8 * sizeof(float);
    if (net->stage_bytes < bytes) {
        FAV_HIP(hipDeviceSynchronize());
        (void)hipFree(net->stage); net->stage = nullptr; net->stage_bytes = 0;
        FAV_HIP(hipMalloc(reinterpret_cast<void**>(&net->stage), bytes));
        net->stage_bytes = bytes;
    }
    float* in8 = net->stage;
    int rc = launch_nchw_to_nhwc_pad(in7, net->in_channels, H, W, net->pad, 8, in8, st); if (rc) return rc;
    return net->forward_padded(in8, H, W, nullptr, out3, st);
}

// nn.SpatialConvolution [+ nn.InstanceNormalization [+ nn.ReLU]] as a stand-alone operator (tests / ops)
extern "C" int fav_conv2d_nchw_f32(const float* in, int Cin, int H, int W, const float* weight, const float* bias, int Cout,
                                   int k, int stride, int pad, const float* gamma, const float* beta, float eps, int relu,
                                   float* out, fav_hipstream_t stream)
{
    FAV_REQUIRE(in && weight && out && Cin > 0 && Cout > 0 && k > 0 && stride > 0 && pad >= 0, "fav_conv2d_nchw_f32: bad argument");
    int rc = ensure_device(); if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cinp = (Cin + 3) / 4 * 4, coutp = (Cout + 31) / 32 * 32, kpad = (k * k * cinp + 31) / 32 * 32;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    FAV_REQUIRE(OH > 0 && OW > 0, "fav_conv2d_nchw_f32: empty output");
    const int M = OH * OW, mb = conv_mblocks(OH, OW);
    Layer L; L.cin = Cin; L.cout = Cout; L.k = k;
    L.w.resize((size_t)Cin * Cout * k * k);
    FAV_HIP(hipMemcpy(L.w.data(), weight, L.w.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float> wre, hb((size_t)coutp, 0.f);
    repack_weights(L, cinp, coutp, kpad, wre);
    if (bias) FAV_HIP(hipMemcpy(hb.data(), bias, (size_t)Cout * sizeof(float), hipMemcpyDeviceToHost));
    float *dw = nullptr, *db = nullptr, *din = nullptr, *dout = nullptr, *dpart = nullptr, *dsc = nullptr, *dsh = nullptr;
    auto cleanup = [&]() { (void)hipFree(dw); (void)hipFree(db); (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dpart); (void)hipFree(dsc); (void)hipFree(dsh); };
    rc = dev_upload(wre, 0, &dw); if (rc) { cleanup(); return rc; }
    rc = dev_upload(hb, 0, &db); if (rc) { cleanup(); return rc; }
    if (hipMalloc(reinterpret_cast<void**>(&din), (size_t)H * W * cinp * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dout), (size_t)M * Cout * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dpart), (size_t)mb * coutp * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dsc), (size_t)coutp * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dsh), (size_t)coutp * 4) != hipSuccess) { cleanup(); return hip_fail(hipErrorOutOfMemory, "hipMalloc"); }
    rc = launch_nchw_to_nhwc_pad(in, Cin, H, W, 0, cinp, din, st);
    ConvLaunch c;
    c.in = din; c.IH = H; c.IW = W; c.IWp = W; c.CIN = cinp; c.wgt = dw; c.bias = db; c.COUT = Cout; c.COUTp = coutp;
    c.KH = c.KW = k; c.stride = stride; c.pad = pad; c.Kpad = kpad; c.OH = OH; c.OW = OW; c.out = dout;
    c.partials = gamma ? dpart : nullptr;
    if (!rc) rc = launch_conv(c, st);
    Affine t;
    if (!rc && gamma) {
        rc = launch_in_finalize(dpart, nullptr, mb, M, CONV_BM, Cout, coutp, gamma, beta, eps, dsc, dsh, st);
        t.scale1 = dsc; t.shift1 = dsh; t.relu1 = relu; t.stages = 1;
    }
    if (!rc) rc = launch_nhwc_to_nchw(dout, M, Cout, t, out, st);
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = hip_fail(hipGetLastError(), "fav_conv2d_nchw_f32");
    cleanup();
    return rc;
}

// ================================================================================================
// C ABI: fused per-frame pipeline
// ================================================================================================
struct fav_stream {
    fav_net* net = nullptr;
    fav_net* img_net = nullptr;  // optional -model_img: stylises frames that have no prior (core.lua:59-66,146)
    int H = 0, W = 0;            // frame (= flow, certainty, network input) size
    int Ho = 0, Wo = 0;          // network output size: H x W when both are multiples of 4, up to 3 px more otherwise (two stride-2
                                 // convolutions, two x2 upsamplings).  The reference keeps and saves the LARGER image and warps it
                                 // with the flow's size (BilinearSamplerBDHW.lua:71: the output takes the grid's size), so does this
    fav_stream_opts opts{};
    float* state = nullptr;      // last_frame_stylized: [3][Ho][Wo] float RGB, unclamped (fav.lua:169)
    bool has_state = false;
    unsigned frame_counter = 0;  // 1-based index of the frame being stylised (key of the uniform-random fill)
    float* in8 = nullptr;        // padded NHWC8 network input
    float* cert_tmp = nullptr; float* cert = nullptr;
    uint8_t* mask = nullptr;     // certainty as the checker writes it (u8 {0,255})
    int* q0_main = nullptr;      // the XCD the caller's queue deals block 0 of a launch to (written by prep_input_kernel, read by the look-ahead mask's long-lived kernels)
    void* ws = nullptr; size_t ws_bytes = 0;
    void* png_ws = nullptr; size_t png_ws_bytes = 0;      // workspace of fav_stream_encode_png (allocated on first use)
    // fav_stream_encode_png_async: the encoder's kernels run on a queue of their own, next to the NEXT frame's network (they fill the
    // tails of its grids instead of standing in front of it).  The state is double-buffered from then on: frame i + 1 is written into
    // the other buffer while frame i's is being encoded, and the frame that comes back to a buffer waits for that buffer's encoder
    hipStream_t png_q = nullptr; hipEvent_t ev_png_in = nullptr;
    float* state_other = nullptr;                         // the buffer `state` is not (null until the first asynchronous encode)
    hipEvent_t png_done = nullptr, png_done_other = nullptr;      // the last encode that read `state` / `state_other` ...
    bool png_pending = false, png_pending_other = false;          // ... if nothing has waited for it yet
    // look-ahead mask (fav_stream_prefetch_mask)
    // two side queues with their own structure workspaces: the masks of frames i+1 and i+2 are computed concurrently
    // (each 4-argument mask contains a ~3 ms sequential fp32 chain, CMatrix::avg), three look-ahead slots
    static constexpr int NSIDE = 2, NPREF = 4;
    hipStream_t side[NSIDE] = {nullptr, nullptr}; void* side_ws[NSIDE] = {nullptr, nullptr}; hipEvent_t ev_in = nullptr;
    float* side_cert_tmp[NSIDE] = {nullptr, nullptr};      // scratch of the certainty preparation (erosion input) on each side queue
    struct Pref { uint8_t* mask = nullptr; float* cert = nullptr; hipEvent_t done = nullptr; bool valid = false;
                  const void *frame = nullptr, *bw = nullptr, *fw = nullptr; int structure = 0;
                  uint32_t retired = 0; };       // host-ordered: retire sequence of the (mask, cert) buffers now in this slot (0: never read)
    Pref pref[NPREF]; int pref_next = 0, side_next = 0;
    // host-ordered look-ahead (fav_stream_set_host_ordered): no event ever enters the caller's queue or the side queues; a one-thread
    // kernel behind the mask pipeline stores a sequence number into host-mapped memory, which fav_stream_next_frame_flow polls
    bool host_ordered = false;
    uint32_t* done_host = nullptr;      // [NPREF], hipHostMalloc
    uint32_t pref_seq[NPREF] = {0, 0, 0, 0}; uint32_t seq_counter = 0;
    // ... and nothing orders a side queue behind the caller's queue either, so the buffers a consumed look-ahead swaps OUT of the stream
    // (read by the previous frames' kernels on the caller's queue) must not be rewritten by a later look-ahead before those kernels are
    // through: the consuming call first enqueues a one-thread kernel on the caller's queue that stores a retire sequence number into
    // host-mapped memory (everything enqueued before it has then finished), and a look-ahead into a slot waits ON THE HOST for the
    // sequence number of the buffers it holds (with NPREF slots and two frames of look-ahead that frame finished long ago: no wait)
    uint32_t* retired_host = nullptr; uint32_t retire_counter = 0;
    hipStream_t last_st = nullptr; bool ran = false;       // the HIP stream of the last forward (forgotten by the destructor)
    ~fav_stream()
    {
        if (net) (void)hipSetDevice(net->device);
        if (net && ran) forget_stream(net->device, last_st);
        if (png_q) { (void)hipStreamSynchronize(png_q); (void)hipStreamDestroy(png_q); }
        if (ev_png_in) (void)hipEventDestroy(ev_png_in);
        if (png_done) (void)hipEventDestroy(png_done);
        if (png_done_other) (void)hipEventDestroy(png_done_other);
        (void)hipFree(state_other);
        for (int i = 0; i < NSIDE; ++i) { if (side[i]) { (void)hipStreamSynchronize(side[i]); (void)hipStreamDestroy(side[i]); } (void)hipFree(side_ws[i]); }
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (done_host) (void)hipHostFree(done_host);       // (retired_host lives in the same allocation)
        for (auto& pf : pref) { if (pf.done) (void)hipEventDestroy(pf.done); (void)hipFree(pf.mask); (void)hipFree(pf.cert); }
        for (int i = 0; i < NSIDE; ++i) (void)hipFree(side_cert_tmp[i]);
        (void)hipFree(q0_main); (void)hipFree(state); (void)hipFree(in8); (void)hipFree(cert_tmp); (void)hipFree(cert); (void)hipFree(mask); (void)hipFree(ws); (void)hipFree(png_ws);
    }
};

extern "C" int fav_net_set_precision(fav_net* net, int mode)
{
    FAV_REQUIRE(net && (mode == 0 || mode == 1), "fav_net_set_precision: mode must be FAV_PRECISION_FP32 or FAV_PRECISION_BF16_OPERANDS");
    if (net->precision != mode) { net->curH = -1; net->acc_dirty = true; }      // the two modes tile (and size the statistics buffers) differently: rebuild the arena
    net->precision = mode;
    return FAV_OK;
}

// internal accessors for the other host units (vr.cpp)
namespace fav {
int net_device(const fav_net* n) { return n->device; }
int net_pad(const fav_net* n) { return n->pad; }
int net_in_channels(const fav_net* n) { return n->in_channels; }
void net_out_size(const fav_net* n, int H, int W, int* Ho, int* Wo) { n->out_size(H, W, Ho, Wo); }
int net_forward_padded(fav_net* n, const float* in8, int H, int W, float* out_planar, hipStream_t st) { return n->forward_padded(in8, H, W, out_planar, nullptr, st); }
}  // namespace fav

// While look-ahead masks are in flight the persistent / stream-K convolution grids leave SIDE_CUS CUs unclaimed
// (fav_net::reserve_cus): the side queues' kernels (among them a ~3 ms single-wave sequential chain) find free CUs, and a
// statically scheduled network block is never kept off the chip by them.
static const int SIDE_CUS = getenv("FAV_SIDE_CUS") ? std::max(0, atoi(getenv("FAV_SIDE_CUS"))) : 4;      // (tuning: read once; 8 until round 4, when a
                                                                                                          //  mask was 1.5 ms of mostly sequential kernels: profiles/r4b_4arg_*)
// one side queue carries every look-ahead since round 4 (a mask is 0.6 ms of short kernels, two in flight fit a 1.8 ms frame back to
// back; two queues measured 541-543 frames/s against 546-548: profiles/r04c_4arg_knobs_ab.log)
static const int NSIDE_USED = getenv("FAV_SIDE_QUEUES") ? std::max(1, std::min(2, atoi(getenv("FAV_SIDE_QUEUES")))) : 1;
// the look-ahead mask's long-lived kernels (the recursive-filter passes) are packed onto the reserved CUs (launch_structure's pack_cus;
// FAV_SIDE_PACK=0: one block per wave, the form of rounds 4-5)
static const int SIDE_PACK = diag_env("FAV_SIDE_PACK") ? std::max(0, atoi(diag_env("FAV_SIDE_PACK"))) : -1;
static hipError_t create_side_stream(hipStream_t* st)
{
    // (confining the side queues with a CU mask -- hipExtStreamCreateWithCUMask -- measured slower than leaving the CUs free, rounds 2-3)
    // FAV_SIDE_CU_MASK=<comma-separated bit numbers>: experiment, round 6
    if (const char* m = diag_env("FAV_SIDE_CU_MASK")) {
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const char* p = m; *p;) { const int b = atoi(p); if (b >= 0 && b < 256) words[b / 32] |= 1u << (b % 32); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        return hipExtStreamCreateWithCUMask(st, 8, words);
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

extern "C" int fav_stream_create(fav_net* net, int H, int W, const fav_stream_opts* o, fav_stream** out)
{
    FAV_REQUIRE(net && out && H > 0 && W > 0, "fav_stream_create: bad argument");
    FAV_REQUIRE(net->pad < H && net->pad < W, "fav_stream_create: %dx%d is smaller than the reflection padding %d", W, H, net->pad);
    int Ho, Wo; net->out_size(H, W, &Ho, &Wo);
    FAV_REQUIRE(Ho >= 1 && Wo >= 1, "frame size %dx%d is too small for the architecture", W, H);
    FAV_HIP(hipSetDevice(net->device));
    fav_stream* s = new fav_stream();
    s->net = net; s->H = H; s->W = W; s->Ho = Ho; s->Wo = Wo;
    if (o) s->opts = *o; else { s->opts.border_mode = FAV_BORDER_STN; s->opts.occlusions_min_filter = 7; s->opts.invert_occlusion = 0; s->opts.fix_occlusions = 0; s->opts.fill_random = 0; s->opts.seed = 0; }
    if (s->opts.occlusions_min_filter < 1) s->opts.occlusions_min_filter = 1;
    const size_t n = (size_t)H * W;
    s->ws_bytes = structure_workspace_bytes(W, H);
    if (hipMalloc(reinterpret_cast<void**>(&s->state), (size_t)3 * Ho * Wo * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->in8), (size_t)(H + 2 * net->pad) * (W + 2 * net->pad) * 32) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->cert_tmp), n * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->cert), n * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->mask), n) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->q0_main), 64) != hipSuccess || hipMemset(s->q0_main, 0, 64) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipMalloc(&s->ws, s->ws_bytes) != hipSuccess) { delete s; return hip_fail(hipErrorOutOfMemory, "hipMalloc(stream buffers)"); }
    for (auto& pf : s->pref)
        if (hipMalloc(reinterpret_cast<void**>(&pf.mask), n) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&pf.cert), n * 4) != hipSuccess || hipEventCreateWithFlags(&pf.done, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
            delete s; return hip_fail(hipErrorOutOfMemory, "look-ahead slots"); }
    for (int i = 0; i < fav_stream::NSIDE; ++i)
        if (create_side_stream(&s->side[i]) != hipSuccess || hipMalloc(&s->side_ws[i], s->ws_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->side_cert_tmp[i]), n * 4) != hipSuccess) {
            delete s; return hip_fail(hipErrorOutOfMemory, "side queues"); }
    *out = s;
    return FAV_OK;
}

extern "C" void fav_stream_destroy(fav_stream* s) { delete s; }

// the buffer the next frame is written into becomes `state`.  Synchronous encodes only: the one buffer, in place (the frame's own
// input was assembled from it before the network starts).  With an asynchronous encode possibly reading `state`: the other buffer,
// behind the encode that read THAT one two frames ago (long finished)
static int state_for_writing(fav_stream* s, hipStream_t st)
{
    if (!s->state_other) return FAV_OK;
    std::swap(s->state, s->state_other);
    std::swap(s->png_done, s->png_done_other);
    std::swap(s->png_pending, s->png_pending_other);
    if (s->png_pending) { FAV_HIP(hipStreamWaitEvent(st, s->png_done, 0)); s->png_pending = false; }
    return FAV_OK;
}

// the forward that was to fill the buffer state_for_writing() switched to has failed: `state` names the last COMPLETE frame again
static void state_writing_failed(fav_stream* s)
{
    if (!s->state_other) return;
    std::swap(s->state, s->state_other);
    std::swap(s->png_done, s->png_done_other);
    std::swap(s->png_pending, s->png_pending_other);
}

static int stream_finish(fav_stream* s, float* out_rgb_f32, uint8_t* out_rgb8_hwc, hipStream_t st)
{
    const size_t n = (size_t)s->Ho * s->Wo;
    s->has_state = true;
    if (out_rgb_f32) FAV_HIP(hipMemcpyAsync(out_rgb_f32, s->state, 3 * n * 4, hipMemcpyDeviceToDevice, st));
    if (out_rgb8_hwc) return launch_quantize_rgb8(s->state, out_rgb8_hwc, s->Ho, s->Wo, st);
    return FAV_OK;
}

extern "C" int fav_stream_first_frame(fav_stream* s, const uint8_t* frame_rgb_hwc, float* out_rgb_f32, uint8_t* out_rgb8_hwc,
                                      fav_hipstream_t stream)
{
    FAV_REQUIRE(s && frame_rgb_hwc, "fav_stream_first_frame: null argument");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    ++s->frame_counter;
    // the image model sees only the three content channels (core.lua:146): no fill there
    int rc = launch_prep_input(frame_rgb_hwc, nullptr, 0, 0, nullptr, nullptr, s->opts.border_mode, s->H, s->W, s->net->pad, s->in8, st,
                               s->img_net ? 0 : s->opts.fill_random, s->opts.seed, s->frame_counter);
    if (rc) return rc;
    s->last_st = st; s->ran = true;
    fav_net* fn = s->img_net ? s->img_net : s->net;      // image model: 3 content channels (the zero prior / mask planes meet zero weights)
    rc = state_for_writing(s, st); if (rc) return rc;
    rc = fn->forward_padded(s->in8, s->H, s->W, s->state, nullptr, st); if (rc) { state_writing_failed(s); return rc; }
    return stream_finish(s, out_rgb_f32, out_rgb8_hwc, st);
}

extern "C" int fav_stream_set_image_net(fav_stream* s, fav_net* image_net)
{
    FAV_REQUIRE(s, "fav_stream_set_image_net: null stream");
    if (image_net) {
        FAV_REQUIRE(image_net->device == s->net->device, "fav_stream_set_image_net: the image model lives on another device");
        FAV_REQUIRE(image_net->pad == s->net->pad, "fav_stream_set_image_net: image model pads %d px, video model %d px (both read the same padded input)", image_net->pad, s->net->pad);
        int Ho, Wo; image_net->out_size(s->H, s->W, &Ho, &Wo);
        FAV_REQUIRE(Ho == s->Ho && Wo == s->Wo, "fav_stream_set_image_net: the image model maps %dx%d to %dx%d, the video model to %dx%d", s->W, s->H, Wo, Ho, s->Wo, s->Ho);
    }
    s->img_net = image_net;
    return FAV_OK;
}

// cert_ready: the certainty (mask options + erosion applied) is already in s->cert (computed ahead on a side queue)
// input_ready: ... and the network input in s->in8 as well (fav_stream_next_frame_flow's fused check + assembly; frame_counter advanced)
static int stream_next(fav_stream* s, const uint8_t* frame, const float* bw, const uint8_t* mask, float* out_f32, uint8_t* out_u8,
                       hipStream_t st, bool cert_ready = false, bool input_ready = false)
{
    FAV_REQUIRE(s->has_state, "fav_stream_next_frame: no previous stylised frame (call fav_stream_first_frame or fav_stream_set_state first)");
    int rc = FAV_OK;
    if (!input_ready) {
        TraceRange tr_pre("fav:certainty+warp+assemble");
        if (!cert_ready)
            rc = launch_cert_prepare(mask, bw, s->opts.invert_occlusion, s->opts.fix_occlusions, s->opts.border_mode,
                                     s->opts.occlusions_min_filter, s->cert_tmp, s->cert, s->H, s->W, st);
        if (rc) return rc;
        ++s->frame_counter;
        rc = launch_prep_input(frame, s->state, s->Ho, s->Wo, bw, s->cert, s->opts.border_mode, s->H, s->W, s->net->pad, s->in8, st,
                               s->opts.fill_random, s->opts.seed, s->frame_counter, s->q0_main);
        if (rc) return rc;
    }
    s->last_st = st; s->ran = true;
    rc = state_for_writing(s, st); if (rc) return rc;       // (the prior was read from the previous state above)
    { TraceRange tr_net("fav:network"); rc = s->net->forward_padded(s->in8, s->H, s->W, s->state, nullptr, st); }
    if (rc) { state_writing_failed(s); return rc; }
    return stream_finish(s, out_f32, out_u8, st);
}

extern "C" int fav_stream_next_frame_cert(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                                          const uint8_t* cert_pgm, float* out_rgb_f32, uint8_t* out_rgb8_hwc,
                                          fav_hipstream_t stream)
{
    FAV_REQUIRE(s && frame_rgb_hwc && backward_flo && cert_pgm, "fav_stream_next_frame_cert: null argument");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    FAV_HIP(hipMemcpyAsync(s->mask, cert_pgm, (size_t)s->H * s->W, hipMemcpyDeviceToDevice, st));
    return stream_next(s, frame_rgb_hwc, backward_flo, s->mask, out_rgb_f32, out_rgb8_hwc, st);
}

extern "C" int fav_stream_next_frame_flow(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                                          const float* forward_flo, int use_structure, float* out_rgb_f32,
                                          uint8_t* out_rgb8_hwc, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && frame_rgb_hwc && backward_flo && forward_flo, "fav_stream_next_frame_flow: null argument");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // makeOptFlow_deepflow.sh:59: consistencyChecker backward_i_j.flo forward_j_i.flo reliable_i_j.pgm [frame_i.ppm]
    for (auto& pf : s->pref)
        if (pf.valid && pf.frame == frame_rgb_hwc && pf.bw == backward_flo && pf.fw == forward_flo && pf.structure == (use_structure != 0)) {
            // the mask was computed ahead of time on the side stream
            pf.valid = false;
            if (s->host_ordered) {
                // the mask pipeline of this slot ends in a store of its sequence number to host-mapped memory: wait for it HERE, on the
                // host (it was started a frame ago: normally no wait at all), then enqueue -- no dependency between the queues
                const int slot = (int)(&pf - s->pref);
                volatile uint32_t* flag = &s->done_host[slot];
                for (int spins = 0; *flag != s->pref_seq[slot]; ++spins) {
                    usleep(50);
                    if ((spins & 2047) == 2047) { const hipError_t e = hipStreamQuery(s->side[0]); if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "look-ahead queue"); }
                }
                // the buffers about to leave the stream were read by kernels already in the caller's queue: mark the point behind them
                pf.retired = ++s->retire_counter;
                int rcf = launch_store_flag(s->retired_host, pf.retired, st); if (rcf) return rcf;
            } else FAV_HIP(hipStreamWaitEvent(st, pf.done, 0));
            std::swap(s->mask, pf.mask);
            std::swap(s->cert, pf.cert);          // mask -> certainty (options, erosion) was done on the side queue as well
            return stream_next(s, frame_rgb_hwc, backward_flo, s->mask, out_rgb_f32, out_rgb8_hwc, st, true);
        }
    // not prefetched: compute inline on the caller's stream (own workspace)
    TraceRange tr_mask("fav:consistency mask");
    const float* structure = nullptr; const float* avg = nullptr;
    if (use_structure) {
        int rc = launch_structure(frame_rgb_hwc, s->W, s->H, s->ws, s->ws_bytes, &structure, &avg, st); if (rc) return rc;
    }
    // check + certainty options + erosion + input assembly in ONE tile kernel (round 5; the mask byte and the eroded certainty of every pixel
    // are still written: fav_stream_last_mask); FAV_NO_CHECK_PREP: the check and the assembly as two launches (rounds 3-4)
    static const bool fused_prep = diag_env("FAV_NO_CHECK_PREP") == nullptr;      // (tuning: read once)
    if (fused_prep && s->has_state) {
        ++s->frame_counter;
        int rcp = launch_check_prep(frame_rgb_hwc, s->state, s->Ho, s->Wo, backward_flo, forward_flo, structure, avg, s->mask, s->cert,
                                    s->opts.invert_occlusion, s->opts.fix_occlusions, s->opts.border_mode, s->opts.occlusions_min_filter,
                                    s->H, s->W, s->net->pad, s->in8, st, s->opts.fill_random, s->opts.seed, s->frame_counter);
        if (rcp) return rcp;
        return stream_next(s, frame_rgb_hwc, backward_flo, s->mask, out_rgb_f32, out_rgb8_hwc, st, true, true);
    }
    int rc = launch_check_cert(backward_flo, forward_flo, structure, avg, s->mask, s->opts.invert_occlusion, s->opts.fix_occlusions, s->opts.border_mode,
                               s->opts.occlusions_min_filter, s->cert, s->H, s->W, st);
    if (rc) return rc;
    return stream_next(s, frame_rgb_hwc, backward_flo, s->mask, out_rgb_f32, out_rgb8_hwc, st, true);
}

extern "C" int fav_stream_prefetch_mask(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                                        const float* forward_flo, int use_structure, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && frame_rgb_hwc && backward_flo && forward_flo, "fav_stream_prefetch_mask: null argument");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the 4-argument mask holds long sequential chains: from now on the network's persistent grids leave the side queues their CUs;
    // the 3-argument mask + certainty (25 us of small kernels) fits into the grids' own tails
    if (use_structure) s->net->reserve_cus = SIDE_CUS;
    fav_stream::Pref& pf = s->pref[s->pref_next];
    s->pref_next = (s->pref_next + 1) % fav_stream::NPREF;
    const int q = s->side_next; s->side_next = (s->side_next + 1) % NSIDE_USED;
    hipStream_t sd = s->side[q];
    if (!s->host_ordered) {
        FAV_HIP(hipEventRecord(s->ev_in, st));                 // inputs are complete at this point of the caller's stream
        FAV_HIP(hipStreamWaitEvent(sd, s->ev_in, 0));          // (work enqueued on `stream` AFTER this call is not waited for)
    } else if (pf.retired) {                                   // host-ordered: the caller has SEEN the inputs complete (fav.h) ...
        // ... and the slot's buffers were read by frames on the caller's queue: those must be through before a side queue rewrites them
        volatile uint32_t* flag = s->retired_host;
        for (int spins = 0; (int32_t)(*flag - pf.retired) < 0; ++spins) {
            usleep(50);
            if ((spins & 2047) == 2047) { const hipError_t e = hipStreamQuery(st); if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "look-ahead: the caller's queue"); }
        }
        pf.retired = 0;
    }
    const float* structure = nullptr; const float* avg = nullptr;
    if (use_structure) {
        int rc = launch_structure(frame_rgb_hwc, s->W, s->H, s->side_ws[q], s->ws_bytes, &structure, &avg, sd, SIDE_PACK >= 0 ? SIDE_PACK : SIDE_CUS, s->q0_main); if (rc) return rc;
    }
    // mask + certainty of the frame (mask options, fix_occlusions warp of ones, erosion): depends on the flows and the stream's options only
    int rc = launch_check_cert(backward_flo, forward_flo, structure, avg, pf.mask, s->opts.invert_occlusion, s->opts.fix_occlusions, s->opts.border_mode,
                               s->opts.occlusions_min_filter, pf.cert, s->H, s->W, sd);
    if (rc) return rc;
    if (s->host_ordered) {
        const int slot = (int)(&pf - s->pref);
        s->pref_seq[slot] = ++s->seq_counter;
        rc = launch_store_flag(&s->done_host[slot], s->pref_seq[slot], sd); if (rc) return rc;
    } else FAV_HIP(hipEventRecord(pf.done, sd));
    pf.valid = true; pf.frame = frame_rgb_hwc; pf.bw = backward_flo; pf.fw = forward_flo; pf.structure = use_structure != 0;
    return FAV_OK;
}

extern "C" int fav_stream_get_state(fav_stream* s, float* state_rgb_f32, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && state_rgb_f32 && s->has_state, "fav_stream_get_state: no state");
    FAV_HIP(hipSetDevice(s->net->device));
    FAV_HIP(hipMemcpyAsync(state_rgb_f32, s->state, (size_t)3 * s->Ho * s->Wo * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return FAV_OK;
}

extern "C" int fav_stream_set_state(fav_stream* s, const float* state_rgb_f32, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && state_rgb_f32, "fav_stream_set_state: null argument");
    FAV_HIP(hipSetDevice(s->net->device));
    if (s->png_pending) { FAV_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), s->png_done, 0)); s->png_pending = false; }
    FAV_HIP(hipMemcpyAsync(s->state, state_rgb_f32, (size_t)3 * s->Ho * s->Wo * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    s->has_state = true;
    return FAV_OK;
}

extern "C" int fav_stream_wait_png(fav_stream* s, fav_hipstream_t stream)
{
    FAV_REQUIRE(s, "fav_stream_wait_png: null stream");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->png_pending) FAV_HIP(hipStreamWaitEvent(st, s->png_done, 0));
    if (s->png_pending_other) FAV_HIP(hipStreamWaitEvent(st, s->png_done_other, 0));
    return FAV_OK;
}

extern "C" int fav_stream_encode_png(fav_stream* s, void* png_out, size_t capacity, uint32_t* png_bytes_out, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && s->has_state, "fav_stream_encode_png: no stylised frame yet");
    FAV_HIP(hipSetDevice(s->net->device));
    if (!s->png_ws) {
        s->png_ws_bytes = png_workspace_bytes(s->Wo, s->Ho);
        FAV_HIP(hipMalloc(&s->png_ws, s->png_ws_bytes));
    }
    if (s->png_q) { int rc = fav_stream_wait_png(s, stream); if (rc) return rc; }      // (one workspace: behind the asynchronous encodes)
    return launch_png_encode(nullptr, s->state, s->Wo, s->Ho, png_out, capacity, png_bytes_out, s->png_ws, s->png_ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int fav_stream_encode_png_async(fav_stream* s, void* png_out, size_t capacity, uint32_t* png_bytes_out, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && s->has_state, "fav_stream_encode_png_async: no stylised frame yet");
    FAV_HIP(hipSetDevice(s->net->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!s->png_ws) {
        s->png_ws_bytes = png_workspace_bytes(s->Wo, s->Ho);
        FAV_HIP(hipMalloc(&s->png_ws, s->png_ws_bytes));
    }
    if (!s->png_q) {
        // all or nothing: a stream with the queue but without the second state buffer would let the next frame overwrite what is being encoded
        const unsigned ef = hipEventDisableTiming | hipEventDisableSystemFence;
        hipStream_t q = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr; float* other = nullptr;
        const hipError_t e = [&]() -> hipError_t {
            hipError_t r;
            if ((r = hipStreamCreateWithFlags(&q, hipStreamNonBlocking)) != hipSuccess) return r;
            if ((r = hipEventCreateWithFlags(&e0, ef)) != hipSuccess) return r;
            if ((r = hipEventCreateWithFlags(&e1, ef)) != hipSuccess) return r;
            if ((r = hipEventCreateWithFlags(&e2, ef)) != hipSuccess) return r;
            return hipMalloc(reinterpret_cast<void**>(&other), (size_t)3 * s->Ho * s->Wo * 4);
        }();
        if (e != hipSuccess) {
            if (q) (void)hipStreamDestroy(q);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (e2) (void)hipEventDestroy(e2);
            (void)hipFree(other);
            return hip_fail(e, "fav_stream_encode_png_async: encoder queue / second state buffer");
        }
        s->png_q = q; s->ev_png_in = e0; s->png_done = e1; s->png_done_other = e2; s->state_other = other;
        // from now on the encoder's kernels run NEXT TO the following frame's network, like the look-ahead masks do: the network's
        // persistent grids leave them SIDE_CUS CUs and the generic kernel's hand-off between co-resident blocks is off (timed_conv)
        s->net->reserve_cus = std::max(s->net->reserve_cus, SIDE_CUS);
        if (s->img_net) s->img_net->reserve_cus = std::max(s->img_net->reserve_cus, SIDE_CUS);
    }
    FAV_HIP(hipEventRecord(s->ev_png_in, st));                 // the frame is complete at this point of the caller's queue
    FAV_HIP(hipStreamWaitEvent(s->png_q, s->ev_png_in, 0));
    int rc = launch_png_encode(nullptr, s->state, s->Wo, s->Ho, png_out, capacity, png_bytes_out, s->png_ws, s->png_ws_bytes, s->png_q);
    if (rc) return rc;
    FAV_HIP(hipEventRecord(s->png_done, s->png_q));
    s->png_pending = true;
    return FAV_OK;
}

extern "C" int fav_stream_set_host_ordered(fav_stream* s, int on)
{
    FAV_REQUIRE(s, "fav_stream_set_host_ordered: null stream");
    FAV_HIP(hipSetDevice(s->net->device));
    if (on && !s->done_host) {
        FAV_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->done_host), 128, hipHostMallocDefault));
        for (int i = 0; i < 32; ++i) s->done_host[i] = 0u;
        s->retired_host = s->done_host + 16;               // its own 64-byte line
    }
    for (auto& pf : s->pref) { pf.valid = false; pf.retired = 0; }     // look-aheads in flight in the other mode are dropped (their queues drain on their own)
    s->host_ordered = on != 0;
    return FAV_OK;
}

extern "C" int fav_stream_output_size(const fav_stream* s, int* Ho, int* Wo)
{
    FAV_REQUIRE(s && Ho && Wo, "fav_stream_output_size: null argument");
    *Ho = s->Ho; *Wo = s->Wo;
    return FAV_OK;
}

extern "C" int fav_stream_get_input_f32(const fav_stream* s, float* in7, fav_hipstream_t stream)
{
    FAV_REQUIRE(s && in7 && s->frame_counter > 0, "fav_stream_get_input_f32: no frame has been assembled yet");
    FAV_HIP(hipSetDevice(s->net->device));
    return launch_unpad_input(s->in8, s->H, s->W, s->net->pad, in7, static_cast<hipStream_t>(stream));
}

extern "C" const uint8_t* fav_stream_last_mask(const fav_stream* s) { return s ? s->mask : nullptr; }
