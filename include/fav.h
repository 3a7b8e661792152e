/*
 * fav.h -- C ABI of libfav: the MI355X-native (gfx950, hand-written HIP) implementation of the
 * per-frame hot path of manuelruder/fast-artistic-videos:
 *
 *   frames + backward .flo (+ forward .flo | .pgm certainty) + .t7 weights
 *     -> warp previous stylised frame -> occlusion/consistency mask -> 7-channel assembly
 *     -> conv-InstanceNorm-ReLU residual transformer network -> deprocess -> stylised frame
 *
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * tree).  Conventions:
 *   - plain C, no exceptions / longjmp across the boundary; return 0 (FAV_OK) or a negative
 *     fav_status; fav_last_error() gives a thread-local message.
 *   - all data pointers are DEVICE pointers unless the parameter name ends in `_host`;
 *     the caller owns every buffer it passes.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns
 *     without synchronising (NULL = the default stream).
 *   - concurrency: a fav_net owns one activation arena and one stream-K workspace, so it executes ONE forward at a time --
 *     everything that shares a net (its fav_streams, a fav_vr) must be enqueued on the same HIP stream or be ordered by the
 *     caller.  The persistent / stream-K convolution kernels size their grids to the whole device (minus the CUs reserved for
 *     the library's own look-ahead queues) and hand partial tiles between co-resident blocks: two forwards must therefore not
 *     run CONCURRENTLY on one device.  Inside one process the library sees to that itself: when a forward is enqueued on another
 *     HIP stream than the previous forward on that device, the host first waits for that previous stream to drain (the ONE case in
 *     which an entry point synchronises), so several nets / streams are SERIALISED, never wrong.  Before a HIP stream that carried
 *     forwards is destroyed, call fav_stream_destroy on the fav_streams that ran on it or fav_net_forget_stream (either drains it
 *     and drops the library's note of the handle).  Across processes it cannot: one process per GPU (as
 *     bench.py and the launcher do), or fav_net_set_shared_device.  Handles are not locked: do not call into the same handle
 *     from two host threads at once.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     FAV_ENODEVICE.
 */
#ifndef FAV_H
#define FAV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fav_hipstream_t;

enum fav_status {
    FAV_OK = 0,
    FAV_EINVAL = -1,       /* bad argument (shape, null pointer, ...) */
    FAV_EIO = -2,          /* file could not be opened / read / written */
    FAV_EFORMAT = -3,      /* malformed .t7 / .flo / pnm */
    FAV_EUNSUPPORTED = -4, /* valid input, but outside what the kernels cover (e.g. asymmetric reflection padding) */
    FAV_EHIP = -5,         /* HIP runtime error */
    FAV_ENODEVICE = -6     /* no HIP device: the product path has no CPU fallback */
};

/* border policy of the warp: the reference has two (fast_artistic_video/utils.lua:141-149) */
enum fav_border {
    FAV_BORDER_STN = 0, /* GPU path: stnbdhw/BilinearSamplerBDHW.cu:48-109, each tap zeroed individually */
    FAV_BORDER_CPU = 1  /* CPU path: image.warp(img, flow, 'bilinear', true, 'pad', 0) */
};

const char* fav_last_error(void);
int fav_version(void);
/* number of HIP devices, or FAV_ENODEVICE */
int fav_device_count(void);

/* ---- A2: optical-flow warp -------------------------------------------------------------------
 * Replaces nn.BilinearSamplerBDHW():forward({img, flow}) (stnbdhw/BilinearSamplerBDHW.lua:54-82,
 * native cunn_BilinearSamplerBDHW_updateOutput, stnbdhw/BilinearSamplerBDHW.cu:111-152) and the
 * CPU branch of utils.warp_image (fast_artistic_video/utils.lua:147).
 * img [B][C][H][W], flow [B][2][Ho][Wo] with flow[:,0]=dy, flow[:,1]=dx (pixel offsets),
 * out [B][C][Ho][Wo]; all contiguous fp32. */
int fav_warp_bdhw_f32(const float* img, const float* flow, float* out,
                      int B, int C, int H, int W, int Ho, int Wo,
                      int border_mode, fav_hipstream_t stream);

/* ---- A3/A4: forward-backward consistency mask ------------------------------------------------
 * Replaces the process `consistencyChecker flow1.flo flow2.flo out.pgm [image.ppm]`
 * (consistencyChecker/consistencyChecker.cpp:136-171; checkConsistency :80-134, computeCorners
 * :39-78).  flow1_flo / flow2_flo are the .flo payloads: [H][W][2] interleaved (u,v) fp32.
 * rgb_hwc is the P6 payload [H][W][3] u8, or NULL for the 3-argument mode.  out is the PGM
 * payload: [H][W] u8 in {0,255}, bit-exact with the reference binary.
 * workspace: fav_consistency_workspace_bytes(W,H,with_structure) bytes of device memory (may be
 * NULL when that returns 0). */
size_t fav_consistency_workspace_bytes(int W, int H, int with_structure);
int fav_consistency_u8(const float* flow1_flo, const float* flow2_flo, const uint8_t* rgb_hwc,
                       uint8_t* out, int W, int H, void* workspace, size_t workspace_bytes,
                       fav_hipstream_t stream);

/* ---- A5: certainty erosion -------------------------------------------------------------------
 * Replaces utils.min_filter(cert, r) (fast_artistic_video/utils.lua:161-169):
 * 1 - maxpool_{r x r, stride 1, pad r/2}(1 - cert), window truncated at the borders. cert [H][W]. */
int fav_min_filter_f32(const float* cert, float* out, int H, int W, int r, fav_hipstream_t stream);

/* ---- A6/A7: VGG preprocessing + 7-channel input assembly --------------------------------------
 * Replaces run_next_image's input construction (fast_artistic_video_core.lua:161-171) and the first
 * frame's (:133-138) with fill_occlusions = vgg-mean; preprocess.lua:57-62.
 * frame_rgb [3][H][W] RGB in [0,1]; warped_rgb [3][H][W] or NULL (first frame);
 * cert [H][W] (already min-filtered) or NULL; in7 [7][H][W]. */
int fav_assemble_input_f32(const float* frame_rgb, const float* warped_rgb, const float* cert,
                           float* in7, int H, int W, fav_hipstream_t stream);

/* ---- A8: the transformer network ---------------------------------------------------------------
 * Replaces torch.load(path).model + model:forward(input) (fast_artistic_video_core.lua:38-57,172);
 * layer semantics of fast_artistic_video/models_video.lua:10-140, InstanceNormalization.lua:33-53,
 * ShaveImage.lua:9-16, train_video.lua:319-325. */
typedef struct fav_net fav_net;

/* parse a Torch7 .t7 checkpoint on the host and upload the weights to `device` */
int fav_net_create(const char* t7_path_host, int device, fav_net** out);
/* host-side: flatten a .t7 checkpoint into a self-describing blob (what rank 0 broadcasts over
 * RCCL in the multi-GPU launcher).  Call with blob_host=NULL to query the size. */
int fav_net_pack_host(const char* t7_path_host, void* blob_host, size_t capacity, size_t* bytes);
/* create a net from a packed blob (host memory) */
int fav_net_create_from_blob(const void* blob_host, size_t bytes, int device, fav_net** out);
void fav_net_destroy(fav_net* net);
/* The stream-K convolution grids hand partial tiles between co-resident blocks; a hand-off that cannot complete (another
 * context holding compute units of the device) times out inside the kernel and sets a host-visible word.  Call this AFTER the
 * work of a frame has completed (event / stream synchronised): FAV_EHIP means the frames computed since the last check are
 * wrong.  fav_stylize checks before every PNG is queued and at exit; fav_net_forward / fav_stream_* also fail on their next
 * call.  Clears the word. */
int fav_net_check(fav_net* net);
/* shared != 0: the device is NOT exclusively this network's (other processes / contexts compute on it).  The convolutions then
 * use data-parallel grids only -- one block per tile, nothing handed between blocks, no co-residency assumed -- at some cost in
 * load balance.  fav_net_check switches a network to this mode by itself after a hand-off has timed out. */
int fav_net_set_shared_device(fav_net* net, int shared);
/* `stream` (a hipStream_t that forwards of this network's device were enqueued on) is about to be destroyed by its owner: drains it
 * if it is the one the library would otherwise wait on before the next forward on another stream (concurrency note above). */
int fav_net_forget_stream(fav_net* net, fav_hipstream_t stream);
/* human-readable layer list (one line per layer) -- used to cross-check the .t7 reader */
int fav_net_describe_host(const fav_net* net, char* buf_host, size_t capacity);
/* host-only (no device needed): parse a .t7 checkpoint and write the same layer list text */
int fav_t7_describe_host(const char* t7_path_host, char* buf_host, size_t capacity);
/* number of fp32 parameters (conv weights+biases+IN gamma/beta) */
long long fav_net_param_count(const fav_net* net);
/* output size for an H x W input (equals H x W when H, W are multiples of 4) */
int fav_net_output_size(const fav_net* net, int H, int W, int* Ho, int* Wo);
/* in7 [1][7][H][W] -> out3 [1][3][Ho][Wo] (150*tanh(...), BGR mean-subtracted space), fp32 NCHW.
 * Internal buffers are (re)sized on first use for a given H x W (allocation happens outside the
 * stream order only then). */
int fav_net_forward(fav_net* net, const float* in7, float* out3, int H, int W, fav_hipstream_t stream);

/* per-convolution timing with HIP events recorded on the forward's stream (bench / roofline use).
 * enable: every later forward records one event pair per convolution launch.  read: synchronises the
 * recorded events and returns, per convolution in network order, the summed milliseconds, the number of
 * launches, the useful MACs of one launch (no padding) and an id of the kernel instance that ran it (1 = row-folded
 * last layer, 8 = first-layer kernel, 300+N = halo-resident 3x3 kernel with N output channels, otherwise the N tile
 * 128/64/32 of the generic implicit-GEMM kernel); then clears the accumulators.  Arrays hold up to `capacity` entries. */
int fav_net_profile_enable(fav_net* net, int on);
int fav_net_profile_read_host(fav_net* net, int capacity, int* count, double* ms_sum, int* launches,
                              double* macs_per_launch, int* ntile);

/* operator-level entry (nn.SpatialConvolution [+ nn.InstanceNormalization [+ nn.ReLU]]), NCHW fp32;
 * weight [Cout][Cin][k][k], bias [Cout] or NULL, gamma/beta [Cout] or NULL (no IN).
 * Runs the same MFMA implicit-GEMM kernel the network uses.  Allocates temporaries: test/ops use. */
int fav_conv2d_nchw_f32(const float* in, int Cin, int H, int W,
                        const float* weight, const float* bias, int Cout, int k, int stride, int pad,
                        const float* gamma, const float* beta, float eps, int relu,
                        float* out, fav_hipstream_t stream);

/* ---- the fused per-frame pipeline (A1..A10 minus file I/O) -------------------------------------
 * Replaces one iteration of run_fast_neural_video's loop (fast_artistic_video_core.lua:194-211)
 * with the video CLI's callbacks (fast_artistic_video.lua:93-172).  Holds the recurrent state
 * last_frame_stylized (float, unclamped: fast_artistic_video.lua:169) on the device. */
/* Arithmetic of the network.  FAV_PRECISION_FP32 (default) is the parity mode: fp32 operands, exact fp32 MFMA.  The optional
 * fast mode FAV_PRECISION_BF16_OPERANDS rounds the operands of the halo-resident 3x3 convolutions (the
 * residual blocks and c3s1-64: 71 % of the FLOPs) to bf16 on their way into LDS, accumulates in fp32 and keeps every
 * activation in fp32; it is NOT bit-compatible with the reference and is gated by PSNR against the fp32 oracle in the tests. */
enum fav_precision { FAV_PRECISION_FP32 = 0, FAV_PRECISION_BF16_OPERANDS = 1 };
int fav_net_set_precision(fav_net* net, int mode);

typedef struct fav_stream fav_stream;

typedef struct fav_stream_opts {
    int border_mode;           /* enum fav_border */
    int occlusions_min_filter; /* -occlusions_min_filter (default 7) */
    int invert_occlusion;      /* -invert_occlusion */
    int fix_occlusions;        /* -fix_occlusions */
    int fill_random;           /* -fill_occlusions uniform-random (1) | vgg-mean (0): core.lua:108-117.  The reference's
                                  torch.rand is unseeded; here u = counter RNG(seed, frame counter, channel, y, x), see fav_vr */
    unsigned seed;
} fav_stream_opts;

/* H x W: size of the frames (= flows = certainty maps).  The stylised frame has the NETWORK's output size Ho x Wo
 * (fav_stream_output_size): equal to H x W when both are multiples of 4, up to 3 pixels larger otherwise (two stride-2
 * convolutions, two x2 upsamplings).  As in the reference, that larger image is what is saved and kept as the recurrent state, and
 * the next frame's warp samples it on the flow's H x W grid (stnbdhw/BilinearSamplerBDHW.lua:71: the output takes the grid's size;
 * stylizeVideo_deepflow.sh:72-78 lets the user pick any w:h, e.g. 854x480).  Every out_rgb_f32 below is [3][Ho][Wo], every
 * out_rgb8_hwc [Ho][Wo][3]. */
int fav_stream_create(fav_net* net, int H, int W, const fav_stream_opts* opts_host, fav_stream** out);
int fav_stream_output_size(const fav_stream* s, int* Ho, int* Wo);
void fav_stream_destroy(fav_stream* s);
/* -model_img <file>: a separate 3-channel image model (fast_artistic_video_core.lua:59-66,146) that stylises frames
 * without a prior (fav_stream_first_frame).  NULL restores 'self' (the video model with an all-occluded prior).  The
 * image model must use the same leading reflection padding as the video model.  Not owned by the stream. */
int fav_stream_set_image_net(fav_stream* s, fav_net* image_net);
/* first frame / -create_inconsistent: core.lua:121-158 with model_img == 'self'.
 * frame_rgb_hwc: P6 payload [H][W][3] u8.  out_rgb_f32 [3][H][W] float RGB (deprocessed, unclamped)
 * and/or out_rgb8_hwc [H][W][3] u8 (image.save quantisation); either may be NULL. */
int fav_stream_first_frame(fav_stream* s, const uint8_t* frame_rgb_hwc,
                           float* out_rgb_f32, uint8_t* out_rgb8_hwc, fav_hipstream_t stream);
/* next frame with a precomputed certainty map (the .pgm written by consistencyChecker):
 * backward_flo [H][W][2] (u,v) payload of backward_[i]_{i-1}.flo, cert_pgm [H][W] u8. */
int fav_stream_next_frame_cert(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                               const uint8_t* cert_pgm, float* out_rgb_f32, uint8_t* out_rgb8_hwc,
                               fav_hipstream_t stream);
/* next frame with the consistency check fused on the GPU: forward_flo = forward_{i-1}_[i].flo.
 * use_structure != 0 selects the 4-argument (image-structure) mode of the checker. */
int fav_stream_next_frame_flow(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                               const float* forward_flo, int use_structure,
                               float* out_rgb_f32, uint8_t* out_rgb8_hwc, fav_hipstream_t stream);
/* optional look-ahead: start computing the consistency mask of a FUTURE frame on an internal side stream (the mask
 * depends only on that frame and its two flows, not on the recurrent state), so the order-preserving 4-argument
 * structure pass overlaps the network of the current frame.  The inputs must already be complete on `stream`.  The next
 * fav_stream_next_frame_flow call with the same three pointers and mode consumes the prefetched mask. */
int fav_stream_prefetch_mask(fav_stream* s, const uint8_t* frame_rgb_hwc, const float* backward_flo,
                             const float* forward_flo, int use_structure, fav_hipstream_t stream);
/* Host-ordered look-ahead (on != 0): fav_stream_prefetch_mask then records nothing on `stream` and nothing waits on it -- the CALLER
 * guarantees that the three input buffers are complete when it calls (it has seen their upload finish) -- and the matching
 * fav_stream_next_frame_flow waits for the look-ahead on the HOST (a sequence number the mask pipeline's last kernel stores into
 * host-mapped memory; polled with short sleeps, normally already there) instead of making the compute queue wait on an event.
 * Why: on this runtime every event record behind long-running kernels and every dependency between queues keeps a runtime thread
 * spinning until it resolves -- one core per process ("quiet synchronisation").  The look-ahead's OUTPUT buffers are the
 * library's business in both modes: host-ordered, a look-ahead that would rewrite a mask / certainty buffer which frames still in the
 * caller's queue read waits on the host until those frames are through (a retire sequence number stored by a one-thread kernel in the
 * caller's queue; with up to two frames of look-ahead that is never a wait), so a caller may run any number of frames ahead.
 * Default: off (event-ordered, fully asynchronous). */
int fav_stream_set_host_ordered(fav_stream* s, int on);
/* read / overwrite the recurrent state ([3][Ho][Wo] float RGB) -- for -continue_with */
int fav_stream_get_state(fav_stream* s, float* state_rgb_f32, fav_hipstream_t stream);
int fav_stream_set_state(fav_stream* s, const float* state_rgb_f32, fav_hipstream_t stream);
/* test view: the 7-channel network input of the LAST frame as run_next_image assembles it (fast_artistic_video_core.lua:161-171:
 * content | masked warped prior + fill | certainty), [7][H][W] fp32, copied out of the fused kernel's padded buffer */
int fav_stream_get_input_f32(const fav_stream* s, float* in7, fav_hipstream_t stream);
/* device pointer of the last certainty mask used (u8 [H][W], before the min filter) -- tests */
const uint8_t* fav_stream_last_mask(const fav_stream* s);

/* fp32 sum in index order with the rounding of every partial sum: bit-identical to `float s = 0; for (i) s += x[i];`, the
 * arithmetic of CMatrix::avg (consistencyChecker/CMatrix.h:1245-1251) that the 4-argument checker depends on.  Evaluated in
 * parallel (exact parity-transducer scan; kernels_consistency.hip).  x, sum_out: device pointers. */
int fav_sequential_sum_f32(const float* x, size_t n, float* sum_out, fav_hipstream_t stream);

/* ---- temporal-consistency metric -------------------------------------------------
 * The third number of func_eval (fast_artistic_video.lua:128-151, -evaluate without the VGG terms):
 * MSE(warp(prev_stylised, flow) * cert, cur_stylised * cert) over 3*H*W elements (nn.MSECriterion).  prev / cur: [3][H][W]
 * float RGB (what fav_stream_get_state returns), backward_flo: .flo payload, cert_pgm: u8 [H][W].  Synchronises `stream`;
 * the value is written to host memory. */
int fav_temporal_loss_host(const float* prev_rgb, const float* cur_rgb, const float* backward_flo, const uint8_t* cert_pgm,
                           int H, int W, int border_mode, double* loss_host, fav_hipstream_t stream);

/* ---- 360-degree cube-map orchestration --------------------------------------------
 * Replaces the callbacks fast_artistic_video_vr.lua passes to run_fast_neural_video (fast_artistic_video_core.lua:189-229):
 * func_load_cert (:204-237), func_make_last_frame_warped (:239-302), func_is_single_image (:304-310), func_save_image /
 * blend_other_sides (:454-559), with the static maps of fast_artistic_video/vr_helper.lua:3-184.  Faces are hplus x wplus
 * (face + overlap); face index i is 1-based and frame-major, mode = (i-1) % 6 walks the processing order of file ids
 * {6,1,2,5,3,4} (:103); the caller maps ids to files.  `-fill_occlusions uniform-random` is an unseeded torch.rand in the
 * reference (core.lua:108-117); here it is a documented counter RNG keyed by (seed, i, channel, y, x). */
typedef struct fav_vr fav_vr;
typedef struct fav_vr_opts {
    int overlap_w, overlap_h;        /* -overlap_pixel_w / _h (fast_artistic_video_vr.lua:42-43); even, > 10 */
    int occlusions_min_filter;       /* :36 */
    int median_filter;               /* :50; 0, 3 or 5 */
    int fill_random;                 /* -fill_occlusions uniform-random (1) | vgg-mean (0) */
    unsigned seed;                   /* of the documented RNG */
    int create_inconsistent;         /* :39 */
    int create_inconsistent_border;  /* :40 */
    int out_equi_w, out_equi_h;      /* :46-47; 0 = no equirectangular output */
    int border_mode;                 /* fav_border of every warp (the reference's GPU path: FAV_BORDER_STN) */
} fav_vr_opts;
int fav_vr_create(fav_net* video_net, fav_net* image_net_or_null, int hplus, int wplus, const fav_vr_opts* opts, fav_vr** out);
void fav_vr_destroy(fav_vr* v);
/* one cube face.  frame: u8 [hplus][wplus][3]; backward_flo ([H][W][2] .flo payload) and cert_pgm (u8 [H][W]) are required
 * from the second frame on (i >= 7) and ignored before; out_rgb_f32 ([3][H][W], may be NULL) receives the stylised face */
int fav_vr_face(fav_vr* v, int i, const uint8_t* frame_rgb_hwc, const float* backward_flo, const uint8_t* cert_pgm,
                float* out_rgb_f32, fav_hipstream_t stream);
/* after the sixth face of a frame: post-blend (the blended faces become the next frame's warp sources), median filter,
 * equirectangular image (u8 [out_equi_h][out_equi_w][3]) and / or cube-map strip (u8 [cube_h][cube_w][3]); either may be NULL */
int fav_vr_finish_frame(fav_vr* v, uint8_t* equi_rgb8_hwc, uint8_t* cubemap_rgb8_hwc, fav_hipstream_t stream);
int fav_vr_output_sizes(const fav_vr* v, int* equi_w, int* equi_h, int* cube_w, int* cube_h, int* filtered_w, int* filtered_h);
/* float views for tests: which = 0 this frame's raw face k, 1 blended face k, 2 median-filtered face k, 3 equirectangular
 * image, 4 cube-map strip (device to device copy) */
int fav_vr_get_f32(const fav_vr* v, int which, int k, float* out_dev, fav_hipstream_t stream);
/* the static maps, computed on the host (no device needed): kind 0..3 = perspective map left/right/top/bottom
 * ([2][hplus][wplus], `overlap` = the crop along that axis), 4 = cube->equirectangular map ([2][out_h][out_w]) */
int fav_vr_map_host(int kind, int hplus, int wplus, int overlap, int median_filter, int out_w, int out_h, float* out_host);

/* ---- host-side formats (A1, A9) ----------------------------------------------------------------
 * .flo: flowFileLoader.lua:14-34 / consistencyChecker.cpp:16-36 (tag read, not validated);
 * P6/P5 8-bit binary PNM; PNG writer (RGB8, zlib).  Buffers are malloc'ed; free with fav_free_host. */
int fav_read_flo_host(const char* path, float** uv_out, int* W, int* H);
int fav_read_pnm_host(const char* path, uint8_t** data_out, int* W, int* H, int* channels);
/* the same readers into a caller-provided (e.g. pinned) buffer: no allocation, FAV_EINVAL if the payload does not fit */
int fav_read_flo_into_host(const char* path, float* uv_buf, size_t capacity_floats, int* W, int* H);
int fav_read_pnm_into_host(const char* path, uint8_t* buf, size_t capacity_bytes, int* W, int* H, int* channels);
int fav_write_pgm_host(const char* path, const uint8_t* data, int W, int H);
int fav_write_png_rgb8_host(const char* path, const uint8_t* rgb_hwc, int W, int H, int zlib_level);
void fav_free_host(void* p);

/* ---- A9 on the GPU: image.save("<prefix>-%05d.png", img) (fast_artistic_video.lua:160-170) -----------------------------------
 * The BYTES OF THE PNG FILE are produced on the device (Sub filter; one deflate block per image row -- run-length matches, the
 * cheapest of twelve ready-made Huffman codes / the fixed code / stored, by exact bit count -- re-aligned by an empty stored block;
 * Adler-32 and CRC-32 combined from per-row parts: csrc/kernels_png.hip, csrc/png_tables.cpp), so the host only write()s them: the
 * zlib path above costs ~25 ms per 1280x720 frame and core.  The files decode to exactly the bytes fav_stream_* writes to
 * out_rgb8_hwc (clamp to [0,1], x255, truncate); sizes are within a few per cent of zlib's level-1 run-length mode on image rows.
 *   png_out        device-ACCESSIBLE memory of fav_png_capacity(W, H) bytes, 4-byte aligned: device memory or host-mapped pinned
 *                  memory (hipHostMalloc: the packed bytes then cross PCIe once and are complete when the stream reaches the point
 *                  after the call -- use an event WITH the system-scope fence);
 *   png_bytes_out  device-accessible uint32: the file size;
 *   workspace      fav_png_workspace_bytes(W, H) bytes of device memory, 16-byte aligned, owned by the caller, private to the call
 *                  until the stream has passed it.
 * fav_png_encode_rgb8: rgb_hwc = u8 [H][W][3];  fav_png_encode_f32: planar float RGB [3][H][W] (what fav_stream_get_state returns),
 * quantisation fused.  Width <= 9000; any height whose file stays below 4 GiB.  rgb_hwc needs no padding or alignment:
 * nothing outside its H*W*3 bytes is read. */
size_t fav_png_capacity(int W, int H);
size_t fav_png_workspace_bytes(int W, int H);
int fav_png_encode_rgb8(const uint8_t* rgb_hwc, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes_out,
                        void* workspace, size_t workspace_bytes, fav_hipstream_t stream);
int fav_png_encode_f32(const float* rgb_planar, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes_out,
                       void* workspace, size_t workspace_bytes, fav_hipstream_t stream);
/* host-only (no device needed): the encoder's Huffman codes, for tools that restate its bit layout (oracle/png_model.py).
 * count tables of table_bytes bytes each: uint32 sym[277] = (code length << 16) | bit-reversed code; uint32 hdr[40] = the
 * dynamic-block header behind BFINAL / BTYPE, LSB first; uint32 hdr_bits, btype, dist_len, dist_code.  The last table is the
 * fixed code of RFC 1951 3.2.6.  out_host may be NULL to query count and table_bytes. */
int fav_png_tables_host(void* out_host, size_t capacity, int* count, int* table_bytes);
/* host-only (no device needed): crc32(A || B) from crc32(A), crc32(B) and |B| in bytes, by the table scheme the encoder's kernels use
 * to raise a row's CRC to its position in the IDAT chunk (any 32-bit length) -- lets the CPU suite pin it against zlib */
uint32_t fav_png_crc32_combine_host(uint32_t crc_a, uint32_t crc_b, uint32_t len_b);
/* the stream's current stylised frame (the recurrent state) as a PNG file, with the stream's own workspace */
int fav_stream_encode_png(fav_stream* s, void* png_out, size_t capacity, uint32_t* png_bytes_out, fav_hipstream_t stream);
/* the same file, produced NEXT TO whatever `stream` is given afterwards (normally the next frame's network: the encoder's short kernels
 * fill the tails of its grids instead of standing in front of it): enqueued on a queue the fav_stream owns, behind everything `stream`
 * holds at the time of the call.  The stream's state is double-buffered from the first such call on -- the next frame is written
 * next to the one being encoded, and the frame after it waits (on the device) for the encoder of the buffer it returns to -- so the
 * fav_stream_* calls may follow immediately.  png_out / png_bytes_out are complete when *png_bytes_out (set it to 0 first; e.g.
 * host-mapped memory) becomes the file's size, or behind fav_stream_wait_png on a queue of the caller's.  One encode per frame;
 * successive encodes run in order.  Same bytes as fav_stream_encode_png.  From the first call on the stream's network(s) run the way they
 * do next to the look-ahead masks (a few CUs left to the side queue, no hand-off between co-resident blocks of the generic kernel): the
 * encoder shares the device with the following forward.  A forward that fails leaves the state at the last complete frame.
 * User: `fav_stylize -png_overlap 1` (measured: no gain over the default). */
int fav_stream_encode_png_async(fav_stream* s, void* png_out, size_t capacity, uint32_t* png_bytes_out, fav_hipstream_t stream);
/* makes `stream` wait (on the device) for the asynchronous encodes issued so far */
int fav_stream_wait_png(fav_stream* s, fav_hipstream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAV_H */
