// api.cpp -- C ABI glue: error reporting, device check, operator-level entry points (A2, A3/A4, A5, A6/A7)
// and the host-side file formats (A1, A9).  See include/fav.h for the reference interfaces replaced.
#include <unistd.h>
#include <zlib.h>

#include <cctype>
#include <cstdlib>
#include <cstring>

#include "fav_internal.h"

#include <dlfcn.h>

namespace fav {

namespace {
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const Roctx& roctx()
{
    static const Roctx r = [] {
        Roctx x;
        if (!getenv("FAV_ROCTX")) return x;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (x.push && x.pop) return x;
                x = Roctx();
            }
        }
        return x;
    }();
    return r;
}
}  // namespace

bool TraceRange::enabled() { return roctx().push != nullptr; }
TraceRange::TraceRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
TraceRange::~TraceRange() { if (on) roctx().pop(); }


static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return FAV_EHIP;
}

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s): libfav has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return FAV_ENODEVICE;
    }
    return FAV_OK;
}

}  // namespace fav

using namespace fav;

extern "C" const char* fav_last_error(void) { return g_err; }
extern "C" int fav_version(void) { return 100; }

extern "C" int fav_device_count(void)
{
    int rc = ensure_device();
    if (rc) return rc;
    int n = 0;
    (void)hipGetDeviceCount(&n);
    return n;
}

extern "C" int fav_warp_bdhw_f32(const float* img, const float* flow, float* out, int B, int C, int H, int W, int Ho, int Wo,
                                 int border_mode, fav_hipstream_t stream)
{
    // shape contract of BilinearSamplerBDHW.lua:26-42
    FAV_REQUIRE(img && flow && out, "fav_warp_bdhw_f32: null pointer");
    FAV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "fav_warp_bdhw_f32: non-positive dimension");
    FAV_REQUIRE(border_mode == FAV_BORDER_STN || border_mode == FAV_BORDER_CPU, "fav_warp_bdhw_f32: unknown border mode %d", border_mode);
    int rc = ensure_device(); if (rc) return rc;
    return launch_warp(img, flow, out, B, C, H, W, Ho, Wo, border_mode, static_cast<hipStream_t>(stream));
}

extern "C" size_t fav_consistency_workspace_bytes(int W, int H, int with_structure)
{
    return with_structure ? structure_workspace_bytes(W, H) : 0;
}

extern "C" int fav_consistency_u8(const float* flow1_flo, const float* flow2_flo, const uint8_t* rgb_hwc, uint8_t* out, int W,
                                  int H, void* workspace, size_t workspace_bytes, fav_hipstream_t stream)
{
    FAV_REQUIRE(flow1_flo && flow2_flo && out && W > 0 && H > 0, "fav_consistency_u8: bad argument");
    int rc = ensure_device(); if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* structure = nullptr; const float* avg = nullptr;
    if (rgb_hwc) {
        rc = launch_structure(rgb_hwc, W, H, workspace, workspace_bytes, &structure, &avg, st);
        if (rc) return rc;
    }
    return launch_consistency(flow1_flo, flow2_flo, structure, avg, out, W, H, st);
}

extern "C" int fav_min_filter_f32(const float* cert, float* out, int H, int W, int r, fav_hipstream_t stream)
{
    FAV_REQUIRE(cert && out && H > 0 && W > 0 && r >= 1, "fav_min_filter_f32: bad argument");
    int rc = ensure_device(); if (rc) return rc;
    return launch_min_filter_f32(cert, out, H, W, r, static_cast<hipStream_t>(stream));
}

// ---- A9 on the GPU: the bytes of the PNG file (kernels_png.hip)
extern "C" size_t fav_png_capacity(int W, int H) { return (W > 0 && H > 0) ? png_capacity(W, H) : 0; }
extern "C" size_t fav_png_workspace_bytes(int W, int H) { return (W > 0 && H > 0) ? png_workspace_bytes(W, H) : 0; }

extern "C" uint32_t fav_png_crc32_combine_host(uint32_t crc_a, uint32_t crc_b, uint32_t len_b) { return png_crc32_combine_host(crc_a, crc_b, len_b); }

extern "C" int fav_png_encode_rgb8(const uint8_t* rgb_hwc, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes_out,
                                   void* workspace, size_t workspace_bytes, fav_hipstream_t stream)
{
    FAV_REQUIRE(rgb_hwc, "fav_png_encode_rgb8: null image");
    int rc = ensure_device(); if (rc) return rc;
    return launch_png_encode(rgb_hwc, nullptr, W, H, png_out, capacity, png_bytes_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int fav_png_encode_f32(const float* rgb_planar, int W, int H, void* png_out, size_t capacity, uint32_t* png_bytes_out,
                                  void* workspace, size_t workspace_bytes, fav_hipstream_t stream)
{
    FAV_REQUIRE(rgb_planar, "fav_png_encode_f32: null image");
    int rc = ensure_device(); if (rc) return rc;
    return launch_png_encode(nullptr, rgb_planar, W, H, png_out, capacity, png_bytes_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int fav_assemble_input_f32(const float* frame_rgb, const float* warped_rgb, const float* cert, float* in7, int H,
                                      int W, fav_hipstream_t stream)
{
    FAV_REQUIRE(frame_rgb && in7 && H > 0 && W > 0, "fav_assemble_input_f32: bad argument");
    FAV_REQUIRE((warped_rgb == nullptr) == (cert == nullptr), "fav_assemble_input_f32: warped_rgb and cert must both be given or both be NULL");
    int rc = ensure_device(); if (rc) return rc;
    return launch_assemble(frame_rgb, warped_rgb, cert, in7, H, W, static_cast<hipStream_t>(stream));
}

// left-to-right fp32 sum, bit-identical to `float s = 0; for (i) s += x[i];` (CMatrix::avg, CMatrix.h:1245-1251), evaluated
// with the parallel parity-transducer scan of kernels_consistency.hip
extern "C" int fav_sequential_sum_f32(const float* x, size_t n, float* sum_out, fav_hipstream_t stream)
{
    FAV_REQUIRE(x && sum_out && n > 0, "fav_sequential_sum_f32: bad argument");
    int rc = ensure_device(); if (rc) return rc;
    return launch_sequential_sum(x, n, sum_out, static_cast<hipStream_t>(stream));
}

// temporal-consistency metric of the reference's -evaluate mode (fast_artistic_video.lua:128-151) without the VGG terms
extern "C" int fav_temporal_loss_host(const float* prev_rgb, const float* cur_rgb, const float* backward_flo, const uint8_t* cert_pgm,
                                      int H, int W, int border_mode, double* loss_host, fav_hipstream_t stream)
{
    FAV_REQUIRE(prev_rgb && cur_rgb && backward_flo && cert_pgm && loss_host && H > 0 && W > 0, "fav_temporal_loss_host: bad argument");
    int rc = ensure_device(); if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* part = nullptr;
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(&part), 256 * sizeof(double)));
    rc = launch_temporal_loss(prev_rgb, cur_rgb, backward_flo, cert_pgm, border_mode, H, W, part, st);
    double h[256];
    if (!rc && hipMemcpyAsync(h, part, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipMemcpyAsync");
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipStreamSynchronize");
    (void)hipFree(part);
    if (rc) return rc;
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += h[i];
    *loss_host = s / (3.0 * (double)H * (double)W);
    return FAV_OK;
}

// ================================================================================================
// host-side formats
// ================================================================================================
extern "C" void fav_free_host(void* p) { free(p); }

// flowFileLoader.lua:14-34 / consistencyChecker.cpp:16-36: float tag (not validated), int32 W, int32 H,
// then H*W interleaved (u, v) little-endian float32
// dst == null: malloc the payload (returned through *uv_out); else read into dst (capacity in floats)
static int read_flo(const char* path, float** uv_out, float* dst, size_t capacity, int* W, int* H)
{
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("Could not open %s", path); return FAV_EIO; }
    float tag; int w = 0, h = 0;
    if (fread(&tag, 4, 1, f) != 1 || fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1 || w <= 0 || h <= 0 ||
        (long long)w * h > (1ll << 28)) {
        fclose(f); set_error("%s: bad .flo header", path); return FAV_EFORMAT; }
    const size_t n = (size_t)w * h * 2;
    if (dst && n > capacity) { fclose(f); set_error("%s: %dx%d flow does not fit the caller's buffer", path, w, h); return FAV_EINVAL; }
    float* d = dst ? dst : static_cast<float*>(malloc(n * sizeof(float)));
    if (!d) { fclose(f); set_error("out of host memory"); return FAV_EIO; }
    if (fread(d, sizeof(float), n, f) != n) { fclose(f); if (!dst) free(d); set_error("%s: truncated .flo payload", path); return FAV_EFORMAT; }
    fclose(f);
    if (uv_out) *uv_out = d;
    *W = w; *H = h;
    return FAV_OK;
}

extern "C" int fav_read_flo_host(const char* path, float** uv_out, int* W, int* H)
{
    FAV_REQUIRE(path && uv_out && W && H, "fav_read_flo_host: null argument");
    return read_flo(path, uv_out, nullptr, 0, W, H);
}

extern "C" int fav_read_flo_into_host(const char* path, float* uv_buf, size_t capacity_floats, int* W, int* H)
{
    FAV_REQUIRE(path && uv_buf && W && H, "fav_read_flo_into_host: null argument");
    return read_flo(path, nullptr, uv_buf, capacity_floats, W, H);
}

static bool pnm_token(FILE* f, char* buf, size_t cap)
{
    int c = fgetc(f);
    for (;;) {
        while (c != EOF && isspace(c)) c = fgetc(f);
        if (c == '#') { while (c != EOF && c != '\n') c = fgetc(f); continue; }
        break;
    }
    size_t n = 0;
    while (c != EOF && !isspace(c) && n + 1 < cap) { buf[n++] = (char)c; c = fgetc(f); }
    buf[n] = 0;
    return n > 0;    // the single whitespace after the token has been consumed
}

// binary P6 / P5, maxval 255 (what ffmpeg and consistencyChecker write; image.load accepts the same)
static int read_pnm(const char* path, uint8_t** data_out, uint8_t* dst, size_t capacity, int* W, int* H, int* channels)
{
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("Could not open %s", path); return FAV_EIO; }
    char tok[64];
    int ch = 0;
    if (pnm_token(f, tok, sizeof tok)) { if (!strcmp(tok, "P6")) ch = 3; else if (!strcmp(tok, "P5")) ch = 1; }
    int w = 0, h = 0, maxv = 0;
    if (ch && pnm_token(f, tok, sizeof tok)) w = atoi(tok);
    if (ch && pnm_token(f, tok, sizeof tok)) h = atoi(tok);
    if (ch && pnm_token(f, tok, sizeof tok)) maxv = atoi(tok);
    if (!ch || w <= 0 || h <= 0 || maxv != 255 || (long long)w * h > (1ll << 28)) {
        fclose(f); set_error("%s: not a binary 8-bit P5/P6 file", path); return FAV_EFORMAT; }
    const size_t n = (size_t)w * h * ch;
    if (dst && n > capacity) { fclose(f); set_error("%s: %dx%dx%d image does not fit the caller's buffer", path, w, h, ch); return FAV_EINVAL; }
    uint8_t* d = dst ? dst : static_cast<uint8_t*>(malloc(n));
    if (!d) { fclose(f); set_error("out of host memory"); return FAV_EIO; }
    if (fread(d, 1, n, f) != n) { fclose(f); if (!dst) free(d); set_error("%s: truncated image payload", path); return FAV_EFORMAT; }
    fclose(f);
    if (data_out) *data_out = d;
    *W = w; *H = h; *channels = ch;
    return FAV_OK;
}

extern "C" int fav_read_pnm_host(const char* path, uint8_t** data_out, int* W, int* H, int* channels)
{
    FAV_REQUIRE(path && data_out && W && H && channels, "fav_read_pnm_host: null argument");
    return read_pnm(path, data_out, nullptr, 0, W, H, channels);
}

extern "C" int fav_read_pnm_into_host(const char* path, uint8_t* buf, size_t capacity_bytes, int* W, int* H, int* channels)
{
    FAV_REQUIRE(path && buf && W && H && channels, "fav_read_pnm_into_host: null argument");
    return read_pnm(path, nullptr, buf, capacity_bytes, W, H, channels);
}

// CMatrix::writeToPGM header (CMatrix.h:1064): "P5\n%d %d\n255\n"; written to a temp file and renamed so
// a polling consumer never sees a partial mask (the reference pre-writes an all-255 file, consistencyChecker.cpp:152)
extern "C" int fav_write_pgm_host(const char* path, const uint8_t* data, int W, int H)
{
    FAV_REQUIRE(path && data && W > 0 && H > 0, "fav_write_pgm_host: bad argument");
    // (the temporary name is per process: two writers of one output -- a caller that gave up on the checker's resident helper next to
    //  the helper itself -- never share a half-written file; whoever renames last wins with complete bytes)
    std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) { set_error("Could not open %s for writing", tmp.c_str()); return FAV_EIO; }
    fprintf(f, "P5\n%d %d\n255\n", W, H);
    const bool ok = fwrite(data, 1, (size_t)W * H, f) == (size_t)W * H;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path) != 0) { unlink(tmp.c_str()); set_error("write to %s failed", path); return FAV_EIO; }
    return FAV_OK;
}

static void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

static void png_chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t len)
{
    put_be32(out, (uint32_t)len);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(len + 4)));
}

// image.save("<prefix>-%05d.png") (fast_artistic_video.lua:160-167): 8-bit RGB PNG.  libpng is not
// available in the target image; the container is written directly on top of zlib's deflate.
extern "C" int fav_write_png_rgb8_host(const char* path, const uint8_t* rgb_hwc, int W, int H, int zlib_level)
{
    FAV_REQUIRE(path && rgb_hwc && W > 0 && H > 0, "fav_write_png_rgb8_host: bad argument");
    const size_t stride = (size_t)W * 3;
    // per-thread scratch that persists across calls: a writer pool would otherwise fault in ~8 MB of fresh pages per frame
    // on every thread, and the page-fault path serialises on the process's address-space lock
    static thread_local std::vector<uint8_t> raw, comp, out;
    raw.resize((stride + 1) * H);
    const int level = zlib_level < 0 ? 1 : (zlib_level > 9 ? 9 : zlib_level);
    // Level 0 stores the rows (filter None).  Otherwise every row takes the Sub filter (x - left neighbour, per byte over 3-byte
    // pixels); at the CLI's default level 1 the deflate strategy is Z_RLE: on 1280x720 frames 25 ms per frame and core instead of
    // 85 ms with zlib's default strategy at level 1, and SMALLER files on smooth, textured and noise-like content alike (the
    // filter removes what LZ77 at level 1 would have looked for).  Levels >= 2 keep zlib's default strategy behind the filter.
    for (int y = 0; y < H; ++y) {
        uint8_t* dst = raw.data() + (stride + 1) * y;
        const uint8_t* src = rgb_hwc + stride * y;
        if (level == 0) { dst[0] = 0; memcpy(dst + 1, src, stride); continue; }
        dst[0] = 1;                  // filter type 1 (Sub)
        dst[1] = src[0]; dst[2] = src[1]; dst[3] = src[2];
        for (size_t x = 3; x < stride; ++x) dst[1 + x] = (uint8_t)(src[x] - src[x - 3]);
    }
    uLongf clen = compressBound((uLong)raw.size()) + 64;
    if (comp.size() < clen) comp.resize(clen);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, level == 1 ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) { set_error("zlib deflateInit2 failed"); return FAV_EIO; }
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = comp.data(); zs.avail_out = (uInt)clen;
    const int zrc = deflate(&zs, Z_FINISH);
    clen = zs.total_out;
    deflateEnd(&zs);
    if (zrc != Z_STREAM_END) { set_error("zlib deflate failed"); return FAV_EIO; }
    out.clear();
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)W); put_be32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    png_chunk(out, "IDAT", comp.data(), clen);
    png_chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("Could not open %s for writing", path); return FAV_EIO; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    if (fclose(f) != 0 || !ok) { set_error("write to %s failed", path); return FAV_EIO; }
    return FAV_OK;
}
