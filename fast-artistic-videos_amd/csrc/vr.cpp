// vr.cpp -- 360-degree cube-map orchestration around the unchanged warp (A2) and network (A8) kernels:
// the callbacks fast_artistic_video_vr.lua hands to run_fast_neural_video (core.lua:189-229), as one C-ABI object.
//
//   fav_vr_create       : static perspective warp maps + masks + gradient masks, once (fast_artistic_video_vr.lua:164-198,
//                         vr_helper.lua:3-96), optional cube->equirectangular map (vr_helper.lua:99-184)
//   fav_vr_face         : one cube face = func_load_cert (:204-237) -> min filter (core.lua:207) -> func_make_last_frame_warped
//                         (:239-302) -> input assembly + network (core.lua:121-180) -> last_segments[mode] (:525)
//   fav_vr_finish_frame : blend_other_sides (:454-509) -> prev_last_segments, median filter (utils.lua:151-159), equirectangular
//                         image and cube-map strip (:527-557)
// Faces are planar fp32 [3][hplus][wplus] on the device; six faces per frame in processing order (file ids 6,1,2,5,3,4 =
// modes 0..5; the caller maps ids to files).  The maps are built on the host in double with exactly the operation order of
// the Lua source (Lua numbers are doubles; this unit is compiled for the host without FMA contraction).
#include <cmath>
#include <cstring>
#include <vector>

#include "fav_internal.h"

using namespace fav;

namespace {

inline long trunc_index(double v) { return (long)v; }       // luaL_checklong [recalled]: a fractional tensor index truncates

void map_width(double height, double crop, double* width, double* ov)
{
    const double oversize = crop / 2;                                            // vr_helper.lua:4-5
    double w = height / 2 / ((2 * oversize + height) / height);                 // :6
    const double mrf = (w + oversize) / w;                                       // :7
    w = w - (mrf - 1) / mrf * oversize;                                          // :8
    *width = w; *ov = oversize;
}

// kind 0 left, 1 right, 2 top, 3 bottom.  out: [2][H][W] (dy, dx), 99999 where the map is undefined
void perspective_map(int kind, int H, int W, int crop, std::vector<float>& out)
{
    std::vector<double> m((size_t)2 * H * W, 99999.0);
    auto at = [&](int pl, long y, long x) -> double& { return m[((size_t)pl * H + y) * W + x]; };
    double width, ov;
    if (kind == 0) {                                                              // vr_helper.lua:3-25
        map_width(H, crop, &width, &ov);
        const double mid_y = (double)H / 2;
        for (double x = width - crop + 1; x <= width; x = x + 1) {
            const double rh = (x + ov) / width, rw = (x + ov) / width;
            const long col = trunc_index(x - (width - crop) + W - crop) - 1;
            if (col < 0 || col >= W) continue;
            for (int y = 1; y <= H; ++y) {
                at(0, y - 1, col) = (mid_y - y) * (-1 / rh + 1);
                at(1, y - 1, col) = (width - x - ov) * (rw - 1) / rw - W + crop;
            }
        }
    } else if (kind == 1) {                                                       // :27-48
        map_width(H, crop, &width, &ov);
        const double mid_y = (double)H / 2;
        for (int x = 1; x <= crop; ++x) {
            const double rh = (width - x + ov) / width, rw = (width - x + ov) / width;
            for (int y = 1; y <= H; ++y) {
                at(0, y - 1, x - 1) = (mid_y - y) * (-1 / rh + 1);
                at(1, y - 1, x - 1) = -(x - ov) * (rw - 1) / rw + W - crop;
            }
        }
    } else if (kind == 2) {                                                       // :50-72 (height = map extent along y)
        map_width(W, crop, &width, &ov);                                          // "height" in the Lua source
        const double height = width, mid_x = (double)W / 2;
        for (double y = height - crop + 1; y <= height; y = y + 1) {
            const double rw = (y + ov) / height, rh = (y + ov) / height;
            const long row = trunc_index(y - (height - crop) + H - crop) - 1;
            if (row < 0 || row >= H) continue;
            for (int x = 1; x <= W; ++x) {
                at(0, row, x - 1) = (height - y - ov) * (rh - 1) / rh - H + crop;
                at(1, row, x - 1) = (mid_x - x) * (-1 / rw + 1);
            }
        }
    } else {                                                                      // :75-96
        map_width(W, crop, &width, &ov);
        const double height = width, mid_x = (double)W / 2;
        for (int y = 1; y <= crop; ++y) {
            const double rw = (height - y + ov) / height, rh = (height - y + ov) / height;
            for (int x = 1; x <= W; ++x) {
                at(0, y - 1, x - 1) = -(y - ov) * (rh - 1) / rh + H - crop;
                at(1, y - 1, x - 1) = (mid_x - x) * (-1 / rw + 1);
            }
        }
    }
    out.resize(m.size());
    for (size_t i = 0; i < m.size(); ++i) out[i] = (float)m[i];
}

// vr_helper.lua:99-184.  out: [2][out_h][out_w] offsets (dy, dx) into the strip f,l,r,b,u,d
void equirect_map(double w_plus, double h_plus, double overlap_w, double overlap_h, int out_w, int out_h, std::vector<float>& out)
{
    out.resize((size_t)2 * out_h * out_w);
    const double cw = w_plus - overlap_w, ch = h_plus - overlap_h;
    const double pi = 3.14159265358979323846;        // math.pi
    for (int j = 0; j < out_h; ++j) {
        const double v = 1 - ((double)j / out_h), theta = v * pi;
        for (int i = 0; i < out_w; ++i) {
            const double u = (double)i / out_w, phi = u * 2 * pi;
            const double x = std::sin(phi) * std::sin(theta) * -1, y = std::cos(theta), z = std::cos(phi) * std::sin(theta) * -1;
            const double a = std::fmax(std::fmax(std::fabs(x), std::fabs(y)), std::fabs(z));
            const double xa = x / a, ya = y / a, za = z / a;
            double xp, yp, xo;
            if (xa == 1)       { xp = (((za + 1) / 2) - 1) * cw; xo = 2 * w_plus; yp = ((ya + 1) / 2) * ch; }
            else if (xa == -1) { xp = ((za + 1) / 2) * cw;       xo = 1 * w_plus; yp = ((ya + 1) / 2) * ch; }
            else if (ya == 1)  { xp = ((xa + 1) / 2) * cw;       xo = 5 * w_plus; yp = (((za + 1) / 2) - 1) * ch; }
            else if (ya == -1) { xp = ((xa + 1) / 2) * cw;       xo = 4 * w_plus; yp = ((za + 1) / 2) * ch; }
            else if (za == 1)  { xp = ((xa + 1) / 2) * cw;       xo = 0 * w_plus; yp = ((ya + 1) / 2) * ch; }
            else if (za == -1) { xp = (((xa + 1) / 2) - 1) * cw; xo = 3 * w_plus; yp = ((ya + 1) / 2) * ch; }
            else { xp = 0; yp = 0; xo = 0; }
            xp = std::fabs(xp); yp = std::fabs(yp);
            xp = xp + xo + overlap_w / 2;
            yp = yp + 0 + overlap_h / 2;
            out[((size_t)0 * out_h + j) * out_w + i] = (float)(yp - j);
            out[((size_t)1 * out_h + j) * out_w + i] = (float)(xp - i);
        }
    }
}

// utils.lua make_gradient_mask_{w,h}_{inc,dec}: i/(n+1) ramps (doubles), embedded in an H x W plane of zeros
void gradient_masks(int H, int W, int gw, int gh, std::vector<double> g[4])
{
    for (int k = 0; k < 4; ++k) g[k].assign((size_t)H * W, 0.0);
    for (int y = 0; y < H; ++y) {
        for (int i = 1; i <= gw; ++i) {
            g[0][(size_t)y * W + (i - 1)] = (double)(gw + 1 - i) / (gw + 1);            // left: decreasing from the edge
            g[1][(size_t)y * W + (W - gw + i - 1)] = (double)i / (gw + 1);              // right: increasing to the edge
        }
    }
    for (int x = 0; x < W; ++x) {
        for (int i = 1; i <= gh; ++i) {
            g[2][(size_t)(i - 1) * W + x] = (double)(gh + 1 - i) / (gh + 1);             // top
            g[3][(size_t)(H - gh + i - 1) * W + x] = (double)i / (gh + 1);               // bottom
        }
    }
}

int upload(const std::vector<float>& h, float** d)
{
    FAV_HIP(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(float)));
    FAV_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return FAV_OK;
}

}  // namespace

struct fav_vr {
    fav_net* vid = nullptr; fav_net* img = nullptr;
    int H = 0, W = 0;                     // hplus, wplus
    fav_vr_opts o{};
    size_t n = 0;
    float* map[4] = {nullptr, nullptr, nullptr, nullptr};      // left, right, top, bottom: [2][H][W]
    float* mask[4] = {nullptr, nullptr, nullptr, nullptr};     // warp(ones, map): [H][W]
    float* mask_all = nullptr; float* mask_all_div = nullptr;
    float* grad[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // left, right, top, bottom, left_right, all (fp32)
    float* anti_all = nullptr;            // fp32(1 - grad_all)
    float* equi_map = nullptr;
    float* last[6] = {}; float* prev[6] = {}; float* filt[6] = {};
    bool have_last[6] = {}; bool have_prev = false;
    float* tmp_rot = nullptr; float* tmp_warp = nullptr; float* border = nullptr; float* lfw = nullptr; float* prior = nullptr;
    float* flow_lua = nullptr; float* cert_tmp = nullptr; float* cert = nullptr; float* in8 = nullptr;
    float* strip = nullptr; float* equi = nullptr; float* cube = nullptr;
    int FH = 0, FW = 0;                   // filtered face size
    int cube_h = 0, cube_w = 0;
    ~fav_vr()
    {
        if (vid) (void)hipSetDevice(net_device(vid));
        for (auto p : map) (void)hipFree(p);
        for (auto p : mask) (void)hipFree(p);
        for (auto p : grad) (void)hipFree(p);
        for (auto p : last) (void)hipFree(p);
        for (auto p : prev) (void)hipFree(p);
        for (auto p : filt) (void)hipFree(p);
        float* rest[] = {mask_all, mask_all_div, anti_all, equi_map, tmp_rot, tmp_warp, border, lfw, prior, flow_lua, cert_tmp, cert,
                         in8, strip, equi, cube};
        for (auto p : rest) (void)hipFree(p);
    }
};

extern "C" int fav_vr_map_host(int kind, int hplus, int wplus, int overlap, int median_filter, int out_w, int out_h, float* out_host)
{
    FAV_REQUIRE(out_host && hplus > 0 && wplus > 0, "fav_vr_map_host: bad argument");
    std::vector<float> m;
    if (kind >= 0 && kind <= 3) {
        FAV_REQUIRE(overlap > 0 && overlap < (kind < 2 ? wplus : hplus), "fav_vr_map_host: overlap %d does not fit", overlap);
        perspective_map(kind, hplus, wplus, overlap, m);
    } else if (kind == 4) {
        FAV_REQUIRE(out_w > 0 && out_h > 0, "fav_vr_map_host: equirectangular size");
        const int r = median_filter / 2;                        // fast_artistic_video_vr.lua:191-192 (argument order as there)
        equirect_map(hplus - 2 * r, wplus - 2 * r, overlap - r, overlap - r, out_w, out_h, m);
    } else { set_error("fav_vr_map_host: kind %d", kind); return FAV_EINVAL; }
    std::memcpy(out_host, m.data(), m.size() * sizeof(float));
    return FAV_OK;
}

extern "C" int fav_vr_create(fav_net* video_net, fav_net* image_net_or_null, int hplus, int wplus, const fav_vr_opts* opts,
                             fav_vr** out)
{
    FAV_REQUIRE(video_net && opts && out && hplus > 0 && wplus > 0, "fav_vr_create: bad argument");
    const fav_vr_opts& o = *opts;
    FAV_REQUIRE(o.overlap_w > 10 && o.overlap_h > 10 && o.overlap_w < wplus && o.overlap_h < hplus && !(o.overlap_w & 1) && !(o.overlap_h & 1),
                "fav_vr_create: overlap %dx%d must be even, > 10 (gradient width = overlap - 10) and smaller than the face", o.overlap_w, o.overlap_h);
    FAV_REQUIRE(o.median_filter == 0 || (o.median_filter >= 3 && o.median_filter <= 5 && (o.median_filter & 1)), "fav_vr_create: -median_filter %d (0, 3 or 5)", o.median_filter);
    FAV_REQUIRE(net_in_channels(video_net) == 7, "fav_vr_create: the video model must take 7 input channels");
    int Ho, Wo; net_out_size(video_net, hplus, wplus, &Ho, &Wo);
    FAV_REQUIRE(Ho == hplus && Wo == wplus, "fav_vr_create: a %dx%d face gives a %dx%d output (sizes must be multiples of 4)", wplus, hplus, Wo, Ho);
    if (image_net_or_null) {
        FAV_REQUIRE(net_device(image_net_or_null) == net_device(video_net) && net_pad(image_net_or_null) == net_pad(video_net),
                    "fav_vr_create: image model on another device or with another reflection padding");
        net_out_size(image_net_or_null, hplus, wplus, &Ho, &Wo);
        FAV_REQUIRE(Ho == hplus && Wo == wplus, "fav_vr_create: the image model changes the face size");
    }
    FAV_HIP(hipSetDevice(net_device(video_net)));
    fav_vr* v = new fav_vr();
    v->vid = video_net; v->img = image_net_or_null; v->H = hplus; v->W = wplus; v->o = o;
    if (v->o.occlusions_min_filter < 1) v->o.occlusions_min_filter = 1;
    const size_t n = (size_t)hplus * wplus; v->n = n;
    const int pad = net_pad(video_net);
    auto fail = [&](int rc) { delete v; return rc; };
    auto dmalloc = [&](float** p, size_t floats) { return hipMalloc(reinterpret_cast<void**>(p), floats * sizeof(float)) == hipSuccess; };
    // ---- static maps and masks (fast_artistic_video_vr.lua:164-198)
    const int crop[4] = {o.overlap_w, o.overlap_w, o.overlap_h, o.overlap_h};
    std::vector<float> ones(n, 1.f); float* d_ones = nullptr;
    int rc = upload(ones, &d_ones); if (rc) return fail(rc);
    for (int k = 0; k < 4; ++k) {
        std::vector<float> m; perspective_map(k, hplus, wplus, crop[k], m);
        rc = upload(m, &v->map[k]); if (rc) { (void)hipFree(d_ones); return fail(rc); }
        if (!dmalloc(&v->mask[k], n)) { (void)hipFree(d_ones); return fail(hip_fail(hipErrorOutOfMemory, "vr masks")); }
        rc = launch_warp(d_ones, v->map[k], v->mask[k], 1, 1, hplus, wplus, hplus, wplus, o.border_mode, nullptr);
        if (rc) { (void)hipFree(d_ones); return fail(rc); }
    }
    FAV_HIP(hipDeviceSynchronize());
    (void)hipFree(d_ones);
    {   // mask_all_div = max(l + r + t + b, 1), mask_all = min(.., 1) (:176-177) -- tiny, done on the host in fp32
        std::vector<float> mk[4], div(n), all(n);
        for (int k = 0; k < 4; ++k) { mk[k].resize(n); FAV_HIP(hipMemcpy(mk[k].data(), v->mask[k], n * 4, hipMemcpyDeviceToHost)); }
        for (size_t i = 0; i < n; ++i) {
            volatile float s = mk[0][i] + mk[1][i]; s = s + mk[2][i]; s = s + mk[3][i];      // left + right + top + bottom
            div[i] = s > 1.f ? (float)s : 1.f; all[i] = s < 1.f ? (float)s : 1.f;
        }
        rc = upload(div, &v->mask_all_div); if (rc) return fail(rc);
        rc = upload(all, &v->mask_all); if (rc) return fail(rc);
    }
    {   // gradient masks (:179-187), doubles cast to fp32 when used (:type(dtype))
        std::vector<double> g[4];
        gradient_masks(hplus, wplus, o.overlap_w - 10, o.overlap_h - 10, g);
        std::vector<float> f(n), anti(n);
        for (int k = 0; k < 4; ++k) { for (size_t i = 0; i < n; ++i) f[i] = (float)g[k][i]; rc = upload(f, &v->grad[k]); if (rc) return fail(rc); }
        for (size_t i = 0; i < n; ++i) f[i] = (float)std::fmax(g[0][i], g[1][i]);
        rc = upload(f, &v->grad[4]); if (rc) return fail(rc);
        for (size_t i = 0; i < n; ++i) {
            const double a = std::fmax(std::fmax(g[0][i], g[1][i]), std::fmax(g[2][i], g[3][i]));
            f[i] = (float)a; anti[i] = (float)(1.0 - a);                        // :456: csub in double, then :type(dtype)
        }
        rc = upload(f, &v->grad[5]); if (rc) return fail(rc);
        rc = upload(anti, &v->anti_all); if (rc) return fail(rc);
    }
    const int r = o.median_filter / 2;
    v->FH = hplus - 2 * r; v->FW = wplus - 2 * r;
    const int ovw = o.overlap_w / 2 - r, ovh = o.overlap_h / 2 - r;               // :515-516
    const int ch = hplus - 2 * ovh, cw = wplus - 2 * ovw;                          // crop {ov+1, plus-ov}
    if (ovw < 0 || ovh < 0 || hplus - ovh > v->FH || wplus - ovw > v->FW) { delete v; set_error("fav_vr_create: overlap/2 must be >= 3*floor(median/2) for the cube-map crop"); return FAV_EINVAL; }
    v->cube_h = (ch == cw) ? ch : 0; v->cube_w = (ch == cw) ? 6 * cw : 0;       // rotated top/bottom faces need square crops
    bool ok = true;
    for (int k = 0; k < 6; ++k) ok = ok && dmalloc(&v->last[k], 3 * n) && dmalloc(&v->prev[k], 3 * n) && dmalloc(&v->filt[k], 3 * (size_t)v->FH * v->FW);
    ok = ok && dmalloc(&v->tmp_rot, 3 * n) && dmalloc(&v->tmp_warp, 3 * n) && dmalloc(&v->border, 3 * n) && dmalloc(&v->lfw, 3 * n) &&
         dmalloc(&v->prior, 3 * n) && dmalloc(&v->flow_lua, 2 * n) && dmalloc(&v->cert_tmp, n) && dmalloc(&v->cert, n) &&
         dmalloc(&v->in8, (size_t)(hplus + 2 * pad) * (wplus + 2 * pad) * 8);
    if (v->cube_w) ok = ok && dmalloc(&v->cube, 3 * (size_t)v->cube_h * v->cube_w);
    if (o.out_equi_w > 0 && o.out_equi_h > 0) {
        std::vector<float> m;
        equirect_map(hplus - 2 * r, wplus - 2 * r, o.overlap_w - r, o.overlap_h - r, o.out_equi_w, o.out_equi_h, m);   // :191-192
        rc = upload(m, &v->equi_map); if (rc) return fail(rc);
        ok = ok && dmalloc(&v->strip, 3 * (size_t)v->FH * 6 * v->FW) && dmalloc(&v->equi, 3 * (size_t)o.out_equi_w * o.out_equi_h);
    }
    if (!ok) return fail(hip_fail(hipErrorOutOfMemory, "hipMalloc(vr buffers)"));
    *out = v;
    return FAV_OK;
}

extern "C" void fav_vr_destroy(fav_vr* v) { delete v; }

// border += warp(rot(src), map) [/ div]
static int add_border(fav_vr* v, const float* src, int rot, int map_k, bool divide, bool first, hipStream_t st)
{
    const float* s = src;
    if (rot) {
        FAV_REQUIRE(v->H == v->W, "cube-map faces must be square for the rotated neighbours");
        int rc = launch_vr_rotate(src, v->tmp_rot, v->H, v->W, rot, st); if (rc) return rc;
        s = v->tmp_rot;
    }
    int rc = launch_warp(s, v->map[map_k], v->tmp_warp, 1, 3, v->H, v->W, v->H, v->W, v->o.border_mode, st); if (rc) return rc;
    return launch_vr_accum(v->border, v->tmp_warp, divide ? v->mask_all_div : nullptr, v->n, first ? 1 : 0, st);
}

extern "C" int fav_vr_face(fav_vr* v, int i, const uint8_t* frame_rgb_hwc, const float* backward_flo, const uint8_t* cert_pgm,
                           float* out_rgb_f32, fav_hipstream_t stream)
{
    FAV_REQUIRE(v && frame_rgb_hwc && i >= 1, "fav_vr_face: bad argument");
    FAV_HIP(hipSetDevice(net_device(v->vid)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mode = (i - 1) % 6, H = v->H, W = v->W, pad = net_pad(v->vid);
    const size_t n = v->n;
    const bool single = v->o.create_inconsistent ? (i % 6 == 1) : (i == 1);      // fast_artistic_video_vr.lua:304-310
    const bool temporal = i >= 7 && !v->o.create_inconsistent;
    enum { L = 0, R = 1, T = 2, B = 3 };
    int rc;
    if (single) {
        rc = launch_vr_prep(frame_rgb_hwc, nullptr, nullptr, v->img ? 0 : v->o.fill_random, v->o.seed, (unsigned)i, H, W, pad, v->in8, st);
        if (rc) return rc;
        rc = net_forward_padded(v->img ? v->img : v->vid, v->in8, H, W, v->last[mode], st); if (rc) return rc;
    } else {
        if (temporal) FAV_REQUIRE(backward_flo && cert_pgm && v->have_prev, "fav_vr_face: face %d needs the flow, the certainty and a finished previous frame", i);
        for (int k = 0; k < mode && !v->o.create_inconsistent_border; ++k)
            FAV_REQUIRE(v->have_last[k], "fav_vr_face: faces must be processed in order (face %d of this frame is missing)", k);
        // ---- func_load_cert (:204-237) + min filter (core.lua:207)
        const bool bd = !v->o.create_inconsistent_border;
        const float* ml = bd && (mode == 1 || mode >= 3) ? v->mask[L] : nullptr;
        const float* mr = bd && (mode >= 2) ? v->mask[R] : nullptr;
        const float* mt = bd && (mode >= 4) ? v->mask[T] : nullptr;
        const float* mb = bd && (mode >= 4) ? v->mask[B] : nullptr;
        rc = launch_vr_cert(temporal ? cert_pgm : nullptr, ml, mr, mt, mb, v->cert_tmp, n, st); if (rc) return rc;
        rc = launch_min_filter_f32(v->cert_tmp, v->cert, H, W, v->o.occlusions_min_filter, st); if (rc) return rc;
        // ---- func_make_last_frame_warped (:239-302)
        bool have_border = false;
        if (bd) {
            float** S = v->last;
            if (mode == 1)      { rc = add_border(v, S[0], 0, L, false, true, st); }
            else if (mode == 2) { rc = add_border(v, S[0], 0, R, false, true, st); }
            else if (mode == 3) { rc = add_border(v, S[1], 0, L, false, true, st); if (!rc) rc = add_border(v, S[2], 0, R, false, false, st); }
            else if (mode == 4) {
                rc = add_border(v, S[1], 1, L, true, true, st);
                if (!rc) rc = add_border(v, S[2], 2, R, true, false, st);
                if (!rc) rc = add_border(v, S[3], 0, T, true, false, st);
                if (!rc) rc = add_border(v, S[0], 3, B, true, false, st);
            } else if (mode == 5) {
                rc = add_border(v, S[1], 2, L, true, true, st);
                if (!rc) rc = add_border(v, S[2], 1, R, true, false, st);
                if (!rc) rc = add_border(v, S[0], 3, T, true, false, st);
                if (!rc) rc = add_border(v, S[3], 0, B, true, false, st);
            }
            if (rc) return rc;
            have_border = mode >= 1;
        }
        if (!have_border) FAV_HIP(hipMemsetAsync(v->border, 0, 3 * n * sizeof(float), st));
        const float* prior = v->border;
        if (temporal) {
            rc = launch_vr_flo_to_lua(backward_flo, v->flow_lua, n, st); if (rc) return rc;
            rc = launch_warp(v->prev[mode], v->flow_lua, v->lfw, 1, 3, H, W, H, W, v->o.border_mode, st); if (rc) return rc;
            if (mode == 0) prior = v->lfw;
            else {
                // grad_masks = {right, left, left_right, all, all}; masks = {left, right, left + right, all, all}   (:286-287)
                const float* g = mode == 1 ? v->grad[1] : mode == 2 ? v->grad[0] : mode == 3 ? v->grad[4] : v->grad[5];
                const float* m = mode == 1 ? v->mask[L] : mode == 2 ? v->mask[R] : mode == 3 ? v->mask[L] : v->mask_all;
                const float* m2 = mode == 3 ? v->mask[R] : nullptr;
                rc = launch_vr_prior(v->lfw, v->border, g, v->cert, m, m2, v->prior, n, st); if (rc) return rc;
                prior = v->prior;
            }
        }
        // ---- run_next_image (core.lua:161-180)
        rc = launch_vr_prep(frame_rgb_hwc, prior, v->cert, v->o.fill_random, v->o.seed, (unsigned)i, H, W, pad, v->in8, st); if (rc) return rc;
        rc = net_forward_padded(v->vid, v->in8, H, W, v->last[mode], st); if (rc) return rc;
    }
    v->have_last[mode] = true;
    if (out_rgb_f32) FAV_HIP(hipMemcpyAsync(out_rgb_f32, v->last[mode], 3 * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FAV_OK;
}

extern "C" int fav_vr_finish_frame(fav_vr* v, uint8_t* equi_rgb8_hwc, uint8_t* cubemap_rgb8_hwc, fav_hipstream_t stream)
{
    FAV_REQUIRE(v, "fav_vr_finish_frame: null handle");
    for (int k = 0; k < 6; ++k) FAV_REQUIRE(v->have_last[k], "fav_vr_finish_frame: face %d of this frame is missing", k);
    FAV_HIP(hipSetDevice(net_device(v->vid)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    enum { L = 0, R = 1, T = 2, B = 3 };
    float** S = v->last;
    // blend_other_sides (:454-509): {source face, rotation, map} x 4 per face, summed as combineSides does
    struct Src { int face, rot, map; };
    static const Src plan[6][4] = {
        {{1, 0, R}, {2, 0, L}, {4, 3, B}, {5, 3, T}},
        {{0, 0, L}, {3, 0, R}, {4, 2, B}, {5, 1, T}},
        {{0, 0, R}, {3, 0, L}, {4, 1, B}, {5, 2, T}},
        {{1, 0, L}, {2, 0, R}, {4, 0, B}, {5, 0, T}},
        {{0, 3, B}, {1, 1, L}, {2, 2, R}, {3, 0, T}},
        {{0, 3, T}, {1, 2, L}, {2, 1, R}, {3, 0, B}},
    };
    int rc;
    for (int f = 0; f < 6; ++f) {
        for (int q = 0; q < 4; ++q) { rc = add_border(v, S[plan[f][q].face], plan[f][q].rot, plan[f][q].map, true, q == 0, st); if (rc) return rc; }
        rc = launch_vr_blend(S[f], v->border, v->grad[5], v->anti_all, v->prev[f], v->n, st); if (rc) return rc;
    }
    v->have_prev = true;
    for (int k = 0; k < 6; ++k) v->have_last[k] = false;
    // :529-536 median filter of every blended face
    for (int f = 0; f < 6; ++f) {
        if (v->o.median_filter > 0) { rc = launch_vr_median(v->prev[f], v->filt[f], v->H, v->W, v->o.median_filter, st); if (rc) return rc; }
        else FAV_HIP(hipMemcpyAsync(v->filt[f], v->prev[f], 3 * v->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    const int FH = v->FH, FW = v->FW;
    if (equi_rgb8_hwc) {                                                         // :538-541
        FAV_REQUIRE(v->equi_map, "fav_vr_finish_frame: created without an equirectangular output size");
        const int rot[6] = {0, 0, 0, 0, 3, 3};
        for (int f = 0; f < 6; ++f) { rc = launch_vr_strip(v->filt[f], FH, FW, 0, 0, FH, FW, rot[f], v->strip, FH, 6 * FW, f * FW, st); if (rc) return rc; }
        rc = launch_warp(v->strip, v->equi_map, v->equi, 1, 3, FH, 6 * FW, v->o.out_equi_h, v->o.out_equi_w, v->o.border_mode, st); if (rc) return rc;
        rc = launch_quantize_rgb8(v->equi, equi_rgb8_hwc, v->o.out_equi_h, v->o.out_equi_w, st); if (rc) return rc;
    }
    if (cubemap_rgb8_hwc) {                                                      // :542-552: faces 4, 1, rot90(5), rotMinus90(6), 3, 2
        FAV_REQUIRE(v->cube, "fav_vr_finish_frame: the cube-map strip needs square cropped faces");
        const int r = v->o.median_filter / 2, ovw = v->o.overlap_w / 2 - r, ovh = v->o.overlap_h / 2 - r;
        const int ch = v->H - 2 * ovh, cw = v->W - 2 * ovw;
        const int order[6] = {3, 0, 4, 5, 2, 1}, rot[6] = {0, 0, 1, 2, 0, 0};
        for (int q = 0; q < 6; ++q) { rc = launch_vr_strip(v->filt[order[q]], FH, FW, ovh, ovw, ch, cw, rot[q], v->cube, v->cube_h, v->cube_w, q * cw, st); if (rc) return rc; }
        rc = launch_quantize_rgb8(v->cube, cubemap_rgb8_hwc, v->cube_h, v->cube_w, st); if (rc) return rc;
    }
    return FAV_OK;
}

extern "C" int fav_vr_output_sizes(const fav_vr* v, int* equi_w, int* equi_h, int* cube_w, int* cube_h, int* filt_w, int* filt_h)
{
    FAV_REQUIRE(v, "fav_vr_output_sizes: null handle");
    if (equi_w) *equi_w = v->equi_map ? v->o.out_equi_w : 0;
    if (equi_h) *equi_h = v->equi_map ? v->o.out_equi_h : 0;
    if (cube_w) *cube_w = v->cube_w;
    if (cube_h) *cube_h = v->cube_h;
    if (filt_w) *filt_w = v->FW;
    if (filt_h) *filt_h = v->FH;
    return FAV_OK;
}

// which: 0 last_segments[k] (this frame's raw faces), 1 prev_last_segments[k] (blended faces of the finished frame),
//        2 median-filtered faces ([3][filt_h][filt_w]), 3 equirectangular float image, 4 cube-map float strip
extern "C" int fav_vr_get_f32(const fav_vr* v, int which, int k, float* out_dev, fav_hipstream_t stream)
{
    FAV_REQUIRE(v && out_dev && k >= 0 && k < 6, "fav_vr_get_f32: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* src = nullptr; size_t cnt = 0;
    if (which == 0) { src = v->last[k]; cnt = 3 * v->n; }
    else if (which == 1) { src = v->prev[k]; cnt = 3 * v->n; }
    else if (which == 2) { src = v->filt[k]; cnt = 3 * (size_t)v->FH * v->FW; }
    else if (which == 3) { src = v->equi; cnt = v->equi ? 3 * (size_t)v->o.out_equi_w * v->o.out_equi_h : 0; }
    else if (which == 4) { src = v->cube; cnt = v->cube ? 3 * (size_t)v->cube_h * v->cube_w : 0; }
    FAV_REQUIRE(src && cnt, "fav_vr_get_f32: nothing of kind %d", which);
    FAV_HIP(hipMemcpyAsync(out_dev, src, cnt * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FAV_OK;
}
