// fav_stylize_vr -- drop-in for `th fast_artistic_video_vr.lua ...` (fast_artistic_video_vr.lua:21-76,561-591) on MI355X:
// stylises a 360-degree video given as six overlapping cube faces per frame.  Same single-dash flags, same file patterns
// (-input_pattern with two integers: frame, face id; flow / occlusion patterns with {..} = frame-1, [..] = frame and a
// remaining %d = face id, :105-115), faces processed in the order of ids {6,1,2,5,3,4} (:103), outputs
// "<prefix>-%05d_equi.png" / "<prefix>-%05d_cubemap.png" (:541,552).  All compute goes through libfav's C ABI (fav_vr_*);
// there is no CPU backend.  Not provided (rejected with a message): -evaluate, -backward, -smooth_certainty and
// -continue_with > 1 (in the reference that option reloads per-face PNGs which func_save_image no longer writes, :521-523).
// Additive flags: -precision <fp32|bf16>, -warp_border <stn|cpu>, -poll_timeout <sec>, -png_level <0..9>, -seed <n> (uniform-random fill), -timing <0|1>.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <string>
#include <vector>

#include "../../include/fav.h"

namespace {

[[noreturn]] void die(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); exit(1); }
void check(int rc, const char* what) { if (rc) die(std::string(what) + ": " + fav_last_error()); }
void hipc(hipError_t e, const char* what) { if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e)); }
bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

std::string fmt1(const std::string& pattern, int a) { char b[4096]; snprintf(b, sizeof b, pattern.c_str(), a); return b; }
std::string fmt2(const std::string& pattern, int a, int c) { char b[4096]; snprintf(b, sizeof b, pattern.c_str(), a, c); return b; }

// getFormatedFlowFileName (fast_artistic_video_vr.lua:105-115): {fmt} <- from, [fmt] <- to, then the remaining %d <- face id
std::string flow_name(const std::string& pattern, int from, int to, int face)
{
    std::string out;
    for (size_t p = 0; p < pattern.size();) {
        const char c = pattern[p];
        if (c == '{' || c == '[') {
            const size_t e = pattern.find(c == '{' ? '}' : ']', p + 1);
            if (e != std::string::npos) { out += fmt1(pattern.substr(p + 1, e - p - 1), c == '{' ? from : to); p = e + 1; continue; }
        }
        out += c; ++p;
    }
    return fmt1(out, face);
}

void wait_for_file(const std::string& path, double timeout_s)      // utils.lua:74-80, bounded
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    bool announced = false; long long last = -1;
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && st.st_size > 0) { if ((long long)st.st_size == last) return; last = (long long)st.st_size; }
        else if (!announced) { printf("Waiting for file \"%s\"\n", path.c_str()); fflush(stdout); announced = true; }
        if (std::chrono::duration<double>(clk::now() - t0).count() > timeout_s) die("timed out waiting for " + path);
        usleep(last >= 0 && !announced ? 2000 : 50000);
    }
}

void mkdirs_for(const std::string& path)
{
    for (size_t p = 1; p < path.size(); ++p) if (path[p] == '/') mkdir(path.substr(0, p).c_str(), 0777);
}

}  // namespace

int main(int argc, char** argv)
{
    std::map<std::string, std::string> v = {
        {"input_pattern", ""}, {"flow_pattern", ""}, {"occlusions_pattern", ""}, {"model_img", ""}, {"model_vid", ""},
        {"start_frame", "1"}, {"continue_with", "1"}, {"num_frames", "9999"}, {"occlusions_min_filter", "7"},
        {"fill_occlusions", "vgg-mean"}, {"overlap_pixel_w", "20"}, {"overlap_pixel_h", "20"}, {"output_prefix", "out"},
        {"out_equi_w", "768"}, {"out_equi_h", "768"}, {"median_filter", "3"}, {"gpu", "-1"}, {"backend", "cuda"}, {"use_cudnn", "1"},
        {"cudnn_benchmark", "0"}, {"evaluation_file", "evaluation.txt"}, {"flow_pattern_eval", ""}, {"occlusions_pattern_eval", ""},
        {"content_weights", "1.0"}, {"content_layers", "16"}, {"loss_network", "models/vgg16.t7"}, {"style_image", ""},
        {"style_image_size", "256"}, {"style_weights", "5.0"}, {"style_layers", "4,9,16,23"}, {"style_target_type", "gram"},
        {"warp_border", "stn"}, {"poll_timeout", "600"}, {"png_level", "1"}, {"seed", "1"}, {"timing", "0"}, {"precision", "fp32"}};
    std::map<std::string, bool> b = {
        {"invert_occlusions", false}, {"fix_occlusions", false}, {"smooth_certainty", false}, {"create_inconsistent", false},
        {"create_inconsistent_border", false}, {"backward", false}, {"out_equi", false}, {"out_cubemap", false}, {"evaluate", false},
        {"no_consistency_eval", false}, {"invert_occlusions_eval", false}, {"backward_eval", false}, {"fix_occlusions_eval", false}};
    for (int a = 1; a < argc; ++a) {                       // torch.CmdLine: -flag value | -boolflag
        std::string k = argv[a];
        if (k.size() < 2 || k[0] != '-') die("unknown argument " + k);
        k = k.substr(1);
        if (b.count(k)) { b[k] = true; continue; }
        if (!v.count(k)) die("unknown option -" + k);
        if (a + 1 >= argc) die("missing value for -" + k);
        v[k] = argv[++a];
    }
    auto I = [&](const char* k) { return atoi(v[k].c_str()); };
    if (v["input_pattern"].empty()) die("Must give -input_pattern");                                               // :564-566
    if (!b["create_inconsistent"] && (v["flow_pattern"].empty() || v["occlusions_pattern"].empty()))
        die("Must give -flow_pattern and -occlusions_pattern");                                                     // :567-569
    if (I("gpu") < 0) die("-gpu -1: this build has no CPU backend (the CPU restatement under oracle/ is test infrastructure); pass -gpu <id>");
    if (b["evaluate"]) die("-evaluate (perceptual / edge losses) is outside the hot-path scope");
    if (b["backward"]) die("-backward is not provided for the cube-map pipeline");
    if (b["smooth_certainty"]) die("-smooth_certainty is not provided");
    if (I("continue_with") != 1) die("-continue_with > 1 reloads per-face PNGs that the reference no longer writes (fast_artistic_video_vr.lua:521-523); not provided");
    if (v["model_vid"].empty()) die("Must give -model_vid");
    if (v["fill_occlusions"] != "vgg-mean" && v["fill_occlusions"] != "uniform-random") die("-fill_occlusions must be vgg-mean or uniform-random");
    const bool timing = I("timing") != 0;
    static const int proc_order[6] = {6, 1, 2, 5, 3, 4};                                                           // :103

    hipc(hipSetDevice(I("gpu")), "hipSetDevice");
    fav_net* vid = nullptr; fav_net* img = nullptr;
    if (fav_net_create(v["model_vid"].c_str(), I("gpu"), &vid)) die(std::string("ERROR: Could not load model from ") + v["model_vid"] + " (" + fav_last_error() + ")");
    if (v["precision"] != "fp32" && v["precision"] != "bf16") die("-precision must be fp32 or bf16");
    check(fav_net_set_precision(vid, v["precision"] == "bf16" ? FAV_PRECISION_BF16_OPERANDS : FAV_PRECISION_FP32), "fav_net_set_precision");
    if (!v["model_img"].empty() && v["model_img"] != "self")
        if (fav_net_create(v["model_img"].c_str(), I("gpu"), &img)) die(std::string("ERROR: Could not load model from ") + v["model_img"] + " (" + fav_last_error() + ")");

    hipStream_t st; hipc(hipStreamCreate(&st), "hipStreamCreate");
    fav_vr* vr = nullptr;
    int W = 0, H = 0, ew = 0, eh = 0, cw = 0, ch = 0;
    uint8_t *d_frame = nullptr, *d_cert = nullptr, *d_equi = nullptr, *d_cube = nullptr; float* d_flow = nullptr;
    std::vector<uint8_t> h_equi, h_cube;
    std::future<void> writer;
    const int start = I("start_frame"), total = I("num_frames") * 6;                                              // :571
    const auto t_all = std::chrono::steady_clock::now();
    int frames_done = 0;
    for (int i = 1; i <= total; ++i) {
        const int mode = (i - 1) % 6, file_idx = (i - 1) / 6 + start, face = proc_order[mode];                     // :155-156
        const std::string img_path = fmt2(v["input_pattern"], file_idx, face);
        if (!file_exists(img_path)) break;                                                                          // :160
        uint8_t* rgb = nullptr; int w = 0, h = 0, c = 0;
        check(fav_read_pnm_host(img_path.c_str(), &rgb, &w, &h, &c), "reading the face image");
        if (c != 3) die(img_path + ": not a P6 image");
        if (!vr) {
            W = w; H = h;
            fav_vr_opts o{};
            o.overlap_w = I("overlap_pixel_w"); o.overlap_h = I("overlap_pixel_h"); o.occlusions_min_filter = I("occlusions_min_filter");
            o.median_filter = I("median_filter"); o.fill_random = v["fill_occlusions"] == "uniform-random"; o.seed = (unsigned)I("seed");
            o.create_inconsistent = b["create_inconsistent"]; o.create_inconsistent_border = b["create_inconsistent_border"];
            o.out_equi_w = b["out_equi"] ? I("out_equi_w") : 0; o.out_equi_h = b["out_equi"] ? I("out_equi_h") : 0;
            o.border_mode = v["warp_border"] == "cpu" ? FAV_BORDER_CPU : FAV_BORDER_STN;
            check(fav_vr_create(vid, img, H, W, &o, &vr), "fav_vr_create");
            check(fav_vr_output_sizes(vr, &ew, &eh, &cw, &ch, nullptr, nullptr), "fav_vr_output_sizes");
            if (b["out_cubemap"] && !cw) die("-out_cubemap needs square cropped faces");
            hipc(hipMalloc(reinterpret_cast<void**>(&d_frame), (size_t)W * H * 3), "hipMalloc");
            hipc(hipMalloc(reinterpret_cast<void**>(&d_cert), (size_t)W * H), "hipMalloc");
            hipc(hipMalloc(reinterpret_cast<void**>(&d_flow), (size_t)W * H * 8), "hipMalloc");
            if (ew) { hipc(hipMalloc(reinterpret_cast<void**>(&d_equi), (size_t)ew * eh * 3), "hipMalloc"); h_equi.resize((size_t)ew * eh * 3); }
            if (b["out_cubemap"]) { hipc(hipMalloc(reinterpret_cast<void**>(&d_cube), (size_t)cw * ch * 3), "hipMalloc"); h_cube.resize((size_t)cw * ch * 3); }
        } else if (w != W || h != H) die(img_path + ": face size changed");
        hipc(hipMemcpyAsync(d_frame, rgb, (size_t)W * H * 3, hipMemcpyHostToDevice, st), "H2D frame");
        const bool temporal = i >= 7 && !b["create_inconsistent"];
        float* flo = nullptr; uint8_t* cert = nullptr;
        if (temporal) {                                                                                             // :225-229, :274-278
            const std::string cp = flow_name(v["occlusions_pattern"], file_idx - 1, file_idx, face);
            const std::string fp = flow_name(v["flow_pattern"], file_idx - 1, file_idx, face);
            wait_for_file(cp, atof(v["poll_timeout"].c_str())); wait_for_file(fp, atof(v["poll_timeout"].c_str()));
            int cw_ = 0, ch_ = 0, cc = 0, fw = 0, fh = 0;
            check(fav_read_pnm_host(cp.c_str(), &cert, &cw_, &ch_, &cc), "reading the certainty");
            check(fav_read_flo_host(fp.c_str(), &flo, &fw, &fh), "reading the flow");
            if (cc != 1 || cw_ != W || ch_ != H || fw != W || fh != H) die("flow / certainty size does not match the face: " + fp);
            hipc(hipMemcpyAsync(d_cert, cert, (size_t)W * H, hipMemcpyHostToDevice, st), "H2D cert");
            hipc(hipMemcpyAsync(d_flow, flo, (size_t)W * H * 8, hipMemcpyHostToDevice, st), "H2D flow");
        }
        const auto t0 = std::chrono::steady_clock::now();
        check(fav_vr_face(vr, i, d_frame, temporal ? d_flow : nullptr, temporal ? d_cert : nullptr, nullptr, st), "fav_vr_face");
        hipc(hipStreamSynchronize(st), "sync");
        if (timing) printf("Elapsed time for stylizing face %d of frame %d: %.4f\n", face, file_idx, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        fav_free_host(rgb); fav_free_host(flo); fav_free_host(cert);
        if (mode == 5) {                                                                                            // :527-557
            if (writer.valid()) writer.get();
            check(fav_vr_finish_frame(vr, d_equi, b["out_cubemap"] ? d_cube : nullptr, st), "fav_vr_finish_frame");
            if (d_equi) hipc(hipMemcpyAsync(h_equi.data(), d_equi, h_equi.size(), hipMemcpyDeviceToHost, st), "D2H");
            if (d_cube) hipc(hipMemcpyAsync(h_cube.data(), d_cube, h_cube.size(), hipMemcpyDeviceToHost, st), "D2H");
            hipc(hipStreamSynchronize(st), "sync");
            const int out_idx = (i - 1) / 6 + 1;                                                                    // :517 (not offset by -start_frame)
            const std::string prefix = v["output_prefix"]; const int lvl = I("png_level");
            std::vector<uint8_t> e = h_equi, cb = h_cube;
            writer = std::async(std::launch::async, [=]() {
                char name[4096];
                if (!e.empty()) {
                    snprintf(name, sizeof name, "%s-%05d_equi.png", prefix.c_str(), out_idx); mkdirs_for(name);
                    if (fav_write_png_rgb8_host(name, e.data(), ew, eh, lvl)) die(std::string("writing ") + name + ": " + fav_last_error());
                }
                if (!cb.empty()) {
                    snprintf(name, sizeof name, "%s-%05d_cubemap.png", prefix.c_str(), out_idx); mkdirs_for(name);
                    if (fav_write_png_rgb8_host(name, cb.data(), cw, ch, lvl)) die(std::string("writing ") + name + ": " + fav_last_error());
                }
            });
            ++frames_done;
        }
    }
    if (writer.valid()) writer.get();
    if (timing && frames_done) {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
        printf("%d frames (x6 faces) in %.3f s: %.2f frames/s\n", frames_done, s, frames_done / s);
    }
    if (vr) fav_vr_destroy(vr);
    if (img) fav_net_destroy(img);
    fav_net_destroy(vid);
    (void)hipFree(d_frame); (void)hipFree(d_cert); (void)hipFree(d_flow); (void)hipFree(d_equi); (void)hipFree(d_cube);
    return 0;
}
