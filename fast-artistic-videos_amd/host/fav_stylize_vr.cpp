// fav_stylize_vr -- drop-in for `th fast_artistic_video_vr.lua ...` (fast_artistic_video_vr.lua:21-76,561-591) on MI355X:
// stylises a 360-degree video given as six overlapping cube faces per frame.  Same single-dash flags, same file patterns
// (-input_pattern with two integers: frame, face id; flow / occlusion patterns with {..} = frame-1, [..] = frame and a
// remaining %d = face id, :105-115), faces processed in the order of ids {6,1,2,5,3,4} (:103), outputs
// "<prefix>-%05d_equi.png" / "<prefix>-%05d_cubemap.png" (:541,552).  All compute goes through libfav's C ABI (fav_vr_*);
// there is no CPU backend.  Not provided (rejected with a message): -evaluate, -backward, -smooth_certainty and
// -continue_with > 1 (in the reference that option reloads per-face PNGs which func_save_image no longer writes, :521-523).
// Additive flags: -precision <fp32|bf16>, -warp_border <stn|cpu>, -poll_timeout <sec>, -poll_settle <sec> (host/fav_poll.h), -png_level <0..9>, -seed <n> (uniform-random fill), -timing <0|1>,
// and -- several 360-degree videos on several GPUs (BASELINE config 5; the faces of one frame depend on each other through the
// border priors, so the unit of sharding is the video) -- -streams <a,b,...> / -gpus <n> / -force_dist / -dry_run exactly as in
// fav_stylize (host/fav_launcher.h): %S in -input_pattern, -flow_pattern, -occlusions_pattern and -output_prefix is the stream's
// name, stream s -> worker process s mod n, rank 0 parses the checkpoints and ncclBroadcast (RCCL) hands the packed blobs on.
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
#include "fav_launcher.h"
#include "fav_poll.h"

namespace {

using favl::die;
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

// utils.lua:74-80 with the settle rule of fav_poll.h: wait, read, poll again while the file reads short under its producer
template <class Reader>
void read_polled(const std::string& path, const favp::Poll& p, const char* what, Reader&& reader)
{
    bool timed_out = false;
    const int rc = favp::read_when_complete(path, p, reader, &timed_out);
    if (timed_out) die("timed out waiting for " + path);
    check(rc, what);
}

void mkdirs_for(const std::string& path)
{
    for (size_t p = 1; p < path.size(); ++p) if (path[p] == '/') mkdir(path.substr(0, p).c_str(), 0777);
}

// one 360-degree video: the loop of run_fast_neural_video with fast_artistic_video_vr.lua's callbacks (:103-302,454-591)
int run_video(std::map<std::string, std::string>& v, std::map<std::string, bool>& b, fav_net* vid, fav_net* img, double* seconds_out)
{
    auto I = [&](const char* k) { return atoi(v[k].c_str()); };
    const bool timing = I("timing") != 0;
    static const int proc_order[6] = {6, 1, 2, 5, 3, 4};                                                           // :103
    hipStream_t st; hipc(hipStreamCreate(&st), "hipStreamCreate");
    fav_vr* vr = nullptr;
    int W = 0, H = 0, ew = 0, eh = 0, cw = 0, ch = 0;
    uint8_t *d_frame = nullptr, *d_cert = nullptr, *d_equi = nullptr, *d_cube = nullptr; float* d_flow = nullptr;
    std::vector<uint8_t> h_equi, h_cube;
    // -png_encoder gpu (default): the two output images leave the device as finished PNG files (fav_png_encode_rgb8); host: zlib, -png_level
    const bool gpu_png = v["png_encoder"] == "gpu";
    uint8_t *d_png_e = nullptr, *d_png_c = nullptr; uint32_t* d_png_n = nullptr; void* d_png_ws = nullptr; size_t png_ws_bytes = 0, cap_e = 0, cap_c = 0;
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
            if (gpu_png && (ew > 9000 || (b["out_cubemap"] && cw > 9000))) die("-png_encoder gpu encodes rows of up to 9000 pixels; pass -png_encoder host for these output sizes");
            if (gpu_png) {
                if (ew) { cap_e = fav_png_capacity(ew, eh); hipc(hipMalloc(reinterpret_cast<void**>(&d_png_e), cap_e), "hipMalloc"); h_equi.resize(cap_e); png_ws_bytes = std::max(png_ws_bytes, fav_png_workspace_bytes(ew, eh)); }
                if (b["out_cubemap"]) { cap_c = fav_png_capacity(cw, ch); hipc(hipMalloc(reinterpret_cast<void**>(&d_png_c), cap_c), "hipMalloc"); h_cube.resize(cap_c); png_ws_bytes = std::max(png_ws_bytes, fav_png_workspace_bytes(cw, ch)); }
                hipc(hipMalloc(reinterpret_cast<void**>(&d_png_n), 16), "hipMalloc");
                if (png_ws_bytes) hipc(hipMalloc(&d_png_ws, png_ws_bytes), "hipMalloc");
            }
        } else if (w != W || h != H) die(img_path + ": face size changed");
        hipc(hipMemcpyAsync(d_frame, rgb, (size_t)W * H * 3, hipMemcpyHostToDevice, st), "H2D frame");
        const bool temporal = i >= 7 && !b["create_inconsistent"];
        float* flo = nullptr; uint8_t* cert = nullptr;
        if (temporal) {                                                                                             // :225-229, :274-278
            const std::string cp = flow_name(v["occlusions_pattern"], file_idx - 1, file_idx, face);
            const std::string fp = flow_name(v["flow_pattern"], file_idx - 1, file_idx, face);
            favp::Poll poll; poll.timeout_s = atof(v["poll_timeout"].c_str()); poll.settle_s = std::max(0.0, atof(v["poll_settle"].c_str()));
            int cw_ = 0, ch_ = 0, cc = 0, fw = 0, fh = 0;
            read_polled(cp, poll, "reading the certainty", [&] { return fav_read_pnm_host(cp.c_str(), &cert, &cw_, &ch_, &cc); });
            read_polled(fp, poll, "reading the flow", [&] { return fav_read_flo_host(fp.c_str(), &flo, &fw, &fh); });
            if (cc != 1 || cw_ != W || ch_ != H || fw != W || fh != H) die("flow / certainty size does not match the face: " + fp);
            hipc(hipMemcpyAsync(d_cert, cert, (size_t)W * H, hipMemcpyHostToDevice, st), "H2D cert");
            hipc(hipMemcpyAsync(d_flow, flo, (size_t)W * H * 8, hipMemcpyHostToDevice, st), "H2D flow");
        }
        const auto t0 = std::chrono::steady_clock::now();
        check(fav_vr_face(vr, i, d_frame, temporal ? d_flow : nullptr, temporal ? d_cert : nullptr, nullptr, st), "fav_vr_face");
        hipc(hipStreamSynchronize(st), "sync");
        // a stream-K hand-off that timed out: fail at THIS face, before any output derived from it is written (fav.h: the check
        // comes before a PNG is queued); the image model stylises the faces without a prior
        check(fav_net_check(vid), "stylising a face");
        if (img) check(fav_net_check(img), "stylising a face (image model)");
        if (timing) printf("Elapsed time for stylizing face %d of frame %d: %.4f\n", face, file_idx, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        fav_free_host(rgb); fav_free_host(flo); fav_free_host(cert);
        if (mode == 5) {                                                                                            // :527-557
            if (writer.valid()) writer.get();
            check(fav_vr_finish_frame(vr, d_equi, b["out_cubemap"] ? d_cube : nullptr, st), "fav_vr_finish_frame");
            uint32_t png_n[2] = {0, 0};
            if (gpu_png) {
                if (d_equi) check(fav_png_encode_rgb8(d_equi, ew, eh, d_png_e, cap_e, d_png_n, d_png_ws, png_ws_bytes, st), "fav_png_encode_rgb8");
                if (d_cube) check(fav_png_encode_rgb8(d_cube, cw, ch, d_png_c, cap_c, d_png_n + 1, d_png_ws, png_ws_bytes, st), "fav_png_encode_rgb8");   // (same stream: the workspace is free again)
                hipc(hipMemcpyAsync(png_n, d_png_n, 8, hipMemcpyDeviceToHost, st), "D2H");
                hipc(hipStreamSynchronize(st), "sync");
                if (png_n[0] > cap_e || png_n[1] > cap_c) die("fav_png_encode_rgb8 returned an impossible size");
                if (d_equi) hipc(hipMemcpyAsync(h_equi.data(), d_png_e, png_n[0], hipMemcpyDeviceToHost, st), "D2H");
                if (d_cube) hipc(hipMemcpyAsync(h_cube.data(), d_png_c, png_n[1], hipMemcpyDeviceToHost, st), "D2H");
            } else {
                if (d_equi) hipc(hipMemcpyAsync(h_equi.data(), d_equi, h_equi.size(), hipMemcpyDeviceToHost, st), "D2H");
                if (d_cube) hipc(hipMemcpyAsync(h_cube.data(), d_cube, h_cube.size(), hipMemcpyDeviceToHost, st), "D2H");
            }
            hipc(hipStreamSynchronize(st), "sync");
            check(fav_net_check(vid), "finishing a frame");
            const int out_idx = (i - 1) / 6 + 1;                                                                    // :517 (not offset by -start_frame)
            const std::string prefix = v["output_prefix"]; const int lvl = I("png_level");
            std::vector<uint8_t> e = h_equi, cb = h_cube;
            if (gpu_png) { e.resize(d_equi ? png_n[0] : 0); cb.resize(d_cube ? png_n[1] : 0); }
            writer = std::async(std::launch::async, [=]() {
                char name[4096];
                auto put = [&](const char* nm, const std::vector<uint8_t>& bytes) {          // the bytes ARE the file (GPU encoder)
                    FILE* f = fopen(nm, "wb");
                    if (!f || fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size() || fclose(f)) die(std::string("writing ") + nm + " failed");
                };
                if (!e.empty()) {
                    snprintf(name, sizeof name, "%s-%05d_equi.png", prefix.c_str(), out_idx); mkdirs_for(name);
                    if (gpu_png) put(name, e);
                    else if (fav_write_png_rgb8_host(name, e.data(), ew, eh, lvl)) die(std::string("writing ") + name + ": " + fav_last_error());
                }
                if (!cb.empty()) {
                    snprintf(name, sizeof name, "%s-%05d_cubemap.png", prefix.c_str(), out_idx); mkdirs_for(name);
                    if (gpu_png) put(name, cb);
                    else if (fav_write_png_rgb8_host(name, cb.data(), cw, ch, lvl)) die(std::string("writing ") + name + ": " + fav_last_error());
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
    check(fav_net_check(vid), "stylising the video");
    if (img) check(fav_net_check(img), "stylising the video (image model)");
    (void)hipFree(d_frame); (void)hipFree(d_cert); (void)hipFree(d_flow); (void)hipFree(d_equi); (void)hipFree(d_cube);
    (void)hipFree(d_png_e); (void)hipFree(d_png_c); (void)hipFree(d_png_n); (void)hipFree(d_png_ws);
    (void)fav_net_forget_stream(vid, st);      // the library must not wait on this handle before the next video's first forward
    hipStreamDestroy(st);
    *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
    return frames_done;
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
        {"warp_border", "stn"}, {"poll_timeout", "600"}, {"poll_settle", "1.0"}, {"png_level", "1"}, {"png_encoder", "gpu"}, {"seed", "1"}, {"timing", "0"}, {"precision", "fp32"},
        {"streams", ""}, {"gpus", "1"}, {"force_dist", "0"}, {"dry_run", "0"}, {"pin_workers", "1"}, {"worker_rank", "-1"}, {"worker_world", "0"}, {"rccl_id_file", ""}};
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
    if (v["precision"] != "fp32" && v["precision"] != "bf16") die("-precision must be fp32 or bf16");
    if (v["png_encoder"] != "gpu" && v["png_encoder"] != "host") die("-png_encoder must be gpu or host");
    const bool timing = I("timing") != 0, dry = I("dry_run") != 0;
    std::vector<std::string> streams = favl::split_list(v["streams"]);
    const bool named = !streams.empty();
    if (!named) streams.push_back("");
    int world = std::max(1, I("gpus"));
    const int rank = I("worker_rank");
    static const char* const path_opts[] = {"input_pattern", "flow_pattern", "occlusions_pattern", "output_prefix"};
    if (named && streams.size() > 1 && v["output_prefix"].find("%S") == std::string::npos)
        die("-streams: -output_prefix must contain %S (the videos would overwrite each other's frames)");
    if (rank < 0 && (world > 1 || I("force_dist"))) {                                   // launcher: one worker process per GPU
        if (!dry) {
            std::vector<std::string> models{v["model_vid"]};
            if (!v["model_img"].empty() && v["model_img"] != "self") models.push_back(v["model_img"]);
            favl::validate_models(models);                        // before any worker exists
            const int ndev = fav_device_count();
            if (ndev <= 0) die(std::string("ERROR: ") + fav_last_error());
            if (I("gpu") + world > ndev) die("-gpus " + v["gpus"] + " from -gpu " + v["gpu"] + ": only " + std::to_string(ndev) + " devices");
        }
        std::string xdir;
        const int worst = favl::spawn_workers(argc, argv, world, I("pin_workers") != 0, &xdir);
        if (timing && !dry && worst == 0) favl::print_aggregate(xdir, world, streams.size());
        favl::remove_exchange_dir(xdir, world);
        return worst;
    }
    const bool dist = rank >= 0;
    if (dist) world = I("worker_world");
    const int device = I("gpu") + (dist ? rank : 0);
    std::vector<std::string> mine;
    for (size_t s_ = 0; s_ < streams.size(); ++s_) if (!dist || (int)(s_ % (size_t)world) == rank) mine.push_back(streams[s_]);
    if (dry) {
        favl::dry_run_failure_hook(rank);
        std::string js = "{\"rank\": " + std::to_string(std::max(rank, 0)) + ", \"world\": " + std::to_string(dist ? world : 1) + ", \"device\": " + std::to_string(device) + ", \"streams\": [";
        for (size_t k = 0; k < mine.size(); ++k) {
            js += std::string(k ? ", " : "") + "{\"name\": " + favl::json_str(mine[k]);
            for (const char* po : path_opts) js += std::string(", \"") + po + "\": " + favl::json_str(named ? favl::subst_stream(v[po], mine[k]) : v[po]);
            js += "}";
        }
        printf("%s]}\n", js.c_str());
        return 0;
    }

    hipc(hipSetDevice(device), "hipSetDevice");
    fav_net* vid = nullptr; fav_net* img = nullptr;
    const bool want_img = !v["model_img"].empty() && v["model_img"] != "self";
    if (dist) {
        favl::load_models_dist(rank, world, v["rccl_id_file"], device, v["model_vid"], want_img ? v["model_img"] : std::string(), &vid, &img, mine.size(), 1);
    } else {
        if (fav_net_create(v["model_vid"].c_str(), device, &vid)) die(std::string("ERROR: Could not load model from ") + v["model_vid"] + " (" + fav_last_error() + ")");
        if (want_img && fav_net_create(v["model_img"].c_str(), device, &img)) die(std::string("ERROR: Could not load model from ") + v["model_img"] + " (" + fav_last_error() + ")");
    }
    check(fav_net_set_precision(vid, v["precision"] == "bf16" ? FAV_PRECISION_BF16_OPERANDS : FAV_PRECISION_FP32), "fav_net_set_precision");
    int frames = 0; double seconds = 0;
    for (const std::string& name : mine) {
        std::map<std::string, std::string> vs = v;
        if (named) for (const char* po : path_opts) vs[po] = favl::subst_stream(v[po], name);
        double sec = 0;
        frames += run_video(vs, b, vid, img, &sec);
        seconds += sec;
    }
    if (dist && timing) favl::write_worker_result(v["rccl_id_file"], rank, frames, seconds);
    if (img) fav_net_destroy(img);
    fav_net_destroy(vid);
    return 0;
}
