// fav_stylize -- drop-in for `th fast_artistic_video.lua ...` (fast_artistic_video.lua:21-67,174-189 +
// fast_artistic_video_core.lua:34-229) on MI355X.  Same single-dash flags, same file-name patterns
// ({..} = index i-1, [..] = index i; fast_artistic_video.lua:70-77), same 1-based frame loop that stops
// at the first missing frame (core.lua:196-197), same "<prefix>-%05d.png" outputs (fav.lua:161).
//
// Host structure: loader threads read/decode frames i+1.. (PPM, .flo, .pgm) into pinned staging while the GPU works on
// frame i; uploads run on their own queue; the host enqueues frame i before it waits for frame i-1 (so the GPU never idles
// between frames); PNG deflate + file writes run on a small pool.  All compute
// goes through libfav's C ABI (fav_stream_*); there is no CPU backend (-gpu -1 is rejected).
//
// Additive flags (not in the reference): -forward_flow_pattern <pat> (run the consistency check on the
// GPU instead of reading .pgm files), -structure <0|1> (4-argument checker mode, default 1 as in
// makeOptFlow_deepflow.sh:59), -warp_border <stn|cpu>, -poll_timeout <sec>, -poll_settle <sec> (1.0: a file younger than this is
//   taken only once it has not been modified for that long -- what utils.lua:79's `sleep 1` buys; host/fav_poll.h),
// -png_overlap <0|1> (0, default; 1: with -png_encoder gpu the encoder's kernels run on a queue of their own next to the following frame's
//   network -- fav_stream_encode_png_async -- instead of in front of it: +0.5 % in HBM, but the events between the queues cost this
//   process 1.6 ms of CPU per frame and the file -> PNG rate does not move, profiles/bench_r5e.log against bench_r5f.log)
// -png_encoder <gpu|host> (gpu, default: the PNG file's bytes are produced on the device, the host only write()s them -- fixed-Huffman
// deflate, larger files; host: zlib on the writer threads at -png_level <0..9>, ~25 ms per 1280x720 frame and core),
// -precision <fp32|bf16> (fp32 = parity mode, default; bf16 = optional fast mode, see include/fav.h),
// -seed <n> (key of the documented RNG behind -fill_occlusions uniform-random, unseeded in the reference),
// -writers <n>, -timing <0|1>, -temporal_eval_file <path> (the temporal-consistency number of -evaluate, fav.lua:128-151,
// with the frame's own flow and certainty: one line of ';'-separated per-frame values, one line with their mean).
//
// Several videos on several GPUs (BASELINE config 4; the reference runs one `th` process per video, stylizeVideo_deepflow.sh:87-96,
// and its only device hook is utils.setup_gpu, fast_artistic_video/utils.lua:43-66):
//   -streams <a,b,c,...>  independent videos; every path option (-input_pattern, -flow_pattern, -forward_flow_pattern,
//                         -occlusions_pattern, -output_prefix, -temporal_eval_file) has the token %S replaced by the stream's name
//   -gpus <n>             one worker PROCESS per GPU (devices -gpu .. -gpu+n-1), stream s -> worker s mod n, streams of a worker run
//                         back to back.  Rank 0 parses the checkpoint(s) once and broadcasts the packed blob (fav_net_pack_host) with
//                         RCCL (ncclBroadcast over xGMI; ncclCommInitRank + a unique-id file); no other collective: a stream's
//                         frame i needs its own frame i-1 only.  Each worker gets host-threads / n PNG writers and loaders.
//   -shared_gpu 1         the GPU is shared with other processes: data-parallel convolution grids (fav_net_set_shared_device)
//   -force_dist 1         take the worker / RCCL path even for -gpus 1;  -dry_run 1: workers print their assignment and exit
//                         without touching a device (plumbing test).
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <fcntl.h>
#include <sys/resource.h>
#include <time.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fav.h"
#include "fav_launcher.h"
#include "fav_poll.h"

namespace {

struct Opt {
    std::map<std::string, std::string> v;
    std::map<std::string, bool> b;
    std::string s(const char* k) const { return v.at(k); }
    int i(const char* k) const { return atoi(v.at(k).c_str()); }
    double d(const char* k) const { return atof(v.at(k).c_str()); }
    bool f(const char* k) const { return b.at(k); }
};

using favl::die;

// diagnostic switches exist in -DFAV_DIAG builds of this executable only (csrc/fav_internal.h says why)
const char* diag_env(const char* name)
{
#ifdef FAV_DIAG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

void check(int rc, const char* what)
{
    if (rc) die(std::string(what) + ": " + fav_last_error());
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

std::string fmt_int(const std::string& pattern, int i)
{
    char buf[4096];
    snprintf(buf, sizeof buf, pattern.c_str(), i);
    return buf;
}

// getFormatedFlowFileName (fast_artistic_video.lua:70-77): {fmt} <- fromIndex, [fmt] <- toIndex
std::string flow_name(const std::string& pattern, int from, int to)
{
    std::string out;
    for (size_t p = 0; p < pattern.size();) {
        const char c = pattern[p];
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            const size_t e = pattern.find(close, p + 1);
            if (e != std::string::npos) {
                out += fmt_int(pattern.substr(p + 1, e - p - 1), c == '{' ? from : to);
                p = e + 1;
                continue;
            }
        }
        out += c; ++p;
    }
    return out;
}

// utils.wait_for_file (fast_artistic_video/utils.lua:74-80), bounded, with the settle rule of fav_poll.h
void wait_for_file(const std::string& path, const favp::Poll& p)
{
    if (favp::wait_for_file(path, p) != favp::WAIT_OK) die("timed out waiting for " + path);
}

// wait, read, and poll again while the file reads short under a producer that is still writing it (fav_poll.h)
template <class Reader>
void read_polled(const std::string& path, const favp::Poll& p, Reader&& reader)
{
    bool timed_out = false;
    const int rc = favp::read_when_complete(path, p, reader, &timed_out);
    if (timed_out) die("timed out waiting for " + path);
    check(rc, path.c_str());
}

void mkdirs_for(const std::string& path)
{
    for (size_t p = 1; p < path.size(); ++p)
        if (path[p] == '/') mkdir(path.substr(0, p).c_str(), 0777);
}

// minimal PNG reader for -continue_with (8-bit RGB / RGBA / grey, non-interlaced)
bool read_png_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& W, int& H)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> d;
    uint8_t buf[65536]; size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + got);
    fclose(f);
    if (d.size() < 33 || memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8)) return false;
    auto be32 = [&](size_t p) { return (uint32_t)d[p] << 24 | (uint32_t)d[p + 1] << 16 | (uint32_t)d[p + 2] << 8 | d[p + 3]; };
    size_t p = 8; int ctype = -1, depth = 0, interlace = 0; std::vector<uint8_t> idat;
    while (p + 12 <= d.size()) {
        const uint32_t len = be32(p); const char* ty = (const char*)&d[p + 4];
        if (p + 12 + len > d.size()) return false;
        if (!memcmp(ty, "IHDR", 4)) { W = (int)be32(p + 8); H = (int)be32(p + 12); depth = d[p + 16]; ctype = d[p + 17]; interlace = d[p + 20]; }
        else if (!memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), d.begin() + p + 8, d.begin() + p + 8 + len);
        else if (!memcmp(ty, "IEND", 4)) break;
        p += 12 + len;
    }
    const int ch = ctype == 2 ? 3 : ctype == 6 ? 4 : ctype == 0 ? 1 : 0;
    if (!ch || depth != 8 || interlace || W <= 0 || H <= 0) return false;
    const size_t stride = (size_t)W * ch;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) return false;
    std::vector<uint8_t> img(stride * H);
    for (int y = 0; y < H; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* s = &raw[(stride + 1) * y + 1];
        uint8_t* o = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)ch ? o[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int pr = 0;
            if (ft == 1) pr = a; else if (ft == 2) pr = b; else if (ft == 3) pr = (a + b) / 2;
            else if (ft == 4) { const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            o[x] = (uint8_t)(s[x] + pr);
        }
    }
    rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i)
        for (int c = 0; c < 3; ++c) rgb[i * 3 + c] = img[i * ch + (ch == 1 ? 0 : c)];
    return true;
}

// tiny thread pool for PNG encode + write
class Pool {
public:
    explicit Pool(int n, int device = -1) { for (int i = 0; i < n; ++i) th_.emplace_back([this, device] { if (device >= 0) (void)hipSetDevice(device); run(); }); }
    ~Pool() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(f)); ++pending_; } cv_.notify_one(); }
    void wait_below(size_t n) { std::unique_lock<std::mutex> l(m_); done_.wait(l, [&] { return pending_ <= n; }); }
private:
    void run()
    {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return stop_ || !q_.empty(); }); if (q_.empty()) return; f = std::move(q_.front()); q_.pop_front(); }
            f();
            { std::lock_guard<std::mutex> l(m_); --pending_; }
            done_.notify_all();
        }
    }
    std::vector<std::thread> th_; std::deque<std::function<void()>> q_; std::mutex m_; std::condition_variable cv_, done_;
    size_t pending_ = 0; bool stop_ = false;
};

// Pinned PNG output slots with explicit ownership: a slot is taken before the D2H copy of a frame is enqueued and released by
// the writer task once the file is on disk.  (Counting pending tasks is not enough: tasks finish out of order, so the oldest
// slot -- the next one a round-robin hands out -- may still be read by its writer.)
class Slots {
public:
    void add(uint8_t* p) { std::lock_guard<std::mutex> l(m_); free_.push_back(p); }
    uint8_t* take() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !free_.empty(); }); uint8_t* p = free_.back(); free_.pop_back(); return p; }
    uint8_t* try_take() { std::lock_guard<std::mutex> l(m_); if (free_.empty()) return nullptr; uint8_t* p = free_.back(); free_.pop_back(); return p; }
    void give(uint8_t* p) { { std::lock_guard<std::mutex> l(m_); free_.push_back(p); } cv_.notify_one(); }
private:
    std::mutex m_; std::condition_variable cv_; std::vector<uint8_t*> free_;
};

struct FrameIn {           // everything frame i needs from disk
    int index = 0; bool ok = false; bool single = false;
    uint8_t* frame = nullptr; int W = 0, H = 0;
    float* bw = nullptr; float* fw = nullptr; uint8_t* cert = nullptr;
    void release() { fav_free_host(frame); fav_free_host(bw); fav_free_host(fw); fav_free_host(cert); frame = nullptr; bw = fw = nullptr; cert = nullptr; }
};


struct StreamResult { int frames = 0; double seconds = 0, wait_loader = 0, wait_gpu = 0, wait_png = 0, setup = 0, tail = 0, cpu_s = 0, cpu_loaders = 0, cpu_writers = 0, cpu_writer_sync = 0, cpu_main = 0; long long png_bytes = 0; };

// Waiting for an event WITHOUT occupying a core: hipEventSynchronize spins on this runtime even for events created with
// hipEventBlockingSync (measured: 1.8 ms of CPU per 1.8 ms frame in the thread that waits, profiles/r03g_e2e.log), so the host polls
// with short sleeps instead.  The host runs a frame ahead of the GPU, so the added latency (<= the sleep) is never on the GPU's path.
hipError_t wait_event_sleeping(hipEvent_t ev, unsigned sleep_us = 100)
{
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        usleep(sleep_us);
    }
}

double thread_cpu_seconds()
{
    struct timespec ts;
    if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts)) return 0.0;
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// CPU time of every live thread of the process (name: seconds), for -timing: the HIP runtime's own threads show up here
std::string thread_cpu_report()
{
    std::string out = "[";
    DIR* d = opendir("/proc/self/task");
    if (!d) return "[]";
    const double tick = 1.0 / (double)sysconf(_SC_CLK_TCK);
    bool first = true;
    while (struct dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        const std::string p = std::string("/proc/self/task/") + e->d_name + "/stat";
        FILE* f = fopen(p.c_str(), "r");
        if (!f) continue;
        char buf[1024]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
        const char* lp = strchr(buf, '('); const char* rp = strrchr(buf, ')');
        if (!lp || !rp) continue;
        std::string comm(lp + 1, rp);
        unsigned long ut = 0, stt = 0; char state;
        // fields after ')': state ppid pgrp session tty tpgid flags minflt cminflt majflt cmajflt utime stime
        if (sscanf(rp + 2, "%c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &state, &ut, &stt) != 3) continue;
        const double sec = (ut + stt) * tick;
        if (sec < 0.02) continue;
        char item[128]; snprintf(item, sizeof item, "%s{\"thread\": \"%s\", \"cpu_s\": %.2f}", first ? "" : ", ", comm.c_str(), sec);
        out += item; first = false;
    }
    closedir(d);
    return out + "]";
}

double process_cpu_seconds()
{
    struct rusage ru;
    if (getrusage(RUSAGE_SELF, &ru)) return 0.0;
    return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
}

struct Pinned { uint8_t* frame = nullptr; float* bw = nullptr; float* fw = nullptr; uint8_t* cert = nullptr; };

// Everything frame i needs from disk (fast_artistic_video.lua:93-110,153-158): the frame, and for a frame with a predecessor the backward
// flow plus either the certainty file (the reference's path: polled, fav.lua:102) or the forward flow (fused check).
// pd != null: decode straight into that pinned staging set of pin_px pixels (no allocation, no second copy)
FrameIn load_frame_inputs(const Opt& o, const favp::Poll& poll, int i, bool first_of_run, const Pinned* pd, size_t pin_px)
{
    const bool fused_check = !o.s("forward_flow_pattern").empty();
    FrameIn in; in.index = pd ? -i - 1 : i;              // negative index marks "pinned, do not free"
    const std::string fp = fmt_int(o.s("input_pattern"), i);
    if (!file_exists(fp)) return in;                                                                     // fav.lua:93-97 -> nil -> break
    int ch;
    if (pd) { check(fav_read_pnm_into_host(fp.c_str(), pd->frame, pin_px * 3, &in.W, &in.H, &ch), fp.c_str()); in.frame = pd->frame; }
    else check(fav_read_pnm_host(fp.c_str(), &in.frame, &in.W, &in.H, &ch), fp.c_str());
    if (ch != 3) die(fp + ": expected a colour (P6) frame");
    in.single = (i == 1) || o.f("create_inconsistent") || first_of_run;                                 // fav.lua:172
    if (!in.single) {
        const std::string fl = flow_name(o.s("flow_pattern"), i - 1, i);                               // fav.lua:100,154
        int w, h;
        if (fused_check) {
            const std::string ff = flow_name(o.s("forward_flow_pattern"), i - 1, i);
            read_polled(ff, poll, [&] { if (pd) { in.fw = pd->fw; return fav_read_flo_into_host(ff.c_str(), pd->fw, pin_px * 2, &w, &h); }
                                        return fav_read_flo_host(ff.c_str(), &in.fw, &w, &h); });
            if (w != in.W || h != in.H) die(ff + ": size differs from the frame");
        } else {
            const std::string cp = flow_name(o.s("occlusions_pattern"), i - 1, i);
            int cch = 0;
            read_polled(cp, poll, [&] { if (pd) { in.cert = pd->cert; return fav_read_pnm_into_host(cp.c_str(), pd->cert, pin_px, &w, &h, &cch); }   // fav.lua:102
                                        return fav_read_pnm_host(cp.c_str(), &in.cert, &w, &h, &cch); });
            if (cch != 1 || w != in.W || h != in.H) die(cp + ": expected a P5 mask of the frame's size");
        }
        read_polled(fl, poll, [&] { if (pd) { in.bw = pd->bw; return fav_read_flo_into_host(fl.c_str(), pd->bw, pin_px * 2, &w, &h); }
                                    return fav_read_flo_host(fl.c_str(), &in.bw, &w, &h); });
        if (w != in.W || h != in.H) die(fl + ": size differs from the frame");
    }
    in.ok = true;
    return in;
}

// One video: the loop of run_fast_neural_video (core.lua:189-229) with the video CLI's callbacks (fav.lua:93-172).
// `net` / `net_img` live on the current device; `nwriters` PNG threads.
void run_stream(const Opt& o, fav_net* net, fav_net* net_img, int nwriters, StreamResult* res)   // nwriters: the budget; adjusted below
{
    const bool fused_check = !o.s("forward_flow_pattern").empty();
    const int border = o.s("warp_border") == "cpu" ? FAV_BORDER_CPU : FAV_BORDER_STN;
    const int num_frames = o.i("num_frames");
    const bool backward = o.f("backward");
    const int start = backward ? num_frames - 1 : o.i("continue_with"), end = backward ? 1 : num_frames, inc = backward ? -1 : 1;   // core:189-191
    favp::Poll poll; poll.timeout_s = o.d("poll_timeout"); poll.settle_s = std::max(0.0, o.d("poll_settle"));

    size_t pin_px = 0;                       // capacity of a pinned staging set in pixels (0: none yet)
    auto load = [&](int i, bool first_of_run, const Pinned* pd) { return load_frame_inputs(o, poll, i, first_of_run, pd, pin_px); };

    // -png_encoder gpu (default): the PNG file's bytes are produced on the device (fav_stream_encode_png: Sub filter + fixed-Huffman /
    // run-length deflate per row + Adler-32 / CRC-32 combine), leave as ONE exact-size DMA and the writer thread only write()s them;
    // -png_encoder host: the 8-bit frame is downloaded and deflated by zlib on the writer threads (-png_level; ~25 ms per frame and core)
    const bool gpu_png = o.s("png_encoder") == "gpu";
    const bool png_overlap = o.i("png_overlap") != 0;
    // QUIET synchronisation (gpu_png, no 4-argument look-ahead): nothing but kernels ever enters the compute queue.  On this runtime
    // a marker behind long-running kernels (hipEventRecord on the compute stream) or a device-side dependency between queues
    // (hipStreamWaitEvent, a copy in front of kernels in one stream) is resolved by a runtime thread that SPINS until the marker
    // fires -- one core per process, 1.6 ms of CPU per 1.8 ms frame (scripts/runtime_thread_bench.hip, profiles/r03k_runtime_thread.log:
    // kernels only 0.03 cores in the background, + one event record per frame 0.70, + a cross-stream dependency 0.74).  So: uploads are
    // DMA copies on their own queue and the HOST checks (sleeping poll on that queue's event) that frame i's inputs have arrived before
    // it enqueues frame i's kernels -- they were requested a frame earlier; the frame's last kernel writes the PNG size into host-mapped
    // memory, which the host polls; the PNG bytes then leave as one DMA on a third queue.  The 4-argument look-ahead (its mask pipeline
    // runs a frame ahead on the library's side queues) is started after the host has seen that frame's upload finish and is waited
    // for on the host as well (fav_stream_set_host_ordered): no event there either.
    const bool quiet = gpu_png;       // (the 4-argument look-ahead runs host-ordered in this mode: fav_stream_set_host_ordered)
    int cur_device = 0; (void)hipGetDevice(&cur_device);
    const double cpu0 = process_cpu_seconds();
    if (gpu_png && o.i("writers") <= 0) nwriters = std::min(nwriters, 4);      // they only write() finished files: 0.5 ms per frame
    Pool writers(nwriters, cur_device);
    // compute queue; upload queue (the next frame's inputs travel while this frame computes); download queue (the 8-bit frame leaves
    // while the next frame computes: on the compute queue the 2.8 MB copy held back the next frame's kernels for its whole duration)
    // gpu_png: the file's bytes must not sit behind the NEXT frame's size word on the download queue (that one waits for the next
    // frame's kernels: the writer thread then waited a whole frame for a 0.1 ms DMA): finish() enqueues them first
    // (three queues besides the library's two look-ahead queues: a process gets four hardware queues by default, and the 4-argument
    //  mode lost 20 % when a sixth stream made its side queues share one with the compute queue)
    hipStream_t st, st_copy, st_down;
    if (hipStreamCreate(&st) != hipSuccess || hipStreamCreate(&st_copy) != hipSuccess || hipStreamCreate(&st_down) != hipSuccess) die("hipStreamCreate failed");
    hipStream_t const st_data = st_down;     // quiet mode: st_down carries nothing else; otherwise finish() enqueues frame i-1's bytes BEFORE frame i's size word
    constexpr int NDEV = 4;                  // device input sets: frame i (in use) + up to two frames of look-ahead + one being refilled
    hipEvent_t ev_up[NDEV], ev_done[2], ev_out[2];
    for (auto& e : ev_out) if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) die("hipEventCreate failed");
    for (auto& e : ev_up) if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) die("hipEventCreate failed");
    // events the HOST waits on: blocking (the waiting thread sleeps instead of spinning -- a spinning wait costs one core per waiter,
    // 1.9 ms of CPU per frame for the main thread alone at 530 frames/s) and with the system-scope fence (the host reads what they cover)
    for (auto& e : ev_done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) die("hipEventCreate failed");
    fav_stream* fs = nullptr;
    int W = 0, H = 0;                        // frame size
    int Wo = 0, Ho = 0;                      // size of the stylised frames: the network's output (H x W when both are multiples of 4)
    struct Dev { uint8_t* frame = nullptr; uint8_t* cert = nullptr; float* bw = nullptr; float* fw = nullptr; };
    Dev dev[NDEV];
    uint8_t* d_out8s[2] = {nullptr, nullptr};    // frames alternate: frame i + 2 is enqueued after the host has seen frame i's download complete
    // gpu_png: device PNG buffers (two, alternating like d_out8s), their sizes on the device and in pinned host memory, and per pinned
    // output slot the event of the exact-size copy into it
    uint8_t* d_png[2] = {nullptr, nullptr}; uint32_t* d_png_size[2] = {nullptr, nullptr}; uint32_t* h_png_size = nullptr;
    hipEvent_t png_copy_ev[2] = {nullptr, nullptr};       // last copy OUT of d_png[k]: the next encode into it waits for this
    size_t png_cap = 0;
    std::map<uint8_t*, hipEvent_t> slot_ev;
    std::atomic<long long> png_bytes_total{0}, cpu_loaders_us{0}, cpu_writers_us{0}, cpu_writer_sync_us{0};
    const double cpu_main0 = thread_cpu_seconds();
    float *d_prev = nullptr, *d_cur = nullptr; std::vector<double> temporal;      // -temporal_eval_file
    const int nslots = nwriters + 2;   // pinned output slots in flight to the PNG pool (deflate ~55 ms/frame/thread)
    std::vector<uint8_t*> h_out;       // allocated on demand, at most nslots
    size_t slot_bytes = 0;
    Slots slots;
    auto new_slot = [&]() {
        uint8_t* p = nullptr;
        if (hipHostMalloc((void**)&p, slot_bytes, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc failed");
        if (gpu_png) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) die("hipEventCreate failed"); slot_ev[p] = e; }
        h_out.push_back(p);
        return p;
    };


    // continue_with > 1: reload the previous stylised PNG as the recurrent state (8-bit; the reference's
    // video CLI does not reload anything and fails, fast_artistic_video.lua:89,153-156)
    std::vector<float> resume_state;
    bool have_resume = false;
    if (!backward && start > 1 && !o.f("create_inconsistent")) {
        std::vector<uint8_t> prev; int pw, ph;
        char nm[4096]; snprintf(nm, sizeof nm, "%s-%05d.png", o.s("output_prefix").c_str(), start - 1);
        if (read_png_rgb8(nm, prev, pw, ph)) {
            resume_state.resize((size_t)pw * ph * 3);
            for (size_t i = 0; i < (size_t)pw * ph; ++i)
                for (int c = 0; c < 3; ++c) resume_state[(size_t)c * pw * ph + i] = prev[i * 3 + c] / 255.0f;
            W = -pw; H = -ph; have_resume = true;    // validated against the first frame below
        } else {
            fprintf(stderr, "warning: %s not found; frame %d is stylised without a prior\n", nm, start);
        }
    }

    // Host pipeline: DEPTH loader threads read + decode frames i+1.. while the GPU works on frame i; decoded
    // inputs are copied into pinned staging sets so the H2D copies are true async DMA.
    constexpr int DEPTH = 6;
    Pinned pin[DEPTH + 1];
    bool pinned_ready = false;
    auto load_pinned = [&](int i, bool first_of_run, int set) {
        if (pinned_ready && !pin[set].frame) {           // first use of this staging set: pinned here, by the loader thread that owns it
            Pinned& p = pin[set];
            if (hipHostMalloc((void**)&p.frame, pin_px * 3, hipHostMallocDefault) || hipHostMalloc((void**)&p.bw, pin_px * 8, hipHostMallocDefault) ||
                (fused_check ? hipHostMalloc((void**)&p.fw, pin_px * 8, hipHostMallocDefault) : hipHostMalloc((void**)&p.cert, pin_px, hipHostMallocDefault))) die("hipHostMalloc failed");
        }
        return load(i, first_of_run, pinned_ready ? &pin[set] : nullptr);
    };
    auto idx_ok = [&](int i) { return backward ? i >= end : i <= end; };
    // persistent loader threads (a std::thread per frame cost ~0.1 ms of CPU each); no more of them than CPUs to run them on
    Pool loaders(std::min(DEPTH, std::max(2, favl::effective_cpus())), cur_device);
    std::deque<std::pair<int, std::future<void>>> inflight;      // (slot set, completion of its load)
    std::vector<FrameIn> ready(DEPTH + 1);
    int next_to_issue = start, sets_used = 0;
    auto issue = [&]() {
        while ((int)inflight.size() < DEPTH && idx_ok(next_to_issue)) {
            const int i = next_to_issue, set = sets_used % (DEPTH + 1);
            const bool fo = (i == start) && !have_resume && start != 1;
            auto done_p = std::make_shared<std::promise<void>>();
            inflight.emplace_back(set, done_p->get_future());
            loaders.submit([&, i, fo, set, done_p] {
                const double c0 = thread_cpu_seconds();
                ready[set] = load_pinned(i, fo, set);
                cpu_loaders_us += (long long)(1e6 * (thread_cpu_seconds() - c0));
                done_p->set_value();
            });
            next_to_issue += inc; ++sets_used;
        }
    };

    bool first = true;
    const auto t_begin = std::chrono::steady_clock::now();
    double t_wait_load = 0, t_gpu = 0, t_wait_writer = 0;
    int done = 0, dset = 0;
    auto upload = [&](const FrameIn& f, int set) {                // async H2D of one frame's inputs (pinned -> device) on the copy queue
        const size_t n = (size_t)W * H; const Dev& dv = dev[set];
        hipMemcpyAsync(dv.frame, f.frame, n * 3, hipMemcpyHostToDevice, st_copy);
        if (f.bw) hipMemcpyAsync(dv.bw, f.bw, n * 8, hipMemcpyHostToDevice, st_copy);
        if (f.fw) hipMemcpyAsync(dv.fw, f.fw, n * 8, hipMemcpyHostToDevice, st_copy);
        if (f.cert) hipMemcpyAsync(dv.cert, f.cert, n, hipMemcpyHostToDevice, st_copy);
        hipEventRecord(ev_up[set], st_copy);
        if (!quiet) hipStreamWaitEvent(st, ev_up[set], 0);         // everything enqueued on the compute queue from here on sees them
    };
    // the host runs one frame ahead of the GPU: frame i is enqueued before frame i-1's completion is awaited
    struct Pending { bool valid = false; int index = 0; bool single = false; uint8_t* hb = nullptr; int ev = 0;
                     std::chrono::steady_clock::time_point t0; };
    Pending pend;
    auto finish = [&](Pending& pd) {
        if (!pd.valid) return;
        if (quiet) {
            // the frame's last kernel (png_finish_kernel) stores the file size into host-mapped memory behind a system-scope fence
            volatile uint32_t* sz = &h_png_size[pd.ev];
            for (int spins = 0; *sz == 0u; ++spins) {
                usleep(100);
                if ((spins & 1023) == 1023) {                 // every ~0.1 s: has the GPU faulted? (a query, not a marker)
                    const hipError_t e = hipStreamQuery(st);
                    if (e != hipSuccess && e != hipErrorNotReady) die("GPU error while stylising a frame");
                }
            }
        } else if (wait_event_sleeping(ev_done[pd.ev]) != hipSuccess) die("GPU error while stylising a frame");
        check(fav_net_check(net), "stylising a frame");      // a stream-K hand-off that timed out: fail at THIS frame, before its PNG exists
        if (net_img) check(fav_net_check(net_img), "stylising a frame");
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pd.t0).count();
        if (pd.single) printf("Elapsed time for stylizing frame independently:%g\n", ms / 1000.0);           // core:155
        else printf("Elapsed time for stylizing frame:%g\n", ms / 1000.0);                                   // core:177
        char nm[4096]; snprintf(nm, sizeof nm, "%s-%05d.png", o.s("output_prefix").c_str(), pd.index);       // fav.lua:161
        printf("Writing output image to %s\n", nm);
        mkdirs_for(nm);
        const int lvl = o.i("png_level");
        const std::string path = nm; uint8_t* hb = pd.hb; const int w_ = Wo, h_ = Ho;
        Slots* sl = &slots;
        if (gpu_png) {
            // the size word has arrived (ev_done): ONE DMA of exactly the file's bytes, then the writer thread only write()s
            const uint32_t nbytes = h_png_size[pd.ev];
            if (nbytes < 57 || nbytes > png_cap) die("fav_stream_encode_png returned an impossible size");
            hipEvent_t cev = slot_ev[hb];
            if (hipMemcpyAsync(hb, d_png[pd.ev], nbytes, hipMemcpyDeviceToHost, st_data) != hipSuccess || hipEventRecord(cev, st_data) != hipSuccess) die("D2H of the PNG failed");
            png_copy_ev[pd.ev] = cev;
            png_bytes_total += nbytes;
            std::atomic<long long>* cw = &cpu_writers_us;
            std::atomic<long long>* cs = &cpu_writer_sync_us;
            writers.submit([hb, path, nbytes, cev, sl, cw, cs] {
                const double c0 = thread_cpu_seconds();
                if (wait_event_sleeping(cev, 50) != hipSuccess) { fprintf(stderr, "GPU error while downloading %s\n", path.c_str()); exit(1); }
                *cs += (long long)(1e6 * (thread_cpu_seconds() - c0));
                const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
                size_t off = 0;
                while (fd >= 0 && off < nbytes) { const ssize_t k = write(fd, hb + off, nbytes - off); if (k <= 0) break; off += (size_t)k; }
                if (fd < 0 || off != nbytes || close(fd)) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
                *cw += (long long)(1e6 * (thread_cpu_seconds() - c0));
                sl->give(hb);
            });
        } else {
            std::atomic<long long>* cw = &cpu_writers_us;
            writers.submit([hb, path, w_, h_, lvl, sl, cw] {
                const double c0 = thread_cpu_seconds();
                if (fav_write_png_rgb8_host(path.c_str(), hb, w_, h_, lvl)) fprintf(stderr, "%s\n", fav_last_error());
                *cw += (long long)(1e6 * (thread_cpu_seconds() - c0));
                sl->give(hb);                                     // only now may the slot receive another frame
            });
        }
        pd.valid = false;
    };
    auto pop_next = [&](FrameIn& out) {
        if (inflight.empty()) return false;
        const auto tl = std::chrono::steady_clock::now();
        const int set = inflight.front().first;
        inflight.front().second.get(); inflight.pop_front();
        out = ready[set];
        t_wait_load += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl).count();
        return out.ok;
    };
    // the first frame is loaded synchronously (its size sizes every buffer)
    FrameIn cur = load(start, !have_resume && start != 1, nullptr);
    // Look-ahead: frames i+1 .. i+LA are uploaded (device set of ahead[k] = dset + 1 + k) and, with the checker's 4-argument mode, their
    // masks are under way on the side queues while frame i is enqueued.  That mode needs TWO frames: a mask takes 1.5 ms behind a
    // 0.45 ms upload, a frame 1.9 ms -- one frame ahead, every frame waited 0.1 ms for its mask (period 2.05 ms: 460 frames/s;
    // profiles/r03z_cli_4arg_trace.txt)
    const int LA = (fused_check && o.i("structure") != 0) ? 2 : 1;
    std::deque<FrameIn> ahead;
    bool drained = false;
    next_to_issue = start + inc;
    for (int i = start; idx_ok(i) && cur.ok; i += inc) {                                                      // core:196-197
        if (first) {
            const int rW = -W, rH = -H;          // size of the reloaded PNG (-continue_with), if any
            W = cur.W; H = cur.H;
            fav_stream_opts so{border, o.i("occlusions_min_filter"), o.f("invert_occlusion") ? 1 : 0, o.f("fix_occlusions") ? 1 : 0,
                               o.s("fill_occlusions") == "uniform-random" ? 1 : 0, (unsigned)o.i("seed")};
            check(fav_stream_create(net, H, W, &so, &fs), "fav_stream_create");
            check(fav_stream_output_size(fs, &Ho, &Wo), "fav_stream_output_size");
            if (quiet) check(fav_stream_set_host_ordered(fs, 1), "fav_stream_set_host_ordered");
            if (have_resume && (rW != Wo || rH != Ho)) die("-continue_with: the previous PNG's size differs from the stylised frames' (" + std::to_string(Wo) + "x" + std::to_string(Ho) + ")");
            if (!o.s("temporal_eval_file").empty() && (Wo != W || Ho != H))
                die("-temporal_eval_file: the stylised frames (" + std::to_string(Wo) + "x" + std::to_string(Ho) + ") are larger than the flow; the reference's func_eval fails on such sizes as well (fav.lua:128-151)");
            if (net_img) check(fav_stream_set_image_net(fs, net_img), "fav_stream_set_image_net");
            const size_t n = (size_t)W * H, no = (size_t)Wo * Ho;
            if (gpu_png && Wo > 9000) die("-png_encoder gpu encodes rows of up to 9000 pixels (one image row lives in a CU's LDS); pass -png_encoder host for " + std::to_string(Wo) + "-wide frames");
            if (gpu_png) {
                png_cap = fav_png_capacity(Wo, Ho);
                for (int k = 0; k < 2; ++k) if (hipMalloc((void**)&d_png[k], png_cap) || hipMalloc((void**)&d_png_size[k], 16)) die("hipMalloc failed");
                if (hipHostMalloc((void**)&h_png_size, 64, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc failed");
            } else if (hipMalloc((void**)&d_out8s[0], no * 3) || hipMalloc((void**)&d_out8s[1], no * 3)) die("hipMalloc failed");
            for (auto& dv : dev)
                if (hipMalloc((void**)&dv.frame, n * 3) || hipMalloc((void**)&dv.cert, n) || hipMalloc((void**)&dv.bw, n * 8) ||
                    hipMalloc((void**)&dv.fw, n * 8)) die("hipMalloc failed");
            // pinned memory is paid for by the page (~0.2 ms per MB): two output slots now, the others when the pool first falls behind;
            // every loader thread pins its own staging set on first use, in parallel with this thread's set-up
            slot_bytes = gpu_png ? png_cap : no * 3;
            for (int k = 0; k < std::min(2, nslots); ++k) slots.add(new_slot());
            pin_px = n; pinned_ready = true;
            if (have_resume) {
                float* d_state = nullptr;
                if (hipMalloc((void**)&d_state, no * 12) != hipSuccess) die("hipMalloc failed");
                hipMemcpy(d_state, resume_state.data(), no * 12, hipMemcpyHostToDevice);
                check(fav_stream_set_state(fs, d_state, st), "fav_stream_set_state");
                hipStreamSynchronize(st); hipFree(d_state);
            }
            upload(cur, dset);
            hipStreamSynchronize(st_copy);               // the first frame's host buffers are malloc'ed: release them now
            first = false;
            res->setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();     // first frame read + every allocation
        } else if (cur.W != W || cur.H != H) die("frame size changed inside the sequence");
        issue();                                         // keep DEPTH loads in flight
        const auto t0 = std::chrono::steady_clock::now();
        static const bool loop_trace = diag_env("FAV_LOOP_TRACE") != nullptr;      // where a loop iteration's host time goes (frames 100-111)
        double tr_ms[6] = {0, 0, 0, 0, 0, 0};
        auto tr_mark = [&](int k) { if (loop_trace) tr_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        // frames i+1 (i+2): wait for the loader, upload, and start the consistency mask on the side queues so the (sequential, 1.5 ms)
        // 4-argument structure pass overlaps the network of the frames before it
        const Dev& dc = dev[dset];
        auto top_up = [&]() {
        while ((int)ahead.size() < LA && !drained) {
            FrameIn nxt;
            if (!idx_ok(i + inc * ((int)ahead.size() + 1)) || !pop_next(nxt)) { drained = true; break; }
            if (nxt.W != W || nxt.H != H) die("frame size changed inside the sequence");
            const int nset = (dset + 1 + (int)ahead.size()) % NDEV;
            const Dev& dn = dev[nset];
            upload(nxt, nset);
            if (fused_check && !nxt.single && (!quiet || o.i("structure") != 0)) {
                // quiet: the look-ahead may only start on inputs the host has seen arrive (0.35 ms of DMA; the GPU is busy with frame i-1)
                if (quiet && wait_event_sleeping(ev_up[nset], 50) != hipSuccess) die("GPU error while uploading a frame");
                check(fav_stream_prefetch_mask(fs, dn.frame, dn.bw, dn.fw, o.i("structure"), st), "fav_stream_prefetch_mask");
            }
            ahead.push_back(nxt);
        }
        };
        tr_mark(0);      // (the look-ahead is topped up BEHIND the frame's enqueue, also for the first frame: its enqueue is 15 ms of one-time
                         //  work -- activation arena, code objects -- that now runs while the loaders read and pin frames 2, 3, ...)
        if (quiet && wait_event_sleeping(ev_up[dset], 50) != hipSuccess) die("GPU error while uploading a frame");      // requested a frame ago: already there
        uint8_t* const d_out8 = gpu_png ? nullptr : d_out8s[done & 1];
        const bool teval = !o.s("temporal_eval_file").empty();
        if (teval && !d_prev) { if (hipMalloc((void**)&d_prev, (size_t)W * H * 12) || hipMalloc((void**)&d_cur, (size_t)W * H * 12)) die("hipMalloc failed"); }
        if (teval && !cur.single) check(fav_stream_get_state(fs, d_prev, st), "fav_stream_get_state");
        if (cur.single) {
            check(fav_stream_first_frame(fs, dc.frame, nullptr, d_out8, st), "fav_stream_first_frame");       // core:203-204
        } else if (fused_check) {
            check(fav_stream_next_frame_flow(fs, dc.frame, dc.bw, dc.fw, o.i("structure"), nullptr, d_out8, st), "fav_stream_next_frame_flow");
        } else {
            check(fav_stream_next_frame_cert(fs, dc.frame, dc.bw, dc.cert, nullptr, d_out8, st), "fav_stream_next_frame_cert");   // core:206-208
        }
        if (teval) {                                     // fav.lua:128-151 (third number)
            double tl = 0.0;
            if (!cur.single) {
                check(fav_stream_get_state(fs, d_cur, st), "fav_stream_get_state");
                check(fav_temporal_loss_host(d_prev, d_cur, dc.bw, fused_check ? fav_stream_last_mask(fs) : dc.cert, H, W, border, &tl, st), "fav_temporal_loss_host");
            }
            temporal.push_back(tl);
        }
        tr_mark(1);                                      // frame i enqueued
        const auto tw = std::chrono::steady_clock::now();
        uint8_t* hb = slots.try_take();                  // a pinned output slot nobody is reading ...
        if (!hb) hb = (int)h_out.size() < nslots ? new_slot() : slots.take();      // ... a new one while allowed, else wait for the PNG pool
        t_wait_writer += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        Pending now; now.valid = true; now.index = i; now.single = cur.single; now.hb = hb; now.ev = done & 1; now.t0 = t0;
        if (gpu_png) {
            // two frames ago this buffer's bytes left through the download queue: that copy must have finished (it has, long ago)
            if (png_copy_ev[now.ev]) {
                if (quiet) { if (wait_event_sleeping(png_copy_ev[now.ev], 50) != hipSuccess) die("GPU error while downloading a frame"); }
                else hipStreamWaitEvent(st, png_copy_ev[now.ev], 0);
            }
            if (quiet) h_png_size[now.ev] = 0u;          // the frame's last kernel overwrites it with the file size (host-mapped)
            // -png_overlap 1 (quiet mode): on the stream's encoder queue, next to frame i + 1's network -- the host learns of the file
            // through its size word either way
            if (quiet && png_overlap) check(fav_stream_encode_png_async(fs, d_png[now.ev], png_cap, &h_png_size[now.ev], st), "fav_stream_encode_png_async");
            else check(fav_stream_encode_png(fs, d_png[now.ev], png_cap, quiet ? &h_png_size[now.ev] : d_png_size[now.ev], st), "fav_stream_encode_png");
        }
        if (!quiet) hipEventRecord(ev_out[now.ev], st);  // the frame's 8-bit image / PNG is complete on the compute queue ...
        tr_mark(2);                                      // PNG encode enqueued
        top_up();                                        // behind frame i's kernels: the 0.45 ms this waits for the upload are not in front of them
        const auto tg = std::chrono::steady_clock::now();
        finish(pend);                                    // frame i-1: wait, report, hand to the PNG pool -- frame i's kernels are already queued
        t_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count();
        tr_mark(3);
        static const bool loop_trace_start = loop_trace && atoi(diag_env("FAV_LOOP_TRACE")) == 2;       // 2: the first 48 frames instead (start-up)
        if (loop_trace_start && done < 48)
            fprintf(stderr, "loop trace frame %d: at %.3f ms since start: look-ahead %.3f  frame enqueued %.3f  encode enqueued %.3f  previous frame finished %.3f ms\n", i,
                    std::chrono::duration<double, std::milli>(t0 - t_begin).count(), tr_ms[0], tr_ms[1], tr_ms[2], tr_ms[3]);
        if (loop_trace && !loop_trace_start && done >= 100 && done < 112)
            fprintf(stderr, "loop trace frame %d: look-ahead %.3f  frame enqueued %.3f  encode enqueued %.3f  previous frame finished %.3f ms\n", i, tr_ms[0], tr_ms[1], tr_ms[2], tr_ms[3]);
        if (!quiet) {
            hipStreamWaitEvent(st_down, ev_out[now.ev], 0);  // ... and leaves on the download queue
            if (gpu_png) hipMemcpyAsync(&h_png_size[now.ev], d_png_size[now.ev], 4, hipMemcpyDeviceToHost, st_down);      // (the bytes follow in finish(), exactly `size` of them)
            else hipMemcpyAsync(hb, d_out8, (size_t)Wo * Ho * 3, hipMemcpyDeviceToHost, st_down);
            hipEventRecord(ev_done[now.ev], st_down);
        }
        pend = now;
        if (cur.index >= 0) cur.release();               // malloc'ed (first frame); pinned sets are reused
        ++done;
        if (ahead.empty()) break;
        cur = ahead.front(); ahead.pop_front(); dset = (dset + 1) % NDEV;
    }
    finish(pend);
    const auto t_tail = std::chrono::steady_clock::now();       // the GPU is done: what follows is the PNG pool draining
    for (auto& pr : inflight) pr.second.get();
    for (auto& r : ready) if (r.index >= 0) r.release();
    fflush(stdout);
    writers.wait_below(0);
    if (!temporal.empty()) {                             // core.lua:231-238 layout: values joined by ';', then the mean
        FILE* f = fopen(o.s("temporal_eval_file").c_str(), "a");
        if (!f) die("cannot open " + o.s("temporal_eval_file"));
        double sum = 0.0;
        for (size_t k = 0; k < temporal.size(); ++k) { fprintf(f, "%s%.9g", k ? ";" : "", temporal[k]); sum += temporal[k]; }
        fprintf(f, "\n%.9g\n", sum / (double)temporal.size());
        fclose(f);
    }
    res->frames = done;
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    res->tail = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tail).count();
    res->wait_loader = t_wait_load; res->wait_gpu = t_gpu; res->wait_png = t_wait_writer;
    res->cpu_s = process_cpu_seconds() - cpu0; res->png_bytes = png_bytes_total.load();
    res->cpu_loaders = 1e-6 * cpu_loaders_us.load(); res->cpu_writers = 1e-6 * cpu_writers_us.load(); res->cpu_main = thread_cpu_seconds() - cpu_main0;
    res->cpu_writer_sync = 1e-6 * cpu_writer_sync_us.load();
    for (int k = 0; k < 2; ++k) { hipFree(d_png[k]); hipFree(d_png_size[k]); }
    if (h_png_size) hipHostFree(h_png_size);
    for (auto& kv : slot_ev) hipEventDestroy(kv.second);
    fav_stream_destroy(fs);
    hipFree(d_prev); hipFree(d_cur);
    hipFree(d_out8s[0]); hipFree(d_out8s[1]); for (auto& dv : dev) { hipFree(dv.frame); hipFree(dv.cert); hipFree(dv.bw); hipFree(dv.fw); }
    for (auto p : h_out) hipHostFree(p);
    for (auto& p : pin) { hipHostFree(p.frame); hipHostFree(p.bw); hipHostFree(p.fw); hipHostFree(p.cert); }
    for (auto& e : ev_up) hipEventDestroy(e);
    for (auto& e : ev_done) hipEventDestroy(e);
    for (auto& e : ev_out) hipEventDestroy(e);
    hipStreamDestroy(st); hipStreamDestroy(st_copy); hipStreamDestroy(st_down);
}

}  // namespace

int main(int argc, char** argv)
{
    Opt o;
    // fast_artistic_video.lua:21-67 (defaults as there, except -gpu: the reference defaults to the CPU, which this build does not have)
    o.v = {{"model_img", "models/checkpoint-candy-image.t7"}, {"model_vid", "models/checkpoint-candy-video.t7"}, {"num_frames", "9999"}, {"continue_with", "1"},
           {"input_pattern", ""}, {"output_prefix", "out"}, {"flow_pattern", ""}, {"occlusions_pattern", ""},
           {"occlusions_min_filter", "7"}, {"fill_occlusions", "vgg-mean"}, {"median_filter", "3"}, {"scale_factor", "1"},
           {"gpu", "0"}, {"backend", "cuda"}, {"use_cudnn", "1"}, {"cudnn_benchmark", "0"},
           {"flow_pattern_eval", ""}, {"occlusions_pattern_eval", ""}, {"evaluation_file", "evaluation.txt"},
           {"content_weights", "1.0"}, {"content_layers", "16"}, {"loss_network", "models/vgg16.t7"},
           {"style_image", "images/styles/candy.jpg"}, {"style_image_size", "256"}, {"style_weights", "1.0"},
           {"style_layers", "4,9,16,23"}, {"style_target_type", "gram"},
           // additive
           {"forward_flow_pattern", ""}, {"structure", "1"}, {"warp_border", "stn"}, {"poll_timeout", "3600"}, {"poll_settle", "1.0"},
           {"png_level", "1"}, {"png_encoder", "gpu"}, {"png_overlap", "0"}, {"writers", "0"}, {"timing", "0"}, {"temporal_eval_file", ""}, {"seed", "1"}, {"precision", "fp32"},
           {"streams", ""}, {"gpus", "1"}, {"force_dist", "0"}, {"dry_run", "0"}, {"shared_gpu", "0"}, {"pin_workers", "1"},
           // internal (set by the launcher for its workers)
           {"worker_rank", "-1"}, {"worker_world", "0"}, {"rccl_id_file", ""},
           // test hook: -load_probe <i> loads frame i's inputs exactly as the frame loop's loader threads do (polling included), prints one
           // JSON line with a hash of each and exits -- no device is touched (tests/test_cpu_poll.py)
           {"load_probe", "0"}};
    o.b = {{"invert_occlusion", false}, {"fix_occlusions", false}, {"backward", false}, {"create_inconsistent", false},
           {"evaluate", false}, {"invert_occlusion_eval", false}, {"fix_occlusions_eval", false}, {"backward_eval", false}};
    for (int a = 1; a < argc; ++a) {
        if (argv[a][0] != '-') die(std::string("invalid argument: ") + argv[a]);
        const std::string k = argv[a] + 1;
        if (o.b.count(k)) { o.b[k] = true; continue; }
        if (!o.v.count(k)) die("unknown option -" + k);
        if (a + 1 >= argc) die("missing value for -" + k);
        o.v[k] = argv[++a];
    }
    if (o.s("input_pattern").empty()) die("Must give -input_pattern");                                      // fav.lua:177-179
    const bool fused_check = !o.s("forward_flow_pattern").empty();
    if (!o.f("create_inconsistent") && (o.s("flow_pattern").empty() || (o.s("occlusions_pattern").empty() && !fused_check)))
        die("Must give -flow_pattern and -occlusions_pattern");                                              // fav.lua:180-182
    if (o.i("gpu") < 0) die("-gpu -1: this build has no CPU backend (the CPU restatement lives in oracle/ and is test infrastructure only)");
    if (o.f("evaluate")) die("-evaluate needs the VGG-16 perceptual-loss network: outside the hot-path scope (DESIGN.md)");
    if (o.d("scale_factor") != 1.0) die("-scale_factor != 1 is not supported");
    if (o.s("fill_occlusions") != "vgg-mean" && o.s("fill_occlusions") != "uniform-random") die("-fill_occlusions must be vgg-mean or uniform-random");
    if (o.s("png_encoder") != "gpu" && o.s("png_encoder") != "host") die("-png_encoder must be gpu (the file's bytes are produced on the device) or host (zlib, -png_level)");
    if (o.s("precision") != "fp32" && o.s("precision") != "bf16") die("-precision must be fp32 (parity mode) or bf16 (bf16 operands in the 3x3 residual convolutions)");
    if (o.i("load_probe") > 0) {
        favp::Poll poll; poll.timeout_s = o.d("poll_timeout"); poll.settle_s = std::max(0.0, o.d("poll_settle"));
        const auto t0 = std::chrono::steady_clock::now();
        FrameIn in = load_frame_inputs(o, poll, o.i("load_probe"), false, nullptr, 0);
        auto fnv = [](const void* p, size_t n) { unsigned long long h = 1469598103934665603ull; const uint8_t* b = (const uint8_t*)p;
                                                 if (!p) return 0ull; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; };
        const size_t n = (size_t)in.W * in.H;
        size_t cert255 = 0; if (in.cert) for (size_t i = 0; i < n; ++i) cert255 += in.cert[i] == 255;
        printf("{\"ok\": %s, \"W\": %d, \"H\": %d, \"seconds\": %.3f, \"frame\": \"%016llx\", \"cert\": \"%016llx\", \"cert_255\": %zu, \"bw\": \"%016llx\", \"fw\": \"%016llx\"}\n",
               in.ok ? "true" : "false", in.W, in.H, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
               fnv(in.frame, n * 3), fnv(in.cert, n), cert255, fnv(in.bw, n * 8), fnv(in.fw, n * 8));
        return in.ok ? 0 : 1;
    }
    const bool dry = o.i("dry_run") != 0;
    std::vector<std::string> streams = favl::split_list(o.s("streams"));
    const bool named = !streams.empty();
    if (!named) streams.push_back("");
    int world = std::max(1, o.i("gpus"));
    const int rank = o.i("worker_rank");
    static const char* const path_opts[] = {"input_pattern", "flow_pattern", "forward_flow_pattern", "occlusions_pattern", "output_prefix", "temporal_eval_file"};
    if (named && streams.size() > 1 && o.s("output_prefix").find("%S") == std::string::npos)
        die("-streams: -output_prefix must contain %S (the streams would overwrite each other's frames)");

    // ------------------------------------------------------------------------------------------ launcher
    if (rank < 0 && (world > 1 || o.i("force_dist"))) {
        if (!dry) {
            std::vector<std::string> models{o.s("model_vid")};
            if (o.s("model_img") != "self") models.push_back(o.s("model_img"));
            favl::validate_models(models);                        // before any worker exists: ERROR + nonzero exit, never a hang
            const int ndev = fav_device_count();
            if (ndev <= 0) die(std::string("ERROR: ") + fav_last_error());
            if (o.i("gpu") + world > ndev) die("-gpus " + o.s("gpus") + " from -gpu " + o.s("gpu") + ": only " + std::to_string(ndev) + " devices");
        }
        if (dry)      // what the host can sustain at most, whatever the GPUs do (the 8-GPU run explains itself)
            printf("{\"host_ceiling\": %s}\n", favl::host_ceiling_json(o.s("png_encoder") == "gpu" ? favl::HOST_CPU_MS_PER_FRAME_GPU_PNG : favl::HOST_CPU_MS_PER_FRAME_HOST_PNG,
                                                                       "default: measured on the project's MI355X box at 1280x720, fused 3-argument check").c_str());
        std::string xdir;
        const int worst = favl::spawn_workers(argc, argv, world, o.i("pin_workers") != 0, &xdir);
        if (o.i("timing") && !dry && worst == 0) favl::print_aggregate(xdir, world, streams.size());
        favl::remove_exchange_dir(xdir, world);
        return worst;
    }

    // ------------------------------------------------------------------------------------------ worker / single process
    const bool dist = rank >= 0;
    if (dist) world = o.i("worker_world");
    const int device = o.i("gpu") + (dist ? rank : 0);
    std::vector<std::string> mine;
    for (size_t s = 0; s < streams.size(); ++s) if (!dist || (int)(s % (size_t)world) == rank) mine.push_back(streams[s]);     // stream s -> GPU s mod N
    const int nwriters = favl::writer_budget(o.i("writers"), dist ? world : 1, dist && world > 1 && o.i("pin_workers") != 0);
    if (dry) {
        favl::dry_run_failure_hook(rank);
        std::string js = "{\"rank\": " + std::to_string(std::max(rank, 0)) + ", \"world\": " + std::to_string(dist ? world : 1) + ", \"device\": " + std::to_string(device) +
                         ", \"cpus\": " + std::to_string(favl::allowed_cpus().size()) + ", \"writers\": " + std::to_string(nwriters) + ", \"streams\": [";
        for (size_t k = 0; k < mine.size(); ++k) {
            js += std::string(k ? ", " : "") + "{\"name\": " + favl::json_str(mine[k]);
            for (const char* po : path_opts) js += std::string(", \"") + po + "\": " + favl::json_str(named ? favl::subst_stream(o.s(po), mine[k]) : o.s(po));
            js += "}";
        }
        printf("%s]}\n", js.c_str());
        return 0;
    }

    if (fav_device_count() <= 0) die(std::string("ERROR: ") + fav_last_error());
    if (hipSetDevice(device) != hipSuccess) die("cannot select GPU " + std::to_string(device));
    const bool want_img = o.s("model_img") != "self";
    fav_net* net = nullptr; fav_net* net_img = nullptr;                                                      // core.lua:39-66
    favl::DistTimes dist_times;
    if (dist) {
        // rank 0 parses; the other ranks never open the .t7
        favl::load_models_dist(rank, world, o.s("rccl_id_file"), device, o.s("model_vid"), want_img ? o.s("model_img") : std::string(), &net, &net_img,
                               mine.size(), nwriters, &dist_times);
    } else {
        if (fav_net_create(o.s("model_vid").c_str(), device, &net)) die(fav_last_error());                   // core.lua:39-43
        if (want_img && fav_net_create(o.s("model_img").c_str(), device, &net_img)) die(fav_last_error());
    }
    check(fav_net_set_precision(net, o.s("precision") == "bf16" ? FAV_PRECISION_BF16_OPERANDS : FAV_PRECISION_FP32), "fav_net_set_precision");
    if (o.i("shared_gpu")) { check(fav_net_set_shared_device(net, 1), "fav_net_set_shared_device"); if (net_img) check(fav_net_set_shared_device(net_img, 1), "fav_net_set_shared_device"); }
    printf("Model loaded.\n");
    if (net_img) printf("Model loaded.\n");

    int frames = 0; double seconds = 0, cpu_seconds = 0;
    for (const std::string& name : mine) {
        Opt os = o;
        if (named) for (const char* po : path_opts) os.v[po] = favl::subst_stream(o.s(po), name);
        StreamResult r;
        run_stream(os, net, net_img, nwriters, &r);
        frames += r.frames; seconds += r.seconds; cpu_seconds += r.cpu_s;
        if (o.i("timing"))
            printf("{%s\"frames\": %d, \"seconds\": %.4f, \"fps_end_to_end\": %.3f, \"wait_loader_s\": %.3f, \"h2d_gpu_d2h_s\": %.3f, \"wait_png_pool_s\": %.3f, \"setup_s\": %.3f, \"png_tail_s\": %.3f, \"png_writers\": %d, "
                   "\"png_encoder\": \"%s\", \"host_cpu_ms_per_frame\": %.3f, \"cpu_ms_per_frame_loaders\": %.3f, \"cpu_ms_per_frame_writers\": %.3f, \"cpu_ms_per_frame_writers_waiting_for_the_copy\": %.3f, \"cpu_ms_per_frame_main\": %.3f, "
                   "\"png_mb_per_frame\": %.3f, \"usable_cpus\": %d}\n",
                   named ? ("\"stream\": " + favl::json_str(name) + ", \"gpu\": " + std::to_string(device) + ", ").c_str() : "",
                   r.frames, r.seconds, r.frames / std::max(r.seconds, 1e-9), r.wait_loader, r.wait_gpu, r.wait_png, r.setup, r.tail, nwriters,
                   o.s("png_encoder").c_str(), 1e3 * r.cpu_s / std::max(r.frames, 1), 1e3 * r.cpu_loaders / std::max(r.frames, 1), 1e3 * r.cpu_writers / std::max(r.frames, 1), 1e3 * r.cpu_writer_sync / std::max(r.frames, 1),
                   1e3 * r.cpu_main / std::max(r.frames, 1), 1e-6 * (double)r.png_bytes / std::max(r.frames, 1), favl::effective_cpus());
    }
    check(fav_net_check(net), "at exit");
    if (net_img) check(fav_net_check(net_img), "at exit (image model)");
    if (o.i("timing")) printf("thread CPU seconds (live threads, whole process): %s\n", thread_cpu_report().c_str());
    if (dist && o.i("timing")) favl::write_worker_result(o.s("rccl_id_file"), rank, frames, seconds, cpu_seconds, dist_times);
    fflush(stdout);
    fav_net_destroy(net); fav_net_destroy(net_img);
    return 0;
}
