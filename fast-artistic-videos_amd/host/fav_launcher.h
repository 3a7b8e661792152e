// fav_launcher.h -- what the two drop-in CLIs (fav_stylize, fav_stylize_vr) share to run SEVERAL videos on SEVERAL GPUs
// (BASELINE configs 4 and 5; the reference starts one `th` process per video, stylizeVideo_deepflow.sh:87-96 /
// stylizeVRVideo_deepflow.sh:68-83, and picks its device in utils.setup_gpu, fast_artistic_video/utils.lua:43-66):
//   * `-streams a,b,...` + `%S` substitution in the path options, stream s -> worker s mod n;
//   * one worker PROCESS per GPU (fork + exec of the same binary with -worker_rank / -worker_world / -rccl_id_file);
//   * rank 0 parses the checkpoint(s); the packed blob (fav_net_pack_host) reaches the other ranks through ONE collective per
//     model, ncclBroadcast (RCCL over xGMI; ncclCommInitRank + a unique-id file) -- there is no other exchange: a video's frame i
//     needs only its own frame i-1;
//   * host thread budget per worker = usable CPUs (cgroup quota respected) / workers; each worker is pinned to its contiguous share of
//     the launcher's CPU affinity mask (-pin_workers 0 turns that off);
//   * failure handling: the launcher parses the checkpoint(s) itself before forking, reaps with waitpid(-1) and stops every other
//     worker when one fails (a rank blocked in an RCCL call never returns after a peer has gone); workers die with the launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fav.h"

namespace favl {

[[noreturn]] inline void die(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); exit(1); }

inline std::string subst_stream(std::string v, const std::string& name)
{
    for (size_t p = v.find("%S"); p != std::string::npos; p = v.find("%S", p + name.size())) v.replace(p, 2, name);
    return v;
}

inline std::vector<std::string> split_list(const std::string& v)
{
    std::vector<std::string> out; std::string cur;
    for (char c : v) { if (c == ',') { if (!cur.empty()) out.push_back(cur); cur.clear(); } else cur += c; }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

// CPU quota of the cgroup in whole CPUs (rounded up), or a very large number when there is none
inline int quota_cpus()
{
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
        char q[64] = "";
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
    }
    if (quota > 0 && period > 0) return (int)std::max(1ll, (quota + period - 1) / period);
    return 1 << 20;
}

// CPUs the calling process may run on (its affinity mask), in ascending order
inline std::vector<int> allowed_cpus()
{
    std::vector<int> out;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) out.push_back(c);
    return out;
}

// CPUs this process may actually use: its affinity mask, capped by the cgroup CPU quota (containers: the GPU box of this project
// shows 256 hardware threads under a 16-CPU quota -- 32 deflate threads there only fight each other)
inline int effective_cpus()
{
    const int aff = (int)allowed_cpus().size();
    return std::max(1, std::min(aff > 0 ? aff : (int)std::thread::hardware_concurrency(), quota_cpus()));
}

// PNG writer threads of one worker when `world` workers share the host: the worker's share of the affinity mask (already narrowed
// when the launcher pinned it) and of the quota (which cannot be narrowed per process: divide)
inline int writer_budget(int requested, int world, bool pinned = false)
{
    if (requested > 0) return requested;
    world = std::max(1, world);
    int aff = std::max(1, (int)allowed_cpus().size());
    if (!pinned) aff = std::max(1, aff / world);
    const int q = std::max(1, quota_cpus() / world);
    return std::max(2, std::min(32, std::min(aff, q)));
}

inline void ncheck(ncclResult_t r, const char* what) { if (r != ncclSuccess) die(std::string("RCCL: ") + what + ": " + ncclGetErrorString(r)); }

// Watchdog of the job's only collectives.  The communicator is NON-BLOCKING (ncclConfig_t::blocking = 0): ncclCommInitRankConfig and
// ncclBroadcast return at once and the worker polls ncclCommGetAsyncError (and the stream) with a deadline, so a rank that never shows
// up or a half-up xGMI fabric ends the job with a message instead of a hang inside librccl (the launcher then stops the other workers).
// FAV_RCCL_TIMEOUT_S overrides the 60 s deadline.
inline double rccl_timeout_s()
{
    const char* e = getenv("FAV_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0 ? v : 60.0;
}

inline void nccl_wait(ncclComm_t comm, ncclResult_t first, const char* what, int rank, hipStream_t st = nullptr, bool wait_stream = false)
{
    if (first != ncclSuccess && first != ncclInProgress) die(std::string("RCCL: ") + what + ": " + ncclGetErrorString(first));
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = rccl_timeout_s();
    for (;;) {
        ncclResult_t state = ncclSuccess;
        ncheck(ncclCommGetAsyncError(comm, &state), "ncclCommGetAsyncError");
        if (state != ncclSuccess && state != ncclInProgress) die(std::string("RCCL: ") + what + " failed: " + ncclGetErrorString(state));
        if (state == ncclSuccess) {
            if (!wait_stream) return;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return;
            if (q != hipErrorNotReady) die(std::string("RCCL: ") + what + ": " + hipGetErrorString(q));
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            fprintf(stderr, "[rank %d] RCCL: %s did not complete within %.0f s -- a rank is missing or the GPU fabric is not fully up "
                            "(see the other workers' messages; `rocm-smi --showtopo`); aborting the communicator\n", rank, what, limit);
            (void)ncclCommAbort(comm);
            exit(4);
        }
        usleep(500);
    }
}

// rank 0 holds `blob`; on return every rank holds the same bytes.  ONE collective per model: ncclBroadcast of the packed
// checkpoint (6.7 MB, latency-bound) preceded by its 8-byte size.
inline void broadcast_blob(ncclComm_t comm, int rank, std::vector<uint8_t>& blob, hipStream_t st)
{
    unsigned long long n = rank == 0 ? blob.size() : 0, *d_n = nullptr;
    if (hipMalloc((void**)&d_n, 8) != hipSuccess) die("hipMalloc failed");
    hipMemcpy(d_n, &n, 8, hipMemcpyHostToDevice);
    nccl_wait(comm, ncclBroadcast(d_n, d_n, 8, ncclUint8, 0, comm, st), "ncclBroadcast(size)", rank, st, true);
    hipMemcpy(&n, d_n, 8, hipMemcpyDeviceToHost); hipFree(d_n);
    if (n == 0) { blob.clear(); return; }
    uint8_t* d_b = nullptr;
    if (hipMalloc((void**)&d_b, n) != hipSuccess) die("hipMalloc failed");
    if (rank == 0) hipMemcpy(d_b, blob.data(), n, hipMemcpyHostToDevice);
    nccl_wait(comm, ncclBroadcast(d_b, d_b, n, ncclUint8, 0, comm, st), "ncclBroadcast(blob)", rank, st, true);
    blob.resize(n);
    hipMemcpy(blob.data(), d_b, n, hipMemcpyDeviceToHost); hipFree(d_b);
}

inline std::vector<uint8_t> pack_model(const std::string& path)
{
    size_t bytes = 0;
    if (fav_net_pack_host(path.c_str(), nullptr, 0, &bytes)) die(fav_last_error());                         // core.lua:39-43
    std::vector<uint8_t> blob(bytes);
    if (fav_net_pack_host(path.c_str(), blob.data(), blob.size(), &bytes)) die(fav_last_error());
    return blob;
}

inline std::string json_str(const std::string& v) { std::string o = "\""; for (char c : v) { if (c == '"' || c == '\\') o += '\\'; o += c; } return o + "\""; }


// ---- exchange directory -----------------------------------------------------------------------------------------------------
// Everything the launcher and its workers exchange through the file system (the RCCL unique id, the per-rank results) lives in a
// PRIVATE directory: mkdtemp (mode 0700) under $TMPDIR or /tmp, files created with O_CREAT|O_EXCL|O_NOFOLLOW, the directory removed
// by the launcher on every exit path.  (Round 2 reused a predictable, unlinked mkstemp name in world-writable /tmp.)
inline std::string make_exchange_dir()
{
    const char* base = getenv("TMPDIR");
    std::string t = std::string(base && *base ? base : "/tmp") + "/fav_launch_XXXXXX";
    std::vector<char> buf(t.begin(), t.end()); buf.push_back(0);
    if (!mkdtemp(buf.data())) die("cannot create the launcher's exchange directory under " + t);
    return buf.data();
}

inline bool write_private_file(const std::string& path, const void* data, size_t n)
{
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    const bool ok = write(fd, data, n) == (ssize_t)n;
    return close(fd) == 0 && ok;
}

inline void remove_exchange_dir(const std::string& dir, int world)
{
    if (dir.empty()) return;
    unlink((dir + "/id").c_str()); unlink((dir + "/id.tmp").c_str());
    for (int r = 0; r < world; ++r) unlink((dir + "/id.rank" + std::to_string(r)).c_str());
    rmdir(dir.c_str());
}

// worker r of `world` gets the r-th contiguous share of the launcher's CPUs, so that the loaders / PNG writers of different workers
// do not migrate across each other's cores (a quota, unlike an affinity mask, cannot be split this way: writer_budget divides it)
inline std::vector<int> worker_cpu_share(const std::vector<int>& cpus, int r, int world)
{
    if ((int)cpus.size() < world) return cpus;                 // fewer CPUs than workers: everybody shares all of them
    const size_t lo = cpus.size() * (size_t)r / (size_t)world, hi = cpus.size() * (size_t)(r + 1) / (size_t)world;
    return std::vector<int>(cpus.begin() + lo, cpus.begin() + hi);
}

// launcher: one worker process per GPU, each re-executing this binary with its rank; returns the worst exit code.
// validate_models (called by the CLIs before anything else): the checkpoints every worker depends on are parsed by the launcher
// first (host-only), so a typo or a damaged file ends the job with the reader's message before any process waits inside a
// collective.  The first worker that fails takes its siblings down (SIGTERM,
// SIGKILL after a grace period): a rank blocked in ncclCommInitRank / ncclBroadcast never returns once a peer is gone.
// `dir_out` receives the exchange directory (remove_exchange_dir after reading the results).
inline void validate_models(const std::vector<std::string>& models)
{
    for (const std::string& m : models) if (!m.empty()) (void)pack_model(m);      // dies with fav_last_error() (core.lua:39-43)
}

inline int spawn_workers(int argc, char** argv, int world, bool pin, std::string* dir_out)
{
    const std::string dir = make_exchange_dir();
    const std::string idf = dir + "/id";
    const std::vector<int> cpus = allowed_cpus();
    std::vector<pid_t> kids;
    fflush(stdout); fflush(stderr);
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();                             // (exec follows at once: no HIP state is shared)
        if (pid < 0) { for (pid_t k : kids) kill(k, SIGTERM); remove_exchange_dir(dir, world); die("fork failed"); }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGTERM);                 // a worker does not outlive its launcher
            if (pin && world > 1) {
                const std::vector<int> mine = worker_cpu_share(cpus, r, world);
                cpu_set_t set; CPU_ZERO(&set);
                for (int c : mine) CPU_SET(c, &set);
                if (!mine.empty()) (void)sched_setaffinity(0, sizeof set, &set);
            }
            std::vector<std::string> args(argv, argv + argc);
            args.insert(args.end(), {"-worker_rank", std::to_string(r), "-worker_world", std::to_string(world), "-rccl_id_file", idf});
            std::vector<char*> av;
            for (auto& a : args) av.push_back(const_cast<char*>(a.c_str()));
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            perror("execv"); _exit(127);
        }
        kids.push_back(pid);
    }
    int worst = 0; size_t alive = kids.size(); bool killed = false;
    auto t_kill = std::chrono::steady_clock::now();
    while (alive > 0) {
        int stt = 0;
        const pid_t k = waitpid(-1, &stt, killed ? WNOHANG : 0);
        if (k > 0) {
            auto it = std::find(kids.begin(), kids.end(), k);
            if (it == kids.end()) continue;
            *it = -1; --alive;
            const int rc = WIFEXITED(stt) ? WEXITSTATUS(stt) : 128 + (WIFSIGNALED(stt) ? WTERMSIG(stt) : 0);
            if (rc != 0 && !killed) {
                worst = rc;                                   // the first failure is the job's exit code
                fprintf(stderr, "worker %d exited with status %d: stopping the other workers\n", (int)(it - kids.begin()), rc);
                for (pid_t o : kids) if (o > 0) kill(o, SIGTERM);
                killed = true; t_kill = std::chrono::steady_clock::now();
            }
        } else if (k < 0 && errno != EINTR) {
            break;                                            // ECHILD: nothing left to wait for
        } else if (killed) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_kill).count() > 5.0)
                for (pid_t o : kids) if (o > 0) kill(o, SIGKILL);
            usleep(20000);
        }
    }
    *dir_out = dir;
    return worst;
}

// test hook of the reaper (CPU suite, -dry_run only): FAV_TEST_WORKER_FAIL=<rank> makes that worker exit with status 3 and the
// others block like a rank inside a collective whose peer has gone
inline void dry_run_failure_hook(int rank)
{
    const char* e = getenv("FAV_TEST_WORKER_FAIL");
    if (!e || rank < 0) return;
    if (atoi(e) == rank) { fprintf(stderr, "[rank %d] simulated failure\n", rank); exit(3); }
    sleep(120);
}

// Host-side ceiling of a job: the CPUs this job may use / the CPU time one frame costs the host (file reads into pinned memory,
// write() of the PNG, submission) -- what the end-to-end rate cannot exceed however many GPUs work.  Measured on the MI355X box
// of this project at 1280x720 with the fused 3-argument check (profiles/r03*_bench.log, `host_cpu_ms_per_frame`): the defaults below.
constexpr double HOST_CPU_MS_PER_FRAME_GPU_PNG = 3.0;       // -png_encoder gpu: 2.2 file reads (17.5 MB) + 0.5 write (2.7 MB) + 0.2 submission (profiles/e2e_r03r.log)
constexpr double HOST_CPU_MS_PER_FRAME_HOST_PNG = 24.6;     // -png_encoder host -png_level 1 (zlib Sub + Z_RLE)
inline std::string host_ceiling_json(double cpu_ms_per_frame, const char* source)
{
    const int cpus = effective_cpus();
    char buf[512];
    snprintf(buf, sizeof buf, "{\"usable_cpus\": %d, \"host_cpu_ms_per_frame\": %.3f, \"host_bound_ceiling_fps\": %.1f, \"source\": \"%s\"}",
             cpus, cpu_ms_per_frame, cpu_ms_per_frame > 0 ? cpus * 1e3 / cpu_ms_per_frame : 0.0, source);
    return buf;
}

// after the workers: total frames / slowest worker's stylisation time (each worker left "<frames> <seconds> <cpu seconds>" in
// <idf>.rank<r>), and the host-bound ceiling those CPU seconds imply
inline void print_aggregate(const std::string& dir, int world, size_t nstreams)
{
    const std::string idf = dir + "/id";
    int frames = 0; double secs = 0, cpu = 0, bcast_ms = 0, init_ms = 0; std::string per = "";
    for (int r = 0; r < world; ++r) {
        const std::string f = idf + ".rank" + std::to_string(r);
        FILE* fp = fopen(f.c_str(), "r");
        int fr = 0; double sc = 0, cs = 0, bm = 0, im = 0;
        if (fp) { if (fscanf(fp, "%d %lf %lf %lf %lf", &fr, &sc, &cs, &bm, &im) < 2) { fr = 0; sc = 0; cs = 0; } fclose(fp); unlink(f.c_str()); }
        frames += fr; secs = std::max(secs, sc); cpu += cs; bcast_ms = std::max(bcast_ms, bm); init_ms = std::max(init_ms, im);
        per += (r ? ", " : "") + std::to_string(fr ? fr / std::max(sc, 1e-9) : 0.0);
    }
    const double ms = frames ? 1e3 * cpu / frames : 0.0;
    printf("{\"gpus\": %d, \"streams\": %zu, \"frames\": %d, \"seconds\": %.4f, \"fps_end_to_end\": %.3f, \"fps_per_gpu\": [%s], "
           "\"host_cpu_ms_per_frame\": %.3f, \"host_ceiling\": %s, "
           "\"weights\": \"rank 0 parsed the .t7, ncclBroadcast of the packed blob\", \"weight_broadcast_ms\": %.3f, \"rccl_comm_init_ms\": %.1f}\n",
           world, nstreams, frames, secs, secs > 0 ? frames / secs : 0.0, per.c_str(), ms, host_ceiling_json(ms, "measured: CPU seconds of all workers / frames").c_str(),
           bcast_ms, init_ms);
}

struct DistTimes { double bcast_ms = 0, init_ms = 0; };      // the slowest rank's values end up in the aggregate line

inline void write_worker_result(const std::string& idf, int rank, int frames, double seconds, double cpu_seconds = 0.0, DistTimes dt = DistTimes())
{
    const std::string f = idf + ".rank" + std::to_string(rank);
    char line[160]; const int n = snprintf(line, sizeof line, "%d %.6f %.6f %.4f %.3f\n", frames, seconds, cpu_seconds, dt.bcast_ms, dt.init_ms);
    if (!write_private_file(f, line, (size_t)n)) fprintf(stderr, "cannot write %s\n", f.c_str());
}

// worker side: rank 0 parses + packs, everybody receives the blob(s) over RCCL and builds its network(s) on `device`
inline void load_models_dist(int rank, int world, const std::string& idf, int device, const std::string& vid_path, const std::string& img_path,
                             fav_net** net, fav_net** net_img, size_t nstreams, int nwriters, DistTimes* times = nullptr)
{
    ncclUniqueId id;
    std::vector<uint8_t> blob, blob_img;
    if (rank == 0) {
        // parse and pack BEFORE the id is published: if the checkpoint is unusable this rank dies here, the launcher stops the
        // others (they are still polling for the id file, not inside a collective)
        blob = pack_model(vid_path); if (!img_path.empty()) blob_img = pack_model(img_path);
        ncheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
        const std::string tmp = idf + ".tmp";
        if (!write_private_file(tmp, &id, sizeof id)) die("cannot write " + tmp);
        if (rename(tmp.c_str(), idf.c_str())) die("cannot publish " + idf);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        FILE* f = nullptr;
        while (!(f = fopen(idf.c_str(), "rb"))) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120) die("timed out waiting for the RCCL id of rank 0");
            usleep(2000);
        }
        if (fread(&id, sizeof id, 1, f) != 1) die("short RCCL id file");
        fclose(f);
    }
    ncclComm_t comm;
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;                                           // every call returns at once; nccl_wait polls with a deadline
    const auto ti = std::chrono::steady_clock::now();
    const ncclResult_t ir = ncclCommInitRankConfig(&comm, world, id, rank, &cfg);      // (sets `comm` before it returns, also when in progress)
    nccl_wait(comm, ir, "ncclCommInitRank", rank);
    const double ims = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti).count();
    hipStream_t bst; if (hipStreamCreate(&bst) != hipSuccess) die("hipStreamCreate failed");
    const auto tb = std::chrono::steady_clock::now();
    broadcast_blob(comm, rank, blob, bst);
    broadcast_blob(comm, rank, blob_img, bst);
    const double bms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
    if (fav_net_create_from_blob(blob.data(), blob.size(), device, net)) die(fav_last_error());
    if (!blob_img.empty() && fav_net_create_from_blob(blob_img.data(), blob_img.size(), device, net_img)) die(fav_last_error());
    printf("[rank %d/%d gpu %d] weights: %zu B%s via ncclBroadcast from rank 0 in %.2f ms (communicator up in %.0f ms); %zu stream(s), %d PNG writers\n", rank, world, device,
           blob.size(), blob_img.empty() ? "" : " (+ image model)", bms, ims, nstreams, nwriters);
    if (times) { times->bcast_ms = bms; times->init_ms = ims; }
    hipStreamDestroy(bst);
    ncclCommDestroy(comm);
}

}  // namespace favl
