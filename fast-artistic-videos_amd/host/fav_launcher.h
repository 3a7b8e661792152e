// fav_launcher.h -- what the two drop-in CLIs (fav_stylize, fav_stylize_vr) share to run SEVERAL videos on SEVERAL GPUs
// (BASELINE configs 4 and 5; the reference starts one `th` process per video, stylizeVideo_deepflow.sh:87-96 /
// stylizeVRVideo_deepflow.sh:68-83, and picks its device in utils.setup_gpu, fast_artistic_video/utils.lua:43-66):
//   * `-streams a,b,...` + `%S` substitution in the path options, stream s -> worker s mod n;
//   * one worker PROCESS per GPU (fork + exec of the same binary with -worker_rank / -worker_world / -rccl_id_file);
//   * rank 0 parses the checkpoint(s); the packed blob (fav_net_pack_host) reaches the other ranks through ONE collective per
//     model, ncclBroadcast (RCCL over xGMI; ncclCommInitRank + a unique-id file) -- there is no other exchange: a video's frame i
//     needs only its own frame i-1;
//   * host thread budget per worker = usable CPUs (cgroup quota respected) / workers.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

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

// CPUs this process may actually use: hardware threads, capped by the cgroup CPU quota (containers: the GPU box of this project
// shows 256 hardware threads under a 16-CPU quota -- 32 deflate threads there only fight each other)
inline int effective_cpus()
{
    int n = std::max(1, (int)std::thread::hardware_concurrency());
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
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1ll, (quota + period - 1) / period));
    return n;
}

// PNG writer threads of one worker when `world` workers share the host: deflate is the slowest host stage (~25 ms per 1280x720
// frame and core at the default -png_level 1), so a worker gets its share of the usable CPUs (the loaders mostly wait on I/O)
inline int writer_budget(int requested, int world)
{
    if (requested > 0) return requested;
    return std::max(4, std::min(32, effective_cpus() / std::max(1, world)));
}

inline void ncheck(ncclResult_t r, const char* what) { if (r != ncclSuccess) die(std::string("RCCL: ") + what + ": " + ncclGetErrorString(r)); }

// rank 0 holds `blob`; on return every rank holds the same bytes.  ONE collective per model: ncclBroadcast of the packed
// checkpoint (6.7 MB, latency-bound) preceded by its 8-byte size.
inline void broadcast_blob(ncclComm_t comm, int rank, std::vector<uint8_t>& blob, hipStream_t st)
{
    unsigned long long n = rank == 0 ? blob.size() : 0, *d_n = nullptr;
    if (hipMalloc((void**)&d_n, 8) != hipSuccess) die("hipMalloc failed");
    hipMemcpy(d_n, &n, 8, hipMemcpyHostToDevice);
    ncheck(ncclBroadcast(d_n, d_n, 8, ncclUint8, 0, comm, st), "ncclBroadcast(size)");
    hipStreamSynchronize(st);
    hipMemcpy(&n, d_n, 8, hipMemcpyDeviceToHost); hipFree(d_n);
    if (n == 0) { blob.clear(); return; }
    uint8_t* d_b = nullptr;
    if (hipMalloc((void**)&d_b, n) != hipSuccess) die("hipMalloc failed");
    if (rank == 0) hipMemcpy(d_b, blob.data(), n, hipMemcpyHostToDevice);
    ncheck(ncclBroadcast(d_b, d_b, n, ncclUint8, 0, comm, st), "ncclBroadcast(blob)");
    if (hipStreamSynchronize(st) != hipSuccess) die("RCCL broadcast failed");
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


// launcher: one worker process per GPU, each re-executing this binary with its rank; returns the worst exit code.
// `idf` receives the path the workers exchange the RCCL unique id (and, with -timing, their results) through.
inline int spawn_workers(int argc, char** argv, int world, std::string* idf_out)
{
    char idf[] = "/tmp/fav_rccl_id_XXXXXX";
    const int fd = mkstemp(idf);
    if (fd < 0) die("cannot create the RCCL id file");
    close(fd); unlink(idf);                                   // the name is reused: rank 0 creates <name> atomically
    std::vector<pid_t> kids;
    fflush(stdout); fflush(stderr);
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();                             // (exec follows at once: no HIP state is shared)
        if (pid < 0) die("fork failed");
        if (pid == 0) {
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
    int worst = 0;
    for (pid_t k : kids) { int stt = 0; waitpid(k, &stt, 0); const int rc = WIFEXITED(stt) ? WEXITSTATUS(stt) : 128; if (rc > worst) worst = rc; }
    *idf_out = idf;
    return worst;
}

// after the workers: total frames / slowest worker's stylisation time (each worker left "<frames> <seconds>" in <idf>.rank<r>)
inline void print_aggregate(const std::string& idf, int world, size_t nstreams)
{
    int frames = 0; double secs = 0; std::string per = "";
    for (int r = 0; r < world; ++r) {
        const std::string f = idf + ".rank" + std::to_string(r);
        FILE* fp = fopen(f.c_str(), "r");
        int fr = 0; double sc = 0;
        if (fp) { if (fscanf(fp, "%d %lf", &fr, &sc) != 2) { fr = 0; sc = 0; } fclose(fp); unlink(f.c_str()); }
        frames += fr; secs = std::max(secs, sc);
        per += (r ? ", " : "") + std::to_string(fr ? fr / std::max(sc, 1e-9) : 0.0);
    }
    printf("{\"gpus\": %d, \"streams\": %zu, \"frames\": %d, \"seconds\": %.4f, \"fps_end_to_end\": %.3f, \"fps_per_gpu\": [%s], "
           "\"weights\": \"rank 0 parsed the .t7, ncclBroadcast of the packed blob\"}\n",
           world, nstreams, frames, secs, secs > 0 ? frames / secs : 0.0, per.c_str());
}

inline void write_worker_result(const std::string& idf, int rank, int frames, double seconds)
{
    const std::string f = idf + ".rank" + std::to_string(rank);
    if (FILE* fp = fopen(f.c_str(), "w")) { fprintf(fp, "%d %.6f\n", frames, seconds); fclose(fp); }
}

// worker side: rank 0 parses + packs, everybody receives the blob(s) over RCCL and builds its network(s) on `device`
inline void load_models_dist(int rank, int world, const std::string& idf, int device, const std::string& vid_path, const std::string& img_path,
                             fav_net** net, fav_net** net_img, size_t nstreams, int nwriters)
{
    ncclUniqueId id;
    if (rank == 0) {
        ncheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
        const std::string tmp = idf + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) die("cannot write " + tmp);
        fclose(f);
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
    ncheck(ncclCommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    hipStream_t bst; if (hipStreamCreate(&bst) != hipSuccess) die("hipStreamCreate failed");
    std::vector<uint8_t> blob, blob_img;
    if (rank == 0) { blob = pack_model(vid_path); if (!img_path.empty()) blob_img = pack_model(img_path); }
    const auto tb = std::chrono::steady_clock::now();
    broadcast_blob(comm, rank, blob, bst);
    broadcast_blob(comm, rank, blob_img, bst);
    const double bms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
    if (fav_net_create_from_blob(blob.data(), blob.size(), device, net)) die(fav_last_error());
    if (!blob_img.empty() && fav_net_create_from_blob(blob_img.data(), blob_img.size(), device, net_img)) die(fav_last_error());
    printf("[rank %d/%d gpu %d] weights: %zu B%s via ncclBroadcast from rank 0 in %.2f ms; %zu stream(s), %d PNG writers\n", rank, world, device,
           blob.size(), blob_img.empty() ? "" : " (+ image model)", bms, nstreams, nwriters);
    hipStreamDestroy(bst);
    ncclCommDestroy(comm);
    if (rank == 0) unlink(idf.c_str());
}

}  // namespace favl
