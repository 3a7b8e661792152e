// consistencyChecker -- drop-in for the reference's stand-alone executable
//   consistencyChecker <flow1.flo> <flow2.flo> <out.pgm> [<image.ppm>]
// (consistencyChecker/consistencyChecker.cpp:136-171; called by makeOptFlow_deepflow.sh:59-60 and
// video_dataset/make_occlusions.sh:31-36).  Same argv, same PGM bytes, exit code 0; the mask is computed
// on the MI355X through libfav (fav_consistency_u8).  Deliberate deviations: the output is written
// once, atomically (the reference first writes an all-255 placeholder, :152, which races with the
// polling consumer), and failures return a non-zero exit code instead of asserting.
// Device selection: environment variable FAV_GPU (default 0).
//
// Additive: `consistencyChecker -batch <list.txt>` -- every line of the list is one call's arguments
// (`<flow1.flo> <flow2.flo> <out.pgm> [<image.ppm>]`, blank-separated; `-` reads the list from stdin), processed in ONE process:
// a single call pays for a HIP context (tens of milliseconds, more than the reference's whole CPU run of a 1280x720 pair), a list of
// N pairs pays it once.  Each line's output path is printed as in the single-call form, one per line.  INTEGRATION.md section 1
// shows the two-line change to makeOptFlow_deepflow.sh:45-66.
//
// Round 5 -- the single call without that change: a call of the four-argument form costs 0.3 s as a fresh process (HIP runtime start-up:
// profiles/r6*_checker_breakdown.log) against 0.09 / 0.14 s for the reference's CPU binary, and makeOptFlow_deepflow.sh:59-60 makes two
// such calls per frame.  The first call therefore leaves a RESIDENT HELPER behind: it forks, the child detaches, creates the GPU context
// once and serves later calls over a unix-domain socket (same argv, same bytes, same exit codes; one request at a time, like a
// sequence of processes).  A later call connects, hands over its arguments (paths made absolute against ITS working directory), waits
// for the status and prints the output name -- no HIP in the calling process at all.  The helper exits after FAV_CC_IDLE_S seconds
// without a request (default 120) and removes its socket.  FAV_CC_DAEMON=0 switches all of this off (every call computes in its own
// process, as in round 4).  The socket lives in a directory only the calling user can enter (mode 0700, ownership checked):
// $XDG_RUNTIME_DIR/fav-cc or /tmp/fav-cc-<uid>; one helper per GPU (FAV_GPU).  FAV_CC_TIMING=1 prints where a call's time goes.
// Round 6: every request carries the caller's BUILD ID (this executable + the libfav.so it loaded + the FAV_* settings in its environment,
// which the library reads once per process): a helper left behind by an older build, or started under other FAV_* settings, answers
// "stale" and leaves instead of serving the call with the old code; the caller computes in its own process and the next call starts a
// fresh helper.  A request the helper took but never answered (FAV_CC_REPLY_S) is an ERROR (exit code 3), not a second computation of
// the same output next to it; outputs are written through a per-process temporary name and renamed.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <dlfcn.h>
#include <chrono>
#include <csignal>
#include <fcntl.h>
#include <poll.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fav.h"

static int fail(const char* what)
{
    fprintf(stderr, "consistencyChecker: %s: %s\n", what, fav_last_error());
    return 1;
}

namespace {
// one context; device buffers and PINNED host staging kept between pairs of the same size (a pageable read + copy of 17.5 MB costs
// more than the mask itself: 29 ms per pair against 6)
struct Batch {
    int W = 0, H = 0;
    float *d1 = nullptr, *d2 = nullptr; uint8_t *dimg = nullptr, *dout = nullptr; void* ws = nullptr; size_t wsb = 0;
    float *h1 = nullptr, *h2 = nullptr; uint8_t *himg = nullptr, *hout = nullptr;
    void release()
    {
        hipFree(d1); hipFree(d2); hipFree(dimg); hipFree(dout); hipFree(ws); d1 = d2 = nullptr; dimg = dout = nullptr; ws = nullptr;
        hipHostFree(h1); hipHostFree(h2); hipHostFree(himg); hipHostFree(hout); h1 = h2 = nullptr; himg = hout = nullptr;
        W = H = 0;
    }
    static bool flo_size(const char* path, int* w, int* h)      // the 12-byte header (consistencyChecker.cpp:16-36: tag, width, height)
    {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        unsigned char hd[12];
        const bool ok = fread(hd, 1, 12, f) == 12;
        fclose(f);
        if (!ok) return false;
        memcpy(w, hd + 4, 4); memcpy(h, hd + 8, 4);
        return *w > 0 && *h > 0 && (long long)*w * *h < (1ll << 28);
    }
    // one call; a failure's message goes to `err` (the helper hands it back to the caller, who prints it)
    std::string err;
    int failf(const char* what) { err = std::string("consistencyChecker: ") + what + ": " + fav_last_error() + "\n"; return 1; }
    int run(const std::vector<std::string>& a)
    {
        err.clear();
        int w1, h1s;
        if (!flo_size(a[0].c_str(), &w1, &h1s)) { err = "consistencyChecker: cannot read " + a[0] + "\n"; return 1; }
        const size_t n = (size_t)w1 * h1s;
        if (w1 != W || h1s != H) {
            release();
            W = w1; H = h1s;
            wsb = fav_consistency_workspace_bytes(W, H, 1);
            if (hipMalloc((void**)&d1, n * 8) || hipMalloc((void**)&d2, n * 8) || hipMalloc((void**)&dout, n) || hipMalloc((void**)&dimg, n * 3) ||
                (wsb && hipMalloc(&ws, wsb)) || hipHostMalloc((void**)&h1, n * 8, hipHostMallocDefault) || hipHostMalloc((void**)&h2, n * 8, hipHostMallocDefault) ||
                hipHostMalloc((void**)&himg, n * 3, hipHostMallocDefault) || hipHostMalloc((void**)&hout, n, hipHostMallocDefault)) {
                err = "consistencyChecker: out of memory\n"; release(); return 1; }
        }
        int w, h, ch;
        if (fav_read_flo_into_host(a[0].c_str(), h1, n * 2, &w, &h)) return failf(a[0].c_str());
        if (fav_read_flo_into_host(a[1].c_str(), h2, n * 2, &w, &h)) return failf(a[1].c_str());
        if (w != W || h != H) { err = "consistencyChecker: flow sizes differ\n"; return 1; }                  // :144-145
        const bool img = a.size() >= 4;
        if (img) {
            if (fav_read_pnm_into_host(a[3].c_str(), himg, n * 3, &w, &h, &ch)) return failf(a[3].c_str());
            if (w != W || h != H || ch != 3) { err = "consistencyChecker: image must be a P6 of the flow's size\n"; return 1; }
        }
        hipMemcpyAsync(d1, h1, n * 8, hipMemcpyHostToDevice, nullptr);
        hipMemcpyAsync(d2, h2, n * 8, hipMemcpyHostToDevice, nullptr);
        if (img) hipMemcpyAsync(dimg, himg, n * 3, hipMemcpyHostToDevice, nullptr);
        if (fav_consistency_u8(d1, d2, img ? dimg : nullptr, dout, W, H, ws, wsb, nullptr)) return failf("fav_consistency_u8");
        if (hipMemcpyAsync(hout, dout, n, hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
            err = "consistencyChecker: device error\n"; return 1; }
        if (fav_write_pgm_host(a[2].c_str(), hout, W, H)) return failf(a[2].c_str());
        return 0;
    }
};

int batch_main(const char* list)
{
    FILE* f = strcmp(list, "-") == 0 ? stdin : fopen(list, "r");
    if (!f) { fprintf(stderr, "consistencyChecker: cannot open %s\n", list); return 1; }
    if (fav_device_count() <= 0) return fail("device");
    const char* g = getenv("FAV_GPU");
    if (hipSetDevice(g ? atoi(g) : 0) != hipSuccess) { fprintf(stderr, "consistencyChecker: cannot select GPU\n"); return 1; }
    Batch b;
    char line[16384];
    int rc = 0, lineno = 0;
    while (fgets(line, sizeof line, f)) {
        ++lineno;
        std::istringstream is(line);
        std::vector<std::string> a; std::string tok;
        while (is >> tok) a.push_back(tok);
        if (a.empty() || a[0][0] == '#') continue;
        if (a.size() < 3 || a.size() > 4) { fprintf(stderr, "consistencyChecker: %s:%d: expected <flow1.flo> <flow2.flo> <out.pgm> [<image.ppm>]\n", list, lineno); rc = 2; break; }
        if ((rc = b.run(a)) != 0) { fputs(b.err.c_str(), stderr); break; }      // as a sequence of single calls under `set -e`: stop at the first failure
        printf("%s\n", a[2].c_str()); fflush(stdout);
    }
    if (f != stdin) fclose(f);
    b.release();
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------------------
// resident helper (see the header): socket, protocol, client and server sides
// ------------------------------------------------------------------------------------------------------------------------------
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int gpu_index() { const char* g = getenv("FAV_GPU"); return g ? atoi(g) : 0; }

// the socket's directory: created 0700, and only trusted when it is a directory of ours that nobody else can enter
bool socket_paths(std::string& dir, std::string& sock, std::string& lock)
{
    const char* x = getenv("XDG_RUNTIME_DIR");
    char buf[256];
    if (x && *x) snprintf(buf, sizeof buf, "%s/fav-cc", x);
    else snprintf(buf, sizeof buf, "/tmp/fav-cc-%u", (unsigned)getuid());
    dir = buf;
    if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return false;
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return false;
    // one helper per device AS THE CALLER SEES IT: FAV_GPU indexes the devices the runtime's visibility variables leave, so their values are
    // part of the name (a call with another HIP_VISIBLE_DEVICES must not reach a helper that sits on another device)
    char tag[24] = "";
    {
        uint32_t hsh = 2166136261u; bool any = false;
        for (const char* v : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL", "CUDA_VISIBLE_DEVICES"}) {
            const char* e = getenv(v);
            if (e && *e) any = true;
            for (const char* c = e ? e : ""; *c; ++c) { hsh ^= (unsigned char)*c; hsh *= 16777619u; }
            hsh ^= ';'; hsh *= 16777619u;
        }
        if (any) snprintf(tag, sizeof tag, "-%08x", hsh);
    }
    snprintf(buf, sizeof buf, "/gpu%d%s.sock", gpu_index(), tag); sock = dir + buf;
    snprintf(buf, sizeof buf, "/gpu%d%s.lock", gpu_index(), tag); lock = dir + buf;
    return sock.size() < sizeof(((sockaddr_un*)nullptr)->sun_path);
}

bool write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n) { const ssize_t k = send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } c += k; n -= (size_t)k; }
    return true;
}
bool read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n) { const ssize_t k = recv(fd, c, n, 0); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } c += k; n -= (size_t)k; }
    return true;
}
constexpr uint32_t CC_MAGIC = 0x32434346u;      // "FCC2": the request carries the caller's build id
constexpr int32_t CC_STALE = -1000;             // reply: "I am another build / was started under other settings -- I am leaving"

// What a helper IS: this executable, the libfav.so it loaded and the FAV_* settings it was started under (the library reads its
// switches once per process).  A caller that differs in any of them -- a rebuilt bin/consistencyChecker, an upgraded library, another
// FAV_* environment -- must not be served by the old helper: the id travels in every request, a helper that sees a foreign one answers
// CC_STALE and leaves (the caller computes in its own process, the next call starts a fresh helper).
uint64_t build_id()
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    auto mix_file = [&](const char* path) {
        struct stat st;
        if (stat(path, &st) == 0) { mix(&st.st_dev, sizeof st.st_dev); mix(&st.st_ino, sizeof st.st_ino); mix(&st.st_size, sizeof st.st_size); mix(&st.st_mtim, sizeof st.st_mtim); }
        else mix(path, strlen(path));
    };
    mix_file("/proc/self/exe");
    Dl_info di;
    if (dladdr(reinterpret_cast<const void*>(&fav_version), &di) && di.dli_fname) mix_file(di.dli_fname);
    const int v = fav_version(); mix(&v, sizeof v);
    // FAV_* settings, sorted (the order of `environ` is the caller's business); the helper's own knobs and FAV_GPU (part of the socket's name) excepted
    std::vector<std::string> env;
    for (char** e = environ; e && *e; ++e)
        if (strncmp(*e, "FAV_", 4) == 0 && strncmp(*e, "FAV_CC_", 7) != 0 && strncmp(*e, "FAV_GPU=", 8) != 0) env.push_back(*e);
    for (size_t i = 1; i < env.size(); ++i) for (size_t j = i; j > 0 && env[j] < env[j - 1]; --j) std::swap(env[j], env[j - 1]);
    for (const std::string& e : env) mix(e.c_str(), e.size() + 1);
    return h;
}

int connect_helper(const std::string& sock)
{
    const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    sockaddr_un sa{}; sa.sun_family = AF_UNIX; strncpy(sa.sun_path, sock.c_str(), sizeof(sa.sun_path) - 1);
    if (connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof sa) != 0) { close(fd); return -1; }
    return fd;
}

// client: 0 = served (status in *rc); -1 = not served and nothing of this request is under way in the helper (the caller computes in its own
// process: the request could not be sent, the helper is another build, or it rejected the request as malformed); -2 = the request WAS
// handed over and no answer came back (time-out, connection lost): the helper may still be working on it, so the caller reports the
// failure instead of computing the same output a second time next to it
int ask_helper(int fd, const std::vector<std::string>& a, int* rc)
{
    const uint64_t id = build_id();
    uint32_t hd[4] = {CC_MAGIC, (uint32_t)a.size(), (uint32_t)id, (uint32_t)(id >> 32)};
    if (!write_all(fd, hd, sizeof hd)) return -1;
    for (const std::string& s : a) { const uint32_t n = (uint32_t)s.size(); if (!write_all(fd, &n, 4) || !write_all(fd, s.data(), n)) return -1; }
    int32_t status; uint32_t mlen;
    if (!read_all(fd, &status, 4) || !read_all(fd, &mlen, 4) || mlen > (1u << 20)) return -2;
    std::string msg(mlen, '\0');
    if (mlen && !read_all(fd, &msg[0], mlen)) return -2;
    if (status == CC_STALE || status == 2) return -1;      // another build (it is leaving) / a request it would not take: compute here
    if (!msg.empty()) fputs(msg.c_str(), stderr);
    *rc = status;
    return 0;
}

volatile sig_atomic_t g_stop = 0;
void on_term(int) { g_stop = 1; }

// server: never returns.  Detached from the caller (own session, no inherited descriptors, cwd /); one request at a time
// `ready_fd`: the write end of a pipe the starting call waits on -- one byte when the helper listens, end-of-file when it gave up
[[noreturn]] void serve(const std::string& sock, const std::string& lock, int ready_fd)
{
    if (setsid() < 0) _exit(0);
    const int nul = open("/dev/null", O_RDWR);
    if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); if (nul > 2 && nul != ready_fd) close(nul); }
    (void)chdir("/");
    for (int fd = 3; fd < 256; ++fd) if (fd != ready_fd) close(fd);
    // one helper per GPU: the lock is held for the helper's lifetime (a second one, started by a concurrent first call, leaves at once)
    const int lfd = open(lock.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (lfd < 0 || flock(lfd, LOCK_EX | LOCK_NB) != 0) { const char r = 'L'; (void)write(ready_fd, &r, 1); _exit(0); }      // 'L': somebody else is (becoming) the helper
    { char pidbuf[32]; const int n = snprintf(pidbuf, sizeof pidbuf, "%d\n", (int)getpid()); (void)ftruncate(lfd, 0); (void)pwrite(lfd, pidbuf, (size_t)n, 0); }      // (for `kill $(cat gpu0.lock)`)
    unlink(sock.c_str());                                          // (a dead helper's socket: nobody listens, the lock was free)
    const int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un sa{}; sa.sun_family = AF_UNIX; strncpy(sa.sun_path, sock.c_str(), sizeof(sa.sun_path) - 1);
    // the GPU context first, the socket after it: a caller that can connect finds a helper that can compute
    if (ls < 0 || fav_device_count() <= 0 || hipSetDevice(gpu_index()) != hipSuccess || hipFree(nullptr) != hipSuccess) _exit(0);
    const mode_t um = umask(0177);
    const bool bound = bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof sa) == 0 && listen(ls, 16) == 0;
    umask(um);
    if (!bound) _exit(0);
    { const char r = 'R'; (void)write(ready_fd, &r, 1); close(ready_fd); }
    struct sigaction act{}; act.sa_handler = on_term; sigaction(SIGTERM, &act, nullptr); sigaction(SIGINT, &act, nullptr); sigaction(SIGHUP, &act, nullptr);
    signal(SIGPIPE, SIG_IGN);
    const char* idle_s = getenv("FAV_CC_IDLE_S");
    const int idle_ms = (idle_s && atoi(idle_s) > 0 ? atoi(idle_s) : 120) * 1000;
    Batch b;
    const uint64_t my_id = build_id();
    while (!g_stop) {
        pollfd pf{ls, POLLIN, 0};
        const int pr = poll(&pf, 1, idle_ms);
        if (pr == 0) break;                                        // idle: leave
        if (pr < 0) { if (errno == EINTR) continue; break; }
        const int c = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
        if (c < 0) continue;
        timeval tv{10, 0}; setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);       // (a caller that dies mid-request does not hang the helper)
        uint32_t hd[4];
        std::vector<std::string> a;
        bool ok = read_all(c, hd, sizeof hd) && hd[0] == CC_MAGIC && hd[1] >= 3 && hd[1] <= 4;
        const bool stale = ok && (((uint64_t)hd[3] << 32) | hd[2]) != my_id;
        for (uint32_t i = 0; ok && i < hd[1]; ++i) {
            uint32_t n;
            ok = read_all(c, &n, 4) && n > 0 && n < 65536;
            if (ok) { std::string s(n, '\0'); ok = read_all(c, &s[0], n) && s[0] == '/'; a.push_back(s); }
        }
        int32_t status = 2; std::string msg = "consistencyChecker: malformed request to the resident helper\n";
        if (stale) { status = CC_STALE; msg.clear(); g_stop = 1; unlink(sock.c_str()); }      // the caller is another build: make room for its helper
        else if (ok) { status = b.run(a); msg = b.err; }
        const uint32_t mlen = (uint32_t)msg.size();
        (void)(write_all(c, &status, 4) && write_all(c, &mlen, 4) && (mlen == 0 || write_all(c, msg.data(), mlen)));
        close(c);
    }
    unlink(sock.c_str());
    b.release();
    _exit(0);
}

std::string absolute(const char* path)
{
    if (path[0] == '/') return path;
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return path;
    return std::string(cwd) + "/" + path;
}

// the single call through the helper: 0 = done (exit status in *rc), -1 = not available (compute here), -2 = handed over, no answer
int via_helper(int argc, char** argv, int* rc)
{
    const char* off = getenv("FAV_CC_DAEMON");
    if (off && strcmp(off, "0") == 0) return -1;
    std::string dir, sock, lock;
    if (!socket_paths(dir, sock, lock)) return -1;
    std::vector<std::string> a;
    for (int i = 1; i < argc && i <= 4; ++i) a.push_back(absolute(argv[i]));
    int fd = connect_helper(sock);
    if (fd < 0) {
        // nobody there: start one and wait until it listens (its GPU context is what a call in this process would have paid for anyway)
        fflush(stdout); fflush(stderr);
        int pp[2];
        if (pipe(pp) != 0) return -1;
        const pid_t pid = fork();
        if (pid < 0) { close(pp[0]); close(pp[1]); return -1; }
        if (pid == 0) {
            close(pp[0]);
            const pid_t p2 = fork();                                // (double fork: the helper is nobody's zombie)
            if (p2 != 0) _exit(0);
            serve(sock, lock, pp[1]);
        }
        close(pp[1]);
        int st; (void)waitpid(pid, &st, 0);
        // 'R': it listens; 'L': another helper holds the lock (it listens already, or will in a moment); end-of-file: it gave up (no GPU)
        pollfd pf{pp[0], POLLIN, 0};
        char r = 0;
        if (!(poll(&pf, 1, 15000) > 0 && read(pp[0], &r, 1) == 1)) r = 0;
        close(pp[0]);
        if (r == 0) return -1;
        fd = connect_helper(sock);
        if (fd < 0 && r == 'L') {      // the lock holder may still be creating its context
            const double t0 = now_s();
            while (fd < 0 && now_s() - t0 < 3.0) { usleep(5000); fd = connect_helper(sock); }
        }
        if (fd < 0) return -1;
    }
    // (a helper that stops answering -- a hung device -- must not hang its callers: after FAV_CC_REPLY_S seconds, default 300, the call
    //  gives up on it and computes in its own process, which then reports what is wrong with the device)
    const char* rs = getenv("FAV_CC_REPLY_S");
    timeval tv{rs && atoi(rs) > 0 ? atoi(rs) : 300, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    const int r = ask_helper(fd, a, rc);
    close(fd);
    return r;
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc == 3 && strcmp(argv[1], "-batch") == 0) return batch_main(argv[2]);
    if (argc >= 4) {
        int rc = 0;
        const double t0 = now_s();
        const int via = via_helper(argc, argv, &rc);
        if (via == 0) {
            if (rc == 0) printf("%s", argv[3]);      // :166
            if (getenv("FAV_CC_TIMING")) fprintf(stderr, "consistencyChecker timing: through the resident helper %.1f ms\n", (now_s() - t0) * 1e3);
            return rc;
        }
        if (via == -2) {
            fprintf(stderr, "consistencyChecker: the resident helper took the request and did not answer (FAV_CC_REPLY_S); not computing %s a second time next to it.\n"
                            "  FAV_CC_DAEMON=0 computes in the calling process; `kill $(cat <runtime dir>/fav-cc/gpu*.lock)` ends the helper.\n", argv[3]);
            return 3;
        }
    }
    if (argc < 4) {
        fprintf(stderr, "usage: consistencyChecker <flow1.flo> <flow2.flo> <out.pgm> [<image.ppm>]\n"
                        "       consistencyChecker -batch <list.txt|->      (one such argument line per pair, one GPU context for all)\n");
        return 2;
    }
    // (computing in this process: FAV_CC_DAEMON=0, or no helper could be reached)
    const bool timing = getenv("FAV_CC_TIMING") != nullptr;
    double tq = now_s();
    auto lap = [&](const char* what) { if (timing) { const double t = now_s(); fprintf(stderr, "consistencyChecker timing: %-34s %8.2f ms\n", what, (t - tq) * 1e3); tq = t; } };
    float *f1 = nullptr, *f2 = nullptr;
    int w1, h1, w2, h2;
    if (fav_read_flo_host(argv[1], &f1, &w1, &h1)) return fail(argv[1]);
    if (fav_read_flo_host(argv[2], &f2, &w2, &h2)) return fail(argv[2]);
    if (w1 != w2 || h1 != h2) { fprintf(stderr, "consistencyChecker: flow sizes differ\n"); return 1; }   // :144-145
    uint8_t* img = nullptr;
    if (argc >= 5) {
        int wi, hi, ch;
        if (fav_read_pnm_host(argv[4], &img, &wi, &hi, &ch)) return fail(argv[4]);
        if (wi != w1 || hi != h1 || ch != 3) { fprintf(stderr, "consistencyChecker: image must be a P6 of the flow's size\n"); return 1; }
    }
    lap("read the input files");
    if (fav_device_count() <= 0) return fail("device");
    lap("hipInit + device enumeration");
    const char* g = getenv("FAV_GPU");
    if (hipSetDevice(g ? atoi(g) : 0) != hipSuccess) { fprintf(stderr, "consistencyChecker: cannot select GPU\n"); return 1; }
    const size_t n = (size_t)w1 * h1;
    float *d1 = nullptr, *d2 = nullptr; uint8_t *dimg = nullptr, *dout = nullptr; void* ws = nullptr;
    const size_t wsb = fav_consistency_workspace_bytes(w1, h1, img != nullptr);
    if (hipMalloc((void**)&d1, n * 8) || hipMalloc((void**)&d2, n * 8) || hipMalloc((void**)&dout, n) ||
        (img && hipMalloc((void**)&dimg, n * 3)) || (wsb && hipMalloc(&ws, wsb))) { fprintf(stderr, "consistencyChecker: hipMalloc failed\n"); return 1; }
    lap("context + device buffers");
    hipMemcpy(d1, f1, n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d2, f2, n * 8, hipMemcpyHostToDevice);
    if (img) hipMemcpy(dimg, img, n * 3, hipMemcpyHostToDevice);
    lap("host -> device copies (pageable)");
    if (fav_consistency_u8(d1, d2, dimg, dout, w1, h1, ws, wsb, nullptr)) return fail("fav_consistency_u8");
    if (timing) hipDeviceSynchronize();
    lap("code object load + the mask's kernels");
    std::vector<uint8_t> out(n);
    if (hipMemcpy(out.data(), dout, n, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "consistencyChecker: device error\n"); return 1; }
    lap("device -> host copy");
    if (fav_write_pgm_host(argv[3], out.data(), w1, h1)) return fail(argv[3]);
    lap("write the .pgm");
    printf("%s", argv[3]);   // :166
    fav_free_host(f1); fav_free_host(f2); fav_free_host(img);
    hipFree(d1); hipFree(d2); hipFree(dimg); hipFree(dout); hipFree(ws);
    return 0;
}
