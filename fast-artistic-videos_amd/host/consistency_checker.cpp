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
#include <hip/hip_runtime.h>

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
    int run(const std::vector<std::string>& a)
    {
        int w1, h1s;
        if (!flo_size(a[0].c_str(), &w1, &h1s)) { fprintf(stderr, "consistencyChecker: cannot read %s\n", a[0].c_str()); return 1; }
        const size_t n = (size_t)w1 * h1s;
        if (w1 != W || h1s != H) {
            release();
            W = w1; H = h1s;
            wsb = fav_consistency_workspace_bytes(W, H, 1);
            if (hipMalloc((void**)&d1, n * 8) || hipMalloc((void**)&d2, n * 8) || hipMalloc((void**)&dout, n) || hipMalloc((void**)&dimg, n * 3) ||
                (wsb && hipMalloc(&ws, wsb)) || hipHostMalloc((void**)&h1, n * 8, hipHostMallocDefault) || hipHostMalloc((void**)&h2, n * 8, hipHostMallocDefault) ||
                hipHostMalloc((void**)&himg, n * 3, hipHostMallocDefault) || hipHostMalloc((void**)&hout, n, hipHostMallocDefault)) {
                fprintf(stderr, "consistencyChecker: out of memory\n"); release(); return 1; }
        }
        int w, h, ch;
        if (fav_read_flo_into_host(a[0].c_str(), h1, n * 2, &w, &h)) return fail(a[0].c_str());
        if (fav_read_flo_into_host(a[1].c_str(), h2, n * 2, &w, &h)) return fail(a[1].c_str());
        if (w != W || h != H) { fprintf(stderr, "consistencyChecker: flow sizes differ\n"); return 1; }                  // :144-145
        const bool img = a.size() >= 4;
        if (img) {
            if (fav_read_pnm_into_host(a[3].c_str(), himg, n * 3, &w, &h, &ch)) return fail(a[3].c_str());
            if (w != W || h != H || ch != 3) { fprintf(stderr, "consistencyChecker: image must be a P6 of the flow's size\n"); return 1; }
        }
        hipMemcpyAsync(d1, h1, n * 8, hipMemcpyHostToDevice, nullptr);
        hipMemcpyAsync(d2, h2, n * 8, hipMemcpyHostToDevice, nullptr);
        if (img) hipMemcpyAsync(dimg, himg, n * 3, hipMemcpyHostToDevice, nullptr);
        if (fav_consistency_u8(d1, d2, img ? dimg : nullptr, dout, W, H, ws, wsb, nullptr)) return fail("fav_consistency_u8");
        if (hipMemcpyAsync(hout, dout, n, hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
            fprintf(stderr, "consistencyChecker: device error\n"); return 1; }
        if (fav_write_pgm_host(a[2].c_str(), hout, W, H)) return fail(a[2].c_str());
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
        if ((rc = b.run(a)) != 0) break;                          // as a sequence of single calls under `set -e`: stop at the first failure
        printf("%s\n", a[2].c_str()); fflush(stdout);
    }
    if (f != stdin) fclose(f);
    b.release();
    return rc;
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc == 3 && strcmp(argv[1], "-batch") == 0) return batch_main(argv[2]);
    if (argc < 4) {
        fprintf(stderr, "usage: consistencyChecker <flow1.flo> <flow2.flo> <out.pgm> [<image.ppm>]\n"
                        "       consistencyChecker -batch <list.txt|->      (one such argument line per pair, one GPU context for all)\n");
        return 2;
    }
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
    if (fav_device_count() <= 0) return fail("device");
    const char* g = getenv("FAV_GPU");
    if (hipSetDevice(g ? atoi(g) : 0) != hipSuccess) { fprintf(stderr, "consistencyChecker: cannot select GPU\n"); return 1; }
    const size_t n = (size_t)w1 * h1;
    float *d1 = nullptr, *d2 = nullptr; uint8_t *dimg = nullptr, *dout = nullptr; void* ws = nullptr;
    const size_t wsb = fav_consistency_workspace_bytes(w1, h1, img != nullptr);
    if (hipMalloc((void**)&d1, n * 8) || hipMalloc((void**)&d2, n * 8) || hipMalloc((void**)&dout, n) ||
        (img && hipMalloc((void**)&dimg, n * 3)) || (wsb && hipMalloc(&ws, wsb))) { fprintf(stderr, "consistencyChecker: hipMalloc failed\n"); return 1; }
    hipMemcpy(d1, f1, n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d2, f2, n * 8, hipMemcpyHostToDevice);
    if (img) hipMemcpy(dimg, img, n * 3, hipMemcpyHostToDevice);
    if (fav_consistency_u8(d1, d2, dimg, dout, w1, h1, ws, wsb, nullptr)) return fail("fav_consistency_u8");
    std::vector<uint8_t> out(n);
    if (hipMemcpy(out.data(), dout, n, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "consistencyChecker: device error\n"); return 1; }
    if (fav_write_pgm_host(argv[3], out.data(), w1, h1)) return fail(argv[3]);
    printf("%s", argv[3]);   // :166
    fav_free_host(f1); fav_free_host(f2); fav_free_host(img);
    hipFree(d1); hipFree(d2); hipFree(dimg); hipFree(dout); hipFree(ws);
    return 0;
}
