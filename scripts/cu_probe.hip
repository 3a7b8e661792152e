// Where do blocks land?  (round 6 experiment: CU masks of the look-ahead queue)  hipcc --offload-arch=gfx950 scripts/cu_probe.hip -o scripts/cu_probe.bin
// For a stream with a CU mask (bits given on the command line) and for a plain 252-block grid: the (XCC, SE, CU) of every block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void show(const char* what, const std::vector<unsigned>& h, int n)
{
    std::map<unsigned, int> per; std::map<unsigned, int> xccs;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[(xcc << 16) | (se << 8) | (sh << 4) | cu]++; xccs[xcc]++;
    }
    printf("%s: %d blocks on %zu distinct CUs; per XCC:", what, n, per.size());
    for (auto& kv : xccs) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    if (per.size() <= 16) { for (auto& kv : per) printf("   xcc %u se %u sh %u cu %u : %d blocks\n", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf, kv.second); }
}
int main(int argc, char** argv)
{
    unsigned* d; hipMalloc(&d, 8 * 4096);
    std::vector<unsigned> h(2 * 4096);
    hipStream_t plain; hipStreamCreate(&plain);
    for (int n : {252, 256}) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(512), 65536, plain, d, 200000);
        hipStreamSynchronize(plain); hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
        char nm[64]; snprintf(nm, sizeof nm, "plain stream, %d blocks x 512 threads, 64 KB LDS", n); show(nm, h, n);
    }
    for (int a = 1; a < argc; ++a) {
        uint32_t words[8] = {0};
        for (const char* p = argv[a]; *p;) { const int b = atoi(p); if (b >= 0 && b < 256) words[b / 32] |= 1u << (b % 32); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) { printf("mask %s: hipExtStreamCreateWithCUMask failed\n", argv[a]); continue; }
        hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, st, d, 20000);
        hipStreamSynchronize(st); hipMemcpy(h.data(), d, 8 * 64, hipMemcpyDeviceToHost);
        char nm[128]; snprintf(nm, sizeof nm, "mask bits {%s}, 64 blocks x 256 threads", argv[a]); show(nm, h, 64);
        // ... and a 252-block grid on the plain stream WHILE the masked stream is busy: does it avoid the masked CUs?
        hipLaunchKernelGGL(probe, dim3(32), dim3(256), 0, st, d + 2048, 4000000);
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, plain, d, 200000);
        hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8 * 252, hipMemcpyDeviceToHost);
        show("   252-block grid next to it", h, 252);
        hipStreamDestroy(st);
    }
    return 0;
}
