// Round-6 experiment: is the XCD round-robin of workgroup dispatch reset per launch, per queue, or global?
//   hipcc --offload-arch=gfx950 scripts/cu_probe2.hip -o scripts/cu_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void show(const char* what, unsigned* d, int n)
{
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, 4 * n, hipMemcpyDeviceToHost);
    int cnt[8] = {0};
    for (int i = 0; i < n; ++i) cnt[h[i] & 7]++;
    printf("%-64s first block -> XCD %u | per XCD:", what, h[0]);
    for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
    printf("\n");
}
int main()
{
    unsigned *a, *b, *c, *e; hipMalloc(&a, 4 * 8192); hipMalloc(&b, 4 * 8192); hipMalloc(&c, 4 * 8192); hipMalloc(&e, 4 * 8192);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
        printf("--- one stream, back to back: 252, 252, 8, 252, 13, 8 blocks\n");
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, a, 20000);
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, b, 20000);
        hipLaunchKernelGGL(probe, dim3(8), dim3(1024), 0, s1, c, 20000);
        hipStreamSynchronize(s1);
        show("252 (1st)", a, 252); show("252 (2nd)", b, 252); show("8", c, 8);
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, a, 20000);
        hipLaunchKernelGGL(probe, dim3(13), dim3(256), 0, s1, b, 20000);
        hipLaunchKernelGGL(probe, dim3(8), dim3(1024), 0, s1, c, 20000);
        hipStreamSynchronize(s1);
        show("252 (3rd)", a, 252); show("13", b, 13); show("8 (after 13)", c, 8);
        printf("--- two streams: s1 runs 252-block kernels, s2 launches 8 blocks while they run\n");
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, a, 2000000);
        hipLaunchKernelGGL(probe, dim3(8), dim3(1024), 0, s2, c, 20000);
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, b, 2000000);
        hipLaunchKernelGGL(probe, dim3(8), dim3(1024), 0, s2, e, 20000);
        hipDeviceSynchronize();
        show("s1: 252 (long)", a, 252); show("s2: 8 (next to it)", c, 8); show("s1: 252 (long, 2nd)", b, 252); show("s2: 8 (2nd)", e, 8);
        hipLaunchKernelGGL(probe, dim3(5), dim3(256), 0, s2, e, 20000);
        hipLaunchKernelGGL(probe, dim3(252), dim3(512), 65536, s1, a, 200000);
        hipLaunchKernelGGL(probe, dim3(8), dim3(1024), 0, s2, c, 20000);
        hipDeviceSynchronize();
        show("s2: 5", e, 5); show("s1: 252 after s2's 5", a, 252); show("s2: 8 after its own 5", c, 8);
    }
    return 0;
}
