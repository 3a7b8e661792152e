// Micro-benchmark (tuning only): which HIP operations keep the runtime's background thread spinning?
// Each variant runs ~1 s of GPU work in ~1.7 ms "frames" and reports process CPU seconds / wall seconds (main thread sleeps in between).
// build: hipcc -O2 --offload-arch=gfx950 scripts/runtime_thread_bench.hip -o scripts/runtime_thread_bench.bin
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double pcpu() { timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double tcpu() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void copy_kernel(const uint4* src, uint4* dst, size_t n16) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i]; }
__global__ void spin_kernel(float* p, int iters) { float v = p[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n = 7372812;
    float* d; CK(hipMalloc(&d, 1 << 20)); void *din, *dout, *hin, *hout;
    CK(hipMalloc(&din, n)); CK(hipMalloc(&dout, n)); CK(hipHostMalloc(&hin, n, hipHostMallocDefault)); CK(hipHostMalloc(&hout, n, hipHostMallocDefault));
    hipStream_t st, sc, sd; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&sc)); CK(hipStreamCreate(&sd));
    hipEvent_t eu, eo, ed; CK(hipEventCreateWithFlags(&eu, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eo, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ed, hipEventDisableTiming));
    // calibrate: kernel of ~35 us
    int iters = 2000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, d, iters); CK(hipStreamSynchronize(st));
    const int frames = 500;
    const double w0 = wall(), c0 = pcpu(), m0 = tcpu();
    for (int f = 0; f < frames; ++f) {
        if (variant >= 1 && variant <= 3) { CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, sc)); CK(hipEventRecord(eu, sc)); CK(hipStreamWaitEvent(st, eu, 0)); }
        if (variant == 4) { CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, sc)); }                                   // copy, nobody depends on it
        if (variant == 5) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sc, d + 4096, 100); CK(hipEventRecord(eu, sc)); CK(hipStreamWaitEvent(st, eu, 0)); }   // cross-stream dependency, no copy
        if (variant == 6) { CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, st)); }                                   // copy on the compute stream itself
        if (variant == 8) { CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, sc)); CK(hipEventRecord(eu, sc)); while (hipEventQuery(eu) == hipErrorNotReady) usleep(50); }   // copy + event, the HOST waits (sleeping poll), then enqueues
        if (variant == 9) { CK(hipEventRecord(eo, st)); }                                                               // an event record per frame on the compute stream, nobody waits on the device
        if (variant == 10) { CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, sc)); while (hipStreamQuery(sc) == hipErrorNotReady) usleep(50); }   // copy, host polls the STREAM
        if (variant == 7) { hipLaunchKernelGGL(copy_kernel, dim3(16), dim3(256), 0, sc, (const uint4*)hin, (uint4*)din, n / 16); CK(hipEventRecord(eu, sc)); CK(hipStreamWaitEvent(st, eu, 0)); }   // upload by a kernel reading pinned host memory
        for (int k = 0; k < 48; ++k) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, d, iters);
        if (variant == 2 || variant == 3) { CK(hipEventRecord(eo, st)); CK(hipStreamWaitEvent(sd, eo, 0)); CK(hipMemcpyAsync(hout, dout, 2900000, hipMemcpyDeviceToHost, sd)); CK(hipEventRecord(ed, sd)); }
        if (variant == 3) { while (hipEventQuery(ed) == hipErrorNotReady) usleep(100); }
        else if (f % 8 == 7) { while (hipStreamQuery(st) == hipErrorNotReady) usleep(200); }      // keep the queue from running away
    }
    while (hipStreamQuery(st) == hipErrorNotReady || hipStreamQuery(sd) == hipErrorNotReady || hipStreamQuery(sc) == hipErrorNotReady) usleep(200);
    const double w = wall() - w0, c = pcpu() - c0, m = tcpu() - m0;
    const char* names[] = {"kernels only (one stream, polled with sleeps)", "+ 7.4 MB pinned H2D on a copy stream + event dependency per frame",
                           "+ 2.9 MB D2H on a download stream + event dependency per frame", "+ host waits for every frame's download (sleeping poll)",
                           "pinned H2D on a copy stream, NO dependency", "cross-stream event dependency (kernel -> kernel), NO copy", "pinned H2D on the compute stream itself",
                           "upload by a KERNEL reading the pinned buffer + event dependency",
                           "pinned H2D on a copy stream + event record, HOST polls the event before enqueueing the kernels", "kernels + one hipEventRecord per frame (no device-side waiter)",
                           "pinned H2D on a copy stream, HOST polls hipStreamQuery before enqueueing the kernels"};
    printf("variant %d: %-70s wall %.3f s, process cpu %.3f s (main thread %.3f, other threads %.3f) = %.2f cores\n", variant, names[variant], w, c, m, c - m, c / w);
    return 0;
}
