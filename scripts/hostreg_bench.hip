// Micro-benchmark (tuning only): what does it cost the HOST to get a 7.4 MB RAM-backed file into device memory?
//   (a) read() into a pinned staging buffer + hipMemcpyAsync            -- what fav_stylize's loaders do
//   (b) mmap + hipHostRegister + hipMemcpyAsync + hipHostUnregister     -- DMA straight out of the page cache, no CPU copy
//   (c) mmap + hipMemcpy from the pageable mapping                      -- the runtime's own staging / pin-in-place path
// build: hipcc -O2 --offload-arch=gfx950 scripts/hostreg_bench.hip -o gpurun_out/hostreg_bench
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double cpu() { timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t n = 7372812; const int R = 40;
    char* src = (char*)malloc(n); memset(src, 3, n);
    const char* path = "/dev/shm/fav_hostreg_bench.bin";
    int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600); if (write(fd, src, n) != (ssize_t)n) return 1; close(fd);
    void* dev; CK(hipMalloc(&dev, n)); void* pin; CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int pass = 0; pass < 2; ++pass) {
        double w0 = wall(), c0 = cpu();
        for (int i = 0; i < R; ++i) { int f = open(path, O_RDONLY); if (read(f, pin, n) != (ssize_t)n) return 1; close(f); CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
        printf("(a) read + pinned H2D          : %.3f ms wall, %.3f ms cpu per file\n", (wall() - w0) * 1e3 / R, (cpu() - c0) * 1e3 / R);
        w0 = wall(); c0 = cpu();
        for (int i = 0; i < R; ++i) {
            int f = open(path, O_RDONLY); void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, f, 0); close(f);
            if (m == MAP_FAILED) { printf("mmap failed\n"); return 1; }
            hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
            if (e != hipSuccess) { printf("hipHostRegister(mmap of tmpfs) -> %s\n", hipGetErrorString(e)); munmap(m, n); break; }
            CK(hipMemcpyAsync(dev, m, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            CK(hipHostUnregister(m)); munmap(m, n);
        }
        printf("(b) mmap + register + H2D      : %.3f ms wall, %.3f ms cpu per file\n", (wall() - w0) * 1e3 / R, (cpu() - c0) * 1e3 / R);
        w0 = wall(); c0 = cpu();
        for (int i = 0; i < R; ++i) {
            int f = open(path, O_RDONLY); void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, f, 0); close(f);
            CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice)); munmap(m, n);
        }
        printf("(c) mmap + pageable hipMemcpy  : %.3f ms wall, %.3f ms cpu per file\n", (wall() - w0) * 1e3 / R, (cpu() - c0) * 1e3 / R);
        w0 = wall(); c0 = cpu();
        for (int i = 0; i < R; ++i) { int f = open("/dev/shm/fav_hostreg_out.bin", O_CREAT | O_WRONLY | O_TRUNC, 0600); if (write(f, pin, 2900000) != 2900000) return 1; close(f); }
        printf("(d) write() of 2.9 MB to tmpfs : %.3f ms wall, %.3f ms cpu per file\n", (wall() - w0) * 1e3 / R, (cpu() - c0) * 1e3 / R);
    }
    unlink(path); unlink("/dev/shm/fav_hostreg_out.bin");
    return 0;
}
