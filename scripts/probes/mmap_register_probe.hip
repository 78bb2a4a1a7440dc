// Can a page-cache file be DMA'd from directly?  mmap(MAP_SHARED) of a /dev/shm file + hipHostRegister in pieces + hipMemcpyAsync,
// against hipHostMalloc + pread + hipMemcpyAsync.  build: hipcc --offload-arch=gfx950 -O2 -o mmap_register_probe mmap_register_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
static double ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/dev/shm/bzq_probe.bin";
    const size_t piece = (argc > 2 ? atoll(argv[2]) : 64) << 20;
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t n = (size_t)st.st_size & ~(piece - 1);
    double t0 = ms();
    hipFree(0);
    uint8_t* d; hipMalloc(&d, n);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t1 = ms();
    printf("init + hipMalloc(%zu MiB): %.1f ms\n", n >> 20, t1 - t0);
    // A: mmap + register pieces
    uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    double reg = 0, cp = 0;
    bool okA = true;
    double tA = ms();
    for (size_t off = 0; off < n && okA; off += piece) {
        double a = ms();
        hipError_t e = hipHostRegister(m + off, piece, hipHostRegisterDefault);
        double b = ms();
        if (e != hipSuccess) { printf("hipHostRegister failed at %zu: %s\n", off, hipGetErrorString(e)); okA = false; (void)hipGetLastError(); break; }
        reg += b - a;
        hipMemcpyAsync(d + off, m + off, piece, hipMemcpyHostToDevice, s);
    }
    hipStreamSynchronize(s);
    double tA1 = ms();
    if (okA) printf("A mmap+register+copy: total %.1f ms = %.1f GB/s (register calls %.1f ms)\n", tA1 - tA, n / (tA1 - tA) / 1e6, reg);
    if (okA) { double a = ms(); for (size_t off = 0; off < n; off += piece) hipHostUnregister(m + off); printf("  unregister %.1f ms\n", ms() - a); }
    // A2: plain hipMemcpy from the unregistered mapping
    { double a = ms(); hipMemcpy(d, m, n, hipMemcpyHostToDevice); printf("A2 hipMemcpy from the bare mapping: %.1f ms = %.1f GB/s\n", ms() - a, n / (ms() - a) / 1e6); }
    // B: hipHostMalloc + pread + copy
    double tB = ms();
    uint8_t* p; hipHostMalloc(&p, 2 * piece, hipHostMallocDefault);
    double tB1 = ms();
    for (size_t off = 0, k = 0; off < n; off += piece, ++k) {
        uint8_t* buf = p + (k & 1) * piece;
        if (k >= 2) hipStreamSynchronize(s);
        pread(fd, buf, piece, off);
        hipMemcpyAsync(d + off, buf, piece, hipMemcpyHostToDevice, s);
    }
    hipStreamSynchronize(s);
    double tB2 = ms();
    printf("B hipHostMalloc(2 pieces) %.1f ms + pread(1 thread)+copy %.1f ms = %.1f GB/s\n", tB1 - tB, tB2 - tB1, n / (tB2 - tB1) / 1e6);
    // C: hipHostMalloc of a big buffer: what pinning costs
    for (size_t mb : {64, 288}) { double a = ms(); uint8_t* q; hipHostMalloc(&q, mb << 20, hipHostMallocDefault); double b = ms(); printf("C hipHostMalloc(%zu MiB): %.1f ms\n", mb, b - a); hipHostFree(q); }
    // D: anonymous memory + hipHostRegister (pinning without the runtime's allocation)
    { uint8_t* q = (uint8_t*)mmap(nullptr, 288 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); double a = ms(); memset(q, 1, 288 << 20); double b = ms(); hipError_t e = hipHostRegister(q, 288 << 20, hipHostRegisterDefault); double c = ms();
      printf("D anonymous 288 MiB: touch %.1f ms, hipHostRegister %.1f ms (%s)\n", b - a, c - b, hipGetErrorString(e)); }
    return 0;
}
