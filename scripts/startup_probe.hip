// startup_probe.hip -- what a command's first second is made of on a box: HIP start-up, device allocations of tens of GB,
// page-locked host buffers, each timed on its own.  Run before / after a big file has gone through the page cache.
//   hipcc --offload-arch=gfx950 -O2 -o startup_probe scripts/startup_probe.hip ; ./startup_probe [GB per device allocation]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 20.0;
    double t0 = now();
    (void)hipInit(0); (void)hipSetDevice(0);
    void* p = nullptr; (void)hipMalloc(&p, 1 << 20);
    printf("  hip start-up + first allocation   %.3f s\n", now() - t0);
    void* big[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        t0 = now();
        hipError_t e = hipMalloc(&big[i], (size_t)(gb * 1073741824.0));
        printf("  hipMalloc %.0f GB (#%d)              %.3f s%s\n", gb, i, now() - t0, e == hipSuccess ? "" : "  FAILED");
    }
    t0 = now(); (void)hipMemsetAsync(big[0], 0, (size_t)(gb * 1073741824.0), 0); (void)hipDeviceSynchronize();
    printf("  memset of one of them             %.3f s\n", now() - t0);
    for (int i = 0; i < 3; i++) {
        void* h = nullptr; t0 = now();
        (void)hipHostMalloc(&h, 96u << 20, hipHostMallocDefault);
        printf("  hipHostMalloc 96 MiB (#%d)         %.3f s\n", i, now() - t0);
    }
    t0 = now();
    for (int i = 0; i < 4; i++) (void)hipFree(big[i]);
    printf("  hipFree x4                        %.3f s\n", now() - t0);
    t0 = now();
    for (int i = 0; i < 2; i++) (void)hipMalloc(&big[i], (size_t)(gb * 1073741824.0));
    printf("  hipMalloc x2 again                %.3f s\n", now() - t0);
    return 0;
}
