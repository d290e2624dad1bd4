// scripts/membench.hip -- random-access ceilings of one MI355X for the access shapes the k-mer set uses.
// Not product code: a measurement aid for DESIGN.md (what a random-probe hash formulation can reach at most).
//   ./membench [log2_table_bytes=34] [n_ops_log2=30]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* tab, uint64_t mask32 /* slots of 32 B - 1 */, uint64_t n, uint64_t* sink) {
    uint64_t acc = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        uint64_t* s = tab + (mix(g) & mask32) * 4;
        if (MODE == 0) {            // 32-byte agent-scope snapshot
            u32x4 lo, hi;
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(s) : "memory");
            acc += lo.x + hi.w;
        } else if (MODE == 1) {     // plain 32-byte load
            const uint4* p = (const uint4*)s; uint4 a = p[0], b = p[1]; acc += a.x + b.w;
        } else if (MODE == 2) {     // returned 64-bit CAS (always succeeds or not, does not matter)
            uint64_t e = g; __hip_atomic_compare_exchange_strong(s + 2, &e, g + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += e;
        } else if (MODE == 3) {     // fire-and-forget 64-bit add
            __hip_atomic_fetch_add(s + 2, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 4) {     // snapshot then CAS on the same slot (the update path of table_put_wide)
            u32x4 lo, hi;
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(s) : "memory");
            uint64_t e = (uint64_t)hi.x | ((uint64_t)hi.y << 32);
            __hip_atomic_compare_exchange_strong(s + 2, &e, e + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += e;
        } else if (MODE == 5) {     // plain 32-byte read-modify-write (no atomics): what a conflict-free partition would do
            uint4* p = (uint4*)s; uint4 b = p[1]; b.x += 1; p[1] = b;
        } else if (MODE == 6) {     // fire-and-forget 64-bit add, two per slot (cnt + ord shape)
            __hip_atomic_fetch_add(s + 2, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min(s + 3, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void copyk(const uint4* a, uint4* b, uint64_t n) {
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) b[g] = a[g];
}
template <int MODE> float run(uint64_t* tab, uint64_t mask, uint64_t n, uint64_t* sink) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 32), dim3(256), 0, 0, tab, mask, n / 8, sink);   // warm
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k<MODE>, dim3(256 * 32), dim3(256), 0, 0, tab, mask, n, sink); CK(hipEventRecord(b));
    CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char** argv) {
    int lb = argc > 1 ? atoi(argv[1]) : 34, ln = argc > 2 ? atoi(argv[2]) : 30;
    uint64_t bytes = 1ULL << lb, n = 1ULL << ln, mask = bytes / 32 - 1;
    uint64_t *tab, *sink; CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&sink, 8)); CK(hipMemset(tab, 0, bytes));
    const char* names[] = {"snapshot32 sc1", "load32 plain", "cas64 returned", "add64 no-return", "snapshot32+cas64", "plain rmw32", "add64+min64 no-return"};
    float ms[7] = {run<0>(tab, mask, n, sink), run<1>(tab, mask, n, sink), run<2>(tab, mask, n, sink), run<3>(tab, mask, n, sink),
                   run<4>(tab, mask, n, sink), run<5>(tab, mask, n, sink), run<6>(tab, mask, n, sink)};
    for (int i = 0; i < 7; i++) printf("table 2^%d B, %-24s %8.2f ms  %7.2f Gops/s\n", lb, names[i], ms[i], n / ms[i] * 1e-6);
    uint64_t cn = bytes / 2 / 16; hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); hipLaunchKernelGGL(copyk, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)tab, (uint4*)tab + cn, cn); CK(hipEventRecord(b));
    CK(hipEventSynchronize(b)); float cms; CK(hipEventElapsedTime(&cms, a, b));
    printf("table 2^%d B, streaming copy           %8.2f ms  %7.2f GB/s (read+write)\n", lb, cms, bytes / cms * 1e-6);
    return 0;
}
