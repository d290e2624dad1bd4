// scripts/lds_bench.hip -- measurement aid (not part of the library): LDS operation rates per CU under the access patterns
// of skm_count_kernel: random 32-bit atomic adds over S slots (S = 2048 random, S = 48 "hot k-mers"), random 64-bit reads.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, uint32_t nslots) {
    __shared__ unsigned int cnt[4][2048];
    __shared__ unsigned long long key[2][2048];
    for (int i = threadIdx.x; i < 2048; i += 1024) { for (int q = 0; q < 4; q++) cnt[q][i] = 0; key[0][i] = i; key[1][i] = i * 3; }
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t h = ((x >> 10) % nslots) * (2048 / nslots > 0 ? 1 : 1);
        const uint32_t a = (x >> 8) & 3;
        if (MODE == 0) atomicAdd(&cnt[a][h], 1u);                                   // one add
        if (MODE == 1) { atomicAdd(&cnt[a][h], 1u); atomicAdd(&cnt[(a + 1) & 3][h], 1u); }   // two adds
        if (MODE == 2) acc += key[0][h] + key[1][h];                                // two 64-bit reads
        if (MODE == 3) { acc += key[0][h] + key[1][h]; if (acc != 7) { atomicAdd(&cnt[a][h], 1u); atomicAdd(&cnt[(a + 1) & 3][h], 1u); } }   // read, then dependent adds
        if (MODE == 4) atomicMin(&key[0][h], (unsigned long long)x);                // 64-bit min
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = cnt[0][0] + (uint32_t)acc;
}
template <int MODE> void run(const char* name, uint32_t nslots) {
    uint32_t* d; hipMalloc(&d, 4096 * 4);
    const int iters = 2000, grid = 256 * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, 10, nslots);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, iters, nslots);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 1024 * iters;
    printf("%-44s slots %5u: %7.2f ms  %7.1f G lane-ops/s  %.2f cycles per wave-op per CU (2.4 GHz, 256 CUs)\n", name, nslots, ms, ops / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / (ops / 64));
    hipFree(d);
}
int main() {
    for (uint32_t s : {2048u, 512u, 48u, 8u}) {
        run<0>("one ds_add_u32", s);
        run<1>("two ds_add_u32", s);
        run<2>("two ds_read_b64", s);
        run<3>("two reads then two dependent adds", s);
        run<4>("one ds_min_u64", s);
    }
    return 0;
}
