// How many host cores does this container really get?  T threads each run the same fixed integer loop; with C usable
// cores the wall time stays flat up to T = C and grows linearly beyond.  (g++ -O2 -pthread scripts/cpu_scaling_probe.cpp)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
static uint64_t spin(uint64_t n) { uint64_t x = 88172645463325252ULL; for (uint64_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; } return x; }
int main() {
    volatile uint64_t sink = 0;
    for (int T : {1, 8, 16, 32, 64, 128, 256}) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&] { sink = sink + spin(300000000ULL); });
        for (auto& x : th) x.join();
        printf("threads %3d: %.2f s\n", T, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}
