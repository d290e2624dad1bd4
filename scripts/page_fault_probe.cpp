// How fast does this host hand out fresh (zeroed) anonymous memory?  T threads each touch their own 512 MiB mapping,
// with and without transparent huge pages, and with MAP_POPULATE.  (g++ -O2 -pthread scripts/page_fault_probe.cpp)
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double run(int nt, int mode) {        // mode 0: 4 KiB touches, 1: THP + touches, 2: MAP_POPULATE
    const size_t per = (size_t)512 << 20, HP = (size_t)2 << 20;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=] {
        char* p = (char*)mmap(0, per + HP, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (mode == 2 ? MAP_POPULATE : 0), -1, 0);
        char* a = (char*)(((size_t)p + HP - 1) & ~(HP - 1));
        if (mode == 1) madvise(a, per, MADV_HUGEPAGE);
        if (mode != 2) for (size_t i = 0; i < per; i += 4096) a[i] = 1;
        munmap(p, per + HP);
    });
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return nt * 0.5 / dt;
}
int main() {
    for (int nt : {1, 4, 8, 16})
        printf("threads %2d: touch %.2f GB/s, THP touch %.2f GB/s, MAP_POPULATE %.2f GB/s\n", nt, run(nt, 0), run(nt, 1), run(nt, 2));
    return 0;
}
