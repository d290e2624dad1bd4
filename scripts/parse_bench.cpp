// parse_bench.cpp -- how fast the host parses FASTQ on this box, and how that scales with threads (no GPU):
//   g++ -O3 -std=c++17 -pthread -I soapdenovo2_amd/csrc -I include scripts/parse_bench.cpp -lz -o /tmp/parse_bench && /tmp/parse_bench reads.fq
// Every thread parses its own copy of the same 256 MiB stretch of the file six times over (parse_range of host_reads.cpp: records ->
// 2 bits a base); long enough for the scheduler to have spread the threads -- with one pass (80 ms) two to four threads stay on the
// processor that created them and look like no scaling at all.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/resource.h>
#include <sched.h>
#include "../soapdenovo2_amd/csrc/host_reads.cpp"
namespace pg { int host_threads(int n) { return n > 0 ? n : 8; } }
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<char> buf(256u << 20);
    size_t n = fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    size_t last = 0; long lines = 0;
    for (size_t i = 0; i < n; i++) if (buf[i] == '\n') { lines++; if (lines % 4 == 0) last = i + 1; }
    n = last;
    pg::InputFile in; in.max_read_len = 150; in.reverse = false;
    for (int nt : {1, 2, 4, 8, 16, 24, 32}) {
        std::vector<std::vector<char>> copies(nt);
        for (auto& c : copies) { c.assign(buf.begin(), buf.begin() + n); c.resize(n + 64); }
        std::vector<pg::PackedRun> runs(nt);
        std::vector<pg::CodeBuf> codes(nt, pg::CodeBuf(200));
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            std::vector<double> cpu(nt, 0); std::vector<int> where(nt, -1);
            for (int t = 0; t < nt; t++) th.emplace_back([&, t] { for (int again = 0; again < 6; again++) { runs[t].clear(); pg::parse_range(in, true, copies[t].data(), n, codes[t], runs[t]); } struct rusage ru; getrusage(RUSAGE_THREAD, &ru); cpu[t] = ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec + ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec; where[t] = sched_getcpu(); });
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) { printf("%2d thread(s): %.1f ns a record a thread, %.2f GB/s in all; wall %.2fs, cpu of thread 0 %.2fs, cpus:", nt, 1e9 * dt / (6.0 * (double)runs[0].records), 6.0 * (double)n * nt / dt / 1e9, dt, cpu[0]); for (int t = 0; t < nt && t < 16; t++) printf(" %d", where[t]); printf("\n"); }
        }
    }
    return 0;
}
