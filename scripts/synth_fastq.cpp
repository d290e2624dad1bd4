// synth_fastq.cpp -- the synthetic read model of SURVEY.md 8d (uniform random genome, fixed-length reads with uniform
// start, strand flipped with p = 0.5, independent substitution errors) written straight to a single-end FASTQ file by all
// host threads.  Counter-based random numbers (a hash of (seed, stream, index)), so the bytes depend on the arguments only --
// not on the thread count or the machine: the same file can be made in the build container (where the reference binary
// runs for half an hour and its md5s are committed) and on the GPU box (where the executable is timed and compared).
//
//   synth_fastq <out.fq> <genome_len> <n_reads> <read_len> <err> <seed> [threads [min_len]]
//
// min_len (round 6; 0 / absent = every read has read_len bases): read r keeps the first min_len + h(seed, r) mod (read_len - min_len + 1)
// bases of what it would have been -- 3'-trimmed reads, lengths uniform in [min_len, read_len], the same fragments as the untrimmed file.
//
// Record: "@r%09llu\n" bases "\n+\n" 'I' x L "\n" (fixed width while n_reads <= 10^9).  A file whose size is a multiple of
// 32768 would lose its tail in the reference's reader (prlHashReads.c:873-877): one blank is added to the last line then.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

static inline uint64_t mix(uint64_t x) {           // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
struct Rng {                                        // one stream per read
    uint64_t s;
    explicit Rng(uint64_t seed, uint64_t stream) : s(mix(seed * 0x100000001B3ULL ^ mix(stream))) {}
    uint64_t next() { s += 0x9E3779B97F4A7C15ULL; uint64_t x = s; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }
};

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: synth_fastq out.fq genome_len n_reads read_len err seed [threads [min_len]]\n"); return 2; }
    const char* path = argv[1];
    const uint64_t G = strtoull(argv[2], nullptr, 10), N = strtoull(argv[3], nullptr, 10);
    const int L = atoi(argv[4]);
    const double err = atof(argv[5]);
    const uint64_t seed = strtoull(argv[6], nullptr, 10);
    int nt = argc > 7 ? atoi(argv[7]) : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (G <= (uint64_t)L || L < 1 || N < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
    std::vector<uint8_t> genome(G);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&, t] { for (uint64_t i = G * t / nt; i < G * (t + 1) / nt; i++) genome[i] = (uint8_t)(mix(seed ^ (i * 0xD6E8FEB86659FD93ULL)) >> 62); });
        for (auto& th : pool) th.join();
    }
    const int minL = argc > 8 ? atoi(argv[8]) : 0;
    if (minL < 0 || minL > L) { fprintf(stderr, "bad min_len\n"); return 2; }
    auto len_of = [&](uint64_t r) -> int { return minL > 0 && minL < L ? minL + (int)(mix(seed * 0x9FB21C651E98DF25ULL ^ mix(r + 0x51ED27ULL)) % (uint64_t)(L - minL + 1)) : L; };
    char name_probe[32];
    const int name_w = snprintf(name_probe, sizeof name_probe, "@r%09llu\n", (unsigned long long)(N - 1));
    if (snprintf(name_probe, sizeof name_probe, "@r%09llu\n", 0ULL) != name_w) { fprintf(stderr, "n_reads too large for fixed-width names\n"); return 2; }
    const uint64_t rec = (uint64_t)name_w + L + 3 + L + 1;                      // the longest record
    const uint64_t BLOCK = 1 << 16;
    const uint64_t n_blocks = (N + BLOCK - 1) / BLOCK;
    // where every block of reads starts in the file (ragged reads: a pass over the lengths first)
    std::vector<uint64_t> block_at(n_blocks + 1, 0);
    {
        std::atomic<uint64_t> nb{0};
        auto sizes = [&] {
            for (;;) {
                const uint64_t b = nb.fetch_add(1);
                if (b >= n_blocks) break;
                uint64_t bytes = 0;
                const uint64_t lo = b * BLOCK, hi = lo + BLOCK < N ? lo + BLOCK : N;
                if (minL > 0 && minL < L) for (uint64_t r = lo; r < hi; r++) bytes += (uint64_t)name_w + 2 * (uint64_t)len_of(r) + 4;
                else bytes = (hi - lo) * rec;
                block_at[b + 1] = bytes;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(sizes);
        sizes();
        for (auto& th : pool) th.join();
        for (uint64_t b = 0; b < n_blocks; b++) block_at[b + 1] += block_at[b];
    }
    const bool pad = block_at[n_blocks] % 32768 == 0;
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { perror(path); return 1; }
    const uint64_t err_thr = err <= 0 ? 0 : (uint64_t)(err * 18446744073709551616.0);
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto body = [&] {
        std::vector<char> buf(BLOCK * rec + 2);
        for (;;) {
            const uint64_t lo = next.fetch_add(BLOCK);
            if (lo >= N) break;
            const uint64_t hi = lo + BLOCK < N ? lo + BLOCK : N;
            char* p = buf.data();
            for (uint64_t r = lo; r < hi; r++) {
                Rng g(seed, r);
                const uint64_t start = (uint64_t)(((unsigned __int128)g.next() * (G - L)) >> 64);
                const bool flip = g.next() >> 63;
                const int Lr = len_of(r);
                p += sprintf(p, "@r%09llu\n", (unsigned long long)r);
                for (int i = 0; i < Lr; i++) {
                    uint8_t c = flip ? (uint8_t)(genome[start + L - 1 - i] ^ 2) : genome[start + i];
                    if (err_thr) {
                        const uint64_t x = g.next();
                        if (x < err_thr) c = (uint8_t)((c + 1 + (mix(x) % 3)) & 3);
                    }
                    *p++ = "ACTG"[c];
                }
                *p++ = '\n'; *p++ = '+'; *p++ = '\n';
                memset(p, 'I', (size_t)Lr); p += Lr;
                if (pad && r == N - 1) *p++ = ' ';
                *p++ = '\n';
            }
            const size_t bytes = (size_t)(p - buf.data());
            const uint64_t at = block_at[lo / BLOCK];
            size_t done = 0;
            while (done < bytes) {
                const ssize_t w = pwrite(fd, buf.data() + done, bytes - done, (off_t)(at + done));
                if (w <= 0) { failed.store(1); return; }
                done += (size_t)w;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(body);
    body();
    for (auto& th : pool) th.join();
    close(fd);
    if (failed.load()) { fprintf(stderr, "short write on %s\n", path); return 1; }
    return 0;
}
