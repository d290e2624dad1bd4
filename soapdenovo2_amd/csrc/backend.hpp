// backend.hpp -- one source for the device graph stages, two ways to run it.
//
// The stages after pass 1 that live on the device (k-mer-set layout for -a pools, tip clipping, the sharded regroup) are
// written once, as templates over a Backend, in dev_graph.hpp: data-parallel steps are function objects called with an
// index (`be.launch(n, f)`), the primitives between them (radix sort, scans) are the backend's.
//   HipBackend   (backend_hip.hpp, hipcc only): launch = a kernel whose lane i calls f(i); rocPRIM / hipCUB primitives on
//                the backend's stream; memory from hipMalloc.  This is what the product runs.
//   HostBackend  (below, any C++ compiler): launch = host threads each taking a strided share of the indices -- the same
//                function objects, the same atomics (through the hd_* wrappers), so a data race or an order dependence shows
//                up in the CPU tests; std:: primitives.  Used by the `-m "not gpu"` tests (pg_host_emu_* in
//                host_emu.cpp) and nowhere in the product path: there is no CPU fallback, call_pregraph never picks it.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kmer.hpp"

// attribute of a lambda handed to Backend::launch (PG_HD carries `inline`, which a lambda cannot)
#if defined(__HIPCC__)
#define PG_LAMBDA __host__ __device__
#else
#define PG_LAMBDA
#endif

namespace pg {

// ---- atomics on plain words, same spelling on both sides --------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
PG_HD unsigned long long hd_atomic_add(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
PG_HD unsigned int hd_atomic_add(unsigned int* p, unsigned int v) { return atomicAdd(p, v); }
PG_HD unsigned int hd_atomic_exch(unsigned int* p, unsigned int v) { return atomicExch(p, v); }
PG_HD unsigned long long hd_atomic_min(unsigned long long* p, unsigned long long v) { return atomicMin(p, v); }
PG_HD unsigned long long hd_atomic_max(unsigned long long* p, unsigned long long v) { return atomicMax(p, v); }
PG_HD unsigned long long hd_atomic_or(unsigned long long* p, unsigned long long v) { return atomicOr(p, v); }
PG_HD unsigned long long hd_atomic_and(unsigned long long* p, unsigned long long v) { return atomicAnd(p, v); }
PG_HD unsigned long long hd_atomic_cas(unsigned long long* p, unsigned long long expected, unsigned long long desired) { return atomicCAS(p, expected, desired); }
PG_HD unsigned long long hd_atomic_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
inline unsigned long long hd_atomic_add(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned int hd_atomic_add(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned int hd_atomic_exch(unsigned int* p, unsigned int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
inline unsigned long long hd_atomic_min(unsigned long long* p, unsigned long long v) {
    unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return cur;
}
inline unsigned long long hd_atomic_max(unsigned long long* p, unsigned long long v) {
    unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return cur;
}
inline unsigned long long hd_atomic_or(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long hd_atomic_and(unsigned long long* p, unsigned long long v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
inline unsigned long long hd_atomic_cas(unsigned long long* p, unsigned long long expected, unsigned long long desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;
}
inline unsigned long long hd_atomic_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
#endif

// ---- the host backend -----------------------------------------------------------------------------------------------------
struct HostBackend {
    int n_threads;
    explicit HostBackend(int threads = 0) : n_threads(threads > 0 ? threads : 4) {}
    static constexpr bool on_device = false;
    int error = 0;                                       // allocation failures etc. (PG_E*), sticky
    std::string error_text;

    template <typename T> T* alloc(size_t n) { return (T*)malloc(std::max<size_t>(n, 1) * sizeof(T)); }
    void release(void* p) { free(p); }
    template <typename T> void fill(T* p, size_t n, T v) { for (size_t i = 0; i < n; i++) p[i] = v; }
    template <typename T> void to_host(T* dst, const T* src, size_t n) { if (n) memcpy((void*)dst, (const void*)src, n * sizeof(T)); }
    template <typename T> void to_device(T* dst, const T* src, size_t n) { if (n) memcpy((void*)dst, (const void*)src, n * sizeof(T)); }
    template <typename T> void copy(T* dst, const T* src, size_t n) { if (n) memmove((void*)dst, (const void*)src, n * sizeof(T)); }
    void sync() {}

    // ---- places: where the k-mer sets live.  A sharded run keeps set s on the GPU of the rank that owns it; the steps that look at
    // one set at a time (the scans over all slots) run at the set's place, with their own lists and counters there, and hand the
    // lead a gathered result.  On the host every place is this memory; `places` > 1 only makes the CPU tests walk the same
    // bookkeeping (per-place lists, gathers) the sharded device run walks.
    int places = 1;
    int n_places() const { return places; }
    int place_of_set(int s) const { return s % places; }
    template <typename T> T* alloc_at(int, size_t n) { return alloc<T>(n); }
    void release_at(int, void* p) { release(p); }
    template <typename T> void fill_at(int, T* p, size_t n, T v) { fill(p, n, v); }
    template <typename T> void to_host_at(int, T* dst, const T* src, size_t n) { to_host(dst, src, n); }
    template <typename T> void gather_at(int, T* dst_lead, const T* src_place, size_t n) { copy(dst_lead, src_place, n); }
    template <typename F> void launch_at(int, uint64_t n, F f) { launch(n, f); }
    template <typename F> void launch_walks_at(int pl, uint64_t n, F f) { launch(n, f); if (pl >= 0 && pl < 8) walks_at[pl] += n; }
    template <typename V> V view_at(int, V v) const { return v; }
    uint64_t walks_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // append the values f(i) != ~0, i in [0, n), to `list` (any order), counting them all in *cnt
    template <typename F> void append_at(int, uint64_t n, F f, unsigned long long* list, unsigned long long* cnt, unsigned long long cap) {
        launch(n, [=](uint64_t i) {
            const unsigned long long v = f(i);
            if (v == ~0ULL) return;
            const unsigned long long at = hd_atomic_add(cnt, 1ULL);
            if (at < cap) list[at] = v;
        });
    }
    void sync_places() {}
    uint64_t launches_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (kept for symmetry with the device backend's per-place launch counts)

    // f(i) for every i in [0, n): thread t takes i = t, t + T, t + 2T, ... so that neighbouring indices run concurrently,
    // as the lanes of a wavefront do.  The threads are the backend's own and live as long as it does (round 6: a std::thread per
    // launch and thread was 14 of the CPU suite's 15 minutes in the kernel -- a fixed point is thousands of small launches).
    struct Workers {
        std::vector<std::thread> th;
        std::mutex mu;
        std::condition_variable go, done;
        uint64_t round = 0;
        int active = 0, pending = 0;
        bool stop = false;
        void (*fn)(void*, int, int) = nullptr;
        void* arg = nullptr;
        explicit Workers(int n) {
            for (int t = 1; t < n; t++)
                th.emplace_back([this, t] {
                    uint64_t seen = 0;
                    for (;;) {
                        void (*f)(void*, int, int);
                        void* a;
                        int T;
                        {
                            std::unique_lock<std::mutex> g(mu);
                            go.wait(g, [&] { return stop || round != seen; });
                            if (stop) return;
                            seen = round; f = fn; a = arg; T = active;
                        }
                        if (t < T) f(a, t, T);
                        {
                            std::lock_guard<std::mutex> g(mu);
                            if (--pending == 0) done.notify_one();
                        }
                    }
                });
        }
        ~Workers() {
            { std::lock_guard<std::mutex> g(mu); stop = true; }
            go.notify_all();
            for (auto& t : th) t.join();
        }
        void run(void (*f)(void*, int, int), void* a, int T) {             // the caller is thread 0
            {
                std::lock_guard<std::mutex> g(mu);
                fn = f; arg = a; active = T; pending = (int)th.size(); round++;
            }
            go.notify_all();
            f(a, 0, T);
            std::unique_lock<std::mutex> g(mu);
            done.wait(g, [&] { return pending == 0; });
        }
    };
    std::shared_ptr<Workers> workers;
    template <typename F> void launch(uint64_t n, F f) {
        const int T = (int)std::min<uint64_t>((uint64_t)n_threads, std::max<uint64_t>(n, 1));
        if (T <= 1) { for (uint64_t i = 0; i < n; i++) f(i); return; }
        if (!workers) workers = std::make_shared<Workers>(n_threads);
        struct Job { F* f; uint64_t n; } job{&f, n};
        workers->run([](void* a, int t, int TT) {
            Job& j = *(Job*)a;
            F g = *j.f;                                                   // (every thread its own copy, as the per-launch threads had)
            for (uint64_t i = (uint64_t)t; i < j.n; i += (uint64_t)TT) g(i);
        }, &job, T);
    }
    // stable sort of (key, value) pairs by the low `bits` bits of the key
    template <typename KT, typename V> void sort_pairs(const KT* kin, KT* kout, const V* vin, V* vout, uint64_t n, int bits) {
        static_assert(sizeof(KT) == 8, "64-bit keys");
        std::vector<uint64_t> idx(n);
        for (uint64_t i = 0; i < n; i++) idx[i] = i;
        const uint64_t mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
        std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (uint64_t i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
    }
    void inclusive_max(const long long* in, long long* out, uint64_t n) {
        long long m = 0;
        for (uint64_t i = 0; i < n; i++) { m = i ? std::max(m, in[i]) : in[i]; out[i] = m; }
    }
    void exclusive_sum(const unsigned long long* in, unsigned long long* out, uint64_t n) {
        unsigned long long s = 0;
        for (uint64_t i = 0; i < n; i++) { const unsigned long long v = in[i]; out[i] = s; s += v; }
    }
};

}  // namespace pg
