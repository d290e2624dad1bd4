// backend_hip.hpp -- the backend the product runs the device graph stages on (see backend.hpp): one HIP device, one stream.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <string.h>
#include <time.h>

#include <string>
#include <vector>

#include "backend.hpp"
#include "arena.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

// lane i of the grid calls f(i); grid-stride, so any n fits one launch
template <typename F>
__global__ __launch_bounds__(256) void be_launch_kernel(uint64_t n, F f) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) f(i);
}
template <typename T>
__global__ __launch_bounds__(256) void be_fill_kernel(T* p, uint64_t n, T v) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = v;
}
struct BeMaxLL { __host__ __device__ long long operator()(long long a, long long b) const { return a > b ? a : b; } };
// list[...] = the values f(i) != ~0 for i in [0, n), in any order; *cnt counts them all (also those beyond cap).  The workgroup collects
// its hits in LDS and reserves room for ~1000 of them with ONE returned atomic: one address serves about 88 returned atomics per
// microsecond on this chip whatever else the kernel does, and a list of millions of entries appended one atomic apiece was bound by
// exactly that (the dead-end listing of the tip stage: 7 M hits = 80 ms; the vertex listing: 2.6 M = 29 ms).
template <typename F>
__global__ __launch_bounds__(256) void be_append_kernel(uint64_t n, F f, unsigned long long* list, unsigned long long* cnt, unsigned long long cap) {
    constexpr unsigned FLUSH = 1024;
    __shared__ unsigned long long buf[FLUSH + 256];
    __shared__ unsigned int s_n;
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // `held` = the entries in buf, the same number in every lane (the barrier that ends a trip counts the trip's hits).  The workgroup decides to
    // flush on it and never on s_n: a wave a trip ahead adds to s_n while a slower one still reads it, and lanes that disagree about entering
    // flush() pair its barriers up wrongly
    unsigned int held = 0;
    auto flush = [&]() {                                                   // (called by the whole workgroup, behind a barrier)
        if (threadIdx.x == 0 && held) { s_base = atomicAdd(cnt, (unsigned long long)held); s_n = 0; }
        __syncthreads();
        for (unsigned int j = threadIdx.x; j < held; j += 256) { const unsigned long long at = s_base + j; if (at < cap) list[at] = buf[j]; }
        __syncthreads();
        held = 0;
    };
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256; i0 < n; i0 += (uint64_t)gridDim.x * 256) {        // (the same trips for every lane of the workgroup)
        const uint64_t i = i0 + threadIdx.x;
        const unsigned long long v = i < n ? f(i) : ~0ULL;
        if (v != ~0ULL) buf[atomicAdd(&s_n, 1u)] = v;
        held += (unsigned int)__syncthreads_count(v != ~0ULL ? 1 : 0);
        if (held > FLUSH) flush();
    }
    flush();
}

struct HipBackend {
    int device;
    hipStream_t stream;
    static constexpr bool on_device = true;
    int error = 0;
    std::string error_text;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    void* pinned = nullptr;                       // 256 bytes of page-locked host memory: the fixed points read a counter or two per round
    double t_readback = 0;                        // seconds spent in small read-backs (copy + wait), for the verbose lines
    unsigned long long n_readback = 0;
    // PG_HOST_VERBOSE with several places: the device time of the steps that run on the LEAD alone (launch, sort_pairs, the scans) against the steps
    // dealt to the places -- an event before and behind every step, read when the stage is over (lead_share)
    bool timing = false;
    struct Timed { hipEvent_t a, b; int place; };      // place -1: the lead alone
    std::vector<Timed> timed;
    void tick_begin(int pl, hipStream_t st, hipEvent_t& a) { a = nullptr; if (!timing) return; if (hipEventCreate(&a) != hipSuccess) { a = nullptr; return; } (void)pl; (void)hipEventRecord(a, st); }
    void tick_end(int pl, hipStream_t st, hipEvent_t a) {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
        (void)hipEventRecord(b, st);
        timed.push_back(Timed{a, b, pl});
    }
    // device milliseconds of the lead-only steps and of the steps at every place so far (everything must have been waited for); forgets them
    void lead_share(double& lead_ms, std::vector<double>& place_ms) {
        lead_ms = 0; place_ms.assign((size_t)n_places(), 0.0);
        for (Timed& t : timed) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { if (t.place < 0) lead_ms += ms; else if (t.place < (int)place_ms.size()) place_ms[t.place] += ms; }
            (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
        }
        timed.clear();
    }
    HipBackend(int device_, hipStream_t stream_) : device(device_), stream(stream_) {
        (void)hipSetDevice(device);
        if (hipHostMalloc(&pinned, 256, hipHostMallocPortable) != hipSuccess) pinned = nullptr;
    }
    // ---- places (backend.hpp): place p = a device of this process with a stream of its own; set s lives at set_place[s].  Empty:
    // one place, the lead itself.  A step launched at a place runs on that GPU, on memory allocated there; the caller orders it
    // against the lead's stream with sync() / sync_places() (kernel boundaries are also where writes through a peer mapping become
    // visible to the memory's owner).
    struct Place { int device; hipStream_t stream; };
    std::vector<Place> place;
    std::vector<int> set_place;
    std::vector<uint64_t> launches_at;            // launches per place, for the verbose lines and the tests
    std::vector<uint64_t> walks_at;               // walks dealt to a place (launch_walks_at)
    // the set geometry and the CRC table as a place reads them: its own copies when it is another GPU than the lead (a walk looks both up at
    // every step; through the lead's memory every step would cross xGMI twice more)
    struct PlaceView { const uint64_t* geo; const uint32_t* crc_tab; };
    std::vector<PlaceView> place_view;
    void use_places(const std::vector<Place>& pl, const std::vector<int>& of_set, const std::vector<PlaceView>& views = {}) {
        place = pl; set_place = of_set; place_view = views; launches_at.assign(pl.size(), 0); walks_at.assign(pl.size(), 0);
    }
    template <typename V> V view_at(int pl, V v) const {
        if (pl < (int)place_view.size() && place_view[pl].geo) { v.geo = place_view[pl].geo; v.crc_tab = place_view[pl].crc_tab; }
        return v;
    }
    int n_places() const { return place.empty() ? 1 : (int)place.size(); }
    int place_of_set(int s) const { return place.empty() || s >= (int)set_place.size() ? 0 : set_place[s]; }
    int dev_at(int pl) const { return place.empty() ? device : place[pl].device; }
    hipStream_t stream_at(int pl) const { return place.empty() ? stream : place[pl].stream; }
    template <typename T> T* alloc_at(int pl, size_t n) {
        void* p = nullptr;
        (void)hipSetDevice(dev_at(pl));
        const bool good = ok(pg::arena_malloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc (at a place)");
        (void)hipSetDevice(device);
        return good ? (T*)p : nullptr;
    }
    void release_at(int pl, void* p) { if (p) { (void)hipSetDevice(dev_at(pl)); (void)pg::arena_free(p); (void)hipSetDevice(device); } }
    template <typename T> void fill_at(int pl, T* p, size_t n, T v) {
        if (!n || error) return;
        (void)hipSetDevice(dev_at(pl));
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(be_fill_kernel<T>, dim3(grid), dim3(256), 0, stream_at(pl), p, (uint64_t)n, v);
        ok(hipGetLastError(), "fill (at a place)");
        (void)hipSetDevice(device);
    }
    template <typename T> void to_host_at(int pl, T* dst, const T* src, size_t n) {
        if (!n || error) return;
        (void)hipSetDevice(dev_at(pl));
        if (ok(hipMemcpyAsync((void*)dst, (const void*)src, n * sizeof(T), hipMemcpyDeviceToHost, stream_at(pl)), "copy to host (from a place)")) ok(hipStreamSynchronize(stream_at(pl)), "sync");
        (void)hipSetDevice(device);
    }
    // what a place listed, into the lead's memory (a peer copy on the lead's stream; the place's stream has been waited for)
    template <typename T> void gather_at(int, T* dst_lead, const T* src_place, size_t n) {
        if (n && !error) ok(hipMemcpyAsync((void*)dst_lead, (const void*)src_place, n * sizeof(T), hipMemcpyDefault, stream), "gather from a place");
    }
    template <typename F> void launch_at(int pl, uint64_t n, F f) {
        if (!n || error) return;
        (void)hipSetDevice(dev_at(pl));
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20);
        hipEvent_t ev; tick_begin(pl, stream_at(pl), ev);
        hipLaunchKernelGGL(be_launch_kernel<F>, dim3(grid), dim3(256), 0, stream_at(pl), n, f);
        tick_end(pl, stream_at(pl), ev);
        ok(hipGetLastError(), "launch (at a place)");
        if (pl < (int)launches_at.size()) launches_at[pl]++;
        (void)hipSetDevice(device);
    }
    // walks dealt to a place: lane q of the launch calls f(q) (the caller's share starts where it says); counted apart from the per-set scans
    template <typename F> void launch_walks_at(int pl, uint64_t n, F f) {
        if (!n || error) return;
        (void)hipSetDevice(dev_at(pl));
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20);
        hipEvent_t ev; tick_begin(pl, stream_at(pl), ev);
        hipLaunchKernelGGL(be_launch_kernel<F>, dim3(grid), dim3(256), 0, stream_at(pl), n, f);
        tick_end(pl, stream_at(pl), ev);
        ok(hipGetLastError(), "launch (walks at a place)");
        if (pl < (int)walks_at.size()) walks_at[pl] += n;
        (void)hipSetDevice(device);
    }
    // append the values f(i) != ~0, i in [0, n), to `list` at place pl (any order), counting them in *cnt (memory of that place)
    template <typename F> void append_at(int pl, uint64_t n, F f, unsigned long long* list, unsigned long long* cnt, unsigned long long cap) {
        if (!n || error) return;
        (void)hipSetDevice(dev_at(pl));
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 256u * 16u);
        hipEvent_t ev; tick_begin(pl, stream_at(pl), ev);
        hipLaunchKernelGGL(be_append_kernel<F>, dim3(grid), dim3(256), 0, stream_at(pl), n, f, list, cnt, cap);
        tick_end(pl, stream_at(pl), ev);
        ok(hipGetLastError(), "launch (append at a place)");
        if (pl < (int)launches_at.size()) launches_at[pl]++;
        (void)hipSetDevice(device);
    }
    void sync_places() {
        for (int pl = 0; pl < n_places() && !error; pl++) { (void)hipSetDevice(dev_at(pl)); ok(hipStreamSynchronize(stream_at(pl)), "sync (a place)"); }
        (void)hipSetDevice(device);
    }
    HipBackend(const HipBackend&) = delete;
    HipBackend& operator=(const HipBackend&) = delete;
    ~HipBackend() { if (tmp) (void)pg::arena_free(tmp); if (pinned) (void)hipHostFree(pinned); }

    bool ok(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        if (!error) { error = e == hipErrorOutOfMemory ? PG_ENOMEM : PG_ENODEV; error_text = std::string(what) + ": " + hipGetErrorString(e); }
        return false;
    }
    template <typename T> T* alloc(size_t n) {
        void* p = nullptr;
        (void)hipSetDevice(device);
        if (!ok(pg::arena_malloc(&p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc")) return nullptr;
        return (T*)p;
    }
    void release(void* p) { if (p) (void)pg::arena_free(p); }
    template <typename T> void fill(T* p, size_t n, T v) {
        if (!n || error) return;
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(be_fill_kernel<T>, dim3(grid), dim3(256), 0, stream, p, (uint64_t)n, v);
        ok(hipGetLastError(), "fill");
    }
    template <typename T> void to_host(T* dst, const T* src, size_t n) {
        if (!n || error) return;
        if (pinned && n * sizeof(T) <= 256) {                             // a pageable destination goes through a staging kernel and a second copy
            struct timespec a, b;
            clock_gettime(CLOCK_MONOTONIC, &a);
            if (ok(hipMemcpyAsync(pinned, (const void*)src, n * sizeof(T), hipMemcpyDeviceToHost, stream), "copy to host") && ok(hipStreamSynchronize(stream), "sync"))
                memcpy((void*)dst, pinned, n * sizeof(T));
            clock_gettime(CLOCK_MONOTONIC, &b);
            t_readback += (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
            n_readback++;
            return;
        }
        ok(hipMemcpyAsync((void*)dst, (const void*)src, n * sizeof(T), hipMemcpyDeviceToHost, stream), "copy to host");
        ok(hipStreamSynchronize(stream), "sync");
    }
    template <typename T> void to_device(T* dst, const T* src, size_t n) { if (n && !error) { ok(hipMemcpyAsync((void*)dst, (const void*)src, n * sizeof(T), hipMemcpyHostToDevice, stream), "copy to device"); ok(hipStreamSynchronize(stream), "sync"); } }
    template <typename T> void copy(T* dst, const T* src, size_t n) { if (n && !error) ok(hipMemcpyAsync((void*)dst, (const void*)src, n * sizeof(T), hipMemcpyDeviceToDevice, stream), "copy"); }
    void sync() { if (!error) ok(hipStreamSynchronize(stream), "sync"); }

    template <typename F> void launch(uint64_t n, F f) {
        if (!n || error) return;
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20);
        hipEvent_t ev; tick_begin(-1, stream, ev);
        hipLaunchKernelGGL(be_launch_kernel<F>, dim3(grid), dim3(256), 0, stream, n, f);
        tick_end(-1, stream, ev);
        ok(hipGetLastError(), "launch");
    }
    bool scratch(size_t bytes) {
        if (bytes <= tmp_bytes) return true;
        if (tmp) (void)pg::arena_free(tmp);
        tmp = nullptr; tmp_bytes = 0;
        if (!ok(pg::arena_malloc(&tmp, bytes), "hipMalloc (scratch)")) return false;
        tmp_bytes = bytes;
        return true;
    }
    template <typename KT, typename V> void sort_pairs(const KT* kin, KT* kout, const V* vin, V* vout, uint64_t n, int bits) {
        static_assert(sizeof(KT) == 8, "64-bit keys");
        if (!n || error) return;
        size_t need = 0;
        if (!ok((rocprim::radix_sort_pairs<rocprim::default_config, const KT*, KT*, const V*, V*, size_t>(
                    nullptr, need, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, stream)), "radix sort (size)")) return;
        if (!scratch(need)) return;
        hipEvent_t ev; tick_begin(-1, stream, ev);
        ok((rocprim::radix_sort_pairs<rocprim::default_config, const KT*, KT*, const V*, V*, size_t>(
               tmp, need, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, stream)), "radix sort");
        tick_end(-1, stream, ev);
    }
    void inclusive_max(const long long* in, long long* out, uint64_t n) {
        if (!n || error) return;
        size_t need = 0;
        if (!ok(rocprim::inclusive_scan(nullptr, need, in, out, (size_t)n, BeMaxLL(), stream), "scan (size)")) return;
        if (!scratch(need)) return;
        ok(rocprim::inclusive_scan(tmp, need, in, out, (size_t)n, BeMaxLL(), stream), "scan");
    }
    void exclusive_sum(const unsigned long long* in, unsigned long long* out, uint64_t n) {
        if (!n || error) return;
        size_t need = 0;
        if (!ok(rocprim::exclusive_scan(nullptr, need, in, out, 0ULL, (size_t)n, rocprim::plus<unsigned long long>(), stream), "sum (size)")) return;
        if (!scratch(need)) return;
        ok(rocprim::exclusive_scan(tmp, need, in, out, 0ULL, (size_t)n, rocprim::plus<unsigned long long>(), stream), "sum");
    }
};

}  // namespace pg
