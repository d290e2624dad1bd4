// arena.cpp -- see arena.hpp.
#include "arena.hpp"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <algorithm>
#include <mutex>
#include <vector>

#include "arena_list.hpp"
#include "env.hpp"

namespace pg {
namespace {

constexpr int MAX_DEVICES = 64;

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Arena {
    int device = -1;
    bool tried = false, active = false;
    std::mutex list_mu;                             // the free list, the live blocks, the pins
    std::mutex map_mu;                              // the table of physical pieces (held per PIECE: a thread that backs tens of gigabytes does not make one that needs a
                                                    // single piece wait for all of them -- the export array's thread beside pass 1's batches)
    char* base = nullptr;
    size_t reserved = 0, chunk = 0;
    std::atomic<size_t> mapped_bytes{0};
    struct Piece { hipMemGenericAllocationHandle_t h; int state; };   // state: 0 none, 1 being created by somebody, 2 mapped
    std::vector<Piece> pieces;                      // one per `chunk` bytes of the range
    std::condition_variable map_cv;
    std::vector<hipMemAccessDesc> access;
    hipMemAllocationProp prop;
    BlockList blocks;                               // the free list and the live blocks (arena_list.hpp)
    int pins = 0;
    uint64_t n_chunks = 0;
    double map_seconds = 0;

    void init(int dev) {
        tried = true;
        device = dev;
        if (const char* e = env_user("SOAPDENOVO2_AMD_ARENA")) if (atoi(e) == 0) return;
        struct OnDevice {                                                   // (hipMemGetInfo and the reservation speak of the current device)
            int cur = -1, dev;
            explicit OnDevice(int d) : dev(d) { (void)hipGetDevice(&cur); if (cur != dev) (void)hipSetDevice(dev); }
            ~OnDevice() { if (cur >= 0 && cur != dev) (void)hipSetDevice(cur); }
        } on_device(dev);
        int vmm = 0;
        if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev) != hipSuccess || !vmm) return;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return;
        chunk = round_up((size_t)512 << 20, gran);                        // physical memory comes in pieces of this size
        // Twice the device's memory in ADDRESSES: blocks are cut first-fit and a freed hole keeps its physical pieces, so a pattern with a small live
        // block between large ones (a pool that grows, the export array's re-allocation) can need fresh addresses while physical memory is still there --
        // with a range of exactly the device's size it ran out of addresses first and failed where hipMalloc would not.
        reserved = round_up(2 * total_b, chunk);
        if (!reserve_range()) { reserved = round_up(total_b, chunk); if (!reserve_range()) return; }
        // who may touch the memory: this GPU, and every GPU of the process that can reach it (a sharded run probes the k-mer sets of
        // the other ranks through peer mappings)
        int n_dev = 0;
        (void)hipGetDeviceCount(&n_dev);
        for (int d = 0; d < n_dev; d++) {
            int can = d == dev;
            if (d != dev && hipDeviceCanAccessPeer(&can, d, dev) != hipSuccess) can = 0;
            if (!can) continue;
            hipMemAccessDesc a;
            memset(&a, 0, sizeof a);
            a.location.type = hipMemLocationTypeDevice;
            a.location.id = d;
            a.flags = hipMemAccessFlagsProtReadWrite;
            access.push_back(a);
        }
        if (const char* e = env_test("PG_ARENA_KEEP")) if (atoi(e) != 0) pins = 1;      // (test hook: the pieces stay for the life of the process)
        active = true;
    }
    // A fresh virtual range for the arena's next life.  A range whose pieces were given back (trim) is RETIRED, never mapped again: with new physical
    // memory mapped at addresses the GPU had translations for, contexts created behind a trim read back something else than they had just written --
    // five of six runs of the counting tests right behind a process that had released 150 GB failed that way, none with the pieces kept, none with plain
    // hipMalloc (profiles/r05w_arena_remap_hazard.txt).  Address space is not scarce: a retired range costs 288 GB of a 47-bit space.
    std::vector<std::pair<char*, size_t>> retired;
    bool reserve_range() {
        void* p = nullptr;
        if (hipMemAddressReserve(&p, reserved, 0, nullptr, 0) != hipSuccess || !p) { (void)hipGetLastError(); return false; }
        base = (char*)p;
        blocks.reset(reserved);
        pieces.assign(reserved / chunk, Piece{hipMemGenericAllocationHandle_t(), 0});
        return true;
    }

    // physical memory under [off, off + bytes); called without list_mu.  Pieces are created where they are needed, in any order; a piece somebody
    // else is creating is waited for, nothing else is.
    hipError_t ensure_piece(size_t pi) {
        std::vector<hipMemAccessDesc> access_now;                          // (a copy taken under the lock: another thread's fallback below rewrites `access`)
        {
            std::unique_lock<std::mutex> g(map_mu);
            while (pieces[pi].state == 1) map_cv.wait(g);
            if (pieces[pi].state == 2) return hipSuccess;
            pieces[pi].state = 1;
            access_now = access;
        }
        const auto t0 = std::chrono::steady_clock::now();
        hipMemGenericAllocationHandle_t h;
        char* at = base + pi * chunk;
        hipError_t rc = hipMemCreate(&h, chunk, &prop, 0);
        if (rc == hipSuccess) {
            rc = hipMemMap(at, chunk, 0, h, 0);
            if (rc == hipSuccess) {
                rc = hipMemSetAccess(at, chunk, access_now.data(), access_now.size());
                if (rc != hipSuccess && access_now.size() > 1) {
                    // (the peers could not all be granted -- a process that sees GPUs it never opened, one rank a process: this GPU alone then, for this
                    //  and every later piece; a sharded run INSIDE one process would have failed at its peer mappings anyway)
                    (void)hipGetLastError();
                    hipMemAccessDesc self = access_now[0];
                    for (const hipMemAccessDesc& a : access_now) if (a.location.id == device) self = a;
                    rc = hipMemSetAccess(at, chunk, &self, 1);
                    if (rc == hipSuccess) {
                        std::lock_guard<std::mutex> g(map_mu);
                        access.assign(1, self);
                        if (env_user("PG_HOST_VERBOSE")) fprintf(stderr, "arena (device %d): peer access could not be granted, the arena's memory is this GPU's alone\n", device);
                    }
                }
                if (rc != hipSuccess) (void)hipMemUnmap(at, chunk);
            }
            if (rc != hipSuccess) (void)hipMemRelease(h);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> g(map_mu);
            pieces[pi].state = rc == hipSuccess ? 2 : 0;
            if (rc == hipSuccess) { pieces[pi].h = h; n_chunks++; mapped_bytes.fetch_add(chunk, std::memory_order_relaxed); }
            map_seconds += dt;
        }
        map_cv.notify_all();
        if (rc != hipSuccess) (void)hipGetLastError();
        return rc;
    }
    hipError_t ensure_mapped(size_t off, size_t bytes) {
        const size_t p0 = off / chunk, p1 = (off + bytes + chunk - 1) / chunk;
        for (size_t pi = p0; pi < p1; pi++)
            if (ensure_piece(pi) != hipSuccess) return hipErrorOutOfMemory;
        return hipSuccess;
    }
    // list_mu held, nothing allocated, nobody pins: the physical memory goes back to the driver
    void trim() {
        const auto t_trim = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> g(map_mu);
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        size_t at = 0;
        for (size_t pi = 0; pi < pieces.size(); pi++)
            if (pieces[pi].state == 2) { (void)hipMemUnmap(base + pi * chunk, chunk); (void)hipMemRelease(pieces[pi].h); pieces[pi].state = 0; at += chunk; }
        mapped_bytes.store(0, std::memory_order_release);
        retired.emplace_back(base, reserved);                              // (see reserve_range: this range is never mapped again)
        if (!reserve_range()) {                                            // (no address space left: plain hipMalloc from here on -- said aloud, it changes every timing)
            active = false; base = nullptr;
            fprintf(stderr, "soapdenovo2_amd: arena (device %d): no address range for the arena's next life (%zu retired); device memory comes from hipMalloc from here on\n", device, retired.size());
        }
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        if (env_user("PG_HOST_VERBOSE"))
            fprintf(stderr, "arena (device %d): %.2f GB of physical memory in %llu piece(s), created in %.2fs in all; peak in use %.2f GB; %llu block(s) cut, %llu given back; given back to the driver in %.2fs\n", device,
                    (double)at / 1e9, (unsigned long long)n_chunks, map_seconds, (double)blocks.peak / 1e9, (unsigned long long)blocks.n_cut, (unsigned long long)blocks.n_back,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_trim).count());
    }

    // PG_ARENA_TRACE=1: every block of 64 MB and more as it is cut and given back, with what is in use behind it (stderr) -- what the executable's
    // memory plan (pg_host_plan_memory) was written from and is checked against
    void trace(const char* what, size_t bytes, uint64_t in_use_after) const {
        static const bool on = env_user("PG_ARENA_TRACE") != nullptr && atoi(env_user("PG_ARENA_TRACE")) != 0;
        static const auto t0 = std::chrono::steady_clock::now();
        if (on && bytes >= ((size_t)64 << 20))
            fprintf(stderr, "[arena %d] %8.3fs %s %9.3f GB -> in use %9.3f GB\n", device, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what, (double)bytes / 1e9, (double)in_use_after / 1e9);
    }
    hipError_t malloc_(void** out, size_t bytes) {
        size_t off = 0, need = 0;
        {
            std::lock_guard<std::mutex> g(list_mu);
            if (!blocks.cut(bytes, &off, &need)) return hipErrorOutOfMemory;
            trace("cut ", need, blocks.in_use);
        }
        const hipError_t rc = ensure_mapped(off, need);
        if (rc != hipSuccess) {
            std::lock_guard<std::mutex> g(list_mu);
            (void)blocks.give_back(off);
            return rc;
        }
        *out = base + off;
        return hipSuccess;
    }

    bool owns(const void* p) const { return active && (const char*)p >= base && (const char*)p < base + reserved; }

    hipError_t free_block(void* p) {
        // hipFree waits for the device: whoever frees a block behind a kernel that still reads it relies on that
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        const hipError_t rc = hipDeviceSynchronize();
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        std::lock_guard<std::mutex> g(list_mu);
        {
            auto it = blocks.used.find((size_t)((char*)p - base));
            const size_t bytes = it == blocks.used.end() ? 0 : it->second;
            if (!blocks.give_back((size_t)((char*)p - base))) return hipErrorInvalidValue;
            trace("back", bytes, blocks.in_use);
        }
        if (blocks.empty() && pins == 0) trim();
        return rc;
    }
};

std::mutex g_init_mu;
Arena* g_arena[MAX_DEVICES];

Arena* arena_of(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> g(g_init_mu);
    if (!g_arena[dev]) g_arena[dev] = new Arena();          // (never deleted: HIP calls from static destructors run after the runtime is gone)
    if (!g_arena[dev]->tried) g_arena[dev]->init(dev);
    return g_arena[dev];
}

}  // namespace

hipError_t arena_malloc(void** p, size_t bytes) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    Arena* a = arena_of(dev);
    if (!a || !a->active) return hipMalloc(p, bytes);
    rc = a->malloc_(p, bytes);
    // A block of the arena holds whatever its last user (of this process, or -- pieces are not cleared by the driver -- of an earlier one) left there,
    // where hipMalloc hands out zeroes.  PG_ARENA_POISON=1 (test hook) fills every block with 0xA5 so that code that counts on zeroes fails every time,
    // not once in a while.
    static const bool poison = env_test("PG_ARENA_POISON") != nullptr && atoi(env_test("PG_ARENA_POISON")) != 0;
    if (rc == hipSuccess && poison) { rc = hipMemset(*p, 0xA5, bytes); if (rc == hipSuccess) rc = hipDeviceSynchronize(); }     // (the fill is done before any stream of the caller touches the block)
    return rc;
}

hipError_t arena_free(void* p) {
    if (!p) return hipSuccess;
    {
        std::unique_lock<std::mutex> g(g_init_mu);
        for (int d = 0; d < MAX_DEVICES; d++) {
            Arena* a = g_arena[d];
            if (a && a->owns(p)) { g.unlock(); return a->free_block(p); }
        }
    }
    return hipFree(p);
}

void arena_shrink(void* p, size_t bytes) {
    if (!p) return;
    std::unique_lock<std::mutex> g(g_init_mu);
    for (int d = 0; d < MAX_DEVICES; d++) {
        Arena* a = g_arena[d];
        if (a && a->owns(p)) {
            g.unlock();
            std::lock_guard<std::mutex> gl(a->list_mu);
            const size_t off = (size_t)((char*)p - a->base);
            auto it = a->blocks.used.find(off);
            const size_t had = it == a->blocks.used.end() ? 0 : it->second;
            if (a->blocks.shrink(off, bytes)) a->trace("trim", had - a->blocks.used[off], a->blocks.in_use);
            return;
        }
    }
}

hipError_t arena_mem_info(size_t* free_bytes, size_t* total_bytes) {
    const hipError_t rc = hipMemGetInfo(free_bytes, total_bytes);
    if (rc != hipSuccess) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipSuccess;
    Arena* a = arena_of(dev);
    if (a && a->active) {
        std::lock_guard<std::mutex> g(a->list_mu);
        const size_t mapped = a->mapped_bytes.load();
        if (mapped > a->blocks.in_use) *free_bytes += mapped - a->blocks.in_use;
    }
    return hipSuccess;
}

void arena_pin(int device) {
    Arena* a = arena_of(device);
    if (!a || !a->active) return;
    std::lock_guard<std::mutex> g(a->list_mu);
    a->pins++;
}

// The API path (pg_create without a pin of the caller's): the device's arena is pinned for the life of the process, once -- a library or Python
// caller that creates and destroys contexts would otherwise pay an unmap, a fresh reservation and a re-creation of the pieces per cycle and retire
// 2 x 288 GB of address space each time.  pg_device_arena_unpin gives the pin up.
void arena_pin_for_process(int device) {
    Arena* a = arena_of(device);
    if (!a || !a->active) return;
    std::lock_guard<std::mutex> g(a->list_mu);
    if (a->pins == 0) a->pins = 1;
}

void arena_unpin(int device) {
    Arena* a = arena_of(device);
    if (!a || !a->active) return;
    std::lock_guard<std::mutex> g(a->list_mu);
    if (a->pins > 0) a->pins--;
    if (a->pins == 0 && a->blocks.empty() && a->mapped_bytes.load() != 0) a->trim();
}

ArenaStats arena_stats(int device) {
    ArenaStats s;
    memset(&s, 0, sizeof s);
    Arena* a = arena_of(device);
    if (!a || !a->active) return s;
    std::lock_guard<std::mutex> g(a->list_mu);
    s.active = 1;
    s.reserved = a->reserved;
    s.mapped = a->mapped_bytes.load();
    s.in_use = a->blocks.in_use;
    s.peak_in_use = a->blocks.peak;
    s.n_malloc = a->blocks.n_cut;
    s.n_free = a->blocks.n_back;
    s.n_chunks_created = a->n_chunks;
    s.map_seconds = a->map_seconds;
    return s;
}

}  // namespace pg
