// pregraph_kernels.hip -- hand-written CDNA4 (gfx950) kernels for pass 1 of SOAPdenovo2's pregraph and the
// C-ABI device operators declared in include/soapdenovo2_amd.h.
//
// What the reference does per k-mer occurrence (prlHashReads.c:163-259 chopKmer4read, hashFunction.c:155
// hash_kmer, newhash.c:473-528 put_kmerset) is one canonical k-mer, its two flanking bases and one
// insert-or-update into an open-addressed set.  The result per distinct k-mer (8 saturating arc counters,
// saturating total, `single`) is an order-independent reduction; only the slot inside the reference's
// table depends on arrival order.  So the device keeps its own hash set (free layout) plus, per k-mer, the
// 64-bit ordinal of its first occurrence (atomic min) and, per reference set, the ordinal of the last put;
// the host replays the reference layout from those (replay_layout / replay_streamed, host_graph.cpp).
//
// Device set: open addressing, linear probing, power-of-two capacity, one slot =
//   NW = 2:  { key[2], cnt, ord }            32 B, two slots per 64-B line
//   NW = 4:  { key[4], cnt, ord, pad[2] }    64 B
// key[0] (the most significant k-mer word) never has its top two bits set (2K mod 64 <= 62 for odd K), so
// ~0 = EMPTY and ~0-1 = LOCKED are free sentinels.  A new key is installed by CAS(key[0]: EMPTY -> LOCKED),
// returned exchanges for the other words, then an exchange that publishes key[0].  Every access to the set
// is an agent-scope atomic, so the protocol does not depend on workgroup -> XCD placement or on the
// non-coherent per-XCD L2s (MI355X_MICROARCH.md, "Correctness boundaries").
//
// HBM-bound integer/atomic work: no MFMA.  One lane per k-mer occurrence (window extraction from the 2-bit
// packed read instead of a serial roll, so every lane has exactly one probe sequence in flight).
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>

#include "kmer.hpp"
#include "arena.hpp"
#include "env.hpp"
#include "extract.hpp"
#include "device_ctx.hpp"
#include "e2_plan.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

constexpr uint64_t SLOT_EMPTY = ~0ULL;
constexpr uint64_t SLOT_LOCKED = ~0ULL - 1;
constexpr int ITEMS = 8;                 // k-mer occurrences per lane in the count kernels
constexpr int TILE = BLOCK * ITEMS;      // per block

template <int NW> struct SlotWords { static constexpr int value = (NW == 2) ? 4 : 8; };


__device__ __forceinline__ uint64_t aload(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t acas(uint64_t* p, uint64_t expected, uint64_t desired) {
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
__device__ __forceinline__ uint64_t axchg(uint64_t* p, uint64_t v) {
    return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- batch geometry ------------------------------------------------------------------------------------
struct Batch {
    const uint64_t* packed;
    const uint64_t* word_off;    // ragged only
    const uint64_t* kmer_base;   // ragged only, n_reads + 1
    uint64_t n_reads, n_kmers;
    uint32_t uniform_len;        // 0 = ragged
    uint32_t kpr;                // uniform: k-mers per read
    uint32_t wpr;                // uniform: words per read
};

// map a batch-local k-mer index g to (read words, j, len)
__device__ __forceinline__ void locate(const Batch& b, uint64_t g, int K, const uint64_t* lds_base, uint64_t r_lo,
                                       int n_lds, const uint64_t*& rd, int& j, int& len) {
    if (b.uniform_len) {
        uint64_t r = g / b.kpr;
        j = (int)(g - r * b.kpr);
        len = (int)b.uniform_len;
        rd = b.packed + r * b.wpr;
    } else {
        // largest i in [0, n_lds) with lds_base[i] <= g
        int lo = 0, hi = n_lds - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (lds_base[mid] <= g) lo = mid; else hi = mid - 1;
        }
        uint64_t r = r_lo + lo;
        j = (int)(g - lds_base[lo]);
        len = (int)(b.kmer_base[r + 1] - lds_base[lo]) + K - 1;
        rd = b.packed + b.word_off[r];
    }
}

// per block: first read of the tile and the staged prefix sums (ragged batches)
__device__ __forceinline__ void stage_tile(const Batch& b, uint64_t g0, uint64_t* lds_base, uint64_t& r_lo, int& n_lds) {
    if (b.uniform_len) { r_lo = 0; n_lds = 0; return; }
    // r_lo = largest r with kmer_base[r] <= g0 (same search in every lane; addresses are wave-uniform)
    uint64_t lo = 0, hi = b.n_reads - 1;
    while (lo < hi) {
        uint64_t mid = (lo + hi + 1) >> 1;
        if (b.kmer_base[mid] <= g0) lo = mid; else hi = mid - 1;
    }
    r_lo = lo;
    uint64_t avail = b.n_reads - r_lo;                 // every read has >= 1 k-mer, so TILE reads suffice
    n_lds = (int)(avail < (uint64_t)TILE ? avail : (uint64_t)TILE);
    for (int i = threadIdx.x; i < n_lds; i += BLOCK) lds_base[i] = b.kmer_base[r_lo + i];
    __syncthreads();
}

// ---- the set -------------------------------------------------------------------------------------------
template <int NW>
struct Table {
    uint64_t* slots;
    uint64_t mask;       // capacity - 1
};

// insert-or-update one occurrence (put_kmerset + set_new_kmer + update_kmer, newhash.c:473-528,74-140)
template <int NW>
__device__ __forceinline__ void table_put(const Table<NW>& t, const Kmer<NW>& key, int left, int right, uint64_t ord,
                                          DevCounters* ctr) {
    constexpr int SW = SlotWords<NW>::value;
    uint64_t h = kmer_mix<NW>(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask;) {
        uint64_t* s = t.slots + h * SW;
        uint64_t w0 = aload(s);
        if (w0 == SLOT_EMPTY) {
            uint64_t old = acas(s, SLOT_EMPTY, SLOT_LOCKED);
            if (old == SLOT_EMPTY) {
                // owner: fill the slot with returned read-modify-writes, wait for them, then publish key[0]
                uint64_t r = 0;
#pragma unroll
                for (int i = 1; i < NW; i++) r |= axchg(s + i, key.w[i]);
                r |= axchg(s + NW, node_first(left, right));
                r |= axchg(s + NW + 1, ord);
                asm volatile("" ::"v"(r) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                axchg(s, key.w[0]);
                atomicAdd(&ctr->n_distinct, 1ULL);
                return;
            }
            w0 = old;
        }
        if (w0 == SLOT_LOCKED) {            // another lane is filling this slot: look again
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        bool match = (w0 == key.w[0]);
#pragma unroll
        for (int i = 1; i < NW; i++) match = match && (aload(s + i) == key.w[i]);
        if (match) {
            uint64_t cur = aload(s + NW);
            for (;;) {
                uint64_t nxt = node_update(cur, left, right);
                if (nxt == cur) break;                      // saturated and already non-single
                uint64_t old = acas(s + NW, cur, nxt);
                if (old == cur) break;
                cur = old;
            }
            if (ord < aload(s + NW + 1))
                __hip_atomic_fetch_min(s + NW + 1, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        h = (h + 1) & t.mask;
        probes++;
    }
    atomicAdd(&ctr->overflow, 1ULL);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One slot snapshot in a single round trip: two 16-byte agent-scope (sc1) loads.  A word of a slot only ever
// goes from its initial value (~0) to its final key value (cnt / ord keep changing but are only used as CAS
// expectations / skip hints), so a torn or early snapshot is detected by the caller: see table_put_wide.
__device__ __forceinline__ void load_slot32(const uint64_t* s, uint64_t& a0, uint64_t& a1, uint64_t& b0, uint64_t& b1) {
    u32x4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(s)
                 : "memory");
    a0 = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
    a1 = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
    b0 = (uint64_t)hi.x | ((uint64_t)hi.y << 32);
    b1 = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
}

// Same semantics as table_put, two round trips per occurrence in the common (update) case instead of five:
// one 32-byte snapshot {key[0], key[1], cnt, ord}, then the CAS on cnt seeded with the snapshot.  NW = 2 only.
__device__ __forceinline__ void table_put_wide(const Table<2>& t, const Kmer<2>& key, int left, int right, uint64_t ord,
                                               DevCounters* ctr) {
    uint64_t h = kmer_mix<2>(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask;) {
        uint64_t* s = t.slots + h * 4;
        uint64_t w0, w1, cur, o;
        load_slot32(s, w0, w1, cur, o);
        if (w0 == SLOT_EMPTY) {
            uint64_t old = acas(s, SLOT_EMPTY, SLOT_LOCKED);
            if (old == SLOT_EMPTY) {
                uint64_t r = axchg(s + 1, key.w[1]);
                r |= axchg(s + 2, node_first(left, right));
                r |= axchg(s + 3, ord);
                asm volatile("" ::"v"(r) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                axchg(s, key.w[0]);
                atomicAdd(&ctr->n_distinct, 1ULL);
                return;
            }
            // lost the race: the slot is being filled (or was filled) by someone else -- look again
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (w0 == SLOT_LOCKED) { __builtin_amdgcn_s_sleep(1); continue; }
        if (w0 == key.w[0]) {
            // key[1] is written before key[0] is published, but the two halves of the snapshot are not one
            // atomic read: an unwritten (~0) key[1] next to a published key[0] is re-read before it is believed
            if (w1 == ~0ULL) w1 = aload(s + 1);
            if (w1 == key.w[1]) {
                for (;;) {
                    uint64_t nxt = node_update(cur, left, right);
                    if (nxt == cur) break;
                    uint64_t old = acas(s + 2, cur, nxt);
                    if (old == cur) break;
                    cur = old;
                }
                if (ord < o) __hip_atomic_fetch_min(s + 3, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        h = (h + 1) & t.mask;
        probes++;
    }
    atomicAdd(&ctr->overflow, 1ULL);
}

template <int NW, bool WIDE>
__device__ __forceinline__ void table_put_any(const Table<NW>& t, const Kmer<NW>& key, int left, int right, uint64_t ord,
                                              DevCounters* ctr) {
    if constexpr (WIDE && NW == 2) table_put_wide(t, key, left, right, ord, ctr);
    else table_put<NW>(t, key, left, right, ord, ctr);
}

// move one complete node into a fresh table (rehash on growth); keys are unique
template <int NW>
__device__ __forceinline__ void table_move(const Table<NW>& t, const Kmer<NW>& key, uint64_t cnt, uint64_t ord,
                                           DevCounters* ctr) {
    constexpr int SW = SlotWords<NW>::value;
    uint64_t h = kmer_mix<NW>(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask; probes++) {
        uint64_t* s = t.slots + h * SW;
        if (aload(s) == SLOT_EMPTY && acas(s, SLOT_EMPTY, key.w[0]) == SLOT_EMPTY) {
            // nobody looks keys up during a rehash, so the remaining words need no hand-shake
#pragma unroll
            for (int i = 1; i < NW; i++) s[i] = key.w[i];
            s[NW] = cnt;
            s[NW + 1] = ord;
            return;
        }
        h = (h + 1) & t.mask;
    }
    atomicAdd(&ctr->overflow, 1ULL);
}

// ---- kernels -------------------------------------------------------------------------------------------

template <int NW, bool WIDE>
__global__ __launch_bounds__(BLOCK) void count_reads_kernel(Batch b, Table<NW> t, int K, SetParams sp, uint64_t ord_base,
                                                            DevCounters* ctr) {
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned long long set_last[256];
    __shared__ uint64_t lds_base[TILE];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    set_last[threadIdx.x] = 0;
    const uint64_t g0 = (uint64_t)blockIdx.x * TILE;
    uint64_t r_lo; int n_lds;
    stage_tile(b, g0, lds_base, r_lo, n_lds);
    __syncthreads();
    const Kmer<NW> filter = kmer_filter<NW>(K);
#pragma unroll 1
    for (int it = 0; it < ITEMS; it++) {
        const uint64_t g = g0 + (uint64_t)it * BLOCK + threadIdx.x;
        if (g >= b.n_kmers) break;
        const uint64_t* rd; int j, len;
        locate(b, g, K, lds_base, r_lo, n_lds, rd, j, len);
        Occurrence occ;
        Kmer<NW> key = canonical_occurrence<NW>(rd, j, len, K, filter, occ);
        const uint64_t ord = ord_base + g;
        const uint32_t set = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
        atomicMax(&set_last[set], (unsigned long long)(ord + 1));
        table_put_any<NW, WIDE>(t, key, occ.left, occ.right, ord, ctr);
    }
    __syncthreads();
    if (threadIdx.x < sp.P && set_last[threadIdx.x]) atomicMax(&ctr->set_last[threadIdx.x], set_last[threadIdx.x]);
}

template <int NW, bool WIDE>
__global__ __launch_bounds__(BLOCK) void count_records_kernel(const uint64_t* recs, uint64_t n, Table<NW> t, SetParams sp,
                                                              DevCounters* ctr) {
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned long long set_last[256];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    set_last[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; g < n; g += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t* r = recs + g * (NW + 1);
        Kmer<NW> key;
#pragma unroll
        for (int i = 0; i < NW; i++) key.w[i] = r[i];
        const uint64_t meta = r[NW];
        const uint64_t ord = meta >> 6;
        const uint32_t set = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
        atomicMax(&set_last[set], (unsigned long long)(ord + 1));
        table_put_any<NW, WIDE>(t, key, (int)((meta >> 3) & 7), (int)(meta & 7), ord, ctr);
    }
    __syncthreads();
    if (threadIdx.x < sp.P && set_last[threadIdx.x]) atomicMax(&ctr->set_last[threadIdx.x], set_last[threadIdx.x]);
}

// multi-GPU routing, pass A: occurrences per owner
template <int NW>
__global__ __launch_bounds__(BLOCK) void route_count_kernel(Batch b, int K, SetParams sp, int n_owners,
                                                            unsigned long long* counts) {
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned int lcnt[256];
    __shared__ uint64_t lds_base[TILE];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    lcnt[threadIdx.x] = 0;
    const uint64_t g0 = (uint64_t)blockIdx.x * TILE;
    uint64_t r_lo; int n_lds;
    stage_tile(b, g0, lds_base, r_lo, n_lds);
    __syncthreads();
    const Kmer<NW> filter = kmer_filter<NW>(K);
#pragma unroll 1
    for (int it = 0; it < ITEMS; it++) {
        const uint64_t g = g0 + (uint64_t)it * BLOCK + threadIdx.x;
        if (g >= b.n_kmers) break;
        const uint64_t* rd; int j, len;
        locate(b, g, K, lds_base, r_lo, n_lds, rd, j, len);
        Occurrence occ;
        Kmer<NW> key = canonical_occurrence<NW>(rd, j, len, K, filter, occ);
        const uint32_t set = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
        atomicAdd(&lcnt[set % n_owners], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_owners && lcnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)lcnt[threadIdx.x]);
}

// multi-GPU routing, pass B: write records grouped by owner (block-aggregated range reservation)
template <int NW>
__global__ __launch_bounds__(BLOCK) void route_scatter_kernel(Batch b, int K, SetParams sp, uint64_t ord_base, int n_owners,
                                                              const uint64_t* owner_off, unsigned long long* cursor,
                                                              uint64_t* out) {
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned int lcnt[256];
    __shared__ unsigned long long lbase[256];
    __shared__ uint64_t lds_base[TILE];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    lcnt[threadIdx.x] = 0;
    const uint64_t g0 = (uint64_t)blockIdx.x * TILE;
    uint64_t r_lo; int n_lds;
    stage_tile(b, g0, lds_base, r_lo, n_lds);
    __syncthreads();
    const Kmer<NW> filter = kmer_filter<NW>(K);
    Kmer<NW> keys[ITEMS];
    uint64_t metas[ITEMS];
    uint32_t where[ITEMS];   // owner << 24 | rank inside the block
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint64_t g = g0 + (uint64_t)it * BLOCK + threadIdx.x;
        where[it] = 0xFFFFFFFFu;
        if (g < b.n_kmers) {
            const uint64_t* rd; int j, len;
            locate(b, g, K, lds_base, r_lo, n_lds, rd, j, len);
            Occurrence occ;
            keys[it] = canonical_occurrence<NW>(rd, j, len, K, filter, occ);
            metas[it] = ((ord_base + g) << 6) | ((uint64_t)occ.left << 3) | (uint64_t)occ.right;
            const uint32_t owner = set_of_crc(kmer_crc32<NW>(keys[it], crc_tab), sp.P, sp.bias) % n_owners;
            where[it] = (owner << 24) | atomicAdd(&lcnt[owner], 1u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_owners)
        lbase[threadIdx.x] = owner_off[threadIdx.x] + (lcnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)lcnt[threadIdx.x]) : 0ULL);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        if (where[it] != 0xFFFFFFFFu) {
            uint64_t* o = out + (lbase[where[it] >> 24] + (where[it] & 0xFFFFFFu)) * (NW + 1);
#pragma unroll
            for (int i = 0; i < NW; i++) o[i] = keys[it].w[i];
            o[NW] = metas[it];
        }
    }
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void rehash_kernel(const uint64_t* old_slots, uint64_t old_cap, Table<NW> t, DevCounters* ctr) {
    constexpr int SW = SlotWords<NW>::value;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < old_cap; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t* s = old_slots + i * SW;
        if (s[0] == SLOT_EMPTY) continue;
        Kmer<NW> key;
#pragma unroll
        for (int w = 0; w < NW; w++) key.w[w] = s[w];
        table_move<NW>(t, key, s[NW], s[NW + 1], ctr);
    }
}

// thread_delow + thread_mark + freqStat in one scan (prlHashReads.c:953-1132)
template <int NW>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(Table<NW> t, int D, DevCounters* ctr) {
    constexpr int SW = SlotWords<NW>::value;
    __shared__ unsigned int lhist[256];
    lhist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t cap = t.mask + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * BLOCK) {
        uint64_t* s = t.slots + i * SW;
        if (s[0] == SLOT_EMPTY) continue;
        uint64_t cnt = s[NW];
        uint32_t A = (uint32_t)cnt, B = (uint32_t)(cnt >> 32);
        int nin = 0, nout = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t l = (A >> (6 * c)) & 63u, r = (B >> (6 * c)) & 63u;
            if (D > 0 && l > 0 && l <= (uint32_t)D) { A &= ~(63u << (6 * c)); l = 0; }
            if (D > 0 && r > 0 && r <= (uint32_t)D) { B &= ~(63u << (6 * c)); r = 0; }
            nin += l > 0; nout += r > 0;
        }
        if (D > 0 && nin == 0 && nout == 0) B |= B_DELETED;
        if (nin == 1 && nout == 1) B |= B_LINEAR;
        atomicAdd(&lhist[A >> 24], 1u);
        s[NW] = (uint64_t)A | ((uint64_t)B << 32);
    }
    __syncthreads();
    if (lhist[threadIdx.x]) atomicAdd(&ctr->hist[threadIdx.x], (unsigned long long)lhist[threadIdx.x]);
}

// compact stored nodes into records: key words, cnt, set << 56 | first ordinal (wave ballot + prefix count)
template <int NW>
__global__ __launch_bounds__(BLOCK) void export_kernel(Table<NW> t, SetParams sp, uint64_t* out, uint64_t capacity, DevCounters* ctr) {
    constexpr int SW = SlotWords<NW>::value;
    __shared__ uint32_t crc_tab[256];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    __syncthreads();
    const uint64_t cap = t.mask + 1;
    const uint64_t span = (uint64_t)gridDim.x * BLOCK;
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = (uint64_t)blockIdx.x * BLOCK; i0 < cap; i0 += span) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t* s = t.slots + i * SW;
        const bool live = i < cap && s[0] != SLOT_EMPTY;
        const unsigned long long m = __ballot(live);
        if (m == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&ctr->n_export, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, 64);
        if (live) {
            const uint64_t pos = base + __popcll(m & ((1ULL << lane) - 1));
            if (pos < capacity) {
                Kmer<NW> key;
#pragma unroll
                for (int w = 0; w < NW; w++) key.w[w] = s[w];
                const uint32_t set = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
                uint64_t* o = out + pos * (NW + 2);
#pragma unroll
                for (int w = 0; w < NW; w++) o[w] = key.w[w];
                o[NW] = s[NW];
                o[NW + 1] = ((uint64_t)set << PG_ORD_BITS) | (s[NW + 1] & PG_ORD_MASK);
            }
        }
    }
}

}  // namespace pg

// =========================================================================================================
// C ABI
// =========================================================================================================
using namespace pg;

static thread_local std::string g_err;
extern "C" const char* pg_last_error(void) { return g_err.c_str(); }
extern "C" const char* pg_version(void) { return "soapdenovo2_amd 0.1 (gfx950)"; }
void pg_set_error(const std::string& s) { g_err = s; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return (e_ == hipErrorOutOfMemory) ? PG_ENOMEM : PG_ENODEV;                        \
        }                                                                                      \
    } while (0)


static inline uint64_t slot_bytes(int NW) { return (NW == 2 ? 4 : 8) * sizeof(uint64_t); }

extern "C" pg_ctx* pg_create_engine(int device, int K, int mer127, int n_sets, int log2_slots, int engine);
extern "C" pg_ctx* pg_create(int device, int K, int mer127, int n_sets, int log2_slots) {
    int engine = 2;
    if (const char* v = pg::env_user("PG_ENGINE")) engine = atoi(v);
    return pg_create_engine(device, K, mer127, n_sets, log2_slots, engine);
}

// (the rule itself: e2_plan.hpp, shared with the memory plan)
static int parts_for(uint64_t total_kmers, int nw, int n_owners = 1) {
    int shift = 0;
    if (const char* v = pg::env_measure("PG_PARTS_SHIFT")) shift = atoi(v);      // A/B runs: twice / half the partitions
    return pg::parts_for_kmers(total_kmers, nw, n_owners, shift);
}
extern "C" pg_ctx* pg_create_sized(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers);
extern "C" pg_ctx* pg_create_planned(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers, uint64_t expected_reads, uint64_t distinct_here, int n_owners);
extern "C" pg_ctx* pg_create_engine(int device, int K, int mer127, int n_sets, int log2_slots, int engine) {
    return pg_create_sized(device, K, mer127, n_sets, log2_slots, engine, 0);
}
extern "C" pg_ctx* pg_create_sized(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers) {
    return pg_create_planned(device, K, mer127, n_sets, log2_slots, engine, expected_kmers, 0, 0, 1);
}
// ... with everything pg_expect can be told known up front (no second allocation)
extern "C" pg_ctx* pg_create_planned(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers, uint64_t expected_reads, uint64_t distinct_here, int n_owners) {
    int n = 0;
    const bool trace = pg::env_user("PG_STARTUP_TRACE") && atoi(pg::env_user("PG_STARTUP_TRACE"));
    const auto t_in = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count(); };
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_err = "pg_create: no HIP device available"; return nullptr; }
    if (trace) fprintf(stderr, "[ctx]   %-44s %7.3f s\n", "hipGetDeviceCount (runtime start-up)", since());
    if (device < 0 || device >= n) { g_err = "pg_create: bad device ordinal"; return nullptr; }
    const int maxK = mer127 ? 127 : 63;
    if (K < 13 || K > maxK || (K & 1) == 0) { g_err = "pg_create: K must be odd and within 13.." + std::to_string(maxK); return nullptr; }
    if (n_sets < 1 || n_sets > 255) { g_err = "pg_create: n_sets must be 1..255"; return nullptr; }
    if (log2_slots < 10 || log2_slots > 40) { g_err = "pg_create: log2_slots out of range"; return nullptr; }
    if (engine != 1 && engine != 2) { g_err = "pg_create: engine must be 1 (global set) or 2 (partitions)"; return nullptr; }
    if (n_owners < 1 || n_owners > 4096) { g_err = "pg_create: bad number of owners"; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { g_err = "pg_create: hipSetDevice failed"; return nullptr; }
    pg::arena_pin_for_process(device);      // (a caller without a pin of its own -- call_pregraph holds one for the command -- keeps the arena's pieces for the process)
    pg_ctx* c = new pg_ctx();
    c->device = device; c->K = K; c->NW = mer127 ? 4 : 2; c->P = n_sets; c->log2_slots = log2_slots;
    c->ub_distinct = 0; c->finalized = false; c->autogrow = true; c->slots = nullptr; c->ctr = nullptr;
    c->variant = 1;
    c->engine = engine;
    if (engine == 2) { c->n_owners = n_owners; c->hint_distinct = distinct_here; c->hint_reads = expected_reads; }
    if (expected_kmers) { c->hint_kmers = expected_kmers; c->hint_log2_parts = parts_for(expected_kmers, c->NW, c->n_owners); }
    if (const char* v = pg::env_measure("PG_VARIANT")) c->variant = atoi(v);
    if (engine == 2) {
        if (pg::arena_malloc(&c->ctr, sizeof(DevCounters)) != hipSuccess) { g_err = "pg_create: hipMalloc failed"; delete c; return nullptr; }
        if (trace) fprintf(stderr, "[ctx]   %-44s %7.3f s (since the call)\n", "hipSetDevice + first hipMalloc", since());
        (void)hipMemset(c->ctr, 0, sizeof(DevCounters));
        if (e2_create(c) != PG_OK) { e2_destroy(c); (void)pg::arena_free(c->ctr); delete c; return nullptr; }
        (void)hipDeviceSynchronize();
        if (trace) fprintf(stderr, "[ctx]   %-44s %7.3f s (since the call)\n", "context ready", since());
        return c;
    }
    const size_t bytes = ((size_t)1 << log2_slots) * slot_bytes(c->NW);
    if (pg::arena_malloc(&c->slots, bytes) != hipSuccess) { g_err = "pg_create: hipMalloc of the k-mer set failed"; delete c; return nullptr; }
    if (pg::arena_malloc(&c->ctr, sizeof(DevCounters)) != hipSuccess) { g_err = "pg_create: hipMalloc failed"; pg::arena_free(c->slots); delete c; return nullptr; }
    hipMemset(c->slots, 0xFF, bytes);
    hipMemset(c->ctr, 0, sizeof(DevCounters));
    hipDeviceSynchronize();
    return c;
}

// Ragged batches: an upper bound on the read lengths to come (the reference's own lenBuffer has one too: maxReadLen sizes its
// buffers, prlHashReads.c:1251-1263).  The tiles of the super-k-mer cutter are sized for it; without one every ragged batch asks the
// device for its longest read and waits for the answer.  A read longer than the bound fails the pass (pg_finalize reports it).
extern "C" int pg_set_read_len_bound(pg_ctx* c, uint32_t max_len) {
    if (!c) { g_err = "null context"; return PG_EINVAL; }
    if (max_len && max_len < (uint32_t)c->K + 1) { g_err = "pg_set_read_len_bound: the bound is shorter than K + 1"; return PG_EINVAL; }
    c->read_len_bound = max_len;
    return PG_OK;
}

// The partition count of engine 2 follows the number of k-mer occurrences to come (about 8 k of them, i.e. some 400
// super-k-mer records, a partition), not the number of distinct k-mers the export array is sized for.  Before the first
// batch only; a no-op for engine 1.
extern "C" int pg_expect(pg_ctx* c, uint64_t total_kmers, uint64_t total_reads, uint64_t distinct_here, int n_owners) {
    if (!c) { g_err = "null context"; return PG_EINVAL; }
    if (c->engine != 2) return PG_OK;
    if (n_owners < 1 || n_owners > 4096) { g_err = "pg_expect: bad number of owners"; return PG_EINVAL; }
    if (c->batches) { g_err = "pg_expect: batches were already counted"; return PG_ESTATE; }
    const int lp = total_kmers ? parts_for(total_kmers, c->NW, n_owners) : c->hint_log2_parts;
    if (lp == c->e2.log2_global && c->hint_kmers == total_kmers && c->hint_reads == total_reads && c->hint_distinct == distinct_here && c->n_owners == n_owners) return PG_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int old = c->hint_log2_parts, old_own = c->n_owners;
    const uint64_t old_kmers = c->hint_kmers, old_distinct = c->hint_distinct, old_reads = c->hint_reads;
    e2_destroy(c);
    c->hint_log2_parts = lp;
    c->hint_kmers = total_kmers;
    c->hint_distinct = distinct_here;
    c->hint_reads = total_reads;
    c->n_owners = n_owners;
    int rc = e2_create(c);
    if (rc != PG_OK) {                       // e.g. no room for that many open chunks: keep the previous geometry
        e2_destroy(c);
        c->hint_log2_parts = old;
        c->hint_kmers = old_kmers;
        c->hint_distinct = old_distinct;
        c->hint_reads = old_reads;
        c->n_owners = old_own;
        const std::string why = g_err;
        if (e2_create(c) != PG_OK) return PG_ENOMEM;
        g_err = why;                         // (the context goes on as it was: one owner's storage takes any partition id)
    }
    (void)hipDeviceSynchronize();
    return PG_OK;
}
extern "C" int pg_expect_kmers(pg_ctx* c, uint64_t total_kmers) { return pg_expect(c, total_kmers, 0, 0, 1); }

// forget everything counted so far, keep the capacity (bench / repeated runs)
extern "C" int pg_reset(pg_ctx* c, void* stream) {
    if (!c) { g_err = "null context"; return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    (void)pg_ctx_drain(c, st);                                    // (records of a sharded round still in flight: let them land, then forget them)
    if (c->engine == 2) { int rc = e2_reset(c, st); if (rc) return rc; }
    else HIP_TRY(hipMemsetAsync(c->slots, 0xFF, ((size_t)1 << c->log2_slots) * slot_bytes(c->NW), st));
    HIP_TRY(hipMemsetAsync(c->ctr, 0, sizeof(DevCounters), st));
    c->batches = 0;
    c->ub_distinct = 0;
    c->finalized = false;
    return PG_OK;
}

extern "C" int pg_set_autogrow(pg_ctx* c, int on) {
    if (!c) { g_err = "null context"; return PG_EINVAL; }
    c->autogrow = on != 0;
    return PG_OK;
}

extern "C" void pg_destroy(pg_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    (void)pg_ctx_drain(c, nullptr);
    if (c->pending_detach) c->pending_detach(c, c->pending_user);
    if (c->engine == 2) e2_destroy(c);
    if (c->slots) pg::arena_free(c->slots);
    if (c->ctr) pg::arena_free(c->ctr);
    delete c;
}

static Batch make_batch(const pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                        uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers) {
    Batch b;
    b.packed = d_packed; b.word_off = d_word_off; b.kmer_base = d_kmer_base;
    b.n_reads = n_reads; b.n_kmers = n_kmers; b.uniform_len = uniform_len;
    b.kpr = uniform_len ? uniform_len - c->K + 1 : 0;
    b.wpr = uniform_len ? (uint32_t)((uniform_len + 31) / 32) : 0;
    return b;
}

static int check_batch(const pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                       uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers) {
    if (!c || !d_packed) { g_err = "null context or read buffer"; return PG_EINVAL; }
    if (uniform_len) {
        if (uniform_len < (uint32_t)c->K + 1) { g_err = "uniform_len must be >= K + 1"; return PG_EINVAL; }
        if (n_kmers != n_reads * (uint64_t)(uniform_len - c->K + 1)) { g_err = "n_kmers does not match n_reads * (len - K + 1)"; return PG_EINVAL; }
    } else if (!d_word_off || !d_kmer_base) { g_err = "ragged batch needs d_word_off and d_kmer_base"; return PG_EINVAL; }
    if (n_kmers >> PG_ORD_BITS) { g_err = "batch too large"; return PG_EINVAL; }
    return PG_OK;
}

template <int NW>
static int grow_to(pg_ctx* c, int new_log2, hipStream_t st) {
    uint64_t* fresh = nullptr;
    const size_t bytes = ((size_t)1 << new_log2) * slot_bytes(NW);
    HIP_TRY(pg::arena_malloc(&fresh, bytes));
    HIP_TRY(hipMemsetAsync(fresh, 0xFF, bytes, st));
    Table<NW> t{fresh, ((uint64_t)1 << new_log2) - 1};
    const uint64_t old_cap = (uint64_t)1 << c->log2_slots;
    const unsigned grid = (unsigned)std::min<uint64_t>((old_cap + BLOCK - 1) / BLOCK, 1u << 20);
    hipLaunchKernelGGL(rehash_kernel<NW>, dim3(grid), dim3(BLOCK), 0, st, c->slots, old_cap, t, c->ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(pg::arena_free(c->slots));
    c->slots = fresh; c->log2_slots = new_log2;
    return PG_OK;
}

// make room for `incoming` more keys in the worst case; load factor kept <= 0.7
static int ensure_capacity(pg_ctx* c, uint64_t incoming, hipStream_t st) {
    const double LOAD = 0.7;
    if (!c->autogrow) return PG_OK;          // caller sized the set; a full set is reported by pg_distinct / pg_finalize
    uint64_t cap = (uint64_t)1 << c->log2_slots;
    if ((double)(c->ub_distinct + incoming) <= LOAD * (double)cap) { c->ub_distinct += incoming; return PG_OK; }
    // the bound is loose (it counts occurrences): read the true count
    unsigned long long real = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&real, &c->ctr->n_distinct, sizeof real, hipMemcpyDeviceToHost));
    c->ub_distinct = real;
    int want = c->log2_slots;
    while ((double)(real + incoming) > LOAD * (double)((uint64_t)1 << want)) want++;
    if (want > c->log2_slots) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(pg::arena_mem_info(&free_b, &total_b));
        // cannot afford the worst case: fall back to the smallest table that holds what is stored plus
        // a quarter of the batch (a batch of reads is mostly repeats); the overflow flag guards the rest
        while (want > c->log2_slots + 1 && ((size_t)1 << want) * slot_bytes(c->NW) > free_b - (free_b >> 4)) want--;
        if (((size_t)1 << want) * slot_bytes(c->NW) > free_b - (free_b >> 4)) { g_err = "k-mer set does not fit in device memory"; return PG_ENOMEM; }
        int rc = (c->NW == 2) ? grow_to<2>(c, want, st) : grow_to<4>(c, want, st);
        if (rc) return rc;
    }
    c->ub_distinct += incoming;
    return PG_OK;
}

static SetParams set_params(const pg_ctx* c) { return SetParams{(uint32_t)c->P, set_bias((uint32_t)c->P)}; }

extern "C" int pg_count_reads(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                              uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, uint64_t ord_base, void* stream) {
    int rc = check_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    if (rc) return rc;
    if (c->finalized) { g_err = "pg_count_reads after pg_finalize"; return PG_ESTATE; }
    if (n_kmers == 0) return PG_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    c->batches++;
    if (c->engine == 2) return e2_scatter(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers, ord_base, st);
    rc = ensure_capacity(c, n_kmers, st);
    if (rc) return rc;
    Batch b = make_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    const uint64_t grid = (n_kmers + TILE - 1) / TILE;
    if (grid > 0x7FFFFFFFULL) { g_err = "batch too large for one launch"; return PG_EINVAL; }
    const uint64_t mask = ((uint64_t)1 << c->log2_slots) - 1;
    if (c->NW == 2) {
        Table<2> t{c->slots, mask};
        if (c->variant == 1)
            hipLaunchKernelGGL((count_reads_kernel<2, true>), dim3((unsigned)grid), dim3(BLOCK), 0, st, b, t, c->K, set_params(c), ord_base, c->ctr);
        else
            hipLaunchKernelGGL((count_reads_kernel<2, false>), dim3((unsigned)grid), dim3(BLOCK), 0, st, b, t, c->K, set_params(c), ord_base, c->ctr);
    } else {
        Table<4> t{c->slots, mask};
        hipLaunchKernelGGL((count_reads_kernel<4, false>), dim3((unsigned)grid), dim3(BLOCK), 0, st, b, t, c->K, set_params(c), ord_base, c->ctr);
    }
    HIP_TRY(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_count_records(pg_ctx* c, const uint64_t* d_records, uint64_t n, void* stream) {
    if (!c || (!d_records && n)) { g_err = "null argument"; return PG_EINVAL; }
    if (c->finalized) { g_err = "pg_count_records after pg_finalize"; return PG_ESTATE; }
    if (c->engine == 2) { g_err = "pg_count_records needs the global-set engine (pg_create_engine(..., 1))"; return PG_ESTATE; }
    if (n == 0) return PG_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_capacity(c, n, st);
    if (rc) return rc;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 1u << 22);
    const uint64_t mask = ((uint64_t)1 << c->log2_slots) - 1;
    if (c->NW == 2) {
        Table<2> t{c->slots, mask};
        if (c->variant == 1)
            hipLaunchKernelGGL((count_records_kernel<2, true>), dim3(grid), dim3(BLOCK), 0, st, d_records, n, t, set_params(c), c->ctr);
        else
            hipLaunchKernelGGL((count_records_kernel<2, false>), dim3(grid), dim3(BLOCK), 0, st, d_records, n, t, set_params(c), c->ctr);
    } else {
        Table<4> t{c->slots, mask};
        hipLaunchKernelGGL((count_records_kernel<4, false>), dim3(grid), dim3(BLOCK), 0, st, d_records, n, t, set_params(c), c->ctr);
    }
    HIP_TRY(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_route_count(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                              uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, int n_owners, uint64_t* d_counts,
                              void* stream) {
    int rc = check_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    if (rc) return rc;
    if (n_owners < 1 || n_owners > 256 || !d_counts) { g_err = "bad n_owners / d_counts"; return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * n_owners, st));
    if (n_kmers == 0) return PG_OK;
    Batch b = make_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    const uint64_t grid = (n_kmers + TILE - 1) / TILE;
    if (c->NW == 2)
        hipLaunchKernelGGL(route_count_kernel<2>, dim3((unsigned)grid), dim3(BLOCK), 0, st, b, c->K, set_params(c), n_owners, (unsigned long long*)d_counts);
    else
        hipLaunchKernelGGL(route_count_kernel<4>, dim3((unsigned)grid), dim3(BLOCK), 0, st, b, c->K, set_params(c), n_owners, (unsigned long long*)d_counts);
    HIP_TRY(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_route_scatter(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                                uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, uint64_t ord_base, int n_owners,
                                const uint64_t* d_owner_off, uint64_t* d_cursor, uint64_t* d_out, void* stream) {
    int rc = check_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    if (rc) return rc;
    if (n_owners < 1 || n_owners > 256 || !d_owner_off || !d_cursor || !d_out) { g_err = "bad routing arguments"; return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(d_cursor, 0, sizeof(uint64_t) * n_owners, st));
    if (n_kmers == 0) return PG_OK;
    Batch b = make_batch(c, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, n_kmers);
    const uint64_t grid = (n_kmers + TILE - 1) / TILE;
    if (c->NW == 2)
        hipLaunchKernelGGL(route_scatter_kernel<2>, dim3((unsigned)grid), dim3(BLOCK), 0, st, b, c->K, set_params(c), ord_base, n_owners, d_owner_off, (unsigned long long*)d_cursor, d_out);
    else
        hipLaunchKernelGGL(route_scatter_kernel<4>, dim3((unsigned)grid), dim3(BLOCK), 0, st, b, c->K, set_params(c), ord_base, n_owners, d_owner_off, (unsigned long long*)d_cursor, d_out);
    HIP_TRY(hipGetLastError());
    return PG_OK;
}

static int check_overflow(pg_ctx* c, DevCounters& h) {
    HIP_TRY(hipMemcpy(&h, c->ctr, sizeof h, hipMemcpyDeviceToHost));
    if (h.overflow) { g_err = "device k-mer set overflowed (" + std::to_string(h.overflow) + " occurrences lost)"; return PG_ENOMEM; }
    return PG_OK;
}

extern "C" int pg_distinct(pg_ctx* c, uint64_t* out, void* stream) {
    if (!c || !out) { g_err = "null argument"; return PG_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    DevCounters h;
    int rc = check_overflow(c, h);
    if (rc) return rc;
    if (c->engine == 2) {
        if (!c->e2.counted) { g_err = "pg_distinct: the partition engine knows the count only after pg_finalize"; return PG_ESTATE; }
        *out = h.n_export;
        return PG_OK;
    }
    *out = h.n_distinct;
    return PG_OK;
}

extern "C" int pg_skm_route(pg_ctx* c, const uint64_t* d_packed, uint64_t n_reads, uint32_t uniform_len, uint64_t ord_base, int n_owners,
                            uint64_t* d_send_records, uint32_t* d_send_parts, uint64_t capacity_per_owner, uint64_t* d_counts, void* stream) {
    if (!c || !d_packed || !d_send_records || !d_send_parts || !d_counts) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2) { g_err = "pg_skm_route needs the partition engine"; return PG_ESTATE; }
    if (n_owners < 1 || n_owners > 256) { g_err = "bad n_owners"; return PG_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (!uniform_len) { g_err = "pg_skm_route needs a uniform-length batch (pg_count_reads_sharded takes ragged ones)"; return PG_EINVAL; }
    return e2_route(c, d_packed, nullptr, nullptr, n_reads, uniform_len, ord_base, n_owners, d_send_records, d_send_parts, capacity_per_owner,
                    d_counts, (hipStream_t)stream);
}

extern "C" int pg_skm_ingest(pg_ctx* c, const uint64_t* d_records, const uint32_t* d_parts, uint64_t n_records, void* stream) {
    if (!c || ((!d_records || !d_parts) && n_records)) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2) { g_err = "pg_skm_ingest needs the partition engine"; return PG_ESTATE; }
    if (c->finalized) { g_err = "pg_skm_ingest after pg_finalize"; return PG_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    return e2_ingest(c, d_records, d_parts, n_records, (hipStream_t)stream);
}

extern "C" int pg_stats(pg_ctx* c, uint64_t out[8]) {
    if (!c || !out) { g_err = "null argument"; return PG_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    DevCounters h;
    HIP_TRY(hipMemcpy(&h, c->ctr, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = (uint64_t)c->engine;
    out[1] = c->engine == 2 ? h.n_export : h.n_distinct;
    if (c->engine == 2) {
        out[2] = h.n_records;
        out[3] = (uint64_t)c->e2.rs * 8;
        out[4] = 0;
        for (uint32_t q = 0; q < POOL_SUBS; q++) out[4] += h.pool_sub[q * 8];
        out[5] = c->e2.pool_chunks;
        out[6] = (uint64_t)1 << c->e2.log2_parts;
        out[7] = c->e2.out_capacity;
    } else {
        out[6] = (uint64_t)1 << c->log2_slots;
        out[3] = slot_bytes(c->NW);
    }
    return PG_OK;
}

extern "C" int pg_table_info(pg_ctx* c, uint64_t* slots, uint32_t* sbytes) {
    if (!c) { g_err = "null context"; return PG_EINVAL; }
    if (c->engine == 2) {
        if (slots) *slots = c->e2.out_capacity;
        if (sbytes) *sbytes = (uint32_t)((c->NW + 2) * 8);
        return PG_OK;
    }
    if (slots) *slots = (uint64_t)1 << c->log2_slots;
    if (sbytes) *sbytes = (uint32_t)slot_bytes(c->NW);
    return PG_OK;
}

extern "C" int pg_finalize(pg_ctx* c, int delow, uint64_t hist_out[256], uint64_t* set_last_put_out, void* stream) {
    if (!c || !hist_out) { g_err = "null argument"; return PG_EINVAL; }
    if (c->finalized) { g_err = "pg_finalize called twice"; return PG_ESTATE; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = pg_ctx_drain(c, st); if (rc) return rc; }      // the last sharded round's records are appended first
    if (c->engine == 2) {
        int rc = e2_count(c, delow, set_last_put_out != nullptr, st);
        if (rc) return rc;
        DevCounters h;
        HIP_TRY(hipMemcpy(&h, c->ctr, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; i++) hist_out[i] = h.hist[i];
        if (set_last_put_out) for (int i = 0; i < c->P; i++) set_last_put_out[i] = h.set_last[i];
        c->finalized = true;
        return PG_OK;
    }
    const uint64_t cap = (uint64_t)1 << c->log2_slots;
    const unsigned grid = (unsigned)std::min<uint64_t>((cap + BLOCK - 1) / BLOCK, 256u * 32u);
    if (c->NW == 2) {
        Table<2> t{c->slots, cap - 1};
        hipLaunchKernelGGL(finalize_kernel<2>, dim3(grid), dim3(BLOCK), 0, st, t, delow, c->ctr);
    } else {
        Table<4> t{c->slots, cap - 1};
        hipLaunchKernelGGL(finalize_kernel<4>, dim3(grid), dim3(BLOCK), 0, st, t, delow, c->ctr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    DevCounters h;
    int rc = check_overflow(c, h);
    if (rc) return rc;
    for (int i = 0; i < 256; i++) hist_out[i] = h.hist[i];
    if (set_last_put_out) for (int i = 0; i < c->P; i++) set_last_put_out[i] = h.set_last[i];
    c->finalized = true;
    return PG_OK;
}

// Partition engine: the per-set last put on demand (K3 is a second expansion of every record, and the layout replay needs
// its answer only when some set ends exactly at a growth threshold: pg_host_last_put_matters), and the per-set counts
// that decision is taken from.
extern "C" int pg_set_counts(pg_ctx* c, uint64_t out[256], void* stream) {
    if (!c || !out) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2) { g_err = "pg_set_counts needs the partition engine"; return PG_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    return e2_set_counts(c, out, (hipStream_t)stream);
}
extern "C" int pg_last_put(pg_ctx* c, uint64_t* set_last_put_out, void* stream) {
    if (!c || !set_last_put_out) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2) { g_err = "pg_last_put needs the partition engine (the global-set engine returns it from pg_finalize)"; return PG_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    return e2_last_put(c, set_last_put_out, (hipStream_t)stream);
}

// Partition engine only: hand the export array itself over (no copy) and let go of everything else the context holds on
// the device -- the record pool, the chunk table, the cursors.  The caller owns *d_records_out (hipFree) from here on; the
// context can only be destroyed afterwards.  Halves the device memory needed at the hand-over to the graph stages.
extern "C" int pg_export_take_ws(pg_ctx* c, uint64_t** d_records_out, uint64_t* n_out, void** d_workspace_out, uint64_t* workspace_bytes_out) {
    if (!c || !d_records_out || !n_out) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2) { g_err = "pg_export_take needs the partition engine"; return PG_ESTATE; }
    if (!c->e2.counted) { g_err = "pg_export_take: call pg_finalize first"; return PG_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, &c->ctr->n_export, sizeof n, hipMemcpyDeviceToHost));
    *d_records_out = c->e2.out;
    *n_out = n;
    c->e2.out = nullptr;
    c->e2.out_capacity = 0;
    if (d_workspace_out && workspace_bytes_out) {           // the record pool changes hands as well, as scratch memory
        *d_workspace_out = c->e2.pool;
        *workspace_bytes_out = c->e2.pool_chunks * (uint64_t)c->e2.rpc * (uint64_t)c->e2.rs * 8;
        c->e2.pool = nullptr;
    }
    e2_destroy(c);                                       // frees what is left (pool, tables); the context is spent
    c->e2.counted = false;
    return PG_OK;
}
// a read-only look at the partition engine's export array (valid until the next pg_reset / pg_export_take / pg_destroy)
extern "C" int pg_export_peek(pg_ctx* c, const uint64_t** d_records_out, uint64_t* n_out) {
    if (!c || !d_records_out || !n_out) { g_err = "null argument"; return PG_EINVAL; }
    if (c->engine != 2 || !c->e2.counted) { g_err = "pg_export_peek: partition engine after pg_finalize only"; return PG_ESTATE; }
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, &c->ctr->n_export, sizeof n, hipMemcpyDeviceToHost));
    *d_records_out = c->e2.out;
    *n_out = n;
    return PG_OK;
}
extern "C" int pg_export_take(pg_ctx* c, uint64_t** d_records_out, uint64_t* n_out) { return pg_export_take_ws(c, d_records_out, n_out, nullptr, nullptr); }

// hipFree for callers that do not link the HIP runtime themselves (what pg_export_take hands over)
extern "C" void pg_device_free(void* d_ptr) { if (d_ptr) (void)pg::arena_free(d_ptr); }
extern "C" void pg_device_arena_pin(int device) { pg::arena_pin(device); }
extern "C" void pg_device_arena_unpin(int device) { pg::arena_unpin(device); }
extern "C" void pg_device_arena_stats(int device, uint64_t out[8]) {
    const pg::ArenaStats s = pg::arena_stats(device);
    out[0] = (uint64_t)s.active; out[1] = s.reserved; out[2] = s.mapped; out[3] = s.in_use; out[4] = s.peak_in_use; out[5] = s.n_malloc; out[6] = s.n_chunks_created;
    out[7] = (uint64_t)(s.map_seconds * 1e6);
}

extern "C" int pg_export(pg_ctx* c, uint64_t* d_records, uint64_t capacity, uint64_t* n_out, void* stream) {
    if (!c || !d_records || !n_out) { g_err = "null argument"; return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(c->device));
    if (c->engine == 2) {
        if (!c->e2.counted) { g_err = "pg_export: call pg_finalize first"; return PG_ESTATE; }
        unsigned long long n = 0;
        HIP_TRY(hipMemcpy(&n, &c->ctr->n_export, sizeof n, hipMemcpyDeviceToHost));
        if (n > capacity) { g_err = "pg_export: capacity too small"; return PG_EINVAL; }
        HIP_TRY(hipMemcpyAsync(d_records, c->e2.out, n * (uint64_t)(c->NW + 2) * 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        *n_out = n;
        return PG_OK;
    }
    HIP_TRY(hipMemsetAsync(&c->ctr->n_export, 0, sizeof(unsigned long long), st));
    const uint64_t cap = (uint64_t)1 << c->log2_slots;
    const unsigned grid = (unsigned)std::min<uint64_t>((cap + BLOCK - 1) / BLOCK, 256u * 32u);
    if (c->NW == 2) {
        Table<2> t{c->slots, cap - 1};
        hipLaunchKernelGGL(export_kernel<2>, dim3(grid), dim3(BLOCK), 0, st, t, set_params(c), d_records, capacity, c->ctr);
    } else {
        Table<4> t{c->slots, cap - 1};
        hipLaunchKernelGGL(export_kernel<4>, dim3(grid), dim3(BLOCK), 0, st, t, set_params(c), d_records, capacity, c->ctr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    DevCounters h;
    int rc = check_overflow(c, h);
    if (rc) return rc;
    if (h.n_export > capacity) { g_err = "pg_export: capacity too small"; return PG_EINVAL; }
    *n_out = h.n_export;
    return PG_OK;
}
