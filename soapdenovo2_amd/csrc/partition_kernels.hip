// partition_kernels.hip -- pass 1 as "partition, then count in LDS" (engine 2), gfx950.
//
// Same contract as the global-set engine of pregraph_kernels.hip (per distinct canonical k-mer: the reference's
// two node words, first-occurrence ordinal, set id; prlHashReads.c:163-259 + newhash.c:473-528), different
// formulation, chosen because a DRAM-resident set is capped by the chip's random-atomic rate (~24 G ops/s,
// profiles/r01_membench_random_access.log) at ~13 % of the HBM roofline:
//
//   K1  skm_scatter_kernel   one lane per read: cut the read into super-k-mers by minimizer partition
//                            (skm.hpp), append each as a fixed-size record to its partition's stream
//                            (streams = chains of 1.5 KB chunks from one pool; one atomic per record, not per
//                            k-mer: ~20x fewer, and the write side is plain 48-byte stores).
//   K2  skm_count_kernel     one workgroup per partition (persistent grid): expand the partition's records
//                            back into k-mer occurrences and insert them into a set that lives in LDS
//                            (64 KB, word-wise CAS claim from the all-ones pattern, 63-bit key words), then
//                            finalize (-d filter, linear flag, coverage histogram: prlHashReads.c:953-1132)
//                            and emit the distinct k-mers as export records.  All occurrences of a k-mer
//                            are in ONE partition, so its LDS result is final: no global table at all.
//                            A partition that does not fit the LDS set is split by key-hash bits and
//                            re-read (the records are small and L2-resident).
//   K3  skm_lastput_kernel   per reference set the ordinal of the last put (needed by the host layout replay
//                            only in a boundary case, so it is a separate, optional pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <algorithm>

#include "device_ctx.hpp"
#include "extract.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

constexpr uint64_t L_EMPTY = ~0ULL;
constexpr unsigned long long F_POOL = 1, F_CHUNKS = 2, F_OUT = 4, F_SPLIT = 8;

template <int NW> struct E2Cfg;
// LDS slot = KW key words | ord | 10 x u32 counters (L[4], R[4], puts, spare)
template <> struct E2Cfg<2> { static constexpr int PW = 5, KW = 2; };    // LDS slot: 2 key words + ord + 9 counters = 60 B
template <> struct E2Cfg<4> { static constexpr int PW = 7, KW = 5; };    // 5 key words + ord + 9 counters = 84 B

struct E2Dev {
    SkmGeom g;
    uint32_t rpc, maxc;
    uint64_t pool_chunks;
    uint32_t* cursor;
    uint32_t* chunk_tbl;
    uint64_t* pool;
    uint64_t* out;
    uint64_t out_capacity;
};

struct ReadsArg {
    const uint64_t* packed;
    const uint64_t* word_off;
    const uint64_t* kmer_base;
    uint64_t n_reads;
    uint32_t uniform_len, kpr, wpr;
    uint64_t ord_base;
};

// address of record q of partition pid.  The lane that draws the first record of a chunk (q % rpc == 0) takes a
// chunk from the pool and publishes its id; lanes with later records of the same chunk wait for the id.
// Deadlock freedom inside a wavefront: the publish is NOT the other arm of the wait (`if (first) publish; else wait`
// lets the compiler run the wait arm first and starve the publisher of the same wave); every lane runs
// "publish if first" and THEN the wait loop, and a publisher never waits for anything before its store.  The lane
// with q % rpc == 0 drew its number before any lane with a later number of that chunk, so its publish is already
// issued (same or earlier instruction of this wave, or an independent wave).  The wait is bounded anyway.
__device__ __forceinline__ uint64_t* record_slot(const E2Dev& e, uint32_t pid, uint32_t q, DevCounters* ctr, int rw) {
    const uint32_t ci = q / e.rpc, ri = q % e.rpc;
    if (ci >= e.maxc) { atomicOr(&ctr->e2_flags, F_CHUNKS); return nullptr; }
    uint32_t* t = e.chunk_tbl + (uint64_t)pid * e.maxc + ci;
    if (ri == 0) {
        const unsigned long long nc = atomicAdd(&ctr->pool_next, 1ULL) + 1;
        const uint32_t id = nc > e.pool_chunks ? 0xFFFFFFFFu : (uint32_t)nc;     // 0xFFFFFFFF = "pool exhausted", releases the waiters too
        if (nc > e.pool_chunks) atomicOr(&ctr->e2_flags, F_POOL);
        __hip_atomic_store(t, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t c = __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; c == 0 && spin < (1 << 20); spin++) {
        __builtin_amdgcn_s_sleep(2);
        c = __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (c == 0) { atomicOr(&ctr->e2_flags, F_POOL); return nullptr; }
    if (c == 0xFFFFFFFFu) return nullptr;
    return e.pool + ((uint64_t)(c - 1) * e.rpc + ri) * (uint64_t)rw;
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void skm_scatter_kernel(ReadsArg a, E2Dev e, DevCounters* ctr) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1;
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= a.n_reads) return;
    const uint64_t* rd;
    int len;
    uint64_t ord0;
    if (a.uniform_len) {
        rd = a.packed + r * a.wpr; len = (int)a.uniform_len; ord0 = a.ord_base + r * a.kpr;
    } else {
        const uint64_t kb = a.kmer_base[r];
        rd = a.packed + a.word_off[r]; len = (int)(a.kmer_base[r + 1] - kb) + e.g.K - 1; ord0 = a.ord_base + kb;
    }
    skm_split_read(rd, len, e.g, [&](int j0, int n, uint32_t pid) {
        const uint32_t q = atomicAdd(&e.cursor[pid], 1u);
        uint64_t* dst = record_slot(e, pid, q, ctr, RW);
        if (!dst) return;
        uint64_t rec[RW];
        skm_make_record<PW>(rd, len, j0, n, ord0, e.g, rec);
#pragma unroll
        for (int i = 0; i < RW; i++) dst[i] = rec[i];
    });
}

// Tiled K1 for uniform-length reads: no data-dependent control flow per k-mer.  A workgroup takes R reads:
//   A  every m-mer value of the tile into LDS (one lane per m-mer position, 32-bit values)
//   B  sliding-window minimum by doubling: V_k[p] = min(V_{k-1}[p], V_{k-1}[p + 2^(k-1)]), window = two overlapping V_lv
//   C  partition of every k-mer; D  run starts (a local rule, skm.hpp) and run lengths -> compact item list in LDS
//   E  one lane per run: reserve a record slot in the partition's stream, build the record from the staged words,
//      store it with 16-byte writes.
// Only E touches global memory besides the coalesced tile load.  Index arithmetic is 32-bit with multiply-high
// reciprocals (the GPU has neither an integer divider nor a 64-bit multiplier).
struct TileArg { int R, np, lv; uint32_t inv_np, inv_kpr, inv_wpr; };
// multi-GPU: instead of appending to the local partition streams, records go to per-owner send regions (owner =
// partition mod n_owners), `cap` records each, with the partition id alongside; cursor[o] counts what owner o gets.
struct RouteArg { uint64_t* recs; uint32_t* pids; unsigned long long* cursor; uint64_t cap; int n_owners; };
__device__ __forceinline__ uint32_t fastdiv(uint32_t i, uint32_t inv) { return __umulhi(i, inv); }

template <int NW, bool ROUTE>
__global__ __launch_bounds__(BLOCK) void skm_scatter_tiled_kernel(ReadsArg a, E2Dev e, DevCounters* ctr, TileArg ta, RouteArg ro) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int R = ta.R, np = ta.np, lv = ta.lv;
    const int wpr = (int)a.wpr, kpr = (int)a.kpr, ws = wpr + 1, len = (int)a.uniform_len;
    const int nch = (kpr + 62) / 63;                               // 63-k-mer chunks per read (one wave each)
    uint64_t* words = (uint64_t*)smem_raw;                       // R * ws
    unsigned long long* masks = (unsigned long long*)(words + (size_t)R * ws);   // R * nch: run-start bits
    uint32_t* v0 = (uint32_t*)(masks + (size_t)R * nch);          // R * np: m-mer values
    uint32_t* pids = v0 + (size_t)R * np;                         // R * kpr
    uint32_t* items = pids + (size_t)R * kpr;                     // R * kpr  (r << 24 | j << 12 | n)
    __shared__ unsigned int n_items;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const int nr = (int)min((uint64_t)R, a.n_reads - r0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) n_items = 0;
    for (int i = threadIdx.x; i < nr * wpr; i += BLOCK) {
        const int r = (int)fastdiv(i, ta.inv_wpr), k = i - r * wpr;
        words[r * ws + k] = a.packed[r0 * wpr + i];
    }
    for (int r = threadIdx.x; r < nr; r += BLOCK) words[r * ws + wpr] = 0;
    __syncthreads();
    const int m = e.g.m, w = e.g.w;
    for (int i = threadIdx.x; i < nr * np; i += BLOCK) {
        const int r = (int)fastdiv(i, ta.inv_np), p = i - r * np;
        v0[i] = mmer_value(words + r * ws, p, m);
    }
    __syncthreads();
    const int span = 1 << lv;                                      // doubling reaches min over [p, p + span), span <= w
    const int nmax = e.g.nmax;
    // B + C in registers, one wave per chunk of 63 k-mers of a read: lane l holds k-mer j = 63 c - 1 + l (lane 0 only
    // supplies the predecessor of lane 1) and the m-mer values at j, j + 64, j + 128.  The sliding minimum over the
    // w m-mers of a k-mer is lv doubling steps of "min with the value d lanes up" (shuffles, no LDS pass, no barrier),
    // then one more shifted min for the remainder of the window; the ballot of the run starts IS the bit mask.
    for (int wi = wave; wi < nr * nch; wi += BLOCK / 64) {
        const int r = wi / nch, c = wi - r * nch;
        const int j = 63 * c - 1 + lane;
        const uint32_t* sv = v0 + r * np;
        uint32_t x0 = (j >= 0 && j < np) ? sv[j] : 0xFFFFFFFFu;
        uint32_t x1 = (j + 64 < np) ? sv[j + 64] : 0xFFFFFFFFu;
        uint32_t mv;
        if (w <= 65) {
            // a window of at most 65 m-mers never reaches past j + 127: two registers a lane
            for (int k = 0; k < lv; k++) {
                const int d = 1 << k;
                const int srcl = (lane + d) & 63;
                const bool wrap = lane + d >= 64;
                const uint32_t a0 = __shfl(x0, srcl, 64), a1 = __shfl(x1, srcl, 64);
                x0 = min(x0, wrap ? a1 : a0);
                x1 = min(x1, wrap ? 0xFFFFFFFFu : a1);
            }
            mv = x0;                                                // min over [j, j + span)
            const int sft = w - span;                               // the rest of the window: [j + w - span, j + w)
            if (sft > 0) {
                const int srcl = (lane + sft) & 63;
                const uint32_t a0 = __shfl(x0, srcl, 64), a1 = __shfl(x1, srcl, 64);
                mv = min(x0, lane + sft >= 64 ? a1 : a0);
            }
        } else {
            uint32_t x2 = (j + 128 < np) ? sv[j + 128] : 0xFFFFFFFFu;
            for (int k = 0; k < lv && k < 6; k++) {
                const int d = 1 << k;
                const int srcl = (lane + d) & 63;
                const bool wrap = lane + d >= 64;
                const uint32_t a0 = __shfl(x0, srcl, 64), a1 = __shfl(x1, srcl, 64), a2 = __shfl(x2, srcl, 64);
                x0 = min(x0, wrap ? a1 : a0);
                x1 = min(x1, wrap ? a2 : a1);
                x2 = min(x2, wrap ? 0xFFFFFFFFu : a2);
            }
            if (lv >= 7) { x0 = min(x0, x1); x1 = min(x1, x2); }  // a step of 64 is the next register (w >= 128 only)
            mv = x0;
            const int sft = w - span;
            if (sft > 0) {
                const int d = sft & 63;
                const int srcl = (lane + d) & 63;
                const bool wrap = lane + d >= 64;
                const uint32_t a0 = __shfl(x0, srcl, 64), a1 = __shfl(x1, srcl, 64), a2 = __shfl(x2, srcl, 64);
                mv = sft >= 64 ? min(x0, wrap ? a2 : a1) : min(x0, wrap ? a1 : a0);
            }
        }
        const bool valid = lane > 0 ? (j < kpr) : false;
        const uint32_t pid = skm_partition(mv, e.g.log2_parts);
        const uint32_t prev = __shfl_up(pid, 1, 64);
        if (valid) pids[r * kpr + j] = pid;
        const bool start = valid && (j == 0 || pid != prev || j % nmax == 0);
        const unsigned long long mk = __ballot(start);
        if (lane == 0) masks[wi] = mk;
    }
    __syncthreads();
    // D: every run start finds the next start in the bit masks (no per-lane walk) and queues one item
    for (int wi = wave; wi < nr * nch; wi += BLOCK / 64) {
        const int r = wi / nch, c = wi - r * nch;
        const int j = 63 * c - 1 + lane;
        const unsigned long long mk = masks[wi];
        const bool start = (mk >> lane) & 1ULL;
        uint32_t item = 0;
        if (start) {
            int next = kpr;
            const unsigned long long above = lane < 63 ? (mk >> (lane + 1)) : 0ULL;
            if (above) next = j + __ffsll((long long)above);
            else {
                for (int cc = c + 1; cc < nch; cc++) {
                    const unsigned long long mm = masks[r * nch + cc];
                    if (mm) { next = 63 * cc - 1 + __ffsll((long long)mm) - 1; break; }
                }
            }
            item = ((uint32_t)r << 24) | ((uint32_t)j << 12) | (uint32_t)(next - j);
        }
        unsigned int base = 0;
        if (lane == 0 && mk) base = atomicAdd(&n_items, (unsigned int)__popcll(mk));
        base = __shfl(base, 0, 64);
        if (start) items[base + __popcll(mk & ((1ULL << lane) - 1))] = item;
    }
    __syncthreads();
    const int total = (int)n_items;
    __shared__ unsigned int ocnt[256];
    __shared__ unsigned long long obase[256];
    uint32_t* ranks = v0;                                          // the m-mer values are dead by now
    if (ROUTE) {
        ocnt[threadIdx.x] = 0;
        __syncthreads();
        for (int it = threadIdx.x; it < total; it += BLOCK) {
            const uint32_t pk = items[it];
            const uint32_t pid = pids[(int)(pk >> 24) * kpr + (int)((pk >> 12) & 0xFFF)];
            ranks[it] = atomicAdd(&ocnt[pid % (uint32_t)ro.n_owners], 1u);
        }
        __syncthreads();
        if ((int)threadIdx.x < ro.n_owners && ocnt[threadIdx.x])
            obase[threadIdx.x] = atomicAdd(&ro.cursor[threadIdx.x], (unsigned long long)ocnt[threadIdx.x]);
        __syncthreads();
    }
    for (int it = threadIdx.x; it < total; it += BLOCK) {
        const uint32_t pk = items[it];
        const int r = (int)(pk >> 24), j0 = (int)((pk >> 12) & 0xFFF), n = (int)(pk & 0xFFF);
        const uint32_t pid = pids[r * kpr + j0];
        uint64_t* out;
        if (ROUTE) {
            const uint32_t o = pid % (uint32_t)ro.n_owners;
            const unsigned long long at = obase[o] + ranks[it];
            if (at >= ro.cap) { atomicOr(&ctr->e2_flags, F_POOL); continue; }
            ro.pids[(uint64_t)o * ro.cap + at] = pid;
            out = ro.recs + ((uint64_t)o * ro.cap + at) * RW;
        } else {
            const uint32_t q = atomicAdd(&e.cursor[pid], 1u);
            out = record_slot(e, pid, q, ctr, RW);
        }
        if (!out) continue;
        uint64_t rec[RW];
        skm_make_record<PW>(words + r * ws, len, j0, n, a.ord_base + (r0 + (uint64_t)r) * (uint64_t)kpr, e.g, rec);
        ulonglong2* o2 = (ulonglong2*)out;
#pragma unroll
        for (int k = 0; k < RW / 2; k++) o2[k] = make_ulonglong2(rec[2 * k], rec[2 * k + 1]);
    }
}

// multi-GPU, receiving side: append routed records to the local partition streams
template <int NW>
__global__ __launch_bounds__(BLOCK) void skm_ingest_kernel(const uint64_t* recs, const uint32_t* rpids, uint64_t n, E2Dev e, DevCounters* ctr) {
    constexpr int RW = E2Cfg<NW>::PW + 1;
    const uint32_t parts = 1u << e.g.log2_parts;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t pid = rpids[i];
        if (pid >= parts) { atomicOr(&ctr->e2_flags, F_CHUNKS); continue; }
        const uint32_t q = atomicAdd(&e.cursor[pid], 1u);
        uint64_t* out = record_slot(e, pid, q, ctr, RW);
        if (!out) continue;
        const ulonglong2* src = (const ulonglong2*)(recs + i * RW);
        ulonglong2* o2 = (ulonglong2*)out;
#pragma unroll
        for (int k = 0; k < RW / 2; k++) o2[k] = src[k];
    }
}

// ---- the LDS set ---------------------------------------------------------------------------------------------
// Struct of arrays, so lanes that hit different slots hit different banks: key[KW][SLOTS] (63-bit words), ord[SLOTS],
// cnt[9][SLOTS] (u32: L[4], R[4], puts).  Keys and ord start as ~0, counters as 0.  A slot is claimed key word by key
// word: an empty word is taken with CAS(~0 -> mine), a word holding something else means another key owns the slot.
// Counting is plain atomic adds, saturated when the node is emitted: a sum of +1's clipped at the end equals the
// reference's saturating increments (newhash.c:74-106) and, unlike a CAS on packed counters, needs no retry when many
// lanes hit one hot k-mer.  `single` = exactly one put (newhash.c:127,511).
template <int NW, int SLOTS>
struct LdsSet {
    static constexpr int KW = E2Cfg<NW>::KW;
    unsigned long long key[KW][SLOTS];
    unsigned long long ord[SLOTS];
    unsigned int cnt[9][SLOTS];
};

// Returns false when the set is too full (a probe sequence longer than MAXPROBE): the caller aborts the attempt and
// splits the key range.  No shared key counter on this path -- a same-address LDS atomic per new key serialises the
// whole workgroup; the keys are counted once, at emit time.
constexpr int K2_MAXPROBE = 48;
template <int NW, int SLOTS>
__device__ __forceinline__ bool lds_put(LdsSet<NW, SLOTS>& t, const Key63<NW>& key, uint64_t hash, int left, int right, uint64_t ord) {
    constexpr int KW = E2Cfg<NW>::KW;
    uint32_t h = (uint32_t)hash & (SLOTS - 1);
    for (int probes = 0; probes < K2_MAXPROBE; probes++) {
        // all key words of the slot are fetched together (one LDS round trip for the usual case, a hit); only a word
        // that is still empty goes through the claiming CAS
        unsigned long long seen[KW];
#pragma unroll
        for (int i = 0; i < KW; i++) seen[i] = t.key[i][h];
        bool mine = true;
#pragma unroll
        for (int i = 0; i < KW; i++) {
            if (!mine) break;
            unsigned long long cur = seen[i];
            if (cur == L_EMPTY) {
                const unsigned long long old = atomicCAS(&t.key[i][h], L_EMPTY, (unsigned long long)key.w[i]);
                cur = old == L_EMPTY ? (unsigned long long)key.w[i] : old;
            }
            mine = cur == key.w[i];
        }
        if (mine) {
            if (left < 4) atomicAdd(&t.cnt[left][h], 1u);
            if (right < 4) atomicAdd(&t.cnt[4 + right][h], 1u);
            atomicAdd(&t.cnt[8][h], 1u);
            atomicMin(&t.ord[h], (unsigned long long)ord);
            return true;
        }
        h = (h + 1) & (SLOTS - 1);
    }
    return false;
}

__device__ __forceinline__ const uint64_t* record_ptr(const E2Dev& e, uint32_t pid, uint32_t i, int rw) {
    const uint32_t c = e.chunk_tbl[(uint64_t)pid * e.maxc + i / e.rpc];
    if (c == 0 || c == 0xFFFFFFFFu) return nullptr;  // pool ran dry in K1 (flagged there; the run fails in e2_count)
    return e.pool + ((uint64_t)(c - 1) * e.rpc + i % e.rpc) * (uint64_t)rw;
}

// K2.  One workgroup per partition (persistent grid).  The partition's records are taken WIN at a time: staged into
// LDS with coalesced 16-byte copies (so the per-occurrence work never waits on global memory), flattened through a
// prefix sum of their k-mer counts so every lane gets an equal contiguous share of occurrences, expanded and inserted
// into the LDS set.  If the set overflows, the attempt is dropped and the key range is split on a hash bit.
template <int NW, int SLOTS, int THREADS, int WIN>
__global__ __launch_bounds__(THREADS) void skm_count_kernel(E2Dev e, int D, SetParams sp, DevCounters* ctr, int dbg) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1, KW = E2Cfg<NW>::KW, NWAVE = THREADS / 64, PIECES = RW / 2;
    static_assert(WIN <= THREADS, "one record per lane in the flatten step");
    __shared__ LdsSet<NW, SLOTS> set;
    __shared__ __align__(16) uint64_t recs[WIN * RW + 8];                 // + readable padding for the window loads
    __shared__ unsigned int noff[WIN + 1];                                // exclusive prefix sum of the records' k-mer counts
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned int hist[256];
    constexpr int STRIPES = (SLOTS + THREADS - 1) / THREADS;
    __shared__ unsigned int aborted, sp_top, s_mask[40], s_val[40], cur_mask, cur_val, wave_cnt[STRIPES][NWAVE];
    __shared__ unsigned int chunk_ids[256];                                // this partition's chunk list (maxc <= 256)
    __shared__ unsigned long long out_base;
    for (int i = threadIdx.x; i < 256; i += THREADS) { crc_tab[i] = crc32_table_entry(i); hist[i] = 0; }
    if (threadIdx.x < 8) recs[WIN * RW + threadIdx.x] = 0;
    const Kmer<NW> filter = kmer_filter<NW>(e.g.K);
    const int K = e.g.K;
    const uint32_t parts = 1u << e.g.log2_parts;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long my_records = 0;
    unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define K2_TICK(i) do { if (dbg & 2) { const unsigned long long tn_ = clock64(); tp[i] += tn_ - tlast; tlast = tn_; } } while (0)
    // the next partition's record count and chunk list are fetched while the current one is processed
    uint32_t pf_nrec = 0, pf_cid = 0;
    if (blockIdx.x < parts) {
        pf_nrec = e.cursor[blockIdx.x];
        if (threadIdx.x < e.maxc) pf_cid = e.chunk_tbl[(uint64_t)blockIdx.x * e.maxc + threadIdx.x];
    }
    for (uint32_t pid = blockIdx.x; pid < parts; pid += gridDim.x) {
        const uint32_t nrec = pf_nrec, my_cid = pf_cid;
        {
            const uint32_t nxt = pid + gridDim.x;
            if (nxt < parts) {
                pf_nrec = e.cursor[nxt];
                if (threadIdx.x < e.maxc) pf_cid = e.chunk_tbl[(uint64_t)nxt * e.maxc + threadIdx.x];
            }
        }
        const uint32_t usable = min(nrec, e.maxc * e.rpc);               // an overfull partition was flagged by K1
        my_records += usable;
        if (usable == 0) continue;
        __syncthreads();
        if (threadIdx.x < e.maxc) chunk_ids[threadIdx.x] = my_cid;
        if (threadIdx.x == 0) { sp_top = 1; s_mask[0] = 0; s_val[0] = 0; }
        __syncthreads();
        K2_TICK(0);
        while (sp_top > 0) {
            __syncthreads();
            if (threadIdx.x == 0) { sp_top--; cur_mask = s_mask[sp_top]; cur_val = s_val[sp_top]; aborted = 0; }
            for (int i = threadIdx.x; i < SLOTS; i += THREADS) {
#pragma unroll
                for (int q = 0; q < KW; q++) set.key[q][i] = L_EMPTY;
                set.ord[i] = L_EMPTY;
#pragma unroll
                for (int q = 0; q < 9; q++) set.cnt[q][i] = 0;
            }
            __syncthreads();
            K2_TICK(1);
            const uint32_t mask = cur_mask, val = cur_val;
            volatile unsigned int* abort_flag = &aborted;
            for (uint32_t w0 = 0; w0 < usable; w0 += WIN) {
                const uint32_t wn = min((uint32_t)WIN, usable - w0);
                // stage the window's records: 16 bytes per lane and step, consecutive lanes -> consecutive pieces
                for (uint32_t pc = threadIdx.x; pc < wn * PIECES; pc += THREADS) {
                    const uint32_t ri = pc / PIECES, part = pc - ri * PIECES;
                    const uint32_t gi = w0 + ri, cid = chunk_ids[gi / e.rpc];
                    ulonglong2 v = make_ulonglong2(0, 0);
                    if (cid != 0 && cid != 0xFFFFFFFFu)
                        v = ((const ulonglong2*)(e.pool + ((uint64_t)(cid - 1) * e.rpc + gi % e.rpc) * (uint64_t)RW))[part];
                    ((ulonglong2*)recs)[pc] = v;
                }
                __syncthreads();
                K2_TICK(2);
                // flatten: occurrence idx -> (record, t)
                {
                    const unsigned int n = threadIdx.x < wn ? (unsigned int)skm_n(recs[threadIdx.x * RW]) : 0u;
                    unsigned int incl = n;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const unsigned int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
                    if (lane == 63) wave_cnt[0][wave] = incl;
                    __syncthreads();
                    unsigned int base = 0;
#pragma unroll
                    for (int wv = 0; wv < NWAVE; wv++) if (wv < wave) base += wave_cnt[0][wv];
                    if (threadIdx.x < wn) noff[threadIdx.x] = base + incl - n;
                    if (threadIdx.x == THREADS - 1) noff[wn] = base + incl;       // lanes past wn contributed 0
                }
                __syncthreads();
                K2_TICK(3);
                if (!*abort_flag) {
                    const uint32_t total_occ = noff[wn];
                    const uint32_t share = (total_occ + THREADS - 1) / THREADS;
                    const uint32_t idx0 = min(total_occ, threadIdx.x * share), idx1 = min(total_occ, idx0 + share);
                    if (idx0 < idx1) {
                        uint32_t lo = 0, hi = wn - 1;                             // first record of this lane's share
                        while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (noff[mid] <= idx0) lo = mid; else hi = mid - 1; }
                        uint32_t r = lo, next_off = noff[r + 1], roff = 0;
                        const uint64_t* rec = recs;
                        uint64_t hdr = 0;
                        int hl = 0, nb = 0;
                        bool fresh = true;
                        for (uint32_t idx = idx0; idx < idx1; idx++) {
                            while (idx >= next_off) { r++; next_off = noff[r + 1]; fresh = true; }
                            if (fresh) {
                                rec = recs + r * RW;
                                hdr = rec[0];
                                hl = skm_has_left(hdr); nb = skm_record_bases(hdr, K); roff = noff[r];
                                fresh = false;
                                if (*abort_flag) break;
                            }
                            Occurrence occ;
                            const Kmer<NW> key = canonical_occurrence<NW>(rec + 1, hl + (int)(idx - roff), nb, K, filter, occ);
                            const uint64_t hh = kmer_mix<NW>(key);
                            if (((uint32_t)(hh >> 32) & mask) != val) continue;
                            if (dbg & 1) { if (hh == 0x1234) aborted = 1; continue; }      // measurement aid: extraction only
                            if (!lds_put<NW, SLOTS>(set, key63_from_kmer<NW>(key), hh, occ.left, occ.right, skm_ord(hdr) + (uint64_t)(idx - roff))) {
                                aborted = 1;
                                break;
                            }
                        }
                    }
                }
                K2_TICK(4);
                __syncthreads();                                                  // recs / noff are rewritten by the next window
                K2_TICK(5);
            }
            if (aborted) {
                if (dbg & 2) tp[9]++;
                // too many distinct keys for the LDS set: split this key range on the next hash bit and redo both halves
                if (threadIdx.x == 0) {
                    const uint32_t bit = mask + 1;                                    // masks are 2^k - 1
                    if (bit >= (1u << 24) || sp_top + 2 > 40) atomicOr(&ctr->e2_flags, F_SPLIT);
                    else {
                        s_mask[sp_top] = mask | bit; s_val[sp_top] = val; sp_top++;
                        s_mask[sp_top] = mask | bit; s_val[sp_top] = val | bit; sp_top++;
                    }
                }
                __syncthreads();
                continue;
            }
            // ---- emit: finalize every stored node and append it to the export array.  One global atomic per attempt,
            // issued as soon as the live slots are counted so that its latency hides behind the node finalisation.
            bool live[STRIPES];
            unsigned long long bal[STRIPES];
#pragma unroll
            for (int st = 0; st < STRIPES; st++) {
                const int si = st * THREADS + threadIdx.x;
                live[st] = si < SLOTS && set.cnt[8][si] != 0;          // a put is only counted once every key word is claimed
                bal[st] = __ballot(live[st]);
                if (lane == 0) wave_cnt[st][wave] = (unsigned int)__popcll(bal[st]);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned int tot = 0;
                for (int st = 0; st < STRIPES; st++) for (int wv = 0; wv < NWAVE; wv++) tot += wave_cnt[st][wv];
                out_base = atomicAdd(&ctr->n_export, (unsigned long long)tot);
            }
            uint64_t rec_out[STRIPES][NW + 2];
            unsigned int off[STRIPES], cov_bin[STRIPES];
            unsigned int running = 0;
#pragma unroll
            for (int st = 0; st < STRIPES; st++) {
                unsigned int before = 0, total = 0;
#pragma unroll
                for (int wv = 0; wv < NWAVE; wv++) {
                    const unsigned int cw = wave_cnt[st][wv];
                    if (wv < wave) before += cw;
                    total += cw;
                }
                off[st] = running + before + (unsigned int)__popcll(bal[st] & ((1ULL << lane) - 1));
                running += total;
                cov_bin[st] = 0;                                          // 0 = not live (a live node has cov >= 1)
                if (live[st]) {
                    const int si = st * THREADS + threadIdx.x;
                    const unsigned int puts = set.cnt[8][si];
                    Key63<NW> k63;
#pragma unroll
                    for (int w = 0; w < KW; w++) k63.w[w] = set.key[w][si];
                    const Kmer<NW> key = kmer_from_key63<NW>(k63);
                    uint32_t A = min(puts, 255u) << 24, B = puts == 1 ? B_SINGLE : 0u;
                    int nin = 0, nout = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {                               // saturate, then thread_delow + thread_mark
                        uint32_t l = min(set.cnt[c][si], 63u), r = min(set.cnt[4 + c][si], 63u);
                        if (D > 0 && l <= (uint32_t)D) l = 0;
                        if (D > 0 && r <= (uint32_t)D) r = 0;
                        A |= l << (6 * c); B |= r << (6 * c);
                        nin += l > 0; nout += r > 0;
                    }
                    if (D > 0 && nin == 0 && nout == 0) B |= B_DELETED;
                    if (nin == 1 && nout == 1) B |= B_LINEAR;
                    cov_bin[st] = A >> 24;
                    const uint32_t sid = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
#pragma unroll
                    for (int w = 0; w < NW; w++) rec_out[st][w] = key.w[w];
                    rec_out[st][NW] = (uint64_t)A | ((uint64_t)B << 32);
                    rec_out[st][NW + 1] = ((uint64_t)sid << PG_ORD_BITS) | (set.ord[si] & PG_ORD_MASK);
                }
                // coverage histogram: most nodes of a partition share one or two coverage values (1 for error k-mers),
                // so count those per wave instead of hammering one LDS word
                const unsigned long long ones = __ballot(cov_bin[st] == 1);
                if (lane == 0 && ones) atomicAdd(&hist[1], (unsigned int)__popcll(ones));
                if (cov_bin[st] > 1) atomicAdd(&hist[cov_bin[st]], 1u);
            }
            K2_TICK(6);
            __syncthreads();                                              // out_base is in; the set may be cleared after this
            K2_TICK(7);
            const unsigned long long ob = out_base;
#pragma unroll
            for (int st = 0; st < STRIPES; st++) {
                if (live[st]) {
                    const uint64_t pos = ob + off[st];
                    if (pos < e.out_capacity) {
                        uint64_t* o = e.out + pos * (NW + 2);
#pragma unroll
                        for (int w = 0; w < NW + 2; w++) o[w] = rec_out[st][w];
                    } else atomicOr(&ctr->e2_flags, F_OUT);
                }
            }
            K2_TICK(8);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += THREADS) if (hist[i]) atomicAdd(&ctr->hist[i], (unsigned long long)hist[i]);
    if (threadIdx.x == 0 && my_records) atomicAdd(&ctr->n_records, my_records);
    if ((dbg & 2) && threadIdx.x == 0) for (int i = 0; i < 10; i++) atomicAdd(&ctr->phase[i], tp[i]);
#undef K2_TICK
}

// per reference set: 1 + ordinal of the last k-mer occurrence routed to it (see host_graph.cpp, before_put)
template <int NW>
__global__ __launch_bounds__(BLOCK) void skm_lastput_kernel(E2Dev e, SetParams sp, DevCounters* ctr) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1;
    __shared__ uint32_t crc_tab[256];
    __shared__ unsigned long long set_last[256];
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    set_last[threadIdx.x] = 0;
    __syncthreads();
    const Kmer<NW> filter = kmer_filter<NW>(e.g.K);
    const int K = e.g.K;
    const uint32_t parts = 1u << e.g.log2_parts;
    for (uint32_t pid = blockIdx.x; pid < parts; pid += gridDim.x) {
        const uint32_t usable = min(e.cursor[pid], e.maxc * e.rpc);
        for (uint32_t i = threadIdx.x; i < usable; i += BLOCK) {
            const uint64_t* rec = record_ptr(e, pid, i, RW);
            if (!rec) continue;
            skm_expand_record<NW>(rec, K, filter, [&](const Kmer<NW>& key, int, int, uint64_t ord) {
                const uint32_t sid = set_of_crc(kmer_crc32<NW>(key, crc_tab), sp.P, sp.bias);
                atomicMax(&set_last[sid], (unsigned long long)(ord + 1));
            });
        }
    }
    __syncthreads();
    if (threadIdx.x < sp.P && set_last[threadIdx.x]) atomicMax(&ctr->set_last[threadIdx.x], set_last[threadIdx.x]);
}

// =========================================================================================================
// host side of engine 2
// =========================================================================================================
#define E2_TRY(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pg_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            return (e_ == hipErrorOutOfMemory) ? PG_ENOMEM : PG_ENODEV;                        \
        }                                                                                      \
    } while (0)

static E2Dev dev_view(const pg_ctx* c) {
    const E2& s = c->e2;
    return E2Dev{s.g, s.rpc, s.maxc, s.pool_chunks, s.cursor, s.chunk_tbl, s.pool, s.out, s.out_capacity};
}

int e2_create(pg_ctx* c) {
    E2& s = c->e2;
    // partitions: expected distinct / ~1000 so a partition usually fits the LDS set in one attempt
    s.log2_parts = std::max(8, std::min(23, c->log2_slots - 11));
    if (c->hint_log2_parts >= 0) s.log2_parts = std::max(8, std::min(23, c->hint_log2_parts));
    if (const char* v = getenv("PG_LOG2_PARTS")) s.log2_parts = std::max(4, std::min(24, atoi(v)));
    s.g = skm_geometry(c->K, s.log2_parts, c->NW);
    s.rpc = 128;
    const uint64_t parts = (uint64_t)1 << s.log2_parts;
    const uint64_t rec_bytes = (uint64_t)s.g.rw * 8, chunk_bytes = rec_bytes * s.rpc;
    size_t free_b = 0, total_b = 0;
    E2_TRY(hipMemGetInfo(&free_b, &total_b));
    // export array: what a set of 2^log2_slots slots would hold at 70 % load
    s.out_capacity = (uint64_t)(0.7 * (double)((uint64_t)1 << c->log2_slots));
    const uint64_t out_bytes = s.out_capacity * (uint64_t)(c->NW + 2) * 8;
    // record pool: every partition keeps one partly filled chunk, plus the records themselves (about one record per
    // 20 k-mers); default = as much as a set of 2^log2_slots 64-byte slots, capped by what is free
    uint64_t pool_bytes = ((uint64_t)1 << c->log2_slots) * 64 + parts * chunk_bytes * 2;
    if (c->hint_kmers)                       // known input size: 2 / (w + 1) records a k-mer, half as much again, it grows
        pool_bytes = (uint64_t)((double)c->hint_kmers * 3.0 / (double)(s.g.w + 1)) * rec_bytes + parts * chunk_bytes * 2 + ((uint64_t)64 << 20);
    if (const char* v = getenv("PG_POOL_MB")) pool_bytes = (uint64_t)atoll(v) << 20;
    const uint64_t budget = (uint64_t)(free_b * 0.85);
    if (out_bytes + parts * 8 > budget) { pg_set_error("partition engine: export array does not fit in device memory"); return PG_ENOMEM; }
    pool_bytes = std::min<uint64_t>(pool_bytes, (budget - out_bytes) * 9 / 10);
    s.pool_chunks = pool_bytes / chunk_bytes;
    if (s.pool_chunks < parts + 16) { pg_set_error("partition engine: record pool too small for the partition count"); return PG_ENOMEM; }
    // chunk table: up to 2^28 entries in total, at least enough for an even spread x8
    const uint64_t even = (s.pool_chunks + parts - 1) / parts;
    s.maxc = (uint32_t)std::max<uint64_t>(8, std::min<uint64_t>(std::min<uint64_t>(256, ((uint64_t)1 << 28) / parts), even * 16));
    E2_TRY(hipMalloc(&s.cursor, parts * sizeof(uint32_t)));
    E2_TRY(hipMalloc(&s.chunk_tbl, parts * s.maxc * sizeof(uint32_t)));
    E2_TRY(hipMalloc(&s.pool, s.pool_chunks * chunk_bytes + 64));
    E2_TRY(hipMalloc(&s.out, std::max<uint64_t>(out_bytes, 64)));
    E2_TRY(hipMemset(s.cursor, 0, parts * sizeof(uint32_t)));
    E2_TRY(hipMemset(s.chunk_tbl, 0, parts * s.maxc * sizeof(uint32_t)));
    s.counted = false;
    s.est_chunks = 0;
    return PG_OK;
}

void e2_destroy(pg_ctx* c) {
    E2& s = c->e2;
    if (s.cursor) (void)hipFree(s.cursor);
    if (s.chunk_tbl) (void)hipFree(s.chunk_tbl);
    if (s.pool) (void)hipFree(s.pool);
    if (s.out) (void)hipFree(s.out);
    s = E2();
}

int e2_reset(pg_ctx* c, hipStream_t st) {
    E2& s = c->e2;
    const uint64_t parts = (uint64_t)1 << s.log2_parts;
    E2_TRY(hipMemsetAsync(s.cursor, 0, parts * sizeof(uint32_t), st));
    E2_TRY(hipMemsetAsync(s.chunk_tbl, 0, parts * s.maxc * sizeof(uint32_t), st));
    s.counted = false;
    s.est_chunks = 0;
    return PG_OK;
}

// Before a batch: make sure the pool can take it.  A read of k k-mers makes ~ 2k/(w+1) + 1 records on average;
// room is kept for 4x that plus one open chunk per partition.  Chunks are referenced by index, so growing the pool
// is a plain copy into a larger allocation.
static int e2_ensure_pool(pg_ctx* c, uint64_t n_reads, uint64_t n_kmers, hipStream_t st) {
    E2& s = c->e2;
    if (!c->autogrow) return PG_OK;
    const uint64_t parts = (uint64_t)1 << s.log2_parts;
    const uint64_t est_records = 4 * (2 * n_kmers / (uint64_t)(s.g.w + 1) + n_reads) + 64;
    s.est_chunks += est_records / s.rpc + 1;
    const uint64_t need = s.est_chunks + parts + 16;
    if (need <= s.pool_chunks) return PG_OK;
    // the estimate is loose: look at what was really handed out
    E2_TRY(hipStreamSynchronize(st));
    unsigned long long used = 0;
    E2_TRY(hipMemcpy(&used, &c->ctr->pool_next, sizeof used, hipMemcpyDeviceToHost));
    s.est_chunks = used + est_records / s.rpc + 1;
    const uint64_t need2 = s.est_chunks + parts + 16;
    if (need2 <= s.pool_chunks) return PG_OK;
    const uint64_t chunk_bytes = (uint64_t)s.g.rw * 8 * s.rpc;
    const uint64_t fresh_chunks = std::max(need2 * 2, s.pool_chunks * 2);
    uint64_t* fresh = nullptr;
    E2_TRY(hipMalloc(&fresh, fresh_chunks * chunk_bytes + 64));
    E2_TRY(hipMemcpy(fresh, s.pool, std::min<uint64_t>(used, s.pool_chunks) * chunk_bytes, hipMemcpyDeviceToDevice));
    E2_TRY(hipFree(s.pool));
    s.pool = fresh;
    s.pool_chunks = fresh_chunks;
    // a longer chunk list per partition too, if the table allows (rebuild with the wider stride)
    const uint64_t even = (fresh_chunks + parts - 1) / parts;
    const uint32_t want = (uint32_t)std::max<uint64_t>(s.maxc, std::min<uint64_t>(std::min<uint64_t>(256, ((uint64_t)1 << 28) / parts), even * 16));
    if (want > s.maxc) {
        uint32_t* tbl = nullptr;
        E2_TRY(hipMalloc(&tbl, parts * want * sizeof(uint32_t)));
        E2_TRY(hipMemset(tbl, 0, parts * want * sizeof(uint32_t)));
        E2_TRY(hipMemcpy2D(tbl, want * sizeof(uint32_t), s.chunk_tbl, s.maxc * sizeof(uint32_t), s.maxc * sizeof(uint32_t), parts, hipMemcpyDeviceToDevice));
        E2_TRY(hipFree(s.chunk_tbl));
        s.chunk_tbl = tbl;
        s.maxc = want;
    }
    return PG_OK;
}

// launch the tiled K1; returns PG_OK / an error, or 1 when the reads are too long for a tile (caller falls back)
static int launch_tiled(pg_ctx* c, const ReadsArg& a, const RouteArg* route, hipStream_t st) {
    const int np = (int)a.uniform_len - c->e2.g.m + 1;
    const size_t per_read = (size_t)(a.wpr + 1) * 8 + (size_t)((a.kpr + 62) / 63) * 8 + (size_t)np * 4 + (size_t)a.kpr * 8;
    int R = (int)std::min<size_t>(8, (56 * 1024) / per_read);          // small tiles: many resident workgroups hide the LDS / atomic latency
    if (const char* v = getenv("PG_K1_R")) R = std::max(1, std::min(atoi(v), (int)((56 * 1024) / per_read)));
    if (R < 1) return 1;
    int lv = 0;
    while ((2 << lv) <= c->e2.g.w) lv++;
    const uint64_t grid = (a.n_reads + R - 1) / R;
    if (grid > 0x7FFFFFFFULL) { pg_set_error("batch too large for one launch"); return PG_EINVAL; }
    const size_t smem = per_read * R;
    auto inv = [](uint32_t d) { return (uint32_t)(((1ULL << 32) + d - 1) / d); };
    TileArg ta{R, np, lv, inv((uint32_t)np), inv(a.kpr), inv(a.wpr)};
    RouteArg ro{nullptr, nullptr, nullptr, 0, 1};
    if (route) ro = *route;
    if (c->NW == 2) {
        if (route) hipLaunchKernelGGL((skm_scatter_tiled_kernel<2, true>), dim3((unsigned)grid), dim3(BLOCK), smem, st, a, dev_view(c), c->ctr, ta, ro);
        else hipLaunchKernelGGL((skm_scatter_tiled_kernel<2, false>), dim3((unsigned)grid), dim3(BLOCK), smem, st, a, dev_view(c), c->ctr, ta, ro);
    } else {
        if (route) hipLaunchKernelGGL((skm_scatter_tiled_kernel<4, true>), dim3((unsigned)grid), dim3(BLOCK), smem, st, a, dev_view(c), c->ctr, ta, ro);
        else hipLaunchKernelGGL((skm_scatter_tiled_kernel<4, false>), dim3((unsigned)grid), dim3(BLOCK), smem, st, a, dev_view(c), c->ctr, ta, ro);
    }
    E2_TRY(hipGetLastError());
    c->e2.counted = false;
    return PG_OK;
}

// multi-GPU step 1: cut a uniform batch into records grouped by owner (partition mod n_owners)
int e2_route(pg_ctx* c, const uint64_t* d_packed, uint64_t n_reads, uint32_t uniform_len, uint64_t ord_base, int n_owners,
             uint64_t* d_recs, uint32_t* d_pids, uint64_t cap, uint64_t* d_counts, hipStream_t st) {
    if (!uniform_len || uniform_len >= 4096) { pg_set_error("pg_skm_route needs a uniform-length batch"); return PG_EINVAL; }
    if ((ord_base >> (64 - SKM_ORD_SHIFT)) != 0) { pg_set_error("ordinal exceeds the 46 bits of a super-k-mer header"); return PG_EINVAL; }
    ReadsArg a;
    a.packed = d_packed; a.word_off = nullptr; a.kmer_base = nullptr; a.n_reads = n_reads; a.uniform_len = uniform_len;
    a.kpr = uniform_len - c->K + 1; a.wpr = (uniform_len + 31) / 32; a.ord_base = ord_base;
    E2_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * n_owners, st));
    RouteArg ro{d_recs, d_pids, (unsigned long long*)d_counts, cap, n_owners};
    int rc = launch_tiled(c, a, &ro, st);
    if (rc == 1) { pg_set_error("reads too long for the tiled kernel"); return PG_EINVAL; }
    return rc;
}

// multi-GPU step 2: take records another rank cut for this rank's partitions
int e2_ingest(pg_ctx* c, const uint64_t* d_recs, const uint32_t* d_pids, uint64_t n, hipStream_t st) {
    if (n == 0) return PG_OK;
    E2& s = c->e2;
    if (c->autogrow) {                       // room for n more records (+ one open chunk per partition is already counted)
        s.est_chunks += n / s.rpc + 1;
        if (s.est_chunks + ((uint64_t)1 << s.log2_parts) + 16 > s.pool_chunks) {
            int rc = e2_ensure_pool(c, 0, 0, st);
            if (rc) return rc;
        }
    }
    const unsigned grid = (unsigned)std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 1u << 20);
    if (c->NW == 2) hipLaunchKernelGGL(skm_ingest_kernel<2>, dim3(grid), dim3(BLOCK), 0, st, d_recs, d_pids, n, dev_view(c), c->ctr);
    else hipLaunchKernelGGL(skm_ingest_kernel<4>, dim3(grid), dim3(BLOCK), 0, st, d_recs, d_pids, n, dev_view(c), c->ctr);
    E2_TRY(hipGetLastError());
    s.counted = false;
    return PG_OK;
}

int e2_scatter(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base, uint64_t n_reads,
               uint32_t uniform_len, uint64_t n_kmers_hint, uint64_t ord_base, hipStream_t st) {
    {
        const uint64_t nk = uniform_len ? n_reads * (uint64_t)(uniform_len - c->K + 1) : n_kmers_hint;
        int rc = e2_ensure_pool(c, n_reads, nk, st);
        if (rc) return rc;
    }
    if ((ord_base >> (64 - SKM_ORD_SHIFT)) != 0) { pg_set_error("ordinal exceeds the 46 bits of a super-k-mer header"); return PG_EINVAL; }
    ReadsArg a;
    a.packed = d_packed; a.word_off = d_word_off; a.kmer_base = d_kmer_base; a.n_reads = n_reads; a.uniform_len = uniform_len;
    a.kpr = uniform_len ? uniform_len - c->K + 1 : 0;
    a.wpr = uniform_len ? (uniform_len + 31) / 32 : 0;
    a.ord_base = ord_base;
    // tiled kernel for uniform batches whose per-read LDS footprint fits
    int tiled = uniform_len && uniform_len < 4096 && (int)a.kpr < 4096;
    if (const char* v = getenv("PG_K1")) tiled = tiled && atoi(v) != 0;
    if (tiled) {
        int rc = launch_tiled(c, a, nullptr, st);
        if (rc != 1) return rc;
    }
    const uint64_t grid = (n_reads + BLOCK - 1) / BLOCK;
    if (grid > 0x7FFFFFFFULL) { pg_set_error("batch too large for one launch"); return PG_EINVAL; }
    if (c->NW == 2) hipLaunchKernelGGL(skm_scatter_kernel<2>, dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr);
    else hipLaunchKernelGGL(skm_scatter_kernel<4>, dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr);
    E2_TRY(hipGetLastError());
    c->e2.counted = false;
    return PG_OK;
}

int e2_count(pg_ctx* c, int delow, bool want_last_put, hipStream_t st) {
    E2& s = c->e2;
    const SetParams sp{(uint32_t)c->P, set_bias((uint32_t)c->P)};
    E2_TRY(hipMemsetAsync(&c->ctr->n_export, 0, sizeof(unsigned long long), st));
    E2_TRY(hipMemsetAsync(c->ctr->hist, 0, sizeof(unsigned long long) * 256, st));
    E2_TRY(hipMemsetAsync(&c->ctr->n_records, 0, sizeof(unsigned long long), st));
    E2_TRY(hipMemsetAsync(c->ctr->phase, 0, sizeof(unsigned long long) * 12, st));
    const uint32_t parts = 1u << s.log2_parts;
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    const unsigned grid = std::min<unsigned>(parts, (unsigned)n_cu * 8u);           // persistent workgroups, x8 per CU for balance
    int dbg = 0, cfg = 0;
    if (const char* v = getenv("PG_DBG")) dbg = atoi(v);
    if (const char* v = getenv("PG_K2CFG")) cfg = atoi(v);
    // cfg 0: 2048-slot set, 1024 lanes, 512-record windows  -> ~150 KB LDS, one workgroup per CU
    // cfg 1: 1024-slot set,  512 lanes, 256-record windows  ->  ~77 KB LDS, two workgroups per CU
    if (c->NW == 2) {
        if (cfg == 0) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512>), dim3(grid), dim3(1024), 0, st, dev_view(c), delow, sp, c->ctr, dbg);
        else hipLaunchKernelGGL((skm_count_kernel<2, 1024, 512, 256>), dim3(grid), dim3(512), 0, st, dev_view(c), delow, sp, c->ctr, dbg);
    } else {
        if (cfg == 0) hipLaunchKernelGGL((skm_count_kernel<4, 1024, 1024, 512>), dim3(grid), dim3(1024), 0, st, dev_view(c), delow, sp, c->ctr, dbg);
        else hipLaunchKernelGGL((skm_count_kernel<4, 512, 512, 256>), dim3(grid), dim3(512), 0, st, dev_view(c), delow, sp, c->ctr, dbg);
    }
    E2_TRY(hipGetLastError());
    if (want_last_put) {
        E2_TRY(hipMemsetAsync(c->ctr->set_last, 0, sizeof(unsigned long long) * 256, st));
        if (c->NW == 2) hipLaunchKernelGGL(skm_lastput_kernel<2>, dim3(grid), dim3(BLOCK), 0, st, dev_view(c), sp, c->ctr);
        else hipLaunchKernelGGL(skm_lastput_kernel<4>, dim3(grid), dim3(BLOCK), 0, st, dev_view(c), sp, c->ctr);
        E2_TRY(hipGetLastError());
    }
    E2_TRY(hipStreamSynchronize(st));
    DevCounters h;
    E2_TRY(hipMemcpy(&h, c->ctr, sizeof h, hipMemcpyDeviceToHost));
    if (h.e2_flags & F_POOL) { pg_set_error("partition engine: record pool exhausted (raise log2_slots or PG_POOL_MB)"); return PG_ENOMEM; }
    if (h.e2_flags & F_CHUNKS) { pg_set_error("partition engine: one partition outgrew its chunk list (heavily skewed minimizers)"); return PG_ENOMEM; }
    if ((h.e2_flags & F_OUT) && c->autogrow && !(h.e2_flags & (F_POOL | F_CHUNKS | F_SPLIT))) {
        // the streams are intact: count again into an export array that holds everything (n_export is the true count)
        const uint64_t want = h.n_export + h.n_export / 8 + 64;
        uint64_t* fresh = nullptr;
        E2_TRY(hipMalloc(&fresh, want * (uint64_t)(c->NW + 2) * 8));
        E2_TRY(hipFree(s.out));
        s.out = fresh;
        s.out_capacity = want;
        unsigned long long keep = h.e2_flags & ~F_OUT;
        E2_TRY(hipMemcpy(&c->ctr->e2_flags, &keep, sizeof keep, hipMemcpyHostToDevice));
        return e2_count(c, delow, want_last_put, st);
    }
    if (h.e2_flags & F_OUT) { pg_set_error("partition engine: more distinct k-mers than the export array holds (raise log2_slots)"); return PG_ENOMEM; }
    if (h.e2_flags & F_SPLIT) { pg_set_error("partition engine: a partition could not be split to fit the LDS set"); return PG_ENOMEM; }
    if (dbg & 2) {
        static const char* names[10] = {"meta+sync", "clear", "stage", "flatten", "occurrences (own share)", "wait for the slowest lane",
                                        "emit: count+atomic+finalise", "emit: wait for the atomic", "emit: writes", "dropped attempts"};
        unsigned long long tot = 0;
        for (int i = 0; i < 9; i++) tot += h.phase[i];
        for (int i = 0; i < 10; i++) fprintf(stderr, "K2 phase %-32s %14llu  %5.1f%%\n", names[i], h.phase[i], i < 9 ? 100.0 * h.phase[i] / (double)(tot ? tot : 1) : 0.0);
    }
    s.counted = true;
    return PG_OK;
}

}  // namespace pg
