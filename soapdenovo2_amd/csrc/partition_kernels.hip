// partition_kernels.hip -- pass 1 as "partition, then count in LDS" (engine 2), gfx950.
//
// Same contract as the global-set engine of pregraph_kernels.hip (per distinct canonical k-mer: the reference's
// two node words, first-occurrence ordinal, set id; prlHashReads.c:163-259 + newhash.c:473-528), different
// formulation, chosen because a DRAM-resident set is capped by the chip's random-atomic rate (~24 G ops/s,
// profiles/r01_membench_random_access.log) at ~13 % of the HBM roofline:
//
//   K1  skm_scatter_seg_kernel  (uniform-length batches; skm_scatter_kernel = one lane per read for ragged ones)
//                            cut the reads into super-k-mers by minimizer partition (skm.hpp, skm_tile.hpp), append each
//                            as a fixed-size record to its partition's stream (streams = chains of 6 KB chunks from one
//                            pool; one atomic per record, not per k-mer: ~20x fewer, and the write side is plain
//                            48-byte stores).
//   K2  skm_count_kernel     one workgroup per partition (persistent grid): expand the partition's records
//                            back into k-mer occurrences (occ32.hpp) and insert them into a set that lives in LDS
//                            (120 KB, word-wise CAS claim from the all-ones pattern, 63-bit key words), then
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
#include <vector>

#include <chrono>

#include "device_ctx.hpp"
#include "arena.hpp"
#include "env.hpp"
#include "extract.hpp"
#include "occ32.hpp"
#include "skm_tile.hpp"
#include "e2_plan.hpp"
#include "graph_lookup.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

constexpr uint64_t L_EMPTY = ~0ULL;
constexpr unsigned long long F_POOL = 1, F_CHUNKS = 2, F_OUT = 4, F_SPLIT = 8, F_ROUTE = 16;   // F_ROUTE: an owner's send region overflowed (multi-GPU cut)
constexpr unsigned long long F_LEN = 32;   // a read of a ragged batch is longer than the bound its tiles were sized for, or shorter than K + 1

// Slots a probe step looks ahead (word 0 only: lds_put), measured at 200 M reads / configs[1] (profiles/r06_k2_probe_look_ahead_ab.json): 0 -> 1 -> 2 -> 3 slots:
// K = 63 131.0 -> 125.8 -> 124.3 -> 124.5 ms, K = 127 108.2 -> 99.0 -> 97.9 -> 99.3, K = 31 (10 M x 100 bp) 16.8 -> 14.6 -> 14.0 -> 13.5: two, and three for the short
// k-mers of the K = 31 kernel (its partitions are the fullest: 95.8 M distinct k-mers in 2^16 partitions).  -DPG_K2_LOOK=n overrides all of them (A/B builds).
template <int NW> struct E2Cfg;
// LDS slot = KW key words | ord | 5 x u32 of counters (LdsSet below).  Two-word flavour: words of 63 bits, so that no word of a key
// is the empty mark ~0 and a slot is claimed word by word.  Four-word flavour: 254 bits do not fit four such words, and a fifth costs
// 8 of 68 bytes a slot -- the k-mer's own four words instead (the first, the most significant, has its two top bits free: never ~0,
// and bit 63 marks a slot whose other words are still being written; lds_put)
// MAXPROBE: a put gives up behind that many slots and the attempt is dropped (below).  48 until round 5; measured then at 200 M reads (profiles/r05q_k2_maxprobe_ab.json):
// K = 63  48 -> 142.6 ms, 64 -> 139.4, 96 -> 137.3, 128 -> 136.7, 192 -> 136.3, 256 -> 136.4;  K = 127  48 -> 128.6, 96 -> 126.0, 128 -> 126.5, 192 -> 128.1 --
// one partition in fifty was dropped and counted again in two sittings for a probe sequence the set could well have taken.
template <> struct E2Cfg<2> { static constexpr int PW = 5, KW = 2, MAXPROBE = 128; static constexpr bool RAW = false; };   // LDS slot: 2 key words + ord + 20 B of counters = 44 B
template <> struct E2Cfg<4> { static constexpr int PW = 7, KW = 4, MAXPROBE = 96; static constexpr bool RAW = true; };    // 4 key words + ord + 20 B = 60 B: 2048 slots in 120 KB

struct E2Dev {
    SkmGeom g;
    uint32_t rpc, maxc, rpc_log2;        // records per chunk: a power of two
    uint32_t rs, direct;                 // record stride in words; chunks per partition at computed addresses (device_ctx.hpp)
    uint64_t pool_chunks;
    uint32_t* cursor;
    uint32_t* chunk_tbl;
    uint64_t* pool;
    uint64_t* out;
    uint64_t out_capacity;
    uint32_t own_div;                    // ranks that share the job's partition ids (1: this context stores them all)
};

struct ReadsArg {
    const uint64_t* packed;
    const uint64_t* word_off;
    const uint64_t* kmer_base;
    uint64_t n_reads;
    uint32_t uniform_len, kpr, wpr;  // ragged batches through the tiled kernel: uniform_len = 0, kpr / wpr = those of the longest read (max_len)
    uint64_t ord_base;
    uint32_t max_len;                // ragged batches: no read is longer (0 = unknown: the one-lane-a-read kernel)
    // ragged batches cut by length class: a class's reads, gathered -- word_off points at the class's own row of start words, and these at its rows of
    // first k-mers and lengths (null: kmer_base[i], kmer_base[i + 1] of the batch as it lies)
    const uint64_t* cls_kb;
    const int32_t* cls_len;
};

// address of record q of partition pid.  The lane that draws the first record of a chunk (q % rpc == 0) takes a
// chunk from the pool and publishes its id; lanes with later records of the same chunk wait for the id.
// Deadlock freedom inside a wavefront: the publish is NOT the other arm of the wait (`if (first) publish; else wait`
// lets the compiler run the wait arm first and starve the publisher of the same wave); every lane runs
// "publish if first" and THEN the wait loop, and a publisher never waits for anything before its store.  The lane
// with q % rpc == 0 drew its number before any lane with a later number of that chunk, so its publish is already
// issued (same or earlier instruction of this wave, or an independent wave).  The wait is bounded anyway.
__device__ __forceinline__ uint64_t* record_slot(const E2Dev& e, uint32_t pid, uint32_t q, DevCounters* ctr, int) {
    uint32_t ci = q >> e.rpc_log2;
    const uint32_t ri = q & (e.rpc - 1);
    if (ci < e.direct)                                             // chunk ci of partition pid has a fixed place: pool chunk ci * parts + pid
        return e.pool + ((((uint64_t)ci << e.g.log2_parts) + pid) * e.rpc + ri) * (uint64_t)e.rs;
    ci -= e.direct;
    if (ci >= e.maxc) { atomicOr(&ctr->e2_flags, F_CHUNKS); return nullptr; }
    uint32_t* t = e.chunk_tbl + (uint64_t)pid * e.maxc + ci;
    if (ri == 0) {
        const uint32_t nsub = min(POOL_SUBS, 1u << e.g.log2_parts), sub = pid & (nsub - 1);      // (every sub-pool serves as many partitions)
        const unsigned long long nc = ((unsigned long long)e.direct << e.g.log2_parts) + atomicAdd(&ctr->pool_sub[sub * 8], 1ULL) * nsub + sub + 1;
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
    return e.pool + ((uint64_t)(c - 1) * e.rpc + ri) * (uint64_t)e.rs;
}
// the pool chunk (id, 1-based; 0 / 0xFFFFFFFF = none) that holds chunk `ci` of partition pid
__device__ __forceinline__ uint32_t chunk_id_of(const E2Dev& e, uint32_t pid, uint32_t ci) {
    if (ci < e.direct) return (uint32_t)((((uint64_t)ci << e.g.log2_parts) + pid) + 1);
    ci -= e.direct;
    return ci < e.maxc ? e.chunk_tbl[(uint64_t)pid * e.maxc + ci] : 0u;
}

// multi-GPU: instead of appending to the local partition streams, records go to per-owner send regions (owner =
// partition mod n_owners), `cap` records each, with the partition id alongside; cursor[o] counts what owner o gets.
struct RouteArg { uint64_t* recs; uint32_t* pids; unsigned long long* cursor; uint64_t cap; int n_owners; };

// one lane per read: any mix of read lengths (the tiled kernel below takes the uniform batches)
template <int NW, bool ROUTE>
__global__ __launch_bounds__(BLOCK) void skm_scatter_kernel(ReadsArg a, E2Dev e, DevCounters* ctr, RouteArg ro) {
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
        uint64_t* dst;
        if (ROUTE) {
            const uint32_t o = pid % (uint32_t)ro.n_owners;
            const unsigned long long at = atomicAdd(&ro.cursor[o], 1ULL);
            if (at >= ro.cap) { atomicOr(&ctr->e2_flags, F_ROUTE); return; }
            ro.pids[(uint64_t)o * ro.cap + at] = pid;
            dst = ro.recs + ((uint64_t)o * ro.cap + at) * RW;
        } else {
            const uint32_t q = atomicAdd(&e.cursor[pid], 1u);
            dst = record_slot(e, pid, q, ctr, RW);
        }
        if (!dst) return;
        uint64_t rec[RW];
        skm_make_record<PW>(rd, len, j0, n, ord0, e.g, rec);
#pragma unroll
        for (int i = 0; i < RW; i++) dst[i] = rec[i];
    });
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t i, uint32_t inv) { return __umulhi(i, inv); }
// i / d for small i with inv = ceil(2^32 / d); d = 1 has no 32-bit reciprocal
__device__ __forceinline__ uint32_t fastdiv1(uint32_t i, uint32_t d, uint32_t inv) { return d == 1 ? i : __umulhi(i, inv); }

// Tiled K1: every thread does a short serial piece of work on consecutive positions
// (skm_tile.hpp).  (Round 1 had one lane per k-mer and a log-step sliding minimum by wave shuffles: ~480 vector
// instructions per read, profiles/r01_pmc_sq_bench20M_engine2.json; this form measures 1.65x faster,
// profiles/r02_k1k2_rewrite_ab.json.)
//   load  the tile's reads as dword strings (hi dword of every 64-bit word first)
//   A     thread = (read, 16 positions): m-mer values from two dwords, compile-time funnel shifts
//   B     thread = (read, S k-mers): window minima with w + S reads (suffix / core / prefix), partition ids, run-start bits
//   D     thread = the same segment: one LDS atomic reserves its items, every start bit finds the next start
//   E     one lane per run: slot in the partition's stream (or the owner's send region), record from the dword string
// Lanes of a wave take different reads (read index fastest), so the row stride -- forced odd -- is the bank stride.
// RAGGED: the reads of the batch have their own lengths (prlHashReads.c:642-648: lenBuffer[r], any read of K + 1 bases and more is
// chopped the same way).  The tile's rows are as long as the batch's longest read needs; a read's length and first ordinal sit in LDS
// beside its row, a segment past the read's last k-mer has no start bits, and everything else is the uniform kernel.
struct SegArg { int R, np, npad, wsd, nseg, nca; uint32_t inv_R, inv_wpr; };

// W: the window length (m-mers a k-mer) at compile time, with 16-mers (0: whatever the geometry says) -- the loops of phase B
// unroll into loads with immediate offsets, phase A loses its shifts by 32 - 2m.
template <int NW, bool ROUTE, int S, int W = 0, bool RAGGED = false>
__global__ __launch_bounds__(BLOCK) void skm_scatter_seg_kernel(ReadsArg a, E2Dev e, DevCounters* ctr, SegArg sa, RouteArg ro) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int R = sa.R, np = sa.np, npad = sa.npad, wsd = sa.wsd, nseg = sa.nseg, nca = sa.nca;
    const int wpr = (int)a.wpr, kpr = (int)a.kpr, len = (int)a.uniform_len;      // (RAGGED: wpr, kpr, np = the longest read's)
    uint32_t* dw = (uint32_t*)smem_raw;                           // R * wsd   dword strings
    uint32_t* v0 = dw + (size_t)R * wsd;                          // R * npad  m-mer values; the item list after B
    // (odd row strides here too: lanes of a wave take different reads, so the row stride is the bank stride -- nseg * S = 88 and nseg = 8 at 150 bp, K = 63
    //  put a wave's 64 rows on 8 banks each: SQ_LDS_BANK_CONFLICT was 2.2x the LDS issue cycles of this kernel, profiles/r04p_pmc_sq_bench20M.json)
    const int kpad = (nseg * S) | 1;                              // partition ids: whole segments, so a segment stores all S of its ids
    const int mpad = nseg | 1;
    uint32_t* pids = v0 + (size_t)R * npad;                       // R * kpad
    uint32_t* smask = pids + (size_t)R * kpad;                    // R * mpad  run-start bits
    uint32_t* ranks = smask + (size_t)R * mpad;                   // R * kpr   (ROUTE only)
    uint32_t* items = v0;
    // RAGGED: per read its first k-mer among the batch's (8-byte aligned: the rows above are dwords) and its length
    uint64_t* rkb = (uint64_t*)(((uintptr_t)(ranks + (ROUTE ? (size_t)R * kpr : 0)) + 7) & ~(uintptr_t)7);
    int* rlen = (int*)(rkb + R);
    __shared__ unsigned int n_items;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    const int nr = (int)min((uint64_t)R, a.n_reads - r0);
    if (threadIdx.x == 0) n_items = 0;
    if (RAGGED) {
        for (int r = threadIdx.x; r < nr; r += BLOCK) {
            const uint64_t kb = a.cls_len ? a.cls_kb[r0 + r] : a.kmer_base[r0 + r];
            int l = a.cls_len ? a.cls_len[r0 + r] : (int)(a.kmer_base[r0 + r + 1] - kb) + e.g.K - 1;
            if (l > (int)a.max_len || l < e.g.K + 1) { atomicOr(&ctr->e2_flags, F_LEN); l = 0; }      // (a read without k-mers from here on; the batch fails in e2_count)
            rkb[r] = kb; rlen[r] = l;
        }
        __syncthreads();
        // lanes over (read, word of the longest read): a read's words are consecutive lanes, the reads of a tile consecutive in memory
        for (int i = threadIdx.x; i < nr * wpr; i += BLOCK) {
            const int r = (int)fastdiv1(i, wpr, sa.inv_wpr), k = i - r * wpr;
            const uint64_t wd = 32 * k < rlen[r] ? a.packed[a.word_off[r0 + r] + k] : 0ULL;
            *(uint2*)(dw + r * wsd + 2 * k) = make_uint2((uint32_t)(wd >> 32), (uint32_t)wd);
        }
    } else {
        for (int i = threadIdx.x; i < nr * wpr; i += BLOCK) {
            const int r = (int)fastdiv1(i, wpr, sa.inv_wpr), k = i - r * wpr;
            const uint64_t wd = a.packed[r0 * wpr + i];
            *(uint2*)(dw + r * wsd + 2 * k) = make_uint2((uint32_t)(wd >> 32), (uint32_t)wd);
        }
    }
    for (int r = threadIdx.x; r < nr; r += BLOCK)
        for (int k = 2 * wpr; k < wsd; k++) dw[r * wsd + k] = 0;
    __syncthreads();
    const int m = W ? 16 : e.g.m, w = W ? W : e.g.w;
    // tasks are numbered over the full tile (read index fastest), a short last tile just leaves lanes idle
    for (int t = threadIdx.x; t < R * nca; t += BLOCK) {
        const int c = (int)fastdiv1(t, R, sa.inv_R), r = t - c * R;
        if (r < nr) tile_mmer_chunk<(W ? 16 : 0)>(dw + r * wsd, c, m, v0 + r * npad);
    }
    __syncthreads();
    // (Phase B with the block minima of a read shared between its segments -- suffix minima first, a barrier, cores from three block minima instead of
    //  37 values: 29 LDS loads a segment instead of 59 -- was built and measured in round 5: 62.3 ms against 57.0 per 200 M reads on one box,
    //  profiles/r05c_k1_shared_minima_ab.json.  The loads it saves have immediate offsets and pipeline; the barrier and the second pass do not.)
    for (int t = threadIdx.x; t < R * nseg; t += BLOCK) {
        const int seg = (int)fastdiv1(t, R, sa.inv_R), r = t - seg * R;
        if (r >= nr) continue;
        const int j0 = seg * S;
        const int kpr_r = RAGGED ? rlen[r] - e.g.K + 1 : kpr, np_r = RAGGED ? rlen[r] - m + 1 : np;
        const int cnt = min(S, kpr_r - j0);
        if (RAGGED && cnt <= 0) { smask[r * mpad + seg] = 0; continue; }       // behind the read's last k-mer
        uint32_t pid[S];
        const uint32_t mk = tile_segment<S, W>(v0 + r * npad, np_r, j0, cnt, w, e.g.nmax, e.g.part_mul, pid);
        smask[r * mpad + seg] = mk;
#pragma unroll
        for (int i = 0; i < S; i++) pids[r * kpad + j0 + i] = pid[i];
    }
    __syncthreads();                                              // v0 is dead from here: the item list takes its place
    for (int t = threadIdx.x; t < R * nseg; t += BLOCK) {
        const int seg = (int)fastdiv1(t, R, sa.inv_R), r = t - seg * R;
        uint32_t mk = r < nr ? smask[r * mpad + seg] : 0u;
        if (mk) {
            unsigned int at = atomicAdd(&n_items, (unsigned int)__popc(mk));
            while (mk) {
                const int i = __ffs((int)mk) - 1;
                mk &= mk - 1;
                const int j = seg * S + i;
                const int nxt = tile_next_start(smask + r * mpad, seg, nseg, S, i, RAGGED ? rlen[r] - e.g.K + 1 : kpr);
                items[at++] = ((uint32_t)r << 24) | ((uint32_t)j << 12) | (uint32_t)(nxt - j);
            }
        }
    }
    __syncthreads();
    const int total = (int)n_items;
    __shared__ unsigned int ocnt[256];
    __shared__ unsigned long long obase[256];
    if (ROUTE) {
        ocnt[threadIdx.x] = 0;
        __syncthreads();
        for (int it = threadIdx.x; it < total; it += BLOCK) {
            const uint32_t pk = items[it];
            const uint32_t pid = pids[(int)(pk >> 24) * kpad + (int)((pk >> 12) & 0xFFF)];
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
        const uint32_t pid = pids[r * kpad + j0];
        uint64_t* out;
        uint32_t q = 0;
        // the returned atomic on the partition's cursor is asked first and looked at after the record is built
        if (!ROUTE) q = atomicAdd(&e.cursor[pid], 1u);
        uint64_t rec[RW];
        if (RAGGED) tile_make_record<PW>(dw + r * wsd, rlen[r], j0, n, a.ord_base + rkb[r], e.g.K, rec);
        else tile_make_record<PW>(dw + r * wsd, len, j0, n, a.ord_base + (r0 + (uint64_t)r) * (uint64_t)kpr, e.g.K, rec);
        if (ROUTE) {
            const uint32_t o = pid % (uint32_t)ro.n_owners;
            const unsigned long long at = obase[o] + ranks[it];
            if (at >= ro.cap) { atomicOr(&ctr->e2_flags, F_ROUTE); continue; }
            ro.pids[(uint64_t)o * ro.cap + at] = pid;
            out = ro.recs + ((uint64_t)o * ro.cap + at) * RW;
        } else out = record_slot(e, pid, q, ctr, RW);
        if (!out) continue;
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
        // the sender's id is the job's; this rank owns the ids with id mod own_div == its rank and stores them at id / own_div
        const uint32_t pid = e.own_div > 1 ? rpids[i] / e.own_div : rpids[i];
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
// cnt[5][SLOTS] (L0|L1<<16, L2|L3<<16, R0|R1<<16, R2|R3<<16, puts without a left neighbour).  Keys and ord start as ~0,
// counters as 0.  A slot is claimed key word by key word: an empty word is taken with CAS(~0 -> mine), a word holding
// something else means another key owns the slot.
// Counting is plain atomic adds, saturated when the node is emitted: a sum of +1's clipped at the end equals the
// reference's saturating increments (newhash.c:74-106) and, unlike a CAS on packed counters, needs no retry when many
// lanes hit one hot k-mer.  `single` = exactly one put (newhash.c:127,511).
template <int NW, int SLOTS>
struct LdsSet {
    static constexpr int KW = E2Cfg<NW>::KW;
    unsigned long long key[KW][SLOTS];
    unsigned long long ord[SLOTS];
    unsigned int cnt[5][SLOTS];       // 16-bit halves: a window adds at most WIN * nmax <= 512 * 127 to a field, and fields are
                                      // clipped to 255 between the windows of a partition (clip_halves_255)
};
__device__ __forceinline__ unsigned int clip_halves_255(unsigned int x) { return min(x & 0xFFFFu, 255u) | (min(x >> 16, 255u) << 16); }

// A put gives up when the set is too full (a probe sequence longer than E2Cfg<NW>::MAXPROBE): the caller aborts the attempt and
// splits the key range.  No shared key counter and no list of claimed slots on this path: measured again in round 2 (one
// wave-aggregated LDS atomic per step that claims a slot), the put loop lost more than the emit's listing phase costs.
constexpr int K2_MAXSPIN = 4096;                                           // looks at a slot that is being claimed (four-word flavour)

__device__ __forceinline__ const uint64_t* record_ptr(const E2Dev& e, uint32_t pid, uint32_t i, int) {
    const uint32_t c = chunk_id_of(e, pid, i >> e.rpc_log2);
    if (c == 0 || c == 0xFFFFFFFFu) return nullptr;  // pool ran dry in K1 (flagged there; the run fails in e2_count)
    return e.pool + ((uint64_t)(c - 1) * e.rpc + (i & (e.rpc - 1))) * (uint64_t)e.rs;
}

// ---- K2 ------------------------------------------------------------------------------------------------------------
// One workgroup per partition (persistent grid).  The partition's records are taken WIN at a time.  A partition's first window
// is fetched while the previous partition is emitted (through registers: p_ask / p_put below), its further ones where they are needed; exact
// copies of a record are merged (dedupe), the representatives' k-mer counts go through a prefix sum so that every lane
// gets an equal contiguous share of occurrences, and the occurrences are expanded and inserted into the LDS set.  If the
// set overflows, the attempt is dropped and the key range is split on a hash bit.  Then the set is finalised (-d filter,
// linear flag, coverage histogram: prlHashReads.c:953-1132) and emitted as export records.
// The per-occurrence path (round 1 spent ~360 vector instructions on it, profiles/r01_pmc_sq_bench20M_engine2.json):
//   * records are staged as dword strings (hi dword first), a k-mer occurrence is cut out by occ_extract (occ32.hpp):
//     six dword reads, funnel shifts, bit reversal -- 32-bit operations throughout, shift amounts that depend on K
//     alone are wave-uniform;
//   * the slot hash is three 32-bit multiplies instead of three 64-bit ones;
//   * no per-lane cache of "the current record": the offset-table entries (first occurrence | place in the window | flank
//     bits) are simply read again every step, one step ahead of their use, so the step is straight-line code -- with 64
//     lanes a wave crossed a record boundary on nearly every step anyway and paid for the divergent refill each time;
//   * the first record of a lane's share comes from a table the flatten step fills (one lane per record writes the lanes
//     whose share starts inside it) instead of a 9-step binary search per lane and window;
//   * the put counter is gone: every put adds to exactly one of L[0..3] / "no left neighbour", so puts = their sum, and
//     the first-occurrence ordinal is only sent through atomicMin when it is smaller than the value just read with the key
//     (a hot k-mer's lanes hit the same word: same-address LDS atomics serialise, same-address reads broadcast).
template <int NW, int SLOTS, int LOOK>
__device__ __forceinline__ bool lds_put(LdsSet<NW, SLOTS>& t, const uint64_t (&kw)[E2Cfg<NW>::KW], uint32_t hash, uint32_t left, uint32_t right,
                                         uint64_t ord, uint32_t copies) {
    constexpr int KW = E2Cfg<NW>::KW;
    uint32_t h = hash & (SLOTS - 1);
    if constexpr (E2Cfg<NW>::RAW) {
        // The claim is ONE compare-and-swap on word 0 (empty -> mine | L_PENDING); the winner then writes words 1..3 and, released
        // behind them, word 0 without the mark.  Everybody reads word 0 FIRST (volatile: the four reads keep their order, and the LDS
        // serves a wave's operations in order): a clean word 0 therefore comes with final words 1..3.  Whoever meets the mark -- or
        // loses the claim -- looks at the same slot again; the winner never waits for anybody, so this ends.
        // (looking again is not a probe: a lane that keeps meeting a slot another wave is still filling must not run out of the probe budget and
        //  report a full set -- the attempt would be dropped and the key range split for nothing; the spins have their own, larger bound)
        constexpr unsigned long long L_PENDING = 1ULL << 63;
        for (int probes = 0, spins = 0; probes < E2Cfg<NW>::MAXPROBE && spins < K2_MAXSPIN;) {
            unsigned long long seen[KW];
#pragma unroll
            for (int i = 0; i < KW; i++)                                  // (an LDS pointer, said so: a plain volatile one is read through the flat path, one load at a time)
                seen[i] = *(volatile __attribute__((address_space(3))) unsigned long long*)(&t.key[i][h]);
            const unsigned long long so = t.ord[h];
            unsigned long long nxt0[LOOK];                      // (see the two-word flavour below)
#pragma unroll
            for (int q = 0; q < LOOK; q++) nxt0[q] = *(volatile __attribute__((address_space(3))) unsigned long long*)(&t.key[0][(h + 1 + q) & (SLOTS - 1)]);
            bool mine = false, again = false;
            if (seen[0] == L_EMPTY) {
                const unsigned long long old = atomicCAS(&t.key[0][h], L_EMPTY, (unsigned long long)kw[0] | L_PENDING);
                if (old == L_EMPTY) {
#pragma unroll
                    for (int i = 1; i < KW; i++) t.key[i][h] = (unsigned long long)kw[i];
                    __hip_atomic_store(&t.key[0][h], (unsigned long long)kw[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    mine = true;
                } else again = true;
            } else if (seen[0] & L_PENDING) again = true;
            else {
                mine = true;
#pragma unroll
                for (int i = 0; i < KW; i++) mine = mine && seen[i] == kw[i];
            }
            if (mine) {
                atomicAdd(&t.cnt[left < 4 ? left >> 1 : 4][h], left < 4 ? copies << ((left & 1u) * 16u) : copies);
                if (right < 4) atomicAdd(&t.cnt[2 + (right >> 1)][h], copies << ((right & 1u) * 16u));
                if (ord < so) atomicMin(&t.ord[h], (unsigned long long)ord);
                return true;
            }
            if (again) spins++;
            else {
                // (a next slot that is taken -- or being taken: the mark aside, word 0 is final -- by another key is passed by without a step of its own)
                uint32_t adv = 1;
                bool run = true;
#pragma unroll
                for (int q = 0; q < LOOK; q++) { run = run && nxt0[q] != L_EMPTY && (nxt0[q] & ~L_PENDING) != kw[0]; adv += run ? 1u : 0u; }
                h = (h + adv) & (SLOTS - 1);
                probes += (int)adv;
            }
        }
        return false;
    }
    for (int probes = 0; probes < E2Cfg<NW>::MAXPROBE; probes++) {
        unsigned long long seen[KW];
#pragma unroll
        for (int i = 0; i < KW; i++) seen[i] = t.key[i][h];
        const unsigned long long so = t.ord[h];
        // (round 6) word 0 of the NEXT slot comes with this slot's words -- one more 8-byte read in the same wait: a next slot that is taken by another key
        // (word 0 of a slot never changes once it is set) is passed by without a step of its own.  A step is an LDS round trip, a workgroup's occurrence phase
        // ends when its longest probe sequence does, and the others wait for it at the barrier: the sequences through a crowded stretch are half as long.
        unsigned long long nxt0[LOOK];
#pragma unroll
        for (int q = 0; q < LOOK; q++) nxt0[q] = t.key[0][(h + 1 + q) & (SLOTS - 1)];
        bool mine = true;
#pragma unroll
        for (int i = 0; i < KW; i++) {
            if (!mine) break;
            unsigned long long cur = seen[i];
            if (cur == L_EMPTY) {
                const unsigned long long old = atomicCAS(&t.key[i][h], L_EMPTY, (unsigned long long)kw[i]);
                cur = old == L_EMPTY ? (unsigned long long)kw[i] : old;
            }
            mine = cur == kw[i];
        }
        if (mine) {
            // exactly one of L[0..3] / "none" per put
            atomicAdd(&t.cnt[left < 4 ? left >> 1 : 4][h], left < 4 ? copies << ((left & 1u) * 16u) : copies);
            if (right < 4) atomicAdd(&t.cnt[2 + (right >> 1)][h], copies << ((right & 1u) * 16u));
            if (ord < so) atomicMin(&t.ord[h], (unsigned long long)ord);
            return true;
        }
        uint32_t adv = 1;                                                  // ... plus the slots ahead, while they are another key's
        bool run = true;
#pragma unroll
        for (int q = 0; q < LOOK; q++) { run = run && nxt0[q] != L_EMPTY && nxt0[q] != kw[0]; adv += run ? 1u : 0u; }
        h = (h + adv) & (SLOTS - 1);
        probes += (int)adv - 1;
    }
    return false;
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of the
// wave (s_waitcnt vmcnt(0)): K2's global stores (export records) are never read back by the kernel and its global loads are
// waited for where their registers are used, so draining them at each of the ~15 barriers of a partition only serialises
// memory latency with the LDS phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sum over the 64 lanes of a wave on the data-parallel-primitive path: four shifted adds inside each row
// of 16 lanes, then the row totals broadcast forward -- no LDS round trips (the shuffle form is six ds_bpermute)
__device__ __forceinline__ unsigned int wave_inclusive_sum(unsigned int x) {
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

constexpr int pow2_at_least(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// The occurrences of a window are dealt 64 at a time: occurrence idx goes to lane idx mod 64 of whichever wave takes tile idx / 64 off a
// counter in LDS -- a wave that finishes early (shorter probe sequences, fewer lost claims) takes the next tile instead of waiting at the
// barrier.  The flatten step marks where every representative's occurrences start (one bit an occurrence: sbits) and which representative is
// running at each tile's first occurrence (tile_rep0); a lane finds its representative with a population count over its tile's 64 start bits.
// (Rounds 2 - 3 cut the window into equal shares a lane, static or dealt as "virtual lanes" through a share table: 177.8 / 168.7 ms against
//  154.3 for this form, profiles/r04a_k2_vt0_opt1_ab.json; those forms are gone.)
// KS: the kernel for ONE k-mer length (0 = any): the K-only shift amounts of occ_extract become immediates, its two wave-uniform switches -- a
// dozen scalar branches an occurrence -- go away, and the record geometry is the usual one (128 records a chunk, records without padding).
// TIMERS: thread 0's cycles per phase into ctr->phase (instantiated in -DPG_MEASURE builds only).  The sums live in LDS, not in registers, and
// there is a timed instantiation for each K the product has one for: until round 6 the timed kernel was the general one with 25 registers of
// sums -- it spilled to scratch where the product does not (the reload of a window piece's address waited out the piece before it: "ask for
// the next window" read 8 % of a kernel whose product form spends next to nothing there), so its shares were those of another kernel.
template <int NW, int SLOTS, int THREADS, int WIN, bool TIMERS, int KS = 0>
__global__ __launch_bounds__(THREADS) void skm_count_kernel(E2Dev e, int D, SetParams sp, OccConst oc, DevCounters* ctr) {
    // What round 4 measured as switches is fixed here (each of them held wave-uniform state in scalar registers the kernel then spilled to vector
    // lanes -- 116 -> 65 of them when they became constants, 152.5 -> 145.3 ms at K = 63, profiles/r04y_k2_switches_as_constants.json):
    //   * the emit lists the live slots in any order through one returned LDS atomic a wave and stripe (the export order is unspecified anyway);
    //   * a wave asks for its next tile when it has finished the current one;
    //   * no split ahead of a foreseen overflow (round 4 split a four-word key range before counting it at 75 % foreseen load: with the probe limit at 96
    //     instead of 48 it stopped paying -- 126.0 ms with it, 125.4 without, profiles/r05q_k2_maxprobe_ab.json);
    //   * ADAPT (four-word flavour only): the search for exact copies stops where a workgroup finds few (see p_dedupe).
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1, KW = E2Cfg<NW>::KW, NWAVE = THREADS / 64, PIECES = RW / 2, N2 = 2 * NW;
    constexpr int RD = 2 * RW;                                            // dwords a record
    constexpr int PAD = 16;                                               // readable dwords in front of record 0 (a window reaches back 2 NW + 2)
    static_assert(WIN <= THREADS, "one record per lane in the flatten step");
    static_assert(PAD % 4 == 0 && PAD >= 2 * NW + 2, "front padding");
    static_assert(SLOTS % THREADS == 0, "whole stripes");
    constexpr int RL_WORDS = PAD + WIN * RD + 8;
    __shared__ LdsSet<NW, SLOTS> set;
    // two record windows: [hdr lo, hdr hi, payload as a dword string: hi, lo, hi, lo, ...].  While one partition's nodes are
    // emitted (the emit keeps its slot list and its staging area in that partition's dead window), the next partition's
    // first window arrives in the other buffer.
    __shared__ __align__(16) uint32_t rl2[2][RL_WORDS];
    // one scratch area, two lives: the record-dedupe table (DT slots), then the flattening tables
    constexpr int DT = pow2_at_least(2 * WIN);                            // open addressing over the window's records, <= 50 % full
    constexpr int NMAX = KS ? (32 * PW - (KS - 1) - 2 < 127 ? 32 * PW - (KS - 1) - 2 : 127) : 127;   // k-mers a record (skm_geometry)
    constexpr int SB_WORDS = (WIN * NMAX + 31) / 32 + 2;                  // a start bit per occurrence of a window (<= WIN * NMAX)
    constexpr int SB_AT = ((DT > WIN + 1 + WIN ? DT : WIN + 1 + WIN) + 1) & ~1;      // behind the dedupe table and behind tile_rep0; read 64 bits at a time
    constexpr int FL_WORDS = SB_AT + SB_WORDS;
    static_assert(WIN * 127 / 64 + 1 <= 2 * WIN, "tile_rep0 holds a short per tile");
    __shared__ __align__(8) unsigned int fl_raw[FL_WORDS];
    unsigned int* const dtab = fl_raw;                                    // record index + 1 of the slot's first taker, 0 = free
    unsigned int* const noff = fl_raw;                                    // [n_rep + 1] exclusive prefix sum of the representatives' k-mer counts
    unsigned short* const tile_rep0 = (unsigned short*)(fl_raw + WIN + 1);   // [tiles] the representative running at occurrence 64 * tile (lives in the dead dedupe table)
    unsigned int* const sbits = fl_raw + SB_AT;                              // bit idx = occurrence idx is the first of its representative
    __shared__ unsigned int tile_ctr;                                     // next tile of 64 occurrences
    __shared__ unsigned int dcount[WIN];                                  // copies of a representative record in the window
    __shared__ uint32_t crc_tab[4 * 256];                                 // CRC-32 sliced by four (kmer.hpp)
    __shared__ unsigned int hist[256];
    __shared__ unsigned int aborted, s_mask[40], s_val[40], wave_cnt_f[NWAVE], s_nlive, s_tot;
    __shared__ unsigned int chunk_ids2[2][256];                           // the partitions' chunk lists (direct + maxc <= 256)
    const uint32_t nchunks = e.direct + e.maxc;
    __shared__ unsigned long long out_base;
    for (int i = threadIdx.x; i < 1024; i += THREADS) crc_tab[i] = crc32_slice_entry(i >> 8, i & 255);
    for (int i = threadIdx.x; i < 256; i += THREADS) hist[i] = 0;
    if (threadIdx.x == 0) tile_ctr = NWAVE;
    if (threadIdx.x < 2 * PAD) rl2[threadIdx.x / PAD][threadIdx.x % PAD] = 0;
    if (threadIdx.x < 16) rl2[threadIdx.x >> 3][PAD + WIN * RD + (threadIdx.x & 7)] = 0;
    // (a kernel for one K is launched on the usual record geometry only -- 128 records a chunk, records without padding: e2_count -- and has it as constants)
    const int K = KS ? KS : e.g.K;
    const uint32_t RPC = KS ? 128u : e.rpc, RPC_LOG2 = KS ? 7u : e.rpc_log2;
    const uint64_t RS = KS ? (uint64_t)RW : (uint64_t)e.rs;
    const uint32_t parts = 1u << e.g.log2_parts;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long my_records = 0;
    bool dirty = true;                                                   // the LDS set needs a full wipe before the next attempt
    __shared__ unsigned long long tp[TIMERS ? 12 : 1];                    // phase timers: their own instantiations (-DPG_MEASURE); thread 0's alone
    if (TIMERS && threadIdx.x == 0) for (int i = 0; i < 12; i++) tp[i] = 0;
    unsigned long long tlast = TIMERS ? clock64() : 0;
    // Every barrier of this kernel orders LDS traffic only (lds_barrier): its global stores are never read back and its
    // global loads are waited for where their registers are used.
#define K2_SYNC() lds_barrier()
#define K2_TICK(i) do { if (TIMERS) { const unsigned long long tn_ = clock64(); if (threadIdx.x == 0) tp[(i) * TIMERS] += tn_ - tlast; tlast = tn_; } } while (0)

    // ---- prepare a window: stage -> dedupe -> flatten.  Written as barrier-free steps for a group of GS lanes (gtid = lane
    // index in the group, gwave = wave index in the group); the caller puts a barrier between the steps.
    struct Prep { bool is_rep; unsigned int n, incl, n_rep; };
    // The search for exact copies pays where there are copies (K = 63 at 300x: every other record; 155.5 ms against 194.1 without it) and costs where there are few (K = 127 from
    // 150-base reads: a record is most of a read; 162.2 ms against 141.7 without): every workgroup looks at its first 4096 records and stops searching when more than dd_pct % of
    // them represented themselves.  The threshold is where the search's share of the kernel equals what it saves: 70 % for the
    // four-word flavour (a 64-byte record to hash and compare; K = 127 from 150-base reads lies between 70 and 85 %: 142.5 ms off, 162.3 on), 82 % for the two-word one (its 15x case
    // -- 100 M reads over 1 Gb, more than 70 % of the records their own -- still gains 5 % from the search: 218.2 ms on, 229.7 off; profiles/r04m_k2_adaptive_dedupe_ab.json)
    // (the two-word flavour never stopped searching on anything measured -- 300x, 150x, 15x -- and the bookkeeping cost it 2 ms of 154: it always searches)
    constexpr bool ADAPT = NW == 4;
    bool dd_on = true;
    uint32_t dd_rec = 0, dd_rep = 0;
    constexpr uint32_t dd_pct = 70u;
    // stage the window's records: 16 bytes per lane and step; the header word as it is, every payload word high dword first
    auto p_stage = [&](auto gs_, int gtid, int nb, int cl, uint32_t w0, uint32_t wn) {
        constexpr int GS = decltype(gs_)::value;
        uint32_t* const rlb = rl2[nb];
        const unsigned int* const cids = chunk_ids2[cl];
        uint32_t pc_first = (uint32_t)gtid;
        asm volatile("" : "+v"(pc_first));                                 // (what follows from the lane's number alone is computed here, not at kernel start and then spilled:
        for (uint32_t pc = pc_first; pc < wn * PIECES; pc += GS) {         //  a reload from scratch waits for every store in flight)
            const uint32_t ri = pc / PIECES, part = pc - ri * PIECES;
            const uint32_t gi = w0 + ri, cid = cids[gi >> RPC_LOG2];
            ulonglong2 v = make_ulonglong2(0, 0);
            if (cid != 0 && cid != 0xFFFFFFFFu)
                v = ((const ulonglong2*)(e.pool + ((uint64_t)(cid - 1) * RPC + (gi & (RPC - 1))) * RS))[part];
            const uint32_t x0 = part ? (uint32_t)(v.x >> 32) : (uint32_t)v.x, x1 = part ? (uint32_t)v.x : (uint32_t)(v.x >> 32);
            ((uint4*)(rlb + PAD))[pc] = make_uint4(x0, x1, (uint32_t)(v.y >> 32), (uint32_t)v.y);
        }
        for (int i = gtid; i < DT; i += GS) dtab[i] = 0;
        for (int i = gtid; i < WIN; i += GS) dcount[i] = 1;
        for (int i = gtid; i < SB_WORDS; i += GS) sbits[i] = 0;
    };
    // The next window (round 4, late).  global_load_lds looked ideal for it -- no registers, the loads fly while the emit runs -- but the compiler
    // guards every LDS WRITE that follows such a load with s_waitcnt vmcnt(0) (it cannot tell what the load will overwrite), and the emit starts
    // with LDS writes: every wave waited out the trip to HBM at the emit's first store, and, vmcnt counting in order, the next partition's first
    // global read then waited for the acknowledgements of the export stores too -- 1 - 2 us of 18 a partition, in two places.  Now the window's
    // pieces are asked for into REGISTERS when the emit starts (p_ask: two 16-byte pieces a lane, one for the 127-mer flavour) and written to the
    // other buffer, dwords turned round, behind the first finalise and IN FRONT OF the export stores (p_put): the wait is for these loads alone,
    // with a listing and a finalise between question and answer, and the next partition starts on a prepared window.
    constexpr int PPL = (WIN * PIECES + THREADS - 1) / THREADS;
    struct Ahead { ulonglong2 v[PPL]; };
    auto p_ask = [&](int cln, uint32_t wn, Ahead& a) {
        const unsigned int* const cids = chunk_ids2[cln];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            uint32_t pc = threadIdx.x + j * THREADS;
            asm volatile("" : "+v"(pc));
            a.v[j] = make_ulonglong2(0, 0);
            if (pc < wn * PIECES) {
                const uint32_t ri = pc / PIECES, part = pc - ri * PIECES;
                const uint32_t cid = cids[ri >> RPC_LOG2];
                if (cid != 0 && cid != 0xFFFFFFFFu)
                    a.v[j] = ((const ulonglong2*)(e.pool + ((uint64_t)(cid - 1) * RPC + (ri & (RPC - 1))) * RS))[part];
            }
        }
    };
    auto p_put = [&](int nb, uint32_t wn, const Ahead& a) {
        uint32_t* const rlb = rl2[nb];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            uint32_t pc = threadIdx.x + j * THREADS;
            asm volatile("" : "+v"(pc));
            if (pc < wn * PIECES) {
                const uint32_t part = pc % PIECES;
                const ulonglong2 v = a.v[j];
                const uint32_t x0 = part ? (uint32_t)(v.x >> 32) : (uint32_t)v.x, x1 = part ? (uint32_t)v.x : (uint32_t)(v.x >> 32);
                ((uint4*)(rlb + PAD))[pc] = make_uint4(x0, x1, (uint32_t)(v.y >> 32), (uint32_t)v.y);
            }
        }
        for (int i = threadIdx.x; i < DT; i += THREADS) dtab[i] = 0;
        for (int i = threadIdx.x; i < WIN; i += THREADS) dcount[i] = 1;
        for (int i = threadIdx.x; i < SB_WORDS; i += THREADS) sbits[i] = 0;
    };
    // dedupe: at high coverage most records of a partition are exact copies of one another (every read that covers a
    // super-k-mer completely cuts out the same bases with the same flanks).  Copies are found through a small hash table over
    // the window; the first taker of a slot represents the others, which add to its count and lower its first ordinal (the
    // header word: equal records differ in nothing else, so the smaller header is the smaller ordinal).  Only
    // representatives are expanded, each occurrence counting `copies` times: the same sums, the same minima, about half the
    // work.
    auto p_dedupe = [&](int gtid, int nb, uint32_t wn, Prep& ps) {
        uint32_t* const rlb = rl2[nb];
        bool is_rep = (uint32_t)gtid < wn;
        if (is_rep && (!ADAPT || dd_on)) {
            const uint32_t* me = rlb + PAD + gtid * RD;
            uint32_t w[RD - 1];
            w[0] = me[0] & ((1u << SKM_ORD_SHIFT) - 1);                    // n, has_left, has_right
#pragma unroll
            for (int q = 1; q < RD - 1; q++) w[q] = me[q + 1];
            uint32_t hsh = 0x9E3779B1u;                                       // rotate-xor fold, one multiplicative finish
#pragma unroll
            for (int q = 0; q < RD - 1; q++) hsh = alignbit32(hsh, hsh, 27u) ^ w[q];
            hsh *= 0x85EBCA6Bu;
            hsh ^= hsh >> 15;
            uint32_t sl = hsh & (DT - 1);
            // a table entry = the taker's place in the window + 1 (10 bits) under 22 bits of its hash: a record with another hash is passed by
            // without reading it
            const unsigned int tag = (hsh >> 10) << 10;
            static_assert(WIN <= 1023, "a record's place + 1 fits 10 bits");
            for (int probes = 0; probes < DT; probes++) {
                unsigned int v = dtab[sl];
                if (v == 0) {
                    const unsigned int old = atomicCAS(&dtab[sl], 0u, tag | ((unsigned int)gtid + 1u));
                    if (old == 0) break;                                      // first of its kind: it represents the rest
                    v = old;
                }
                if ((v & ~1023u) != tag) { sl = (sl + 1) & (DT - 1); continue; }
                v &= 1023u;
                const uint32_t* it = rlb + PAD + (v - 1) * RD;
                bool same = (it[0] & ((1u << SKM_ORD_SHIFT) - 1)) == w[0];
#pragma unroll
                for (int q = 1; q < RD - 1; q++) same = same && it[q + 1] == w[q];
                if (same) {
                    atomicAdd(&dcount[v - 1], 1u);
                    atomicMin((unsigned long long*)it, *(const unsigned long long*)me);
                    is_rep = false;
                    break;
                }
                sl = (sl + 1) & (DT - 1);
            }
        }
        ps.is_rep = is_rep;
    };
    // flatten: occurrence idx -> (representative, t); every lane of the workgroup gets `share` consecutive occurrences
    auto p_flat1 = [&](int gtid, int gwave, int nb, Prep& ps) {
        ps.n = ps.is_rep ? ((rl2[nb][PAD + gtid * RD] >> 2) & 0x7Fu) : 0u;
        unsigned int incl = ps.n | (ps.is_rep ? 1u << 20 : 0u);                // k-mers below bit 20, representatives above
        incl = wave_inclusive_sum(incl);
        if (lane == 63) wave_cnt_f[gwave] = incl;
        ps.incl = incl;
    };
    // the offsets table: entry k = first occurrence index of the k-th representative | its place in the window << 16 | its
    // flank bits << 25 (at most 512 * 127 < 2^16 occurrences a window): all the occurrence loop needs to address the bases
    // of an occurrence, so the header (ordinal, copy count) is off the critical path.  The representative's copy count goes
    // into the idle bits of its LDS header (a record has at most 127 k-mers, the count field is 16 bits wide).
    auto p_flat2 = [&](auto gs_, int gtid, int gwave, int nb, Prep& ps) {
        constexpr int GW = decltype(gs_)::value / 64;
        unsigned int base = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < GW; wv++) { const unsigned int cw = wave_cnt_f[wv]; if (wv < gwave) base += cw; tot += cw; }
        const unsigned int n_rep = tot >> 20;
        ps.n_rep = n_rep;
        tot &= (1u << 20) - 1;
        if (ps.is_rep) {
            uint32_t* me = rl2[nb] + PAD + gtid * RD;
            const uint32_t h0 = me[0];
            me[0] = (h0 & ~0x3FE00u) | ((dcount[gtid] - 1u) << 9);                   // n < 128 keeps bits 2..8, copies - 1 <= 511
            const unsigned int upto = base + ps.incl, k = (upto >> 20) - 1;          // this representative's rank
            const unsigned int o_hi = upto & ((1u << 20) - 1), o_lo = o_hi - ps.n;
            noff[k] = o_lo | ((unsigned int)gtid << 16) | ((h0 & 3u) << 25);          // bit 25 = has_right, bit 26 = has_left
            atomicOr(&sbits[o_lo >> 5], 1u << (o_lo & 31u));
            // the tiles whose first occurrence lies in this representative (at most two more: a record has <= 127 k-mers)
            for (unsigned int t = (o_lo + 63u) >> 6; (t << 6) < o_hi; t++) tile_rep0[t] = (unsigned short)k;
        }
        if (gtid == 0) { noff[n_rep] = tot; s_tot = tot; tile_ctr = NWAVE; }
    };

    // ---- emit: finalize every stored node and append it to the export array.  The set is a quarter full on average, so
    // the live slots are first listed (e_list_any) and then worked on by dense waves: lane i
    // takes the i-th live slot, finalises it in registers (the -d filter, the linear flag, the coverage histogram:
    // prlHashReads.c:953-1132) and wipes the slot behind it, which is all the clearing the next attempt needs; the record
    // goes out from the registers (e_store).  One global atomic per attempt.  Barrier-free steps as above; `sb` =
    // the window buffer whose storage the list borrows.
    static_assert(RL_WORDS * 4 >= SLOTS * 2, "the list of the live slots fits a window buffer");
    // the list, in any order: a wave reserves room for its live slots of a stripe with one returned atomic on s_nlive (zeroed
    // in front of the barrier that ends the occurrence phase)
    auto e_list_any = [&](auto gs_, int gtid, int sb) {
        constexpr int GS = decltype(gs_)::value, STR = SLOTS / GS;
        unsigned short* live_list = (unsigned short*)rl2[sb];
#pragma unroll
        for (int st = 0; st < STR; st++) {
            const bool live = set.key[0][st * GS + gtid] != L_EMPTY;
            const unsigned long long bal = __ballot(live);
            unsigned int base = 0;
            if (lane == 0 && bal) base = atomicAdd(&s_nlive, (unsigned int)__popcll(bal));
            base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
            if (live) live_list[base + (unsigned int)__popcll(bal & ((1ULL << lane) - 1))] = (unsigned short)(st * GS + gtid);
        }
    };
    // (Rounds 2 - 5 finalised into a staging area in LDS and copied it out as whole 16-byte pieces, coalesced.  The emit runs with every wave of
    //  the workgroup in it at once and is bound by the LDS bytes a key moves -- slot read, slot wipe, CRC entries, staging write, staging read:
    //  round 6 keeps the record in REGISTERS across the one barrier that publishes the export base and stores it from there, (NW + 2) / 2 pieces
    //  a lane with a record's stride between lanes; the L2 puts the lines together.  A lane a live slot, THREADS slots a round: two rounds at most.)
    struct Fin { uint64_t w[NW + 2]; };
    auto e_final = [&](auto gs_, int gtid, int sb, unsigned int c0, unsigned int n_live, Fin& fin) -> bool {
        constexpr int GS = decltype(gs_)::value;
        const unsigned short* live_list = (const unsigned short*)rl2[sb];
        // The export slots come from one global counter every workgroup of the grid adds to: its answer takes a while.  The
        // group's last lane asks now and looks at the answer only after its share of the finalisation.
        // (Written as an instruction: the compiler's atomic optimizer turns an atomicAdd on a uniform address into "one lane adds, the
        //  others read its answer through readfirstlane" -- which waits for the answer on the spot (s_waitcnt vmcnt(0) right behind the
        //  atomic in round 3's code: workgroup thread 0 stood there for a round trip to the memory side while fifteen waves went on to
        //  the barrier and waited for it).  The wait is the s_waitcnt below, in front of the only use.)
        // The asker is the group's LAST lane: its wave has no slot to finalise unless the set is half full, so nothing of its own
        // (a wait the compiler puts in front of one of its memory operations would wait for the atomic too) stands between the question
        // and the answer; the other waves finalise meanwhile.
        unsigned long long ticket = 0;
        if (gtid == GS - 1 && c0 == 0) {
            unsigned long long* const addr = &ctr->n_export;
            const unsigned long long add = (unsigned long long)n_live;
            asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0" : "=v"(ticket) : "v"(addr), "v"(add) : "memory");
        }
        const unsigned int i = c0 + (unsigned int)gtid;
        const bool have = i < n_live;
        if (have) {
            const int si = live_list[i];
            unsigned int cl[4], cr[4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const unsigned int lw = set.cnt[c][si], rw2 = set.cnt[2 + c][si];
                cl[2 * c] = lw & 0xFFFFu; cl[2 * c + 1] = lw >> 16;
                cr[2 * c] = rw2 & 0xFFFFu; cr[2 * c + 1] = rw2 >> 16;
            }
            const unsigned int puts = cl[0] + cl[1] + cl[2] + cl[3] + set.cnt[4][si];
            unsigned long long kws[KW];
#pragma unroll
            for (int w = 0; w < KW; w++) kws[w] = set.key[w][si];
            const unsigned long long first = set.ord[si];
            // wipe the slot: the four-word flavour's claim is one compare-and-swap on word 0, whose winner writes words 1 .. 3 before anybody
            // reads them (lds_put) -- an empty word 0 is all its next attempt needs; the two-word flavour claims word by word
#pragma unroll
            for (int w = 0; w < (E2Cfg<NW>::RAW ? 1 : KW); w++) set.key[w][si] = L_EMPTY;
            set.ord[si] = L_EMPTY;
#pragma unroll
            for (int q = 0; q < 5; q++) set.cnt[q][si] = 0;
            Kmer<NW> key;
            if constexpr (E2Cfg<NW>::RAW) {
#pragma unroll
                for (int w = 0; w < NW; w++) key.w[w] = kws[w];
            } else {
                Key63<NW> k63;
#pragma unroll
                for (int w = 0; w < KW; w++) k63.w[w] = kws[w];
                key = kmer_from_key63<NW>(k63);
            }
            uint32_t A = min(puts, 255u) << 24, B = puts == 1 ? B_SINGLE : 0u;
            int nin = 0, nout = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {                                       // saturate, then thread_delow + thread_mark
                uint32_t l = min(cl[c], 63u), r = min(cr[c], 63u);
                if (D > 0 && l <= (uint32_t)D) l = 0;
                if (D > 0 && r <= (uint32_t)D) r = 0;
                A |= l << (6 * c); B |= r << (6 * c);
                nin += l > 0; nout += r > 0;
            }
            if (D > 0 && nin == 0 && nout == 0) B |= B_DELETED;
            if (nin == 1 && nout == 1) B |= B_LINEAR;
            const uint32_t cov = A >> 24;
            // (round 6 tried the CRC as the xor of one entry per NIBBLE -- kmer_crc32_nibbles: no look-up waits for another, 16-entry tables never
            //  collide -- against this chain of 2 NW rounds: 163.4 ms against 124.7 at K = 127, 142.7 against 135.8 at K = 63,
            //  profiles/r06_k2_crc_nibbles_ab.json.  The emit is bound by the LDS bytes a key moves and the instructions it issues, not by the chain.)
            const uint32_t sid = set_of_crc(kmer_crc32_sliced<NW>(key, crc_tab), sp.P, sp.bias);
            // coverage histogram: most nodes of a partition share one or two coverage values (1 for error k-mers), so count
            // those per wave instead of hammering one LDS word
            const unsigned long long ones = __ballot(cov == 1);
            if (lane == __ffsll((long long)ones) - 1) atomicAdd(&hist[1], (unsigned int)__popcll(ones));
            if (cov > 1) atomicAdd(&hist[cov], 1u);
#pragma unroll
            for (int w = 0; w < NW; w++) fin.w[w] = key.w[w];
            fin.w[NW] = (uint64_t)A | ((uint64_t)B << 32);
            fin.w[NW + 1] = ((uint64_t)sid << PG_ORD_BITS) | (first & PG_ORD_MASK);
        }
        if (gtid == GS - 1 && c0 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");
            out_base = ticket;
        }
        return have;
    };
    auto e_store = [&](int gtid, unsigned int c0, unsigned int n_live, bool have, const Fin& fin) {
        const unsigned long long ob = out_base;
        if (ob + n_live <= e.out_capacity) {
            if (have) {
                ulonglong2* dst = (ulonglong2*)(e.out + (ob + c0 + (unsigned int)gtid) * (NW + 2));      // (NW + 2) * 8 is a multiple of 16
#pragma unroll
                for (int q = 0; q < (NW + 2) / 2; q++) dst[q] = make_ulonglong2(fin.w[2 * q], fin.w[2 * q + 1]);
            }
        } else if (gtid == 0 && c0 == 0) atomicOr(&ctr->e2_flags, F_OUT);
    };
    const std::integral_constant<int, THREADS> whole{};

    // (the chunk list of a partition: its first `direct` chunks lie at computed addresses, the others come from a table.  Only the
    //  table entry is asked for ahead of time -- into a register nothing else writes: were the computed id and the loaded one the
    //  same register, the compiler would make the computing lanes wait for every memory operation in flight before they may overwrite
    //  it, the previous partition's export stores included, at the top of every partition)
    auto chunk_ask = [&](uint32_t pid, uint32_t ci) -> uint32_t {
        uint32_t c2 = ci;
        asm volatile("" : "+v"(c2));                                        // (or the lane's address in the table -- or its index -- is computed at kernel start, kept, spilled, and the
                                                                            //  reload waits, s_waitcnt vmcnt(0), for every store in flight at the top of every partition)
        // always a load, never "0 or a load": a register that is written before it is loaded to makes the write wait for whatever the compiler thinks
        // may still be coming into it.  (Callers ask with ci < direct + maxc; a lane whose chunk lies at a computed address reads entry 0 and ignores it.)
        const uint32_t rel = c2 >= e.direct ? min(c2 - e.direct, e.maxc - 1u) : 0u;
        return e.chunk_tbl[(uint64_t)pid * e.maxc + rel];
    };
    auto chunk_take = [&](uint32_t pid, uint32_t ci, uint32_t asked) -> uint32_t {
        return ci < e.direct ? (uint32_t)((((uint64_t)ci << e.g.log2_parts) + pid) + 1) : asked;
    };
    uint32_t pf_nrec = 0, pf_cid = 0;
    // (the record count is the same for every lane, which would make it a scalar load -- and scalar loads are waited for at
    //  the very next barrier together with the LDS traffic (lgkmcnt); through a vector register it stays in flight until used)
    //  (the INDEX goes through the register: a pointer that does would lose its address space and become a flat load, which waits for everything)
    auto peek_cursor = [&](uint32_t p) { uint32_t pp = p; asm volatile("" : "+v"(pp)); return e.cursor[pp]; };
    // what was asked for the next partition is LOOKED AT in the middle of the current one (nrec_next, cid_next: behind the occurrence phase, where
    // nothing else of this wave is in flight) and carried in registers: a use at the top of the next partition would wait for the export stores
    uint32_t nrec_next = 0, cid_next = 0;
    if (blockIdx.x < parts) {
        pf_nrec = peek_cursor(blockIdx.x);
        if (threadIdx.x < nchunks) pf_cid = chunk_ask(blockIdx.x, threadIdx.x);
        nrec_next = pf_nrec;
        cid_next = chunk_take(blockIdx.x, threadIdx.x, pf_cid);
    }
    int b = 0, cl = 0;                                                    // the current window's buffer, the current partition's chunk list
    bool staged = false;                                                  // its first window is already on its way into rl2[b] (asked for by the previous emit)
    for (uint32_t pid = blockIdx.x; pid < parts; pid += gridDim.x) {
        const uint32_t nrec = nrec_next, my_cid = cid_next;
        const uint32_t nxt = pid + gridDim.x;
        if (nxt < parts) {
            pf_nrec = peek_cursor(nxt);
            if (threadIdx.x < nchunks) pf_cid = chunk_ask(nxt, threadIdx.x);
        }
        const uint32_t usable = min(nrec, nchunks * RPC);              // an overfull partition was flagged by K1
        my_records += usable;
        if (usable == 0) {                                                // (never one that was asked for)
            asm volatile("" ::: "memory");                                // (a branch, not two selects in front of it: they would wait for the loads just asked for)
            nrec_next = pf_nrec;
            cid_next = chunk_take(nxt, threadIdx.x, pf_cid);
            continue;
        }
        if (!staged) {                                                    // (the list's last readers are at least an emit's barriers back)
            if (threadIdx.x < nchunks) chunk_ids2[cl][threadIdx.x] = my_cid;
            K2_SYNC();
        }
        K2_TICK(0);
        // key ranges still to count: (mask, val) on the slot hash; the stack pointer lives in a register of every lane
        int top = 0;
        uint32_t mask = 0, val = 0;
        bool window_ready = false;                                        // rl2[b] holds window 0, prepared
        bool raw = staged;                                                // ... or window 0 as the previous partition's emit wrote it (p_put), tables cleared
        staged = false;
        for (;;) {
            // (no barrier of its own for the flag: whoever read it last did so before the barriers of the emit or of the range
            //  stack, and whoever sets it next does so behind the barriers of the prepare -- or, for a window kept from the
            //  previous range, finds the 0 that is there already)
            if (threadIdx.x == 0) aborted = 0;
            if (dirty) {                                                  // (an emit leaves the set empty: it wipes what it reads)
                for (int i = threadIdx.x; i < SLOTS; i += THREADS) {
#pragma unroll
                    for (int q = 0; q < KW; q++) set.key[q][i] = L_EMPTY;
                    set.ord[i] = L_EMPTY;
#pragma unroll
                    for (int q = 0; q < 5; q++) set.cnt[q][i] = 0;
                }
                dirty = false;
                K2_SYNC();
            }
            K2_TICK(1);
            for (uint32_t w0 = 0; w0 < usable; w0 += WIN) {
                const uint32_t wn = min((uint32_t)WIN, usable - w0);
                if (!(window_ready && w0 == 0)) {
                    Prep ps;
                    if (!raw) p_stage(whole, threadIdx.x, b, cl, w0, wn);   // (raw: p_put wrote the window and cleared the tables)
                    raw = false;
                    K2_SYNC();
                    K2_TICK(2);
                    p_dedupe(threadIdx.x, b, wn, ps);
                    // (a lane knows whether it represents its record as soon as its own probe ends, and a header's k-mer
                    //  count is the same in every copy: the wave prefix needs no barrier in front of it)
                    p_flat1(threadIdx.x, wave, b, ps);
                    K2_TICK(10);
                    K2_SYNC();                                            // dtab is dead from here, counts and minima are final
                    p_flat2(whole, threadIdx.x, wave, b, ps);
                    if (ADAPT && dd_on && dd_rec < 4096u) {                // (the same numbers in every lane)
                        dd_rec += wn; dd_rep += ps.n_rep;
                        if (dd_rec >= 4096u && dd_rep * 100u > dd_rec * dd_pct) dd_on = false;
                    }
                    K2_SYNC();
                    K2_TICK(3);
                }
                window_ready = usable <= WIN;                             // a single window stays good for the other key ranges
                const uint32_t total_occ = s_tot;
                const uint32_t* const rl = rl2[b];
                if (!__hip_atomic_load(&aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                  // tiles of 64 occurrences: the wave's first one is its own number, the next ones come off the counter
                  for (uint32_t tile = (uint32_t)wave;;) {
                    if (tile * 64u >= total_occ || __hip_atomic_load(&aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    const uint32_t idx = tile * 64u + (uint32_t)lane;
                    if (idx < total_occ) {
                        // the representative running at the tile's first occurrence + the starts among the tile's occurrences 1 .. lane
                        const unsigned long long tb = *(const unsigned long long*)(sbits + 2 * tile);
                        const unsigned long long upto = lane == 63 ? ~0ULL : (2ULL << lane) - 1ULL;
                        const uint32_t k = (uint32_t)tile_rep0[tile] + (uint32_t)__popcll(tb & upto & ~1ULL);
                        const uint32_t nk = noff[k], nk1 = noff[k + 1];        // idx lies inside representative k
                        do {
                            const uint32_t o_lo = nk & 0xFFFFu, o_hi = nk1 & 0xFFFFu;
                            const uint32_t* rec = rl + PAD + ((nk >> 16) & 0x1FFu) * RD;
                            const uint32_t hl = (nk >> 26) & 1u, hr = (nk >> 25) & 1u, n = o_hi - o_lo;
                            const uint32_t t = idx - o_lo;
                            uint32_t f[N2], rc[N2], prev, next;
                            if constexpr (KS != 0) {
                                constexpr OccConst ock = occ_const(KS, NW);
                                occ_extract<NW>(rec + 2, (int)(hl + t), KS, ock, f, rc, prev, next);
                            } else occ_extract<NW>(rec + 2, (int)(hl + t), K, oc, f, rc, prev, next);
                            const bool lt = occ_less<N2>(f, rc);
                            const bool hasprev = (hl + t) != 0, hasnext = t + 1 < n + hr;
                            const uint32_t pv = hasprev ? prev : 4u, nx = hasnext ? next : 4u;          // x ^ 2 keeps "none" (bit 2) set
                            const uint32_t left = lt ? pv : (nx ^ 2u), right = lt ? nx : (pv ^ 2u);
                            uint32_t c[N2];
#pragma unroll
                            for (int q = 0; q < N2; q++) c[q] = lt ? f[q] : rc[q];
                            const uint32_t hh = occ_hash<N2>(c);
                            if (((hh >> 11) & mask) != val) break;                  // (another key range's occurrence)
                            uint64_t kw[KW];
                            if constexpr (E2Cfg<NW>::RAW) {
#pragma unroll
                                for (int q = 0; q < KW; q++) kw[q] = ((uint64_t)c[2 * q] << 32) | c[2 * q + 1];
                            } else occ_key63<NW>(c, kw);
                            const uint32_t h_lo = rec[0], h_hi = rec[1];
                            const uint32_t copies = ((h_lo >> 9) & 0x1FFu) + 1u;
                            const uint64_t ord = ((((uint64_t)h_hi << 32) | h_lo) >> SKM_ORD_SHIFT) + t;
#ifdef PG_K2_LOOK
                            constexpr int LOOK = PG_K2_LOOK;
#else
                            constexpr int LOOK = KS == 31 ? 3 : 2;
#endif
                            if (!lds_put<NW, SLOTS, LOOK>(set, kw, hh, left, right, ord, copies)) aborted = 1;
                        } while (0);
                    }
                    unsigned int nt = 0;
                    if (lane == 0) nt = atomicAdd(&tile_ctr, 1u);
                    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)nt);
                  }
                }
                K2_TICK(4);
                // the next partition's chunk list, for the prepare that runs beside this partition's last emit
                {
                    uint32_t asked = pf_cid;
                    asm volatile("" : "+v"(asked));                          // (looked at here, by every lane: the register is known to be at rest when the top of the next
                    if (threadIdx.x < nchunks)                               //  partition writes to it again -- or that write waits for the export stores)
                        chunk_ids2[cl ^ 1][threadIdx.x] = cid_next = chunk_take(nxt, threadIdx.x, asked);
                }
                {
                    uint32_t pfn = pf_nrec;
                    asm volatile("" : "+v"(pfn));
                    nrec_next = pfn;
                }
                if (threadIdx.x == 0) s_nlive = 0;                          // (e_list_any adds to it; its last readers are a barrier back)
                K2_SYNC();                                                  // the window and its tables are rewritten by the next one
                // (every wave is past the tile loop: the counter starts over for the next occurrence phase -- the next window,
                //  or the same window again for another key range -- which is at least one barrier away)
                if (threadIdx.x == 0) tile_ctr = NWAVE;
                if (w0 + WIN < usable) {                                    // more windows add to these counters: keep the halves small
                    for (int i = threadIdx.x; i < SLOTS; i += THREADS) {    // (they saturate at 63 / 255 in the end anyway)
#pragma unroll
                        for (int q = 0; q < 4; q++) set.cnt[q][i] = clip_halves_255(set.cnt[q][i]);
                    }
                }
                K2_TICK(5);
            }
            if (aborted) {                                                // (read behind the window loop's last barrier: the same for every lane)
                if (TIMERS && threadIdx.x == 0) tp[9 * TIMERS]++;
                dirty = true;
                // too many distinct keys for the LDS set: split this key range on the next hash bit and redo both halves
                const uint32_t bit = mask + 1;                            // masks are 2^k - 1
                if (bit >= (1u << 20) || top + 2 > 40) {
                    if (threadIdx.x == 0) atomicOr(&ctr->e2_flags, F_SPLIT);
                } else {
                    if (threadIdx.x == 0) {
                        s_mask[top] = mask | bit; s_val[top] = val;
                        s_mask[top + 1] = mask | bit; s_val[top + 1] = val | bit;
                    }
                    top += 2;
                }
            } else {
                // (the next partition's record count was asked for at the top of this partition and looked at behind the occurrence phase
                //  -- nrec_next: a use right behind the load, or at the next partition's top, makes every wave wait out a trip to HBM and the
                //  acknowledgements of the export stores with it: the "partition header" phase of round 3's profile)
                const uint32_t usable_next = nxt < parts ? min(nrec_next, nchunks * RPC) : 0u;
                // The last emit of a partition borrows its own window (dead now) and, first of all, asks for the next partition's
                // first window (p_ask: into registers; p_put writes it to the other buffer in front of the export stores).  Earlier
                // emits (more key ranges to come) borrow the other buffer, and a single-window partition keeps its prepared window
                // for the remaining ranges.
                const bool ahead = top == 0 && usable_next > 0;
                const int sb = top == 0 ? b : b ^ 1;
                const uint32_t wn_next = min((uint32_t)WIN, usable_next);
                Ahead ah;
                bool put = !ahead;
                if (ahead) { p_ask(cl ^ 1, wn_next, ah); staged = true; }
                K2_TICK(7);
                {
                    e_list_any(whole, threadIdx.x, sb);
                    K2_SYNC();
                    K2_TICK(6);
                    const unsigned int n_live = s_nlive;
                    for (unsigned int c0 = 0; c0 < n_live; c0 += THREADS) {
                        Fin fin;
                        const bool have = e_final(whole, threadIdx.x, sb, c0, n_live, fin);
                        K2_TICK(11);
                        if (c0 == 0) {
                            K2_SYNC();                                        // the export base is there for everybody
                            if (!put) { p_put(b ^ 1, wn_next, ah); put = true; }
                        }
                        e_store(threadIdx.x, c0, n_live, have, fin);
                    }
                    if (!put) p_put(b ^ 1, wn_next, ah);                    // (an emit without a node)
                    K2_TICK(8);
                }
            }
            if (top == 0) break;
            K2_SYNC();                                                    // the pushed ranges are visible
            top--;
            mask = s_mask[top];
            val = s_val[top];
        }
        if (staged) { b ^= 1; cl ^= 1; }
    }
    K2_SYNC();
    for (int i = threadIdx.x; i < 256; i += THREADS) if (hist[i]) atomicAdd(&ctr->hist[i], (unsigned long long)hist[i]);
    if (threadIdx.x == 0 && my_records) atomicAdd(&ctr->n_records, my_records);
    if (TIMERS && threadIdx.x == 0) for (int i = 0; i < 12; i++) atomicAdd(&ctr->phase[i], tp[i * TIMERS]);
#undef K2_TICK
#undef K2_SYNC
}

// ---- pass 2 through the partitions (round 6, opt-in: SOAPDENOVO2_AMD_P2_PARTITIONED=1) ---------------------------------------------------------------
// Pass 2 asks the k-mer sets for the node of every k-mer of every read: 17.6 G lookups at 200 M reads, each a random line of HBM or two, although only 1.15 G
// k-mers are distinct.  All occurrences of a k-mer meet in ONE partition -- what pass 1 is built on -- so the reads are cut into super-k-mer records once more
// (K1, as they lie on the device), and this kernel takes a partition at a time: the distinct k-mers of the partition into the LDS set (claims only), ONE lookup in
// the sets in HBM per distinct k-mer (dense waves over the live slots; the node word goes where pass 1 keeps the first ordinal), then every occurrence finds its
// k-mer's word in LDS and writes it at its ordinal: ans[ordinal - ord_base].  The reads are then threaded over their stretch of `ans` in read order
// (graph_kernels.hip: p2_thread_ordered_kernel).  A simpler sibling of skm_count_kernel: no dedupe of records (every record's ordinals are
// its own), no counters, one window buffer; a set that overflows splits its key range on a hash bit, as there.
struct AnsArg {
    unsigned long long* ans;          // [occurrences of the round] node word of every k-mer occurrence, ~0 = not in the sets
    unsigned long long ord_base;      // ordinal of ans[0]
    const uint64_t* geo;              // the sets' geometry (graph_lookup.hpp: SV_GEO words a set)
    uint32_t P, bias;
};
template <int NW, int SLOTS, int LOOK>
__device__ __forceinline__ int lds_claim(LdsSet<NW, SLOTS>& t, const uint64_t (&kw)[E2Cfg<NW>::KW], uint32_t hash) {
    constexpr int KW = E2Cfg<NW>::KW;
    uint32_t h = hash & (SLOTS - 1);
    if constexpr (E2Cfg<NW>::RAW) {
        constexpr unsigned long long L_PENDING = 1ULL << 63;
        for (int probes = 0, spins = 0; probes < E2Cfg<NW>::MAXPROBE && spins < K2_MAXSPIN;) {
            unsigned long long seen[KW];
#pragma unroll
            for (int i = 0; i < KW; i++) seen[i] = *(volatile __attribute__((address_space(3))) unsigned long long*)(&t.key[i][h]);
            unsigned long long nxt0[LOOK];
#pragma unroll
            for (int q = 0; q < LOOK; q++) nxt0[q] = *(volatile __attribute__((address_space(3))) unsigned long long*)(&t.key[0][(h + 1 + q) & (SLOTS - 1)]);
            bool mine = false, again = false;
            if (seen[0] == L_EMPTY) {
                const unsigned long long old = atomicCAS(&t.key[0][h], L_EMPTY, (unsigned long long)kw[0] | L_PENDING);
                if (old == L_EMPTY) {
#pragma unroll
                    for (int i = 1; i < KW; i++) t.key[i][h] = (unsigned long long)kw[i];
                    __hip_atomic_store(&t.key[0][h], (unsigned long long)kw[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    mine = true;
                } else again = true;
            } else if (seen[0] & L_PENDING) again = true;
            else {
                mine = true;
#pragma unroll
                for (int i = 0; i < KW; i++) mine = mine && seen[i] == kw[i];
            }
            if (mine) return (int)h;
            if (again) spins++;
            else {
                uint32_t adv = 1;
                bool run = true;
#pragma unroll
                for (int q = 0; q < LOOK; q++) { run = run && nxt0[q] != L_EMPTY && (nxt0[q] & ~L_PENDING) != kw[0]; adv += run ? 1u : 0u; }
                h = (h + adv) & (SLOTS - 1);
                probes += (int)adv;
            }
        }
        return -1;
    }
    for (int probes = 0; probes < E2Cfg<NW>::MAXPROBE; probes++) {
        unsigned long long seen[KW];
#pragma unroll
        for (int i = 0; i < KW; i++) seen[i] = t.key[i][h];
        unsigned long long nxt0[LOOK];
#pragma unroll
        for (int q = 0; q < LOOK; q++) nxt0[q] = t.key[0][(h + 1 + q) & (SLOTS - 1)];
        bool mine = true;
#pragma unroll
        for (int i = 0; i < KW; i++) {
            if (!mine) break;
            unsigned long long cur = seen[i];
            if (cur == L_EMPTY) {
                const unsigned long long old = atomicCAS(&t.key[i][h], L_EMPTY, (unsigned long long)kw[i]);
                cur = old == L_EMPTY ? (unsigned long long)kw[i] : old;
            }
            mine = cur == kw[i];
        }
        if (mine) return (int)h;
        uint32_t adv = 1;
        bool run = true;
#pragma unroll
        for (int q = 0; q < LOOK; q++) { run = run && nxt0[q] != L_EMPTY && nxt0[q] != kw[0]; adv += run ? 1u : 0u; }
        h = (h + adv) & (SLOTS - 1);
        probes += (int)adv - 1;
    }
    return -1;
}
// the slot of a key that is in the set (every word final: the claims are a barrier back)
template <int NW, int SLOTS>
__device__ __forceinline__ int lds_find(const LdsSet<NW, SLOTS>& t, const uint64_t (&kw)[E2Cfg<NW>::KW], uint32_t hash) {
    constexpr int KW = E2Cfg<NW>::KW;
    uint32_t h = hash & (SLOTS - 1);
    for (int probes = 0; probes < SLOTS; probes++) {
        bool eq = true;
#pragma unroll
        for (int i = 0; i < KW; i++) eq = eq && t.key[i][h] == (unsigned long long)kw[i];
        if (eq) return (int)h;
        if (t.key[0][h] == L_EMPTY) return -1;
        h = (h + 1) & (SLOTS - 1);
    }
    return -1;
}

template <int NW, int SLOTS, int THREADS, int WIN, int KS>
__global__ __launch_bounds__(THREADS) void skm_answer_kernel(E2Dev e, OccConst oc, AnsArg aa, DevCounters* ctr) {
    constexpr int PW = E2Cfg<NW>::PW, RW = PW + 1, KW = E2Cfg<NW>::KW, NWAVE = THREADS / 64, PIECES = RW / 2, N2 = 2 * NW;
    constexpr int RD = 2 * RW, PAD = 16;
    constexpr int RL_WORDS = PAD + WIN * RD + 8;
    constexpr int NMAX = KS ? (32 * PW - (KS - 1) - 2 < 127 ? 32 * PW - (KS - 1) - 2 : 127) : 127;
    constexpr int SB_WORDS = (WIN * NMAX + 31) / 32 + 2;
    constexpr int SB_AT = (WIN + 1 + WIN + 1) & ~1;
    constexpr int LOOK = KS == 31 ? 3 : 2;
    __shared__ LdsSet<NW, SLOTS> set;                                     // key words; `ord` holds the node word from the lookup phase on
    __shared__ __align__(16) uint32_t rl[RL_WORDS];
    __shared__ __align__(8) unsigned int fl_raw[SB_AT + SB_WORDS];
    unsigned int* const noff = fl_raw;
    unsigned short* const tile_rep0 = (unsigned short*)(fl_raw + WIN + 1);
    unsigned int* const sbits = fl_raw + SB_AT;
    __shared__ unsigned short live_list[SLOTS];
    __shared__ uint32_t crc_tab[4 * 256];
    __shared__ unsigned int tile_ctr, aborted, s_mask[40], s_val[40], wave_cnt_f[NWAVE], s_nlive, s_tot, chunk_ids[256];
    const uint32_t nchunks = e.direct + e.maxc;
    for (int i = threadIdx.x; i < 1024; i += THREADS) crc_tab[i] = crc32_slice_entry(i >> 8, i & 255);
    if (threadIdx.x < PAD) rl[threadIdx.x] = 0;
    if (threadIdx.x < 8) rl[PAD + WIN * RD + threadIdx.x] = 0;
    const int K = KS ? KS : e.g.K;
    const uint32_t RPC = e.rpc, RPC_LOG2 = e.rpc_log2;
    const uint64_t RS = (uint64_t)e.rs;
    const uint32_t parts = 1u << e.g.log2_parts;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long lost = 0;
    __syncthreads();
    // stage window [w0, w0 + wn) of the partition and build its tables: occurrence idx -> (record, t)
    auto prepare = [&](uint32_t w0, uint32_t wn) {
        for (uint32_t pc = threadIdx.x; pc < wn * PIECES; pc += THREADS) {
            const uint32_t ri = pc / PIECES, part = pc - ri * PIECES;
            const uint32_t gi = w0 + ri, cid = chunk_ids[gi >> RPC_LOG2];
            ulonglong2 v = make_ulonglong2(0, 0);
            if (cid != 0 && cid != 0xFFFFFFFFu) v = ((const ulonglong2*)(e.pool + ((uint64_t)(cid - 1) * RPC + (gi & (RPC - 1))) * RS))[part];
            const uint32_t x0 = part ? (uint32_t)(v.x >> 32) : (uint32_t)v.x, x1 = part ? (uint32_t)v.x : (uint32_t)(v.x >> 32);
            ((uint4*)(rl + PAD))[pc] = make_uint4(x0, x1, (uint32_t)(v.y >> 32), (uint32_t)v.y);
        }
        for (int i = threadIdx.x; i < SB_WORDS; i += THREADS) sbits[i] = 0;
        __syncthreads();
        const bool have = threadIdx.x < wn;
        const unsigned int n = have ? ((rl[PAD + threadIdx.x * RD] >> 2) & 0x7Fu) : 0u;
        const unsigned int incl = wave_inclusive_sum(n);
        if (lane == 63) wave_cnt_f[wave] = incl;
        __syncthreads();
        unsigned int base = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < NWAVE; wv++) { const unsigned int cw = wave_cnt_f[wv]; if (wv < wave) base += cw; tot += cw; }
        if (have) {
            const uint32_t h0 = rl[PAD + threadIdx.x * RD];
            const unsigned int o_hi = base + incl, o_lo = o_hi - n;
            noff[threadIdx.x] = o_lo | ((unsigned int)threadIdx.x << 16) | ((h0 & 3u) << 25);
            if (n) {
                atomicOr(&sbits[o_lo >> 5], 1u << (o_lo & 31u));
                for (unsigned int t = (o_lo + 63u) >> 6; (t << 6) < o_hi; t++) tile_rep0[t] = (unsigned short)threadIdx.x;
            }
        }
        if (threadIdx.x == 0) { noff[wn] = tot; s_tot = tot; tile_ctr = NWAVE; }
        __syncthreads();
    };
    // the occurrences of the prepared window, 64 at a time: f(key words, slot hash, ordinal)
    auto for_occurrences = [&](uint32_t mask, uint32_t val, auto&& f) {
        const uint32_t total_occ = s_tot;
        for (uint32_t tile = (uint32_t)wave;;) {
            if (tile * 64u >= total_occ) break;
            const uint32_t idx = tile * 64u + (uint32_t)lane;
            if (idx < total_occ) {
                const unsigned long long tb = *(const unsigned long long*)(sbits + 2 * tile);
                const unsigned long long upto = lane == 63 ? ~0ULL : (2ULL << lane) - 1ULL;
                // (every record has at least one k-mer, hence exactly one start bit: the record running at the tile's first occurrence + the starts behind it)
                const uint32_t k = (uint32_t)tile_rep0[tile] + (uint32_t)__popcll(tb & upto & ~1ULL);
                const uint32_t nk = noff[k], nk1 = noff[k + 1];
                const uint32_t o_lo = nk & 0xFFFFu, o_hi = nk1 & 0xFFFFu;
                const uint32_t* rec = rl + PAD + ((nk >> 16) & 0x1FFu) * RD;
                const uint32_t hl = (nk >> 26) & 1u;
                const uint32_t t = idx - o_lo;
                (void)o_hi;
                uint32_t fw[N2], rc[N2], prev, next;
                if constexpr (KS != 0) {
                    constexpr OccConst ock = occ_const(KS, NW);
                    occ_extract<NW>(rec + 2, (int)(hl + t), KS, ock, fw, rc, prev, next);
                } else occ_extract<NW>(rec + 2, (int)(hl + t), K, oc, fw, rc, prev, next);
                const bool lt = occ_less<N2>(fw, rc);
                uint32_t c[N2];
#pragma unroll
                for (int q = 0; q < N2; q++) c[q] = lt ? fw[q] : rc[q];
                const uint32_t hh = occ_hash<N2>(c);
                if (((hh >> 11) & mask) == val) {
                    uint64_t kw[KW];
                    if constexpr (E2Cfg<NW>::RAW) {
#pragma unroll
                        for (int q = 0; q < KW; q++) kw[q] = ((uint64_t)c[2 * q] << 32) | c[2 * q + 1];
                    } else occ_key63<NW>(c, kw);
                    const uint64_t ord = ((((uint64_t)rec[1] << 32) | rec[0]) >> SKM_ORD_SHIFT) + t;
                    f(kw, hh, ord);
                }
            }
            unsigned int nt = 0;
            if (lane == 0) nt = atomicAdd(&tile_ctr, 1u);
            tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)nt);
        }
    };
    for (uint32_t pid = blockIdx.x; pid < parts; pid += gridDim.x) {
        const uint32_t usable = min(e.cursor[pid], nchunks * RPC);
        if (usable == 0) continue;
        __syncthreads();                                                  // (the previous partition's last readers of the chunk list and the tables)
        if (threadIdx.x < nchunks) chunk_ids[threadIdx.x] = chunk_id_of(e, pid, threadIdx.x);
        int top = 0;
        uint32_t mask = 0, val = 0;
        for (;;) {
            for (int i = threadIdx.x; i < SLOTS; i += THREADS) {
#pragma unroll
                for (int q = 0; q < KW; q++) set.key[q][i] = L_EMPTY;
                set.ord[i] = L_EMPTY;
            }
            if (threadIdx.x == 0) { aborted = 0; s_nlive = 0; }
            __syncthreads();
            // A: the distinct k-mers of the key range into the set
            for (uint32_t w0 = 0; w0 < usable; w0 += WIN) {
                prepare(w0, min((uint32_t)WIN, usable - w0));
                for_occurrences(mask, val, [&](const uint64_t (&kw)[KW], uint32_t hh, uint64_t) {
                    if (__hip_atomic_load(&aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
                    if (lds_claim<NW, SLOTS, LOOK>(set, kw, hh) < 0) aborted = 1;
                });
                __syncthreads();
            }
            if (aborted) {                                                // too many distinct keys: split the range on the next hash bit
                const uint32_t bit = mask + 1;
                if (bit >= (1u << 20) || top + 2 > 40) { if (threadIdx.x == 0) atomicOr(&ctr->e2_flags, F_SPLIT); }
                else {
                    if (threadIdx.x == 0) { s_mask[top] = mask | bit; s_val[top] = val; s_mask[top + 1] = mask | bit; s_val[top + 1] = val | bit; }
                    top += 2;
                }
            } else {
                // B: one lookup in the sets per distinct k-mer, dense waves over the live slots
                for (int st = 0; st < SLOTS / THREADS; st++) {
                    const int si = st * THREADS + threadIdx.x;
                    const bool live = set.key[0][si] != L_EMPTY;
                    const unsigned long long bal = __ballot(live);
                    unsigned int basep = 0;
                    if (lane == 0 && bal) basep = atomicAdd(&s_nlive, (unsigned int)__popcll(bal));
                    basep = (unsigned int)__builtin_amdgcn_readfirstlane((int)basep);
                    if (live) live_list[basep + (unsigned int)__popcll(bal & ((1ULL << lane) - 1))] = (unsigned short)si;
                }
                __syncthreads();
                const unsigned int n_live = s_nlive;
                for (unsigned int i = threadIdx.x; i < n_live; i += THREADS) {
                    const int si = live_list[i];
                    Kmer<NW> key;
                    if constexpr (E2Cfg<NW>::RAW) {
#pragma unroll
                        for (int w = 0; w < NW; w++) key.w[w] = set.key[w][si];
                    } else {
                        Key63<NW> k63;
#pragma unroll
                        for (int w = 0; w < KW; w++) k63.w[w] = set.key[w][si];
                        key = kmer_from_key63<NW>(k63);
                    }
                    const uint32_t sidx = set_of_crc(kmer_crc32_sliced<NW>(key, crc_tab), aa.P, aa.bias);
                    const uint64_t size = aa.geo[SV_GEO * sidx + 1];
                    const uint64_t* base = (const uint64_t*)(uintptr_t)aa.geo[SV_GEO * sidx + 2];
                    uint64_t hc = home_slot<NW>(key, ModConst{size, aa.geo[SV_GEO * sidx + 3], (uint32_t)aa.geo[SV_GEO * sidx + 4]});
                    unsigned long long ab = ~0ULL;
                    for (uint64_t step = 0; step < size; step++) {
                        const uint64_t* nd = base + hc * (NW + 1);
                        uint64_t dd[NW + 1];
#pragma unroll
                        for (int q = 0; q <= NW; q++) dd[q] = sv_word(nd + q);
                        if (dd[0] == SV_EMPTY) break;
                        bool eq = true;
#pragma unroll
                        for (int q = 0; q < NW; q++) eq = eq && dd[q] == key.w[q];
                        if (eq) { ab = dd[NW]; break; }
                        if (++hc == size) hc = 0;
                    }
                    set.ord[si] = ab;
                }
                __syncthreads();
                // C: every occurrence of the range takes its k-mer's word to its ordinal
                for (uint32_t w0 = 0; w0 < usable; w0 += WIN) {
                    if (usable > WIN) prepare(w0, min((uint32_t)WIN, usable - w0));
                    else { if (threadIdx.x == 0) tile_ctr = NWAVE; __syncthreads(); }      // (a single window is still there, with its tables)
                    for_occurrences(mask, val, [&](const uint64_t (&kw)[KW], uint32_t hh, uint64_t ord) {
                        const int si = lds_find<NW, SLOTS>(set, kw, hh);
                        if (si < 0) { lost++; return; }
                        aa.ans[ord - aa.ord_base] = set.ord[si];
                    });
                    __syncthreads();
                }
            }
            if (top == 0) break;
            __syncthreads();
            top--;
            mask = s_mask[top];
            val = s_val[top];
        }
    }
    if (lost) atomicAdd(&ctr->overflow, lost);
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
        const uint32_t usable = min(e.cursor[pid], (e.direct + e.maxc) * e.rpc);
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

// distinct k-mers per reference set, from the export array (the set id is the top byte of a record's last word)
__global__ __launch_bounds__(BLOCK) void set_count_kernel(const uint64_t* out, uint64_t n, int rw, DevCounters* ctr) {
    __shared__ unsigned int cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        atomicAdd(&cnt[out[i * rw + rw - 1] >> PG_ORD_BITS], 1u);
    __syncthreads();
    if (cnt[threadIdx.x]) atomicAdd(&ctr->set_last[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
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
    uint32_t lg = 0; while ((1u << lg) < s.rpc) lg++;
    return E2Dev{s.g, s.rpc, s.maxc, lg, s.rs, s.direct, s.pool_chunks, s.cursor, s.chunk_tbl, s.pool, s.out, s.out_capacity, (uint32_t)std::max(1, c->n_owners)};
}

int e2_create(pg_ctx* c) {
    E2& s = c->e2;
    // every size comes from ONE place, e2_plan (e2_plan.hpp): the executable memory plan (pg_host_plan_memory) calls the same function
    size_t free_b = 0, total_b = 0;
    E2_TRY(pg::arena_mem_info(&free_b, &total_b));
    uint32_t rpc = 128;
    int rs_o = 0, direct_o = -1, test_lp = -1;
    uint64_t pool_mb = 0;
    if (const char* v = env_measure("PG_RPC")) { const int q = atoi(v); if (q == 16 || q == 32 || q == 64 || q == 128) rpc = (uint32_t)q; }
    // record slots: PG_REC_STRIDE=8 gives every 48-byte record of the two-word flavour its own 64-byte line (whole-line writes
    // in K1; the four-word flavour's records are 64 bytes anyway)
    if (const char* v = env_measure("PG_REC_STRIDE")) rs_o = atoi(v);
    // PG_DIRECT_CHUNKS=M: the first M chunks of every partition at computed addresses (M * parts chunks set aside up front)
    if (const char* v = env_measure("PG_DIRECT_CHUNKS")) { const int e = atoi(v); if (e >= 0) direct_o = e; }
    if (const char* v = env_measure("PG_POOL_MB")) pool_mb = (uint64_t)atoll(v);
    if (const char* v = env_test("PG_LOG2_PARTS")) test_lp = atoi(v);
    const E2Plan pl = e2_plan(c->K, c->NW, c->log2_slots, c->hint_kmers, c->hint_reads, c->hint_log2_parts, c->hint_distinct, c->n_owners, free_b, total_b, rpc, rs_o, direct_o, pool_mb, test_lp);
    if (pl.err == 1) { pg_set_error("partition engine: export array does not fit in device memory"); return PG_ENOMEM; }
    if (pl.err == 2) { pg_set_error("partition engine: record pool too small for the partition count"); return PG_ENOMEM; }
    // Two partition counts: the ids of the job (the hash is scaled to part_mul = 2^log2_global of them) and what THIS context stores -- all of
    // them, or, as one of n_owners ranks, those with id mod n_owners == its rank at id / n_owners (skm_ingest_kernel).  Everything that
    // addresses storage -- cursors, chunk table, computed chunk addresses, the counting grid -- goes by g.log2_parts = log2_store.
    s.log2_global = pl.log2_global;
    s.log2_parts = pl.log2_store;
    s.g = skm_geometry(c->K, pl.log2_global, c->NW);
    s.g.log2_parts = pl.log2_store;
    // PG_PARTS_EFF_PCT: only that share of the partition ids is used (the hash is scaled to it): partitions between two powers of two
    if (const char* v = env_measure("PG_PARTS_EFF_PCT")) { const int pct = atoi(v); if (pct >= 50 && pct <= 100) s.g.part_mul = (uint32_t)(((uint64_t)1 << pl.log2_global) * (uint64_t)pct / 100); }
    s.rpc = pl.rpc;
    s.rs = pl.rs;
    s.direct = pl.direct;
    s.maxc = pl.maxc;
    s.pool_chunks = pl.pool_chunks;
    s.out_capacity = pl.out_capacity;
    const uint64_t parts = (uint64_t)1 << s.log2_parts;
    const uint64_t chunk_bytes = pl.chunk_bytes, out_bytes = pl.out_bytes;
    // PG_STARTUP_TRACE=1: what each step of the start-up took (stderr), for boxes on which a command's first second is not its own
    const bool trace = env_user("PG_STARTUP_TRACE") && atoi(env_user("PG_STARTUP_TRACE"));
    auto t_last = std::chrono::steady_clock::now();
    auto step = [&](const char* what, double gb) {
        if (!trace) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[ctx]   %-44s %7.3f s  (%.2f GB)\n", what, std::chrono::duration<double>(t - t_last).count(), gb);
        t_last = t;
    };
    step("geometry, hipMemGetInfo", 0.0);
    E2_TRY(pg::arena_malloc(&s.cursor, parts * sizeof(uint32_t)));
    E2_TRY(pg::arena_malloc(&s.chunk_tbl, parts * s.maxc * sizeof(uint32_t)));
    step("hipMalloc: cursors + chunk table", (double)(parts * (s.maxc + 1) * sizeof(uint32_t)) / 1e9);
    E2_TRY(pg::arena_malloc(&s.pool, s.pool_chunks * chunk_bytes + 64));
    step("hipMalloc: record pool", (double)(s.pool_chunks * chunk_bytes) / 1e9);
    s.out = nullptr; s.out_err = 0;
    if (env_measure("PG_EXPORT_ASYNC") && atoi(env_measure("PG_EXPORT_ASYNC")) == 0) E2_TRY(pg::arena_malloc(&s.out, std::max<uint64_t>(out_bytes, 64)));
    else {
        const int dev = c->device;
        const uint64_t bytes = std::max<uint64_t>(out_bytes, 64);
        E2* const sp = &s;
        s.out_thread = new std::thread([sp, dev, bytes] {
            void* p = nullptr;
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = pg::arena_malloc(&p, bytes);
            sp->out = (uint64_t*)p;
            sp->out_err = (int)e;
        });
    }
    E2_TRY(hipMemset(s.cursor, 0, parts * sizeof(uint32_t)));
    E2_TRY(hipMemset(s.chunk_tbl, 0, parts * s.maxc * sizeof(uint32_t)));
    if (trace) { (void)hipDeviceSynchronize(); step("hipMemset: cursors + chunk table (first device work of the process)", (double)(parts * (s.maxc + 1) * sizeof(uint32_t)) / 1e9); }
    s.counted = false;
    s.est_chunks = 0;
    return PG_OK;
}

// the export array is there (or could not be had)
static int e2_join_out(pg_ctx* c) {
    E2& s = c->e2;
    if (s.out_thread) { s.out_thread->join(); delete s.out_thread; s.out_thread = nullptr; }
    if (s.out_err) {
        pg_set_error(std::string("partition engine: the export array could not be allocated: ") + hipGetErrorString((hipError_t)s.out_err));
        return (hipError_t)s.out_err == hipErrorOutOfMemory ? PG_ENOMEM : PG_ENODEV;
    }
    return PG_OK;
}

void e2_destroy(pg_ctx* c) {
    E2& s = c->e2;
    (void)e2_join_out(c);
    (void)hipSetDevice(c->device);
    if (s.cursor) (void)pg::arena_free(s.cursor);
    if (s.chunk_tbl) (void)pg::arena_free(s.chunk_tbl);
    if (s.pool) (void)pg::arena_free(s.pool);
    if (s.out) (void)pg::arena_free(s.out);
    if (s.rg_perm) (void)pg::arena_free(s.rg_perm);
    if (s.rg_hist) (void)pg::arena_free(s.rg_hist);
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
    // (the input's size is known and the pool was made for it -- pg_expect: a batch is taken at half as much again as expected, not at four times;
    //  at 10 M reads a batch is a fifth of the input and the larger figure grew a pool that was large enough)
    const uint64_t est_records = (c->hint_kmers ? 3 * (2 * n_kmers / (uint64_t)(s.g.w + 1) + n_reads) / 2 : 4 * (2 * n_kmers / (uint64_t)(s.g.w + 1) + n_reads)) + 64;
    s.est_chunks += est_records / s.rpc + 1;
    const uint64_t fixed = (uint64_t)s.direct * parts;                 // chunks at computed addresses: always there, never handed out
    const uint64_t need = s.est_chunks + parts + 16 + fixed;
    if (need <= s.pool_chunks) return PG_OK;
    // the estimate is loose: look at what was really handed out
    E2_TRY(hipStreamSynchronize(st));
    unsigned long long used = 0;                                      // as if every sub-pool were as full as the fullest
    {
        std::vector<unsigned long long> subs(POOL_SUBS * 8);
        E2_TRY(hipMemcpy(subs.data(), c->ctr->pool_sub, subs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < POOL_SUBS; q++) used = std::max(used, subs[q * 8]);
        used *= std::min<uint64_t>(POOL_SUBS, parts);
    }
    s.est_chunks = used + est_records / s.rpc + 1;
    const uint64_t need2 = s.est_chunks + parts + 16 + fixed;
    if (need2 <= s.pool_chunks) return PG_OK;
    const uint64_t chunk_bytes = (uint64_t)s.rs * 8 * s.rpc;
    const uint64_t fresh_chunks = std::max(need2 * 2 - fixed, s.pool_chunks * 2 - fixed);
    if (fresh_chunks >= 0xFFFFFFF0ULL) { pg_set_error("partition engine: more than 2^32 record chunks"); return PG_ENOMEM; }
    uint64_t* fresh = nullptr;
    E2_TRY(pg::arena_malloc(&fresh, fresh_chunks * chunk_bytes + 64));
    E2_TRY(hipMemcpy(fresh, s.pool, std::min<uint64_t>(fixed + used, s.pool_chunks) * chunk_bytes, hipMemcpyDeviceToDevice));
    E2_TRY(pg::arena_free(s.pool));
    s.pool = fresh;
    s.pool_chunks = fresh_chunks;
    // a longer chunk list per partition too, if the table allows (rebuild with the wider stride)
    const uint64_t even = (fresh_chunks + parts - 1) / parts;
    const uint32_t want = (uint32_t)std::max<uint64_t>(s.maxc, std::min<uint64_t>(std::min<uint64_t>(256 - s.direct, ((uint64_t)1 << 29) / parts), even * 16));
    if (want > s.maxc) {
        uint32_t* tbl = nullptr;
        E2_TRY(pg::arena_malloc(&tbl, parts * want * sizeof(uint32_t)));
        E2_TRY(hipMemset(tbl, 0, parts * want * sizeof(uint32_t)));
        E2_TRY(hipMemcpy2D(tbl, want * sizeof(uint32_t), s.chunk_tbl, s.maxc * sizeof(uint32_t), s.maxc * sizeof(uint32_t), parts, hipMemcpyDeviceToDevice));
        E2_TRY(pg::arena_free(s.chunk_tbl));
        s.chunk_tbl = tbl;
        s.maxc = want;
    }
    return PG_OK;
}

template <int NW, bool ROUTE, int W, bool RG>
static void launch_seg_s(int S, dim3 grid, size_t smem, hipStream_t st, const ReadsArg& a, const E2Dev& e, DevCounters* ctr, const SegArg& sa, const RouteArg& ro) {
    switch (S) {
        case 7: hipLaunchKernelGGL((skm_scatter_seg_kernel<NW, ROUTE, 7, W, RG>), grid, dim3(BLOCK), smem, st, a, e, ctr, sa, ro); break;
        case 9: hipLaunchKernelGGL((skm_scatter_seg_kernel<NW, ROUTE, 9, W, RG>), grid, dim3(BLOCK), smem, st, a, e, ctr, sa, ro); break;
        case 11: hipLaunchKernelGGL((skm_scatter_seg_kernel<NW, ROUTE, 11, W, RG>), grid, dim3(BLOCK), smem, st, a, e, ctr, sa, ro); break;
        case 13: hipLaunchKernelGGL((skm_scatter_seg_kernel<NW, ROUTE, 13, W, RG>), grid, dim3(BLOCK), smem, st, a, e, ctr, sa, ro); break;
        default: hipLaunchKernelGGL((skm_scatter_seg_kernel<NW, ROUTE, 15, W, RG>), grid, dim3(BLOCK), smem, st, a, e, ctr, sa, ro); break;
    }
}
// the instantiations with the window length at compile time: K = 63 (48 16-mers a k-mer) and K = 31 (16) in the two-word flavour,
// K = 127 (112) in the four-word one; every other geometry runs the general kernel (PG_K1_W=0: always)
template <int NW, bool ROUTE, bool RG>
static void launch_seg(int S, int m, int w, dim3 grid, size_t smem, hipStream_t st, const ReadsArg& a, const E2Dev& e, DevCounters* ctr, const SegArg& sa, const RouteArg& ro) {
    bool fixed = m == 16 && S <= w;
    if (const char* v = env_measure("PG_K1_W")) fixed = fixed && atoi(v) != 0;
    if (fixed && NW == 2 && w == 48) launch_seg_s<NW, ROUTE, 48, RG>(S, grid, smem, st, a, e, ctr, sa, ro);
    else if (fixed && NW == 2 && w == 16) launch_seg_s<NW, ROUTE, 16, RG>(S, grid, smem, st, a, e, ctr, sa, ro);
    else if (fixed && NW == 4 && w == 112) launch_seg_s<NW, ROUTE, 112, RG>(S, grid, smem, st, a, e, ctr, sa, ro);
    else launch_seg_s<NW, ROUTE, 0, RG>(S, grid, smem, st, a, e, ctr, sa, ro);
}
template <int NW, bool ROUTE>
static void launch_seg_rg(bool ragged, int S, int m, int w, dim3 grid, size_t smem, hipStream_t st, const ReadsArg& a, const E2Dev& e, DevCounters* ctr, const SegArg& sa, const RouteArg& ro) {
    if (ragged) launch_seg<NW, ROUTE, true>(S, m, w, grid, smem, st, a, e, ctr, sa, ro);
    else launch_seg<NW, ROUTE, false>(S, m, w, grid, smem, st, a, e, ctr, sa, ro);
}

// launch the tiled K1; returns PG_OK / an error, or 1 when the reads are too long for a tile (caller falls back)
static int launch_tiled(pg_ctx* c, const ReadsArg& a, const RouteArg* route, hipStream_t st) {
    const SkmGeom& g = c->e2.g;
    const bool ragged = a.uniform_len == 0;                                   // rows for the batch's longest read (a.kpr, a.wpr: its)
    const int kpr = (int)a.kpr, wpr = (int)a.wpr, np = (int)(ragged ? a.max_len : a.uniform_len) - g.m + 1;
    int S = tile_pick_segment(kpr, g.w);
    if (const char* v = env_measure("PG_K1_S")) { const int q = atoi(v); if (q >= 7 && q <= 15 && (q & 1) && (q <= g.w || q == 7)) S = q; }
    const int nseg = (kpr + S - 1) / S, nca = (np + 15) / 16, npad = (16 * nca) | 1, wsd = (2 * wpr + 3) | 1;   // value rows: whole 16-position chunks, odd stride
    const size_t per_read = (size_t)(wsd + npad + ((nseg * S) | 1) + (nseg | 1) + (route ? kpr : 0)) * 4 + (ragged ? 12 : 0);     // (the kernel's rows: dword string, values, partition ids, start bits, ranks; ragged: first k-mer + length)
    int R = std::min<int>(128, std::max(1, BLOCK / nseg));                     // one pass of phase B per tile
    R = (int)std::min<size_t>((size_t)R, (60 * 1024) / per_read);
    // ... and about 25 KB of LDS a tile, i.e. five workgroups a CU: measured at 150 bp (profiles/r03v_k1_tile_sizes.json), K = 63
    // (1016 B a read): 12 / 16 / 20 / 24 / 28 / 32 / 56 reads -> 84.7 / 73.6 / 65.9 / 59.5 - 61.1 / 61.6 / 66.8 - 67.3 / 108.9 ms per
    // 200 M reads; K = 127 (760 B a read): 16 / 24 / 32 / 48 / 80 -> 57.7 / 42.1 / 38.1 / 41.4 / 54.3 ms
    {
        const int r_lds = (int)(((25 * 1024) / per_read) & ~(size_t)7);
        if (r_lds >= 8) R = std::min(R, r_lds);
    }
    if (const char* v = env_measure("PG_K1_R")) R = std::max(1, std::min(atoi(v), (int)std::min<size_t>(128, (60 * 1024) / per_read)));
    if (R < 1) return 1;
    const uint64_t grid = (a.n_reads + R - 1) / R;
    if (grid > 0x7FFFFFFFULL) { pg_set_error("batch too large for one launch"); return PG_EINVAL; }
    auto inv = [](uint32_t d) { return (uint32_t)(((1ULL << 32) + d - 1) / d); };
    SegArg sa{R, np, npad, wsd, nseg, nca, inv((uint32_t)R), inv(a.wpr)};
    RouteArg ro{nullptr, nullptr, nullptr, 0, 1};
    if (route) ro = *route;
    const size_t smem = per_read * R + (ragged ? 8 : 0);                      // (+ the alignment of the 8-byte row)
    if (c->NW == 2) {
        if (route) launch_seg_rg<2, true>(ragged, S, g.m, g.w, dim3((unsigned)grid), smem, st, a, dev_view(c), c->ctr, sa, ro);
        else launch_seg_rg<2, false>(ragged, S, g.m, g.w, dim3((unsigned)grid), smem, st, a, dev_view(c), c->ctr, sa, ro);
    } else {
        if (route) launch_seg_rg<4, true>(ragged, S, g.m, g.w, dim3((unsigned)grid), smem, st, a, dev_view(c), c->ctr, sa, ro);
        else launch_seg_rg<4, false>(ragged, S, g.m, g.w, dim3((unsigned)grid), smem, st, a, dev_view(c), c->ctr, sa, ro);
    }
    E2_TRY(hipGetLastError());
    c->e2.counted = false;
    return PG_OK;
}

// multi-GPU step 1: cut a uniform batch into records grouped by owner (partition mod n_owners)
static int launch_serial(pg_ctx* c, const ReadsArg& a, const RouteArg* route, hipStream_t st) {
    const uint64_t grid = (a.n_reads + BLOCK - 1) / BLOCK;
    if (grid > 0x7FFFFFFFULL) { pg_set_error("batch too large for one launch"); return PG_EINVAL; }
    RouteArg ro{nullptr, nullptr, nullptr, 0, 1};
    if (route) ro = *route;
    if (c->NW == 2) {
        if (route) hipLaunchKernelGGL((skm_scatter_kernel<2, true>), dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr, ro);
        else hipLaunchKernelGGL((skm_scatter_kernel<2, false>), dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr, ro);
    } else {
        if (route) hipLaunchKernelGGL((skm_scatter_kernel<4, true>), dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr, ro);
        else hipLaunchKernelGGL((skm_scatter_kernel<4, false>), dim3((unsigned)grid), dim3(BLOCK), 0, st, a, dev_view(c), c->ctr, ro);
    }
    E2_TRY(hipGetLastError());
    c->e2.counted = false;
    return PG_OK;
}

// The longest read of a ragged batch sizes the rows of its tiles: the caller's bound (pg_set_read_len_bound) or, without one, a
// reduction over kmer_base and one host wait a batch.
__global__ __launch_bounds__(BLOCK) void ragged_max_kernel(const uint64_t* __restrict__ kmer_base, uint64_t n_reads, DevCounters* ctr) {
    unsigned long long mx = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * BLOCK)
        mx = max(mx, (unsigned long long)(kmer_base[r + 1] - kmer_base[r]));
    for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned long long)__shfl_xor((unsigned long long)mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&ctr->ragged_max, mx);
}
// ---- ragged batches by LENGTH CLASS.  A tile's rows are as long as its longest read needs, and a read of 100 bases in a tile sized for 150 leaves a
// third of its lanes idle in every phase: the trimmed reads of round 6's first build cost 1.32 - 1.35 x the uniform reads per BASE.  The order in which
// reads are cut does not matter to anything (a record carries its ordinal, a partition's records arrive in any order anyway), so a batch is cut class by
// class: class = segments of S k-mers a read has; a counting sort of the batch's reads by class (a histogram, one host wait for its 32 counters, a
// scatter) gathers every class's start words, first k-mers and lengths into rows of its own, and every class is a launch of its own with tiles for ITS longest read.
// (A first form handed the cutter a permutation instead: one more dependent load in front of every tile, 75.6 ms against 65.3 without classes.)
constexpr int RG_CLASSES = 32;
__global__ __launch_bounds__(BLOCK) void ragged_class_hist(const uint64_t* __restrict__ kmer_base, uint64_t n_reads, uint32_t S, unsigned int* hist) {
    __shared__ unsigned int h[RG_CLASSES];
    if (threadIdx.x < RG_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t k = kmer_base[r + 1] - kmer_base[r];
        atomicAdd(&h[max(1u, min((uint32_t)((k + S - 1) / S), (uint32_t)RG_CLASSES - 1u))], 1u);      // (a read without k-mers is flagged by the cutter: with the shortest)
    }
    __syncthreads();
    if (threadIdx.x < RG_CLASSES && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// cursor[c] starts at the class's first place in perm; a workgroup reserves its reads of a class with one atomic
__global__ __launch_bounds__(BLOCK) void ragged_class_scatter(const uint64_t* __restrict__ kmer_base, const uint64_t* __restrict__ word_off, uint64_t n_reads, uint32_t S, int K,
                                                             unsigned int* cursor, uint64_t* cls_off, uint64_t* cls_kb, int32_t* cls_len) {
    __shared__ unsigned int h[RG_CLASSES], base[RG_CLASSES];
    for (uint64_t r0 = (uint64_t)blockIdx.x * BLOCK; r0 < n_reads; r0 += (uint64_t)gridDim.x * BLOCK) {
        if (threadIdx.x < RG_CLASSES) h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t r = r0 + threadIdx.x;
        uint32_t cls = 0, my = 0;
        if (r < n_reads) {
            const uint64_t k = kmer_base[r + 1] - kmer_base[r];
            cls = max(1u, min((uint32_t)((k + S - 1) / S), (uint32_t)RG_CLASSES - 1u));
            my = atomicAdd(&h[cls], 1u);
        }
        __syncthreads();
        if (threadIdx.x < RG_CLASSES && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
        __syncthreads();
        if (r < n_reads) {                                            // the class's rows: a launch reads them as it reads a batch's (no indirection in the cutter)
            const unsigned int at = base[cls] + my;
            cls_off[at] = word_off[r];
            cls_kb[at] = kmer_base[r];
            cls_len[at] = (int32_t)(kmer_base[r + 1] - kmer_base[r]) + K - 1;
        }
        __syncthreads();
    }
}

// (PG_K1_RAGGED=0 in a -DPG_MEASURE build: ragged batches through the one-lane-a-read kernel, the round-5 form, for the A/B)
static bool ragged_tiles_wanted() {
    if (const char* v = env_measure("PG_K1_RAGGED")) return atoi(v) != 0;
    return true;
}
static int ragged_geometry(pg_ctx* c, ReadsArg& a, hipStream_t st) {
    uint32_t L = c->read_len_bound;
    if (!L) {
        E2_TRY(hipMemsetAsync(&c->ctr->ragged_max, 0, sizeof(unsigned long long), st));
        const unsigned grid = (unsigned)std::min<uint64_t>((a.n_reads + BLOCK - 1) / BLOCK, 2048);
        hipLaunchKernelGGL(ragged_max_kernel, dim3(grid), dim3(BLOCK), 0, st, a.kmer_base, a.n_reads, c->ctr);
        E2_TRY(hipGetLastError());
        unsigned long long mk = 0;
        E2_TRY(hipMemcpyAsync(&mk, &c->ctr->ragged_max, sizeof(mk), hipMemcpyDeviceToHost, st));
        E2_TRY(hipStreamSynchronize(st));
        L = (uint32_t)std::min<unsigned long long>(mk + (unsigned long long)c->K - 1, 0xFFFFFFFFull);
    }
    a.max_len = L;
    if (L >= (uint32_t)c->K + 1 && L < 4096) { a.kpr = L - c->K + 1; a.wpr = (L + 31) / 32; }
    else a.max_len = 0;                                          // (too long for a tile: the one-lane-a-read kernel)
    return PG_OK;
}

// a ragged batch through the tiles, class by class (see ragged_class_hist); returns what launch_tiled returns
static int launch_ragged_by_class(pg_ctx* c, const ReadsArg& a, const RouteArg* route, hipStream_t st) {
    E2& s = c->e2;
    // Built and measured in round 6, and NOT the product's form (PG_K1_CLASSES=1 in a -DPG_MEASURE build runs it): 200 M reads trimmed to 100 - 150 bp, K1 per pass
    // 65.1 ms as the batches lie, 67.9 ms class by class with gathered rows, 75.6 ms through a permutation (profiles/r06_k1_ragged_length_classes_ab.json) --
    // the idle lanes of a padded tile are not what a trimmed batch pays for; the sort, its host wait and five launches a batch cost more than they save.
    bool by_class = false;
    if (const char* v = env_measure("PG_K1_CLASSES")) by_class = atoi(v) != 0 && a.n_reads >= 4096 && a.n_reads < 0xFFFFFFFFull;
    if (!by_class) return launch_tiled(c, a, route, st);
    int S = tile_pick_segment((int)a.kpr, s.g.w);
    if (const char* v = env_measure("PG_K1_S")) { const int q = atoi(v); if (q >= 7 && q <= 15 && (q & 1) && (q <= s.g.w || q == 7)) S = q; }
    if (!s.rg_hist) E2_TRY(pg::arena_malloc(&s.rg_hist, 2 * RG_CLASSES * sizeof(unsigned int)));
    E2_TRY(hipMemsetAsync(s.rg_hist, 0, RG_CLASSES * sizeof(unsigned int), st));
    const unsigned grid = (unsigned)std::min<uint64_t>((a.n_reads + BLOCK - 1) / BLOCK, 4096);
    hipLaunchKernelGGL(ragged_class_hist, dim3(grid), dim3(BLOCK), 0, st, a.kmer_base, a.n_reads, (uint32_t)S, s.rg_hist);
    E2_TRY(hipGetLastError());
    unsigned int h[RG_CLASSES], first[RG_CLASSES];
    E2_TRY(hipMemcpyAsync(h, s.rg_hist, sizeof h, hipMemcpyDeviceToHost, st));
    E2_TRY(hipStreamSynchronize(st));
    int used = 0;
    for (int q = 0; q < RG_CLASSES; q++) used += h[q] != 0;
    if (used <= 1) return launch_tiled(c, a, route, st);               // one class: the batch as it lies
    if (a.n_reads > s.rg_perm_cap) {                                   // 20 bytes a read: start word, first k-mer, length
        if (s.rg_perm) E2_TRY(pg::arena_free(s.rg_perm));
        s.rg_perm = nullptr;
        s.rg_perm_cap = a.n_reads + a.n_reads / 4;
        E2_TRY(pg::arena_malloc(&s.rg_perm, s.rg_perm_cap * 20));
    }
    uint64_t* const cls_off = (uint64_t*)s.rg_perm;
    uint64_t* const cls_kb = cls_off + s.rg_perm_cap;
    int32_t* const cls_len = (int32_t*)(cls_kb + s.rg_perm_cap);
    unsigned int at = 0;
    for (int q = 0; q < RG_CLASSES; q++) { first[q] = at; at += h[q]; }
    E2_TRY(hipMemcpyAsync(s.rg_hist + RG_CLASSES, first, sizeof first, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ragged_class_scatter, dim3(grid), dim3(BLOCK), 0, st, a.kmer_base, a.word_off, a.n_reads, (uint32_t)S, c->K, s.rg_hist + RG_CLASSES, cls_off, cls_kb, cls_len);
    E2_TRY(hipGetLastError());
    E2_TRY(hipStreamSynchronize(st));                                  // (`first` is on this stack frame)
    for (int q = RG_CLASSES - 1; q >= 1; q--) {                        // the longest first: if ITS tile does not fit, nothing has been launched yet
        if (!h[q]) continue;
        ReadsArg b = a;
        const uint32_t len_q = q == RG_CLASSES - 1 ? a.max_len : std::min<uint32_t>(a.max_len, (uint32_t)q * (uint32_t)S + (uint32_t)c->K - 1);
        b.word_off = cls_off + first[q]; b.cls_kb = cls_kb + first[q]; b.cls_len = cls_len + first[q];
        b.n_reads = h[q];
        b.max_len = len_q; b.kpr = len_q - c->K + 1; b.wpr = (len_q + 31) / 32;
        const int rc = launch_tiled(c, b, route, st);
        if (rc) return rc;
    }
    return PG_OK;
}

// multi-GPU step 1: cut a batch into records grouped by owner (partition mod n_owners).  Uniform batches go through the
// tiled kernel, ragged ones (d_word_off / d_kmer_base given) through the one-lane-per-read kernel.
int e2_route(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base, uint64_t n_reads,
             uint32_t uniform_len, uint64_t ord_base, int n_owners, uint64_t* d_recs, uint32_t* d_pids, uint64_t cap, uint64_t* d_counts,
             hipStream_t st) {
    if (n_reads == 0) {                                          // a rank without reads in this round: nothing for anybody
        E2_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * n_owners, st));
        return PG_OK;
    }
    if (!uniform_len && (!d_word_off || !d_kmer_base)) { pg_set_error("pg_skm_route: a ragged batch needs d_word_off and d_kmer_base"); return PG_EINVAL; }
    if ((ord_base >> (64 - SKM_ORD_SHIFT)) != 0) { pg_set_error("ordinal exceeds the 46 bits of a super-k-mer header"); return PG_EINVAL; }
    ReadsArg a;
    a.packed = d_packed; a.word_off = d_word_off; a.kmer_base = d_kmer_base; a.n_reads = n_reads; a.uniform_len = uniform_len;
    a.kpr = uniform_len ? uniform_len - c->K + 1 : 0; a.wpr = uniform_len ? (uniform_len + 31) / 32 : 0; a.ord_base = ord_base; a.max_len = 0; a.cls_kb = nullptr; a.cls_len = nullptr;
    E2_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * n_owners, st));
    RouteArg ro{d_recs, d_pids, (unsigned long long*)d_counts, cap, n_owners};
    if (!uniform_len && ragged_tiles_wanted()) { int rc = ragged_geometry(c, a, st); if (rc) return rc; }
    if ((uniform_len && uniform_len < 4096 && (int)a.kpr < 4096) || a.max_len) {
        int rc = a.max_len ? launch_ragged_by_class(c, a, &ro, st) : launch_tiled(c, a, &ro, st);
        if (rc != 1) return rc;
    }
    return launch_serial(c, a, &ro, st);
}

// the cut of a batch is about to be repeated with larger owner regions (pg_count_reads_sharded): forget the overflow
__global__ void e2_clear_flag_kernel(DevCounters* ctr, unsigned long long bit) { atomicAnd(&ctr->e2_flags, ~bit); }
int e2_clear_route_overflow(pg_ctx* c, hipStream_t st) {
    hipLaunchKernelGGL(e2_clear_flag_kernel, dim3(1), dim3(1), 0, st, c->ctr, F_ROUTE);
    E2_TRY(hipGetLastError());
    return PG_OK;
}

// multi-GPU step 2: take records another rank cut for this rank's partitions
int e2_ingest(pg_ctx* c, const uint64_t* d_recs, const uint32_t* d_pids, uint64_t n, hipStream_t st) {
    if (n == 0) return PG_OK;
    E2& s = c->e2;
    if (c->autogrow) {                       // room for n more records (+ one open chunk per partition is already counted)
        s.est_chunks += n / s.rpc + 1;
        if (s.est_chunks + ((uint64_t)1 << s.log2_parts) + 16 + ((uint64_t)s.direct << s.log2_parts) > s.pool_chunks) {
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
    if (c->n_owners > 1) { pg_set_error("pg_count_reads: this context stores one rank's share of the partitions (pg_expect with n_owners > 1): its batches go through pg_count_reads_sharded"); return PG_ESTATE; }
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
    a.max_len = 0;
    a.cls_kb = nullptr; a.cls_len = nullptr;
    if (!uniform_len && ragged_tiles_wanted()) { int rc = ragged_geometry(c, a, st); if (rc) return rc; }
    // tiled kernel for the batches whose per-read LDS footprint fits (ragged ones: rows for their longest read)
    int tiled = (uniform_len && uniform_len < 4096 && (int)a.kpr < 4096) || a.max_len;
    if (const char* v = env_measure("PG_K1")) tiled = tiled && atoi(v) != 0;
    if (tiled) {
        int rc = a.max_len ? launch_ragged_by_class(c, a, nullptr, st) : launch_tiled(c, a, nullptr, st);
        if (rc != 1) return rc;
    }
    return launch_serial(c, a, nullptr, st);
}

// pass 2 through the partitions: the node word of every k-mer occurrence of the records in the streams -> d_ans[ordinal - ord_base] (skm_answer_kernel)
int e2_answer(pg_ctx* c, const uint64_t* d_geo, uint32_t P, uint32_t bias, unsigned long long* d_ans, unsigned long long ord_base, hipStream_t st) {
    E2& s = c->e2;
    if (!s.pool) { pg_set_error("the partition streams are gone"); return PG_ESTATE; }
    if (c->n_owners > 1) { pg_set_error("e2_answer: a context that shares its partitions is not taken here"); return PG_ESTATE; }
    const uint32_t parts = 1u << s.log2_parts;
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    const dim3 g(std::min<unsigned>(parts, (unsigned)n_cu * 2u)), b(1024);
    const OccConst oc = occ_const(c->K, c->NW);
    const AnsArg aa{d_ans, ord_base, d_geo, P, bias};
    const bool usual = s.rpc == 128 && s.rs == (uint32_t)s.g.rw;
    (void)usual;
    if (c->NW == 2) {
        if (c->K == 63) hipLaunchKernelGGL((skm_answer_kernel<2, 2048, 1024, 512, 63>), g, b, 0, st, dev_view(c), oc, aa, c->ctr);
        else if (c->K == 31) hipLaunchKernelGGL((skm_answer_kernel<2, 2048, 1024, 512, 31>), g, b, 0, st, dev_view(c), oc, aa, c->ctr);
        else hipLaunchKernelGGL((skm_answer_kernel<2, 2048, 1024, 512, 0>), g, b, 0, st, dev_view(c), oc, aa, c->ctr);
    } else {
        if (c->K == 127) hipLaunchKernelGGL((skm_answer_kernel<4, 2048, 1024, 192, 127>), g, b, 0, st, dev_view(c), oc, aa, c->ctr);
        else hipLaunchKernelGGL((skm_answer_kernel<4, 2048, 1024, 192, 0>), g, b, 0, st, dev_view(c), oc, aa, c->ctr);
    }
    E2_TRY(hipGetLastError());
    return PG_OK;
}
// ... and what the pass left in the counters: flags (pool / chunk list / split) and occurrences whose k-mer was not found in the LDS set (must be 0)
int e2_answer_check(pg_ctx* c, hipStream_t st) {
    E2_TRY(hipStreamSynchronize(st));
    DevCounters* h = new DevCounters();
    const hipError_t rc = hipMemcpy(h, c->ctr, sizeof(DevCounters), hipMemcpyDeviceToHost);
    const unsigned long long flags = h->e2_flags, lost = h->overflow;        // (zeroes if the copy failed: it is reported first)
    delete h;
    E2_TRY(rc);
    if (flags & (F_POOL | F_CHUNKS | F_SPLIT | F_LEN)) { pg_set_error("pass 2 through the partitions: the partition engine gave up (flags " + std::to_string(flags) + ")"); return PG_ENOMEM; }
    if (lost) { pg_set_error("pass 2 through the partitions: " + std::to_string(lost) + " occurrence(s) without their k-mer in the set"); return PG_EINVAL; }
    return PG_OK;
}

// K3: per reference set, 1 + the ordinal of the last k-mer occurrence routed to it -> ctr->set_last (the streams must still exist)
static int e2_last_put_launch(pg_ctx* c, hipStream_t st) {
    E2& s = c->e2;
    if (!s.pool) { pg_set_error("the partition streams are gone (pg_export_take)"); return PG_ESTATE; }
    const SetParams sp{(uint32_t)c->P, set_bias((uint32_t)c->P)};
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    const unsigned grid = std::min<unsigned>(1u << s.log2_parts, (unsigned)n_cu * 8u);
    E2_TRY(hipMemsetAsync(c->ctr->set_last, 0, sizeof(unsigned long long) * 256, st));
    if (c->NW == 2) hipLaunchKernelGGL(skm_lastput_kernel<2>, dim3(grid), dim3(BLOCK), 0, st, dev_view(c), sp, c->ctr);
    else hipLaunchKernelGGL(skm_lastput_kernel<4>, dim3(grid), dim3(BLOCK), 0, st, dev_view(c), sp, c->ctr);
    E2_TRY(hipGetLastError());
    return PG_OK;
}
int e2_last_put(pg_ctx* c, uint64_t* out, hipStream_t st) {
    int rc = e2_last_put_launch(c, st);
    if (rc) return rc;
    E2_TRY(hipStreamSynchronize(st));
    unsigned long long h[256];
    E2_TRY(hipMemcpy(h, c->ctr->set_last, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < c->P; i++) out[i] = h[i];
    return PG_OK;
}
// distinct k-mers per reference set (after e2_count); uses ctr->set_last as scratch
int e2_set_counts(pg_ctx* c, uint64_t out[256], hipStream_t st) {
    E2& s = c->e2;
    if (!s.counted || !s.out) { pg_set_error("pg_set_counts: call pg_finalize first"); return PG_ESTATE; }
    unsigned long long n = 0;
    E2_TRY(hipMemcpy(&n, &c->ctr->n_export, sizeof n, hipMemcpyDeviceToHost));
    unsigned long long keep[256], h[256];
    E2_TRY(hipMemcpy(keep, c->ctr->set_last, sizeof keep, hipMemcpyDeviceToHost));
    E2_TRY(hipMemsetAsync(c->ctr->set_last, 0, sizeof keep, st));
    if (n) {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 256u * 16u);
        hipLaunchKernelGGL(set_count_kernel, dim3(grid), dim3(BLOCK), 0, st, s.out, (uint64_t)n, c->NW + 2, c->ctr);
        E2_TRY(hipGetLastError());
    }
    E2_TRY(hipStreamSynchronize(st));
    E2_TRY(hipMemcpy(h, c->ctr->set_last, sizeof h, hipMemcpyDeviceToHost));
    E2_TRY(hipMemcpy(c->ctr->set_last, keep, sizeof keep, hipMemcpyHostToDevice));
    for (int i = 0; i < 256; i++) out[i] = h[i];
    return PG_OK;
}

int e2_count(pg_ctx* c, int delow, bool want_last_put, hipStream_t st) {
    E2& s = c->e2;
    { const int rc = e2_join_out(c); if (rc) return rc; }
    const SetParams sp{(uint32_t)c->P, set_bias((uint32_t)c->P)};
    E2_TRY(hipMemsetAsync(&c->ctr->n_export, 0, sizeof(unsigned long long), st));
    E2_TRY(hipMemsetAsync(c->ctr->hist, 0, sizeof(unsigned long long) * 256, st));
    E2_TRY(hipMemsetAsync(&c->ctr->n_records, 0, sizeof(unsigned long long), st));
    E2_TRY(hipMemsetAsync(c->ctr->phase, 0, sizeof(unsigned long long) * 12, st));
    const uint32_t parts = 1u << s.log2_parts;
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    unsigned per_cu = 2;                                                            // persistent workgroups: two per CU measured best (1 .. 32 tried; every start of a workgroup builds its tables and wipes the set)
    if (const char* v = env_measure("PG_K2_WG_PER_CU")) per_cu = (unsigned)std::max(1, atoi(v));
    const unsigned grid = std::min<unsigned>(parts, (unsigned)n_cu * per_cu);
    // The kernels: 2048-slot set, 1024 lanes, windows of 512 records (192 in the four-word flavour) -> ~150 KB of LDS, one workgroup a CU; an
    // instantiation for each of K = 31 / 63 / 127 on the usual record geometry, the general kernel for everything else.  (Measured and gone:
    // half-size and quarter-size workgroups, half the lanes on the same set, static and share-table dealing, 1024-slot four-word sets: DESIGN.md §6.)
    const bool usual = s.rpc == 128 && s.rs == (uint32_t)s.g.rw;
    bool timers = false;
#ifdef PG_MEASURE
    timers = env_on(env_measure("PG_K2_TIMERS"));                        // thread 0's cycles per phase -> stderr (below)
#endif
    {
        const OccConst oc = occ_const(c->K, c->NW);
        const dim3 g(grid), b(1024);
        if (c->NW == 2) {
#ifdef PG_MEASURE
            if (timers && usual && c->K == 63) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, true, 63>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else if (timers && usual && c->K == 31) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, true, 31>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else if (timers) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, true>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr); else
#endif
            if (usual && c->K == 63) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, false, 63>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else if (usual && c->K == 31) hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, false, 31>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else hipLaunchKernelGGL((skm_count_kernel<2, 2048, 1024, 512, false>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
        } else {
#ifdef PG_MEASURE
            if (timers && usual && c->K == 127) hipLaunchKernelGGL((skm_count_kernel<4, 2048, 1024, 192, true, 127>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else if (timers) hipLaunchKernelGGL((skm_count_kernel<4, 2048, 1024, 192, true>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr); else
#endif
            if (usual && c->K == 127) hipLaunchKernelGGL((skm_count_kernel<4, 2048, 1024, 192, false, 127>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
            else hipLaunchKernelGGL((skm_count_kernel<4, 2048, 1024, 192, false>), g, b, 0, st, dev_view(c), delow, sp, oc, c->ctr);
        }
    }
    E2_TRY(hipGetLastError());
    if (want_last_put) {
        int rc = e2_last_put_launch(c, st);
        if (rc) return rc;
    }
    E2_TRY(hipStreamSynchronize(st));
    DevCounters h;
    E2_TRY(hipMemcpy(&h, c->ctr, sizeof h, hipMemcpyDeviceToHost));
    if (h.e2_flags & F_LEN) { pg_set_error("partition engine: a ragged batch holds a read longer than the bound given with pg_set_read_len_bound, or shorter than K + 1"); return PG_EINVAL; }
    if (h.e2_flags & F_ROUTE) { pg_set_error("partition engine: an owner's send region overflowed and the cut was not repeated"); return PG_ENOMEM; }
    if (h.e2_flags & F_POOL) { pg_set_error("partition engine: record pool exhausted (raise log2_slots or PG_POOL_MB)"); return PG_ENOMEM; }
    if (h.e2_flags & F_CHUNKS) { pg_set_error("partition engine: one partition outgrew its chunk list (heavily skewed minimizers)"); return PG_ENOMEM; }
    if ((h.e2_flags & F_OUT) && c->autogrow && !(h.e2_flags & (F_POOL | F_CHUNKS | F_SPLIT))) {
        // the streams are intact: count again into an export array that holds everything (n_export is the true count)
        // (the array that was too small goes first: nothing in it is kept, and the two together were what did not fit at configs[4])
        const uint64_t want = h.n_export + h.n_export / 8 + 64;
        uint64_t* fresh = nullptr;
        E2_TRY(pg::arena_free(s.out));
        s.out = nullptr;
        E2_TRY(pg::arena_malloc(&fresh, want * (uint64_t)(c->NW + 2) * 8));
        s.out = fresh;
        s.out_capacity = want;
        unsigned long long keep = h.e2_flags & ~F_OUT;
        E2_TRY(hipMemcpy(&c->ctr->e2_flags, &keep, sizeof keep, hipMemcpyHostToDevice));
        return e2_count(c, delow, want_last_put, st);
    }
    if (h.e2_flags & F_OUT) { pg_set_error("partition engine: more distinct k-mers than the export array holds (raise log2_slots)"); return PG_ENOMEM; }
    if (h.e2_flags & F_SPLIT) { pg_set_error("partition engine: a partition could not be split to fit the LDS set"); return PG_ENOMEM; }
    if (timers) {
        static const char* names[12] = {"partition header", "clear after a dropped attempt", "window: unpack / stage + barrier", "barrier + flatten (prefix sum, tables) + barriers", "occurrences (thread 0's share)",
                                        "wait for the slowest wave + barrier", "emit: list the live slots + barrier", "emit: ask for the next window", "emit: barrier + export stores from registers", "dropped attempts (count)",
                                        "dedupe (hash, probe, compare)", "emit: finalise in registers"};
        unsigned long long tot = 0;
        for (int i = 0; i < 12; i++) if (i != 9) tot += h.phase[i];
        for (int i = 0; i < 12; i++) fprintf(stderr, "K2 phase %-40s %14llu  %5.1f%%\n", names[i], h.phase[i], i != 9 ? 100.0 * h.phase[i] / (double)(tot ? tot : 1) : 0.0);
    }
    s.counted = true;
    // the export array was sized from an estimate (a set of 2^log2_slots slots, or the caller's pg_expect): what the distinct k-mers do not
    // fill goes back to the arena now -- 60 of 96 GB at 200 M reads -- and is there for the layout, the tips, the edges and pass 2
    // (a context whose capacity the caller guarantees -- pg_set_autogrow(0) -- keeps what it was given: it may be reset and filled again)
    if (s.out && c->autogrow && h.n_export < s.out_capacity) {
        const uint64_t keep = std::max<uint64_t>(h.n_export, 1);
        if ((s.out_capacity - keep) * (uint64_t)(c->NW + 2) * 8 >= ((uint64_t)64 << 20)) {
            pg::arena_shrink(s.out, keep * (uint64_t)(c->NW + 2) * 8 + 64);
            s.out_capacity = keep;
        }
    }
    return PG_OK;
}

}  // namespace pg
