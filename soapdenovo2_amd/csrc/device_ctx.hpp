// device_ctx.hpp -- state shared by the two pass-1 engines behind the pg_* device operators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <thread>

#include "kmer.hpp"
#include "skm.hpp"

void pg_set_error(const std::string& s);

namespace pg {

constexpr int BLOCK = 256;

struct DevCounters {
    unsigned long long n_distinct;
    unsigned long long overflow;
    unsigned long long hist[256];
    unsigned long long set_last[256];
    unsigned long long n_export;
    // partition engine
    unsigned long long pool_next;      // (unused since the hand-out counters were split: pool_sub)
    unsigned long long e2_flags;       // bit 0 pool exhausted, bit 1 partition chunk list full, bit 2 output full, bit 3 split exhausted
    unsigned long long n_records;      // super-k-mer records written
    unsigned long long phase[12];      // PG_DBG=2: cycles of workgroup thread 0 per K2 phase (measurement aid)
    unsigned long long ragged_max;     // k-mers of the longest read of a ragged batch (partition_kernels.hip: ragged_geometry)
    // Chunks of the record pool are handed out by POOL_SUBS counters, one 64-byte line each, picked by the partition id: a
    // single counter took every chunk request of the chip -- 7 M returned atomics on one address per 200 M reads, and one
    // address serves about 88 of them per microsecond (MI355X_MICROARCH.md, "dequeue"): that alone was K1's 84 ms.
    // Sub-pool s owns the chunks s + 1, s + 1 + POOL_SUBS, ... (interleaved, so a pool that grows keeps every id valid).
    unsigned long long pool_sub[1024 * 8];
};
constexpr uint32_t POOL_SUBS = 1024;

struct SetParams { uint32_t P, bias; };

// Partition engine (engine 2): super-k-mer streams + per-partition LDS counting.
struct E2 {
    SkmGeom g;
    int log2_parts = 0;           // partitions STORED here (cursors, chunk table, computed chunk addresses, the counting grid)
    int log2_global = 0;          // partition ids of the job (= log2_parts unless the ids are shared out between ranks: pg_expect)
    uint32_t rpc = 128;           // records per chunk
    uint32_t rs = 0;              // words from one record to the next (>= g.rw; 8 = every 48-byte record in its own 64-byte line)
    uint32_t direct = 0;          // the first `direct` chunks of every partition lie at computed addresses (chunk c of partition p
                                  // = pool chunk c * parts + p): no table entry, no pool atomic, nobody waits for a published id
    uint32_t maxc = 0;            // chunk-table entries per partition (the chunks after the direct ones)
    uint64_t pool_chunks = 0;
    uint32_t* cursor = nullptr;   // [parts] records appended
    uint32_t* chunk_tbl = nullptr;// [parts * maxc] chunk id + 1, 0 = not allocated
    uint64_t* pool = nullptr;     // pool_chunks * rpc * rw words (+ padding)
    uint64_t* out = nullptr;      // export records
    uint64_t out_capacity = 0;    // records
    bool counted = false;
    uint64_t est_chunks = 0;      // host estimate of chunks in use (pool growth without a sync per batch)
    // The export array is needed when the partitions are counted, not while the batches are cut: it is allocated by a thread of its
    // own (tens of gigabytes of device memory take the driver up to seconds to hand out on some boxes -- it clears what it gives),
    // joined by whoever needs `out` first (e2_count, e2_destroy).
    std::thread* out_thread = nullptr;
    int out_err = 0;              // hipError_t of that allocation
    // ragged batches cut by length class (partition_kernels.hip: launch_ragged_by_class): the batch's reads sorted by class, the classes' counters
    uint32_t* rg_perm = nullptr;      // per class the gathered rows: start words, first k-mers (8 B each), lengths (4 B), rg_perm_cap reads each
    uint64_t rg_perm_cap = 0;
    unsigned int* rg_hist = nullptr;   // [2 * 32] histogram | cursors
};

}  // namespace pg

struct pg_ctx {
    int device, K, NW, P, log2_slots;
    uint64_t* slots;             // engine 1: the open-addressed set
    pg::DevCounters* ctr;        // device
    uint64_t ub_distinct;        // engine 1: host upper bound on stored keys (avoids a sync per batch)
    bool finalized;
    bool autogrow;
    int variant;                 // engine 1: 0 = word-wise atomic loads, 1 = 32-byte slot snapshot (PG_VARIANT)
    int engine;                  // 1 = global hash set, 2 = super-k-mer partitions counted in LDS (PG_ENGINE)
    int hint_log2_parts = -1;    // engine 2: partition count asked for by pg_expect_kmers (-1 = derive from log2_slots)
    uint64_t hint_kmers = 0;     // engine 2: k-mer occurrences to come, 0 = unknown (sizes the record pool)
    uint64_t batches = 0;        // batches taken since create / reset
    int n_owners = 1;            // engine 2: ranks that share the job's partition ids (pg_expect); this context stores id / n_owners of those it owns
    uint64_t hint_reads = 0;     // engine 2: reads to come (pg_expect), 0 = unknown
    uint64_t hint_distinct = 0;  // engine 2: distinct k-mers expected in this context's export array (pg_expect), 0 = what log2_slots says
    uint32_t read_len_bound = 0; // engine 2: no read of a ragged batch is longer (pg_set_read_len_bound; 0 = ask the device, one host wait a batch)
    pg::E2 e2;
    // the sharded pass 1 (exchange.hip) keeps its last round's records in flight when it returns: whatever consumes the partition
    // streams next (pg_finalize, pg_reset, pg_destroy, ...) has them appended first
    int (*pending_drain)(pg_ctx*, void* user, hipStream_t st) = nullptr;
    void (*pending_detach)(pg_ctx*, void* user) = nullptr;       // the context goes away: whoever holds it forgets it
    void* pending_user = nullptr;
};
inline int pg_ctx_drain(pg_ctx* c, hipStream_t st) { return c && c->pending_drain ? c->pending_drain(c, c->pending_user, st) : 0; }

// engine 2 entry points (partition_kernels.hip)
namespace pg {
int e2_create(pg_ctx* c);
void e2_destroy(pg_ctx* c);
int e2_reset(pg_ctx* c, hipStream_t st);
int e2_scatter(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base, uint64_t n_reads,
               uint32_t uniform_len, uint64_t n_kmers_hint, uint64_t ord_base, hipStream_t st);
int e2_count(pg_ctx* c, int delow, bool want_last_put, hipStream_t st);
int e2_last_put(pg_ctx* c, uint64_t* out, hipStream_t st);
int e2_set_counts(pg_ctx* c, uint64_t out[256], hipStream_t st);
int e2_route(pg_ctx* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base, uint64_t n_reads,
             uint32_t uniform_len, uint64_t ord_base, int n_owners, uint64_t* d_recs, uint32_t* d_pids, uint64_t cap, uint64_t* d_counts,
             hipStream_t st);
int e2_clear_route_overflow(pg_ctx* c, hipStream_t st);
int e2_ingest(pg_ctx* c, const uint64_t* d_recs, const uint32_t* d_pids, uint64_t n, hipStream_t st);
int e2_answer(pg_ctx* c, const uint64_t* d_geo, uint32_t P, uint32_t bias, unsigned long long* d_ans, unsigned long long ord_base, hipStream_t st);
int e2_answer_check(pg_ctx* c, hipStream_t st);
}  // namespace pg
