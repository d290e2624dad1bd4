// e2_plan.hpp -- the sizing rules of the partition engine (engine 2) as ONE pure host function: e2_create allocates what it
// says, pg_host_plan_memory (host_plan.cpp) adds the other stages' blocks to it and answers "does this command fit this GPU"
// without a GPU.  No HIP in here.
//
// What is sized (per context = per rank):
//   * the partition geometry: `log2_global` partition ids for the whole job (about 8 k k-mer occurrences a partition, 4 k in the
//     four-word flavour -- parts_for_kmers), of which this rank STORES those it owns.  Round 6: a rank of a sharded run owns the
//     ids with id mod n_owners == its rank and stores them at id / n_owners -- cursors, chunk table and the chunks at computed
//     addresses are sized for ceil(2^log2_global / n_owners) partitions (rounded up to a power of two), not for all of them: at
//     configs[3] (264 G occurrences, 8 ranks) the spare chunk a partition alone was 103 GB a rank before;
//   * the record pool: `direct` chunks a stored partition at computed addresses (1.25 x the mean partition, at most eight chunks
//     and a quarter of the device), the records expected beyond them and one spare chunk a partition;
//   * the export array: hint_distinct records when the caller says how many distinct k-mers it expects here, else what a set of
//     2^log2_slots slots holds at 70 % load.  (When it is too small the partitions are counted again into one that holds the true
//     count: e2_count.)
#pragma once
#include <stdint.h>

#include <algorithm>

#include "skm.hpp"

namespace pg {

// partition ids of a job of total_kmers k-mer occurrences (all ranks together); cap: 24 bits of ids on one rank, more when the ids are
// shared out (every rank stores at most 2^24 of them)
inline int parts_for_kmers(uint64_t total_kmers, int nw, int n_owners = 1, int shift = 0) {
    int cap = 24;
    for (int n = std::max(1, n_owners); n > 1 && cap < 28; n >>= 1) cap++;
    int lp = 8;
    // about 8 k occurrences a partition (2 k for the 127-mer flavour: its LDS set holds half as many keys, and likes them sparse), to
    // the NEAREST power of two: at 200 M x 150 bp, K = 63 (17.6 G occurrences) 2^21 partitions of 8.4 k beat 2^22 of 4.2 k by 7 % in K2
    // (the 127-mer flavour keeps rounding up: 2^22 partitions of 1.1 k beat 2^21 of 2.3 k by 10 % there)
    // round 4, later: the 127-mer flavour's set holds 2048 keys of four words (one claim a key): 4 k occurrences a partition
    while (lp < cap && (double)((uint64_t)(nw == 4 ? 4096 : 8192) << lp) * (nw == 4 ? 1.0 : 1.4142) < (double)total_kmers) lp++;
    return std::max(8, std::min(cap, lp + shift));
}

struct E2Plan {
    int log2_global = 0;          // partition ids of the job: pid = hash scaled to [0, part_mul), part_mul = 2^log2_global (PG_PARTS_EFF_PCT: fewer)
    int log2_store = 0;           // partitions this context stores (= log2_global with one owner)
    uint32_t rpc = 128, rs = 0, direct = 0, maxc = 0;
    uint64_t pool_chunks = 0, out_capacity = 0;
    uint64_t chunk_bytes = 0, pool_bytes = 0, out_bytes = 0, table_bytes = 0;     // table = cursors + chunk table
    int err = 0;                  // 1: the export array does not fit; 2: the pool is too small for the partition count
    uint64_t bytes() const { return pool_bytes + out_bytes + table_bytes; }
};

// log2_slots: the caller's capacity figure (pg_create); hint_kmers: the job's k-mer occurrences, 0 = unknown; hint_log2_parts: -1 = derive;
// hint_distinct: distinct k-mers expected in THIS context's export array, 0 = from log2_slots; free_b / total_b: the device's memory.
// rpc / rs_override / direct_override / pool_mb / parts_eff: the -DPG_MEASURE knobs (0 / -1 = the product's choice).
// hint_reads: the job's reads, 0 = unknown (every read makes at least one record: it matters where a read has few k-mers -- K = 127 from 150-base reads).
inline E2Plan e2_plan(int K, int NW, int log2_slots, uint64_t hint_kmers, uint64_t hint_reads, int hint_log2_parts, uint64_t hint_distinct, int n_owners, uint64_t free_b, uint64_t total_b,
                      uint32_t rpc = 128, int rs_override = 0, int direct_override = -1, uint64_t pool_mb = 0, int test_log2_parts = -1) {
    E2Plan p;
    n_owners = std::max(1, n_owners);
    p.log2_global = std::max(8, std::min(24, log2_slots - (NW == 4 ? 10 : 11)));
    if (hint_log2_parts >= 0) p.log2_global = std::max(8, std::min(28, hint_log2_parts));
    if (test_log2_parts >= 0) p.log2_global = std::max(4, std::min(24, test_log2_parts));
    {
        const uint64_t store = (((uint64_t)1 << p.log2_global) + (uint64_t)n_owners - 1) / (uint64_t)n_owners;
        p.log2_store = 0;
        while (((uint64_t)1 << p.log2_store) < store) p.log2_store++;
    }
    const SkmGeom g = skm_geometry(K, p.log2_global, NW);
    p.rpc = rpc;
    const uint64_t parts = (uint64_t)1 << p.log2_store;                    // what is stored here
    p.rs = (uint32_t)g.rw;
    if (rs_override >= g.rw && rs_override <= 16 && (rs_override & 1) == 0) p.rs = (uint32_t)rs_override;
    const uint64_t rec_bytes = (uint64_t)p.rs * 8;
    p.chunk_bytes = rec_bytes * p.rpc;
    // records expected: a read of k k-mers makes about 2 k / (w + 1) + 1 of them (a rank's partitions take 1 / n_owners of the job's)
    const double recs_job = (double)hint_kmers * 2.0 / (double)(g.w + 1) + (double)hint_reads;
    const double recs_here = recs_job / (double)n_owners;
    {
        // with the input size known, room for the mean partition, rounded up to whole chunks, at computed addresses (at most eight chunks a
        // partition and a quarter of the device memory).  (Until round 6 the mean left out the record every read makes whatever its length and
        // was taken 1.25 times: the same four chunks at K = 63 from 150-base reads, a third of what K = 127 needs -- whose pool then grew mid-pass.)
        int q = hint_kmers ? (int)((recs_job / (double)g.part_mul + (double)p.rpc - 1) / (double)p.rpc) : 0;
        q = std::min(q, 8);
        if (direct_override >= 0) q = std::min(direct_override, 192);
        while (q > 0 && total_b && (uint64_t)q * parts * p.chunk_bytes > total_b / 4) q--;
        p.direct = (uint32_t)std::max(0, q);
    }
    // export array
    p.out_capacity = hint_distinct ? hint_distinct : (uint64_t)(0.7 * (double)((uint64_t)1 << log2_slots));
    p.out_bytes = p.out_capacity * (uint64_t)(NW + 2) * 8;
    // record pool: every partition keeps one partly filled chunk, plus the records themselves (about one record per 20 k-mers); default =
    // as much as a set of 2^log2_slots 64-byte slots, capped by what is free
    uint64_t pool_bytes = ((uint64_t)1 << log2_slots) * 64 + parts * p.chunk_bytes * 2;
    if (hint_kmers)                          // known input size: the records expected, half as much again (0.6 of them beside the chunks at computed addresses); it grows
        // (with chunks at computed addresses every partition's open chunk lies inside that region already: ONE spare chunk a partition is set aside
        //  for the draws beyond it -- what the check below and e2_ensure_pool's estimate count on -- instead of two)
        pool_bytes = (uint64_t)(recs_here * (p.direct ? 0.6 : 1.5)) * rec_bytes + parts * p.chunk_bytes * (p.direct ? 1 : 2) + ((uint64_t)64 << 20);
    if (pool_mb) pool_bytes = pool_mb << 20;
    const uint64_t budget = (uint64_t)((double)free_b * 0.85);
    if (p.out_bytes + parts * 8 > budget) { p.err = 1; return p; }
    pool_bytes += (uint64_t)p.direct * parts * p.chunk_bytes;
    pool_bytes = std::min<uint64_t>(pool_bytes, (budget - p.out_bytes) * 9 / 10);
    p.pool_chunks = pool_bytes / p.chunk_bytes;
    if (p.pool_chunks < (uint64_t)p.direct * parts + parts + 16) { p.err = 2; return p; }
    if ((((uint64_t)p.direct + 1) << p.log2_store) >= 0xFFFFFFFFULL || p.pool_chunks >= 0xFFFFFFFFULL) p.pool_chunks = std::min<uint64_t>(p.pool_chunks, 0xFFFFFFF0ULL);   // chunk ids are 32 bits
    // chunk table: up to 2^29 entries in total (2 GB), at least enough for an even spread x8
    const uint64_t even = (p.pool_chunks - (uint64_t)p.direct * parts + parts - 1) / parts;
    p.maxc = (uint32_t)std::max<uint64_t>(8, std::min<uint64_t>(std::min<uint64_t>(256 - p.direct, ((uint64_t)1 << 29) / parts), even * 16));
    p.pool_bytes = p.pool_chunks * p.chunk_bytes + 64;
    p.table_bytes = parts * ((uint64_t)p.maxc + 1) * sizeof(uint32_t);
    return p;
}

}  // namespace pg
