// host_plan.cpp -- pg_host_plan_memory: the device memory a `pregraph` command takes on one rank, stage by stage, computed on the host
// from the SAME sizing functions the command and the partition engine allocate by (cmd_plan.hpp, e2_plan.hpp, ref_sizes.hpp).  No GPU is
// touched.  It answers, before anything is allocated, whether a configuration fits a GPU -- BASELINE.json's configs[3] and configs[4]
// (3 G reads on 8 ranks) have never met hardware -- and tests/test_host_plan.py holds it against the arena's measured peaks of the command
// legs (PG_ARENA_TRACE=1, profiles/r06_arena_trace_*.txt).
//
// The stages and what is alive in each (one rank):
//   1  pass 1 + count   cursors + chunk table, record pool, export array AS ALLOCATED, the pass-1 batch buffers, the reads kept for pass 2
//                       (+ the exchange's send / receive regions on a sharded run)
//   2  hand-over        the export array cut back to the distinct k-mers (arena_shrink), the record pool as work space: the sort by (set, first
//                       ordinal) runs inside it when it can hold the sort's work space, else the pool is given back first (cmd_sort_ws_bytes)
//   3  layout           the k-mer sets (the reference's slot images: 24 / 40 B a slot, laid inside the record pool when they fit), the sorted
//                       records, the layout's arrays for the largest set (56 B a key for -a pools, 136 B for growable sets) + the sort scratch
//   4  graph + pass 2   the k-mer sets, the kept reads, the tip / edge lists, pass 2's pre-arc table: smaller than stage 3 at every size measured
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>

#include "cmd_plan.hpp"
#include "e2_plan.hpp"
#include "ref_sizes.hpp"
#include "../../include/soapdenovo2_amd.h"

void pg_set_error(const std::string& s);

namespace {

// the reference's set size: -a pools (prlHashReads.c:369-390), or the size a growable set of n keys ends at (newhash.c:340-455)
uint64_t set_slots(uint64_t n_keys, int a_gb, int n_sets, bool mer127) {
    if (a_gb > 0) {
        const uint64_t init = (uint64_t)((double)a_gb * 1073741824.0 / (double)n_sets / (mer127 ? 40.0 : 24.0));
        uint64_t k = 1;
        while (k * 0xFFFFFFULL < init) k++;
        return pg::ref_next_prime(k * 0xFFFFFFULL);
    }
    // the growth schedule without walking it key by key: sizes double (then grow by 0xFFFFFF) until 0.77 of the size holds the keys
    uint64_t size = pg::ref_next_prime(1024);
    while ((uint64_t)((float)size * 0.77f) < n_keys) {
        uint64_t nn = size;
        do {
            nn = (nn < 0xFFFFFFFULL) ? (nn << 1) : (nn + 0xFFFFFFULL);
            nn |= 1;                                                   // (the next "prime" is within a few dozen of it: the plan does not need the exact one)
        } while ((float)nn * 0.77f < (float)((uint64_t)((float)size * 0.77f) + 1));
        size = nn;
    }
    return size;
}

}  // namespace

extern "C" int pg_host_plan_memory(uint64_t reads_total, uint32_t read_len, uint64_t fastq_bytes, uint64_t distinct_total, int K, int mer127, int n_sets, int a_gb,
                                   int n_ranks, uint64_t device_bytes, uint64_t out[24]) {
    if (!out || reads_total == 0 || read_len < (uint32_t)K + 1 || n_ranks < 1 || n_sets < 1 || K < 13 || device_bytes == 0) { pg_set_error("pg_host_plan_memory: bad argument"); return PG_EINVAL; }
    memset(out, 0, 24 * sizeof(uint64_t));
    const int NW = mer127 ? 4 : 2;
    const uint64_t rec_bytes = (uint64_t)(NW + 2) * 8;
    // what call_pregraph estimates from the file sizes: half the bytes of a FASTQ file are bases, of which (max_rd_len - K + 1) / max_rd_len start a k-mer
    if (!fastq_bytes) fastq_bytes = reads_total * (2ull * read_len + 16);
    const uint64_t est_kmers = (uint64_t)((double)fastq_bytes * 0.5 * (double)(read_len - K + 1) / (double)read_len);
    const uint64_t reads_rank = (reads_total + n_ranks - 1) / n_ranks;
    const uint64_t distinct_rank = (uint64_t)((double)distinct_total / (double)n_ranks * (n_ranks > 1 ? 1.1 : 1.0));   // (partition shares are not quite equal)
    // ---- stage 1
    const int log2_slots = pg::cmd_log2_slots(est_kmers, mer127 != 0, device_bytes);
    const uint64_t export_records = pg::cmd_export_records(est_kmers, mer127 != 0, n_ranks, device_bytes);
    const int lp = pg::parts_for_kmers(est_kmers + (n_ranks > 1 ? 1 : 0), NW, n_ranks);
    const uint64_t est_reads = (uint64_t)((double)fastq_bytes * 0.5 / (double)read_len);
    const pg::E2Plan e2 = pg::e2_plan(K, NW, log2_slots, est_kmers + (n_ranks > 1 ? 1 : 0), est_reads, lp, export_records, n_ranks, (uint64_t)((double)device_bytes * 0.995), device_bytes);
    if (e2.err) { pg_set_error(e2.err == 1 ? "pg_host_plan_memory: the export array does not fit" : "pg_host_plan_memory: the record pool is too small for the partition count"); return PG_ENOMEM; }
    const uint64_t wpr = (read_len + 31) / 32;
    const uint64_t batch = 2 * ((pg::CMD_BATCH_WORDS + 8) * 8 + pg::CMD_BATCH_READS * 8 + (pg::CMD_BATCH_READS + 1) * 8);      // two batches in flight: words, word_off, kmer_base
    uint64_t kept = reads_rank * wpr * 8 + (reads_rank / pg::CMD_BATCH_READS + 1) * 64;                  // the reads of pass 1, one segment a batch
    // (chunks of 1 GiB: the last one is cut whole)
    kept = (kept + ((uint64_t)1 << 30) - 1) >> 30 << 30;
    if (kept > pg::cmd_dev_keep_budget(device_bytes) || n_ranks > 1) kept = 0;                           // over the budget (or a sharded pass 1): the host store
    // the exchange of a sharded pass 1: two slots of send + receive regions, about twice a batch's records over the ranks, for every rank pair
    uint64_t exchange = 0;
    if (n_ranks > 1) {
        const pg::SkmGeom g = pg::skm_geometry(K, e2.log2_global, NW);
        const uint64_t kpr = read_len - K + 1, est_recs = 2 * (pg::CMD_BATCH_READS * kpr) / (uint64_t)(g.w + 1) + pg::CMD_BATCH_READS;
        const uint64_t cap = 2 * est_recs / (uint64_t)n_ranks + 4096;
        exchange = 2 /* slots */ * 2 /* send + receive */ * (uint64_t)n_ranks * cap * ((uint64_t)g.rw * 8 + 4);
    }
    // (an export array made for too few: it is released and the partitions are counted again into one of the true size + 1/8)
    const bool recount = distinct_rank > e2.out_capacity;
    const uint64_t s1 = e2.table_bytes + e2.pool_bytes + std::max(e2.out_bytes, recount ? (distinct_rank + distinct_rank / 8 + 64) * rec_bytes : 0) + batch + kept + exchange;
    // ---- stage 2: the count is done, the export array holds the distinct k-mers and no more
    const uint64_t out_kept = distinct_rank * rec_bytes + 4096;
    // the sort by (set, first ordinal) runs inside the record pool when the pool can hold its work space; a pool that cannot is given back first
    const uint64_t sort_ws = pg::cmd_sort_ws_bytes(distinct_rank, mer127 != 0);
    const bool pool_stays = sort_ws <= e2.pool_bytes;
    const uint64_t sort_extra = pool_stays ? 0 : sort_ws;
    const uint64_t s2 = (pool_stays ? e2.pool_bytes : 0) + out_kept + batch + kept + sort_extra;
    // ---- stage 3: the sets of this rank and the layout's arrays for its largest set
    const int sets_here = (n_sets + n_ranks - 1) / n_ranks;
    const uint64_t per_set = (distinct_total + n_sets - 1) / n_sets;
    const uint64_t slots = set_slots(per_set, a_gb, n_sets, mer127 != 0);
    const uint64_t set_bytes = (uint64_t)sets_here * slots * (mer127 ? 40 : 24);
    const uint64_t n_max = (uint64_t)((double)per_set * 1.02);
    // -a pools: a set at a time, 56 B a key + the sort's scratch; growable sets: two sets side by side, 160 B a key each (dev_rehash.hpp)
    const uint64_t layout_arrays = a_gb > 0 ? n_max * 56 + slots * 8 : 2 * n_max * 160 + slots * 8;
    // the sets go inside the record pool (pass 1 is done with it) when it is still there and they fit; otherwise they are blocks of their own
    const uint64_t sets_outside = pool_stays && set_bytes <= e2.pool_bytes ? 0 : set_bytes;
    const uint64_t s3 = (pool_stays ? e2.pool_bytes : 0) + out_kept + kept + sets_outside + layout_arrays;
    // ---- stage 4: pool and records are gone; sets, reads, the graph's lists and pass 2's table
    const uint64_t edges_guess = distinct_rank / 40 + (1 << 20);                                          // (vertices + edges: a few per cent of the k-mers)
    // pass 2's pre-arc table: a lane's reads meet any edge, so it is made for the GRAPH's edge ids (about one per 40 k-mers), not for the rank's share
    const uint64_t prearc = pg::cmd_prearc_entries(distinct_total / 40 + (1 << 20), device_bytes) * pg::CMD_PREARC_ENTRY_BYTES;
    const uint64_t s4 = set_bytes + kept + batch + edges_guess * 64 + prearc;
    const uint64_t peak = std::max(std::max(s1, s2), std::max(s3, s4));
    out[0] = peak;
    out[1] = peak == s1 ? 1 : peak == s2 ? 2 : peak == s3 ? 3 : 4;
    out[2] = e2.table_bytes; out[3] = e2.pool_bytes; out[4] = e2.out_bytes; out[5] = out_kept; out[6] = kept; out[7] = batch + exchange;
    out[8] = set_bytes; out[9] = layout_arrays; out[10] = sort_extra;
    out[11] = (uint64_t)e2.log2_global; out[12] = (uint64_t)e2.log2_store; out[13] = e2.direct; out[14] = e2.pool_chunks * e2.rpc;
    out[15] = peak <= (uint64_t)((double)device_bytes * 0.97) ? 1 : 0;
    out[16] = s1; out[17] = s2; out[18] = s3; out[19] = s4;
    out[20] = est_kmers; out[21] = export_records; out[22] = slots; out[23] = (uint64_t)log2_slots | (recount ? 1ull << 32 : 0);
    return PG_OK;
}
