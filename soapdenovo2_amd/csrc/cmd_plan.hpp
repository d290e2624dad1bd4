// cmd_plan.hpp -- the figures call_pregraph derives from the size of its input before it touches a GPU, as pure functions: the command
// uses them, and so does the executable memory plan (host_plan.cpp: pg_host_plan_memory).  No HIP in here.
#pragma once
#include <stdint.h>

#include <algorithm>

namespace pg {

// pg_create's log2_slots: room for one distinct k-mer per 6 occurrences at 70 % load, never more than a third of the device memory.  With a
// k-mer estimate it only sizes the DEFAULT partition count and pool (both follow the estimate instead); kept as the capacity figure of the API.
inline int cmd_log2_slots(uint64_t est_kmers, bool mer127, uint64_t total_b) {
    int log2_slots = 24;
    const double rec_bytes = (mer127 ? 6 : 4) * 8.0;
    while (log2_slots < 34 && (double)((uint64_t)1 << log2_slots) * 0.7 < (double)est_kmers / 6.0 &&
           (double)((uint64_t)2 << log2_slots) * 0.7 * rec_bytes <= (double)total_b / 3.0)
        log2_slots++;
    return log2_slots;
}

// distinct k-mers a rank's export array is made for: one per 8 occurrences of the rank's share (a quarter more when the partitions are shared
// out: the shares are not equal), capped at a third of the device.  Rounds 2 - 5 took 0.7 x the next power of two above est / 6 -- 96 GB for
// the 37 GB of configs[2]'s 1.15 G distinct k-mers.  An estimate that turns out too small costs one more counting pass (e2_count counts the
// partitions again into an array of the true size); what the estimate leaves unused goes back to the arena behind the count either way.
inline uint64_t cmd_export_records(uint64_t est_kmers, bool mer127, int n_ranks, uint64_t total_b) {
    const double rec_bytes = (mer127 ? 6 : 4) * 8.0;
    n_ranks = std::max(1, n_ranks);
    double want = (double)est_kmers / 8.0 / (double)n_ranks * (n_ranks > 1 ? 1.25 : 1.0) + (double)(1 << 20);
    want = std::min(want, (double)total_b / 3.0 / rec_bytes);
    return (uint64_t)want;
}

// what the sort of n records by (set, first ordinal) needs as work space: 8-byte keys and 4-byte indices twice, the radix sort's own scratch
// (about as much again as one key + index array), and a copy of the records (sort_records.hip).  A record pool that cannot hold this is
// given back BEFORE the sort -- its records are dead by then -- instead of standing beside a copy of its own size (round 6: at configs[3] a
// rank's pool of 130 GB, its 83 GB of distinct k-mers and their sorted copy did not fit 288 GB together).
inline uint64_t cmd_sort_ws_bytes(uint64_t n, bool mer127) { return n * ((mer127 ? 6 : 4) * 8 + 36) + ((uint64_t)64 << 20); }

// pass 2's pre-arc table of one lane (graph_kernels.hip: add_prearc -- open addressing, 32 bytes an entry): eight entries an edge id (the distinct
// pre-arcs are about one an edge: 26.2 M for 26.4 M edge ids at 200 M reads), as a power of two -- but no more than an eighth of the device while that
// still leaves two entries an edge: a human genome's ~0.5 G edge ids would otherwise ask for 137 GB a lane (34 GB so, at a load of ~0.45; what does not
// fit is counted and fails the command, never dropped).
constexpr uint64_t CMD_PREARC_ENTRY_BYTES = 32;
inline uint64_t cmd_prearc_entries(uint64_t num_ed, uint64_t total_b) {
    uint64_t cap = (uint64_t)1 << 16;
    while (cap < num_ed * 8) cap <<= 1;
    while (total_b && cap * CMD_PREARC_ENTRY_BYTES > total_b / 8 && (cap >> 1) >= num_ed * 2) cap >>= 1;
    return cap;
}

// a pass-1 batch: 64 MiB of packed reads / 2 M reads
constexpr uint64_t CMD_BATCH_WORDS = (uint64_t)1 << 23, CMD_BATCH_READS = (uint64_t)1 << 21;
// the reads of pass 1 stay on the device for pass 2 while they fit this share of it
inline uint64_t cmd_dev_keep_budget(uint64_t total_b) { return total_b / 8; }

}  // namespace pg
