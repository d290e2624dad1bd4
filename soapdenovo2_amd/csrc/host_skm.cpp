// host_skm.cpp -- host twins of the multi-GPU routing step, for tests and tools that have no GPU: the same inline code the
// kernels run (skm.hpp: cutting a read into super-k-mer records, partition and owner of a record, expanding a record into
// k-mer occurrences), called serially.  Nothing in the product path calls these; pass 1 runs on HIP devices only.
#include <stdint.h>

#include <string>

#include "extract.hpp"
#include "skm.hpp"
#include "../../include/soapdenovo2_amd.h"

void pg_set_error(const std::string& s);

namespace {

template <int NW>
int64_t cut(const uint64_t* packed, uint64_t n_reads, uint32_t len, int K, int log2_parts, uint64_t ord_base, int n_owners, uint64_t* recs,
            uint64_t* tags, uint64_t cap) {
    constexpr int PW = NW == 2 ? 5 : 7, RW = PW + 1;
    const pg::SkmGeom g = pg::skm_geometry(K, log2_parts, NW);
    const uint64_t wpr = (len + 31) / 32, kpr = len - K + 1;
    uint64_t n = 0;
    bool full = false;
    for (uint64_t r = 0; r < n_reads && !full; r++) {
        const uint64_t* rd = packed + r * wpr;
        pg::skm_split_read(rd, (int)len, g, [&](int j0, int cnt, uint32_t pid) {
            if (n >= cap) { full = true; return; }
            pg::skm_make_record<PW>(rd, (int)len, j0, cnt, ord_base + r * kpr, g, recs + n * RW);
            tags[n] = ((uint64_t)pid << 8) | (uint64_t)(pid % (uint32_t)n_owners);
            n++;
        });
    }
    if (full) { pg_set_error("pg_host_skm_cut: output arrays too small"); return -1; }
    return (int64_t)n;
}

template <int NW>
int64_t expand(const uint64_t* recs, uint64_t n_recs, int K, uint64_t* out, uint64_t cap) {
    constexpr int RW = (NW == 2 ? 5 : 7) + 1;
    const pg::Kmer<NW> filter = pg::kmer_filter<NW>(K);
    uint64_t n = 0;
    bool full = false;
    for (uint64_t i = 0; i < n_recs; i++)
        pg::skm_expand_record<NW>(recs + i * RW, K, filter, [&](const pg::Kmer<NW>& key, int left, int right, uint64_t ord) {
            if (n >= cap) { full = true; return; }
            uint64_t* o = out + n * (NW + 3);
            for (int w = 0; w < NW; w++) o[w] = key.w[w];
            o[NW] = (uint64_t)left; o[NW + 1] = (uint64_t)right; o[NW + 2] = ord;
            n++;
        });
    if (full) { pg_set_error("pg_host_skm_expand: output array too small"); return -1; }
    return (int64_t)n;
}

}  // namespace

extern "C" int64_t pg_host_skm_cut(const uint64_t* packed, uint64_t n_reads, uint32_t read_len, int K, int mer127, int log2_parts, uint64_t ord_base,
                                   int n_owners, uint64_t* records_out, uint64_t* tags_out, uint64_t capacity) {
    if (!packed || !records_out || !tags_out || n_owners < 1 || n_owners > 256 || (int)read_len < K + 1) { pg_set_error("pg_host_skm_cut: bad argument"); return -1; }
    return mer127 ? cut<4>(packed, n_reads, read_len, K, log2_parts, ord_base, n_owners, records_out, tags_out, capacity)
                  : cut<2>(packed, n_reads, read_len, K, log2_parts, ord_base, n_owners, records_out, tags_out, capacity);
}

extern "C" int64_t pg_host_skm_expand(const uint64_t* records, uint64_t n_records, int K, int mer127, uint64_t* out, uint64_t capacity) {
    if ((!records && n_records) || !out) { pg_set_error("pg_host_skm_expand: bad argument"); return -1; }
    return mer127 ? expand<4>(records, n_records, K, out, capacity) : expand<2>(records, n_records, K, out, capacity);
}

// host twin of the regroup's grouping step (exchange.hip: rg_count / rg_scatter): the rank every record goes to -- the owner
// of its reference set, set s -> rank s mod n_ranks -- the counts per destination, and the records grouped by destination
// (order within a destination as they lay); what the gloo tests move through torch.distributed
extern "C" int pg_host_regroup_plan(const uint64_t* records, uint64_t n_records, int rec_words, int n_ranks, uint64_t* counts_out, uint64_t* grouped_out) {
    if ((!records && n_records) || !counts_out || !grouped_out || rec_words < 3 || rec_words > 6 || n_ranks < 1 || n_ranks > 256) { pg_set_error("pg_host_regroup_plan: bad argument"); return PG_EINVAL; }
    for (int q = 0; q < n_ranks; q++) counts_out[q] = 0;
    for (uint64_t i = 0; i < n_records; i++) counts_out[(records[i * rec_words + rec_words - 1] >> PG_ORD_BITS) % (uint64_t)n_ranks]++;
    uint64_t at[256], run = 0;
    for (int q = 0; q < n_ranks; q++) { at[q] = run; run += counts_out[q]; }
    for (uint64_t i = 0; i < n_records; i++) {
        const uint64_t d = (records[i * rec_words + rec_words - 1] >> PG_ORD_BITS) % (uint64_t)n_ranks;
        for (int w = 0; w < rec_words; w++) grouped_out[at[d] * rec_words + w] = records[i * rec_words + w];
        at[d]++;
    }
    return PG_OK;
}
