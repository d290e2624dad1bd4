// tests/emu_skm.cpp -- CPU harness for the host/device-shared logic of the partition engine (csrc/skm.hpp,
// csrc/extract.hpp).  Test infrastructure only (built by tests/test_skm_emu.py with g++); it runs the same
// inline functions the HIP kernels call, serially: cut reads into super-k-mer records, group by partition,
// expand every record and reduce per partition.  Lets the record format, flank rules, ordinals and the
// "one canonical k-mer -> one partition" invariant be checked against the oracle without a GPU.
#include <stdint.h>
#include <string.h>
#include <map>
#include <vector>
#include <array>
#include "../soapdenovo2_amd/csrc/skm.hpp"
#include "../soapdenovo2_amd/csrc/extract.hpp"

using namespace pg;

template <int NW>
static int64_t run(const uint64_t* packed, int64_t n_reads, int len, int K, int log2_parts, uint64_t* out, int64_t cap,
                   int64_t* n_records, int64_t* max_part_distinct) {
    const SkmGeom g = skm_geometry(K, log2_parts, NW);
    const int wpr = (len + 31) / 32;
    std::vector<std::vector<uint64_t>> parts((size_t)1 << log2_parts);
    int64_t nrec = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const uint64_t* rd = packed + r * wpr;
        const uint64_t ord0 = (uint64_t)r * (uint64_t)(len - K + 1);
        int covered = 0;
        skm_split_read(rd, len, g, [&](int j0, int n, uint32_t pid) {
            if (j0 != covered || n < 1 || n > g.nmax) { nrec = -1000000; }
            covered += n;
            std::vector<uint64_t> rec(g.rw + 2, 0);
            skm_make_record<(NW == 2 ? 5 : 7)>(rd, len, j0, n, ord0, g, rec.data());
            auto& v = parts[pid];
            v.insert(v.end(), rec.begin(), rec.begin() + g.rw);
            nrec++;
        });
        if (covered != len - K + 1) return -1;
    }
    if (nrec < 0) return -2;
    *n_records = nrec;
    const Kmer<NW> filter = kmer_filter<NW>(K);
    struct Node { uint64_t cnt, ord; int part; };
    std::map<std::array<uint64_t, NW>, Node> all;
    int64_t maxd = 0;
    for (size_t p = 0; p < parts.size(); p++) {
        auto& v = parts[p];
        v.resize(v.size() + 4, 0);                       // readable padding for the window loads
        const size_t nr = (v.size() - 4) / g.rw;
        int64_t distinct_here = 0;
        for (size_t i = 0; i < nr; i++) {
            const uint64_t* rec = v.data() + i * g.rw;
            const uint64_t h = rec[0];
            const int n = skm_n(h), hl = skm_has_left(h), nb = skm_record_bases(h, K);
            int t = 0, bad = 0;
            skm_expand_record<NW>(rec, K, filter, [&](const Kmer<NW>& key, int left, int right, uint64_t ord) {
                // the rolling expansion must agree with the window extraction of extract.hpp position by position
                Occurrence occ;
                Kmer<NW> ref = canonical_occurrence<NW>(rec + 1, hl + t, nb, K, filter, occ);
                if (!kmer_eq<NW>(ref, key) || occ.left != left || occ.right != right || ord != skm_ord(h) + (uint64_t)t) bad = 1;
                t++;
                // the 63-bit re-cut used by the LDS set must round-trip
                Kmer<NW> back = kmer_from_key63<NW>(key63_from_kmer<NW>(key));
                if (!kmer_eq<NW>(back, key)) bad = 3;
                Key63<NW> k63 = key63_from_kmer<NW>(key);
                for (int q = 0; q < KeyWords<NW>::value; q++) if (k63.w[q] >> 63) bad = 4;
                std::array<uint64_t, NW> kk;
                for (int q = 0; q < NW; q++) kk[q] = key.w[q];
                auto it = all.find(kk);
                if (it == all.end()) { all[kk] = Node{node_first(left, right), ord, (int)p}; distinct_here++; }
                else {
                    if (it->second.part != (int)p) bad = 5;          // a canonical k-mer must live in ONE partition
                    it->second.cnt = node_update(it->second.cnt, left, right);
                    if (ord < it->second.ord) it->second.ord = ord;
                }
            });
            if (t != n) return -7;
            if (bad) return -10 - bad;
        }
        if (distinct_here > maxd) maxd = distinct_here;
    }
    *max_part_distinct = maxd;
    if ((int64_t)all.size() > cap) return -6;
    int64_t o = 0;
    for (auto& kv : all) {
        for (int q = 0; q < NW; q++) out[o * (NW + 2) + q] = kv.first[q];
        out[o * (NW + 2) + NW] = kv.second.cnt;
        out[o * (NW + 2) + NW + 1] = kv.second.ord;
        o++;
    }
    return o;
}

extern "C" int64_t emu_skm_count(const uint64_t* packed, int64_t n_reads, int len, int K, int mer127, int log2_parts, uint64_t* out,
                                 int64_t cap, int64_t* n_records, int64_t* max_part_distinct) {
    return mer127 ? run<4>(packed, n_reads, len, K, log2_parts, out, cap, n_records, max_part_distinct)
                  : run<2>(packed, n_reads, len, K, log2_parts, out, cap, n_records, max_part_distinct);
}
