// dev_graph.hpp -- the graph stages that live on the device, written once over a Backend (backend.hpp):
//
//   layout_static   SURVEY.md App. C "K6": the slot every k-mer occupies in a static (-a) k-mer set, i.e. the replay of
//                   put_kmerset into a table that never grows (newhash.c:353-366,473-528; prlHashReads.c:369-390).
//   (tips: dev_tips.hpp)
//
// ---- layout_static ---------------------------------------------------------------------------------------------------
// With -a the reference sizes every set once and never rehashes (encap_kmerset only raises the load factor: SURVEY.md
// A.2), so a set's layout is plain first-come-first-served linear probing: the keys arrive in first-occurrence order and
// each takes the first empty slot at or after key mod size (newhash.c:487-528).  That is equivalent to one sweep over the
// slots in which every slot takes, among the keys whose home is at or before it and that are not placed yet, the one that
// arrived FIRST:
//   (by induction over the slots of a probe cluster) let x be the earliest-arrived pending key at slot j, home h <= j.  Every
//   slot in [h, j) went to a key that was pending there together with x and beat it, i.e. arrived before x -- so when x
//   arrived it found h .. j-1 taken and probed on to j; and j was still empty then, because whoever holds j in the end is a
//   pending key at j too and cannot have arrived before the earliest.
// Which slots are occupied does not depend on the order at all: sorted by home h_0 <= h_1 <= ..., the i-th occupied slot is
// q_i = max(h_i, q_{i-1} + 1) = i + max_{j<=i}(h_j - j) -- a prefix maximum.  So:
//   1. home slot of every key of the set (the reference's modulus: 128-bit for the 63-mer build, chained 32-bit chunks for
//      the 127-mer build), stable radix sort of (home, arrival rank) by home;
//   2. prefix maximum of h_j - j: occupied slots and the clusters (a cluster starts where the maximum rises);
//   3. a cluster that runs past the end of the table wraps to slot 0.  Then the table is rotated so that a slot that stays
//      empty comes last (the W keys that wrap take the first W empty slots from 0 on, so the (W+1)-th is one), which is a
//      rotation of the sorted array too, and 2. is done again in the rotated frame, where nothing wraps;
//   4. one lane per cluster sweeps its slots with a min-heap of arrival ranks (its storage is the cluster's own stretch of
//      one array: a heap never holds more than the cluster's keys) and writes the nodes into the set image.
// Records come sorted by (set, first ordinal) (pg_sort_records), so the arrival rank is the record index within its set.
#pragma once
#include <stdint.h>

#include "backend.hpp"
#include "graph_lookup.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

// result codes beside PG_OK / PG_E*: the caller falls back to the host replay
constexpr int K6_UNSUITED = 1;          // a set with >= 2^32 keys, or one that fills its pool completely

template <int NW>
struct K6Sweep {
    const uint64_t* hs;                 // homes, sorted (rotated frame)
    const uint32_t* is;                 // arrival ranks in the same order
    const long long* m;                 // prefix maximum of hs[j] - j
    uint32_t* heap;
    const uint64_t* rec;                // the set's records (NW + 2 words each), in arrival order
    uint64_t* nodes;                    // the set's slot 0 (NW + 1 words a slot)
    uint64_t n, S, origin;              // keys, slots, rotation: frame slot p is table slot (p + origin) mod S
    PG_HD void operator()(uint64_t j) const {
        if (j && m[j] <= m[j - 1]) return;                          // not the first key of a cluster
        uint32_t* hp = heap + j;
        uint64_t hn = 0, nxt = j, p = hs[j];
        for (;;) {
            while (nxt < n && hs[nxt] <= p) {                       // keys whose home is reached: into the heap
                uint64_t c = hn++;
                const uint32_t v = is[nxt++];
                while (c) { const uint64_t par = (c - 1) >> 1; if (hp[par] <= v) break; hp[c] = hp[par]; c = par; }
                hp[c] = v;
            }
            if (!hn) return;                                        // the next key starts its own cluster
            const uint32_t first = hp[0];
            const uint32_t last = hp[--hn];
            if (hn) {
                uint64_t c = 0;
                for (;;) {
                    uint64_t ch = 2 * c + 1;
                    if (ch >= hn) break;
                    if (ch + 1 < hn && hp[ch + 1] < hp[ch]) ch++;
                    if (hp[ch] >= last) break;
                    hp[c] = hp[ch]; c = ch;
                }
                hp[c] = last;
            }
            uint64_t slot = p + origin;
            if (slot >= S) slot -= S;
            const uint64_t* r = rec + (uint64_t)first * (NW + 2);
            uint64_t* nd = nodes + slot * (NW + 1);
#pragma unroll
            for (int w = 0; w <= NW; w++) nd[w] = r[w];
            p++;
        }
    }
};

// nodes: P sets of S slots back to back, every slot's first word preset to SV_EMPTY (the rest 0); records: device (backend)
// memory, sorted by (set, ordinal); per_set_count: host.  Returns PG_OK, K6_UNSUITED (nothing useful written) or PG_E*.
template <class BE, int NW>
int layout_static(BE& be, const uint64_t* records, const uint64_t* per_set_count, int P, uint64_t S, uint64_t* nodes) {
    constexpr int RW = NW + 2;
    uint64_t n_max = 0;
    for (int s = 0; s < P; s++) {
        if (per_set_count[s] >= S || per_set_count[s] >= 0xFFFFFFFFULL) return K6_UNSUITED;
        n_max = std::max(n_max, per_set_count[s]);
    }
    if (!n_max) return PG_OK;
    int bits = 1;
    while (bits < 64 && (S >> bits)) bits++;
    uint64_t* hk = be.template alloc<uint64_t>(n_max);
    uint64_t* hs = be.template alloc<uint64_t>(n_max);
    uint64_t* hr = be.template alloc<uint64_t>(n_max);      // homes in the rotated frame (a wrapping set only)
    uint32_t* iv = be.template alloc<uint32_t>(n_max);
    uint32_t* is = be.template alloc<uint32_t>(n_max);
    uint32_t* ir = be.template alloc<uint32_t>(n_max);
    long long* v = be.template alloc<long long>(n_max);
    long long* m = be.template alloc<long long>(n_max);
    uint32_t* heap = be.template alloc<uint32_t>(n_max);
    unsigned long long* scal = be.template alloc<unsigned long long>(4);
    int rc = PG_OK;
    uint64_t first = 0;
    for (int s = 0; s < P && !be.error; s++) {
        const uint64_t n = per_set_count[s];
        const uint64_t* rec = records + first * RW;
        uint64_t* set_nodes = nodes + (uint64_t)s * S * (NW + 1);
        first += n;
        if (!n) continue;
        const ModConst mc = make_modconst(S);
        be.launch(n, [=] PG_LAMBDA(uint64_t i) {
            Kmer<NW> k;
#pragma unroll
            for (int w = 0; w < NW; w++) k.w[w] = rec[i * RW + w];
            hk[i] = home_slot<NW>(k, mc);
            iv[i] = (uint32_t)i;
        });
        be.sort_pairs(hk, hs, iv, is, n, bits);
        be.launch(n, [=] PG_LAMBDA(uint64_t j) { v[j] = (long long)hs[j] - (long long)j; });
        be.inclusive_max(v, m, n);
        long long m_last = 0;
        be.to_host(&m_last, m + (n - 1), 1);
        if (be.error) break;
        const uint64_t q_last = (uint64_t)((long long)(n - 1) + m_last);
        const uint64_t* hs_use = hs;
        const uint32_t* is_use = is;
        uint64_t origin = 0;
        if (q_last >= S) {
            // the last cluster wraps: W keys go on from slot 0 and take the first W empty slots there, so the (W+1)-th empty
            // slot e stays empty; frame = the table turned so that e comes last.  One lane: W is a handful of keys.
            const uint64_t W = q_last - (S - 1);
            be.launch(1, [=] PG_LAMBDA(uint64_t) {
                uint64_t seen = 0, prev_end = 0, e = 0;
                bool found = false;
                for (uint64_t j = 0; j < n && !found; j++) {
                    const uint64_t q = (uint64_t)((long long)j + m[j]);
                    if (q >= S) break;
                    const uint64_t gap = q - prev_end;                          // empty slots in [prev_end, q)
                    if (seen + gap >= W + 1) { e = prev_end + (W + 1 - seen) - 1; found = true; }
                    seen += gap;
                    prev_end = q + 1;
                }
                if (!found) e = prev_end + (W + 1 - seen) - 1;                     // behind the last cluster that stays below S
                const uint64_t o = e + 1 == S ? 0 : e + 1;
                uint64_t lo = 0, hi = n;                                        // first j with hs[j] >= o
                while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (hs[mid] >= o) hi = mid; else lo = mid + 1; }
                scal[0] = o; scal[1] = lo;
            });
            unsigned long long orot[2] = {0, 0};
            be.to_host(orot, scal, 2);
            if (be.error) break;
            origin = orot[0];
            const uint64_t r = orot[1] % n, o = origin;
            be.launch(n, [=] PG_LAMBDA(uint64_t j) {
                const uint64_t src = j + r >= n ? j + r - n : j + r;
                const uint64_t h = hs[src];
                hr[j] = h >= o ? h - o : h + S - o;
                ir[j] = is[src];
                v[j] = (long long)hr[j] - (long long)j;
            });
            be.inclusive_max(v, m, n);
            be.to_host(&m_last, m + (n - 1), 1);
            if (be.error) break;
            if ((uint64_t)((long long)(n - 1) + m_last) >= S) { rc = PG_EINVAL; be.error_text = "layout_static: the rotated frame still wraps"; break; }
            hs_use = hr; is_use = ir;
        }
        be.launch(n, K6Sweep<NW>{hs_use, is_use, m, heap, rec, set_nodes, n, S, origin});
    }
    be.sync();
    be.release(hk); be.release(hs); be.release(hr); be.release(iv); be.release(is); be.release(ir);
    be.release(v); be.release(m); be.release(heap); be.release(scal);
    if (be.error) return be.error;
    return rc;
}

}  // namespace pg
