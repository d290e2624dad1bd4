// graph_lookup.hpp -- the reference's k-mer-set lookup on a copy of the sets in device (or, for the CPU tests, host) memory.
// search_kmerset (newhash.c:277-318): set = signext(crc32(key)) % thrd_num (hashFunction.c:155), slot = key mod size
// (newhash.c:36-57), linear probing until the key or an empty slot.  Host + device (PG_HD), no LDS: the CRC table and the
// per-set geometry are read through the caches, which is all the graph stages need (they wait for random node probes).
#pragma once
#include <stdint.h>

#include "kmer.hpp"

namespace pg {

constexpr uint64_t SV_EMPTY = ~0ULL;      // first key word of an empty slot (K <= 63 / 127 leaves the top bits of word 0 clear)

// The k-mer sets as the graph stages see them.  A slot is NW + 1 words: key words, then A | B << 32.  Sets are numbered
// in the reference's order; "global slot" g = first[s] + slot, the position in the reference's scan order (set by set,
// slot by slot: node2edge.c:383-406, cutTipPreGraph.c:374-395).  Every set has its own base address, so the sets of one
// view may live in different allocations -- on different GPUs of one process, reached through peer mappings.
struct SetsView {
    const uint64_t* geo;          // per set: first global slot, size, address of its slot 0
    const uint32_t* crc_tab;      // [256]
    uint32_t P, bias;
    int K;
};
PG_HD uint64_t sv_first(const SetsView& v, uint32_t s) { return v.geo[3 * s]; }
PG_HD uint64_t sv_size(const SetsView& v, uint32_t s) { return v.geo[3 * s + 1]; }
PG_HD uint64_t* sv_base(const SetsView& v, uint32_t s) { return (uint64_t*)(uintptr_t)v.geo[3 * s + 2]; }
// the set a global slot lies in (binary search over <= 255 firsts), and the slot's address
PG_HD uint32_t sv_set_of_slot(const SetsView& v, uint64_t g) {
    uint32_t lo = 0, hi = v.P;                           // last s with first[s] <= g
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sv_first(v, mid) <= g) lo = mid; else hi = mid; }
    return lo;
}
template <int NW>
PG_HD uint64_t* sv_node(const SetsView& v, uint64_t g) {
    const uint32_t s = sv_set_of_slot(v, g);
    return sv_base(v, s) + (g - sv_first(v, s)) * (NW + 1);
}

// ((r << 32) | chunk) mod d for r < d, exact for any d < 2^63
PG_HD uint64_t mod_step32(uint64_t r, uint32_t chunk, uint64_t d) {
    if (d <= 0x100000000ULL) return ((r << 32) | chunk) % d;
    for (int b = 31; b >= 0; b--) {
        r = (r << 1) | ((chunk >> b) & 1u);
        if (r >= d) r -= d;
    }
    return r;
}
template <int NW>
PG_HD uint64_t home_slot(const Kmer<NW>& k, uint64_t size) {
    if (NW == 2) {                                   // exact 128-bit modulus (newhash.c:36-57, 63-mer build)
        uint64_t r = k.w[0] % size;
        r = mod_step32(r, (uint32_t)(k.w[1] >> 32), size);
        return mod_step32(r, (uint32_t)k.w[1], size);
    }
    uint64_t t = k.w[0] % size;                      // the 127-mer build folds 32-bit chunks in 64-bit arithmetic
#pragma unroll
    for (int i = 1; i < NW; i++) {
        t = (t << 32 | (k.w[i] >> 32)) % size;
        t = (t << 32 | (k.w[i] & 0xffffffffULL)) % size;
    }
    return t;
}

// the node of a canonical key: its global slot (~0 when absent) and its address
template <int NW>
PG_HD uint64_t sv_find(const SetsView& v, const Kmer<NW>& key, uint64_t*& node) {
    const uint32_t s = set_of_crc(kmer_crc32<NW>(key, v.crc_tab), v.P, v.bias);
    const uint64_t size = sv_size(v, s);
    uint64_t* base = sv_base(v, s);
    uint64_t hc = home_slot<NW>(key, size);
    for (uint64_t step = 0; step < size; step++) {
        uint64_t* nd = base + hc * (NW + 1);
        const uint64_t w0 = nd[0];
        if (w0 == SV_EMPTY) break;
        bool eq = w0 == key.w[0];
#pragma unroll
        for (int i = 1; i < NW; i++) eq = eq && nd[i] == key.w[i];
        if (eq) { node = nd; return sv_first(v, s) + hc; }
        if (++hc == size) hc = 0;
    }
    node = nullptr;
    return ~0ULL;
}

// one step of a walk: the node of the walk-oriented k-mer `word`
template <int NW>
PG_HD bool sv_step(const SetsView& v, const Kmer<NW>& word, uint64_t& slot, uint64_t*& node, bool& smaller) {
    const Kmer<NW> bal = kmer_rc<NW>(word, v.K);
    smaller = !kmer_less<NW>(bal, word);
    slot = sv_find<NW>(v, smaller ? word : bal, node);
    return slot != ~0ULL;
}

PG_HD int count_arcs24(uint32_t w24) { return (int)((w24 & 63u) != 0) + (int)(((w24 >> 6) & 63u) != 0) + (int)(((w24 >> 12) & 63u) != 0) + (int)(((w24 >> 18) & 63u) != 0); }
// the single outgoing base of a linear node in walk orientation (only_out)
PG_HD int linear_out_ab(uint64_t ab, bool smaller) {
    const uint32_t A = (uint32_t)ab, B = (uint32_t)(ab >> 32);
    int ch;
    if (smaller) { for (ch = 0; ch < 4; ch++) if ((B >> (6 * ch)) & 63) break; return ch; }
    for (ch = 0; ch < 4; ch++) if ((A >> (6 * ch)) & 63) break;
    return ch ^ 2;
}

}  // namespace pg
