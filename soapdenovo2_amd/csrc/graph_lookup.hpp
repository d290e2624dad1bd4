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
constexpr int SV_GEO = 5;             // words a set in SetsView::geo
struct SetsView {
    const uint64_t* geo;          // per set: first global slot, size, address of its slot 0, and the size's reciprocal (ModConst: v, s)
    const uint32_t* crc_tab;      // [256]
    uint32_t P, bias;
    int K;
};
PG_HD uint64_t sv_first(const SetsView& v, uint32_t s) { return v.geo[SV_GEO * s]; }
PG_HD uint64_t sv_size(const SetsView& v, uint32_t s) { return v.geo[SV_GEO * s + 1]; }
PG_HD uint64_t* sv_base(const SetsView& v, uint32_t s) { return (uint64_t*)(uintptr_t)v.geo[SV_GEO * s + 2]; }
// a word of a set image.  An address that was read from memory could be anything, and a load through it is a flat load -- which counts as LDS
// traffic too and is waited for together with everything else in flight; the images are device memory (this rank's or a peer's): say so
PG_HD uint64_t sv_word(const uint64_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) uint64_t*)p;
#else
    return *p;
#endif
}
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

// ---- x mod d for a divisor that is used many times (a set's size): division by an invariant integer with a precomputed
// reciprocal (Moeller & Granlund, "Improved division by invariant integers", 2011, algorithm 4: 2-by-1 division).  The
// compiler's 64-bit `%` is a ~200-instruction subroutine on the GPU, and the reference's home slot takes three of them for a
// 63-mer (six for a 127-mer); this form is one 64 x 64 -> 128 multiply, one 64-bit multiply and a few adds a step.
struct ModConst {
    uint64_t d;        // the divisor (0 < d < 2^63)
    uint64_t v;        // floor((2^128 - 1) / (d << s)) - 2^64
    uint32_t s;        // leading zeros of d: d << s has its top bit set
};
inline ModConst make_modconst(uint64_t d) {                   // host side; d = 0 is given an unused 1
    ModConst m;
    if (!d) d = 1;
    m.d = d;
    m.s = (uint32_t)__builtin_clzll(d);
    const uint64_t dn = d << m.s;
    m.v = (uint64_t)(~(unsigned __int128)0 / dn - ((unsigned __int128)1 << 64));
    return m;
}
PG_HD uint64_t mul_hi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// (hi * 2^64 + lo) mod d, for hi < d
PG_HD uint64_t rem128(uint64_t hi, uint64_t lo, const ModConst& m) {
    const uint64_t dn = m.d << m.s;
    const uint64_t u1 = m.s ? (hi << m.s) | (lo >> (64 - m.s)) : hi, u0 = lo << m.s;     // the dividend moved up with the divisor
    uint64_t q0 = m.v * u1, q1 = mul_hi64(m.v, u1);
    q0 += u0;
    q1 += u1 + (q0 < u0 ? 1u : 0u) + 1u;
    uint64_t r = u0 - q1 * dn;
    if (r > q0) r += dn;
    if (r >= dn) r -= dn;
    return r >> m.s;
}
// key mod size as the reference computes it (newhash.c:36-57)
template <int NW>
PG_HD uint64_t home_slot(const Kmer<NW>& k, const ModConst& m) {
    if (NW == 2)                                     // the 63-mer build: the exact 128-bit modulus
        return rem128(rem128(0, k.w[0], m), k.w[1], m);
    // the 127-mer build folds 32-bit chunks in 64-bit arithmetic: `t << 32` drops t's upper half once the size passes 2^32
    uint64_t t = rem128(0, k.w[0], m);
#pragma unroll
    for (int i = 1; i < NW; i++) {
        t = rem128(0, t << 32 | (k.w[i] >> 32), m);
        t = rem128(0, t << 32 | (k.w[i] & 0xffffffffULL), m);
    }
    return t;
}
// the node of a canonical key: its global slot (~0 when absent) and its address
template <int NW>
PG_HD uint64_t sv_find(const SetsView& v, const Kmer<NW>& key, uint64_t*& node) {
    const uint32_t s = set_of_crc(kmer_crc32<NW>(key, v.crc_tab), v.P, v.bias);
    const uint64_t size = sv_size(v, s);
    uint64_t* base = sv_base(v, s);
    uint64_t hc = home_slot<NW>(key, ModConst{size, v.geo[SV_GEO * s + 3], (uint32_t)v.geo[SV_GEO * s + 4]});
    for (uint64_t step = 0; step < size; step++) {
        uint64_t* nd = base + hc * (NW + 1);
        const uint64_t w0 = sv_word(nd);
        if (w0 == SV_EMPTY) break;
        bool eq = w0 == key.w[0];
#pragma unroll
        for (int i = 1; i < NW; i++) eq = eq && sv_word(nd + i) == key.w[i];
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
