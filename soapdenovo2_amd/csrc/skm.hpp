// skm.hpp -- super-k-mer partitioning shared by the HIP kernels of the partition engine and its CPU
// test harness (tests/emu_skm.cpp).  New design, nothing in the reference to mirror: the reference inserts
// every k-mer occurrence into a DRAM-resident table (prlHashReads.c:79-90, newhash.c:473-528), which on a
// GPU is bound by the chip's random-atomic rate.  Here occurrences are first grouped
// so that all occurrences of one canonical k-mer meet in one small partition that is counted in LDS:
//
//   partition(k-mer) = f(min over the canonical m-mers inside the k-mer of hash(m-mer))        (minimizer)
//
// which is a function of the canonical k-mer only (both strands contain the same canonical m-mers), and
// consecutive k-mers of a read mostly share it, so a read is cut into a few runs ("super-k-mers") that are
// stored as one fixed-size record each: a header word and the run's bases (2 bit/base, MSB first) with one
// flanking base on each side where the read has one.  A record is a miniature read: k-mer i of the run is
// position has_left + i of the record's bases, and the flank rules of chopKmer4read (prlHashReads.c:198-257)
// apply unchanged to it.
#pragma once
#include "kmer.hpp"

namespace pg {

// ---- geometry -------------------------------------------------------------------------------------------
struct SkmGeom {
    int K;          // k-mer size
    int m;          // minimizer length (m-mer)
    int w;          // m-mers per k-mer = K - m + 1
    int pw;         // payload words per record
    int rw;         // record words = 1 + pw
    int nmax;       // max k-mers per record
    int log2_parts; // partition ids = 1 << log2_parts (what the cursors, chunk lists and the counting grid are sized for)
    uint32_t part_mul;  // partitions in use <= 1 << log2_parts: pid = (hash * part_mul) >> 32.  1 << log2_parts: all of them (= the
                        // top log2_parts bits of the hash); fewer: every partition holds more, the ids above part_mul stay empty
};

// nw = words per k-mer of the build flavour (2 or 4): fixes the record size so kernels can keep it static
PG_HD SkmGeom skm_geometry(int K, int log2_parts, int nw = 2) {
    SkmGeom g;
    g.K = K;
    // 16 = the most that fits the 32-bit m-mer arithmetic.  It matters at scale: only ~1/25 of the canonical m-mers ever
    // win a window, and all sites that share a winning m-mer share a partition -- with m = 13 (33 M canonical 13-mers,
    // ~1.3 M effective minimizers) a 100 Mb genome already put ~3 sites into every used partition and overflowed the
    // LDS set of 6 % of them.
    g.m = K - 6 < 7 ? 7 : (K - 6 > 16 ? 16 : K - 6);
    g.w = K - g.m + 1;
    g.pw = nw == 2 ? 5 : 7;                       // 160 / 224 bases per record
    g.rw = 1 + g.pw;
    g.nmax = 32 * g.pw - (K - 1) - 2;             // leaves room for both flanks
    if (g.nmax > 127) g.nmax = 127;               // ... and the count fits 7 bits: the counting kernel keeps a copy count beside it
    g.log2_parts = log2_parts;
    g.part_mul = 1u << log2_parts;
    return g;
}

// header = first ordinal << 18 | n_kmers << 2 | has_left << 1 | has_right
constexpr int SKM_ORD_SHIFT = 18;
PG_HD uint64_t skm_header(uint64_t ord_first, int n, int has_left, int has_right) {
    return (ord_first << SKM_ORD_SHIFT) | ((uint64_t)n << 2) | ((uint64_t)has_left << 1) | (uint64_t)has_right;
}
PG_HD uint64_t skm_ord(uint64_t h) { return h >> SKM_ORD_SHIFT; }
PG_HD int skm_n(uint64_t h) { return (int)((h >> 2) & 0xFFFF); }
PG_HD int skm_has_left(uint64_t h) { return (int)((h >> 1) & 1); }
PG_HD int skm_has_right(uint64_t h) { return (int)(h & 1); }

// ---- minimizer value of the m-mer at base p of a packed read ------------------------------------------------
// 64 bits of the read starting at bit `bit` (MSB-first bit string); rd must be readable one word past
PG_HD uint64_t bits_at(const uint64_t* rd, int bit) {
    const int a = bit >> 6, o = bit & 63;
    uint64_t v = rd[a] << o;
    if (o) v |= rd[a + 1] >> (64 - o);
    return v;
}
// 32-bit arithmetic on purpose: the GPU has no 64-bit integer multiplier, and these run once per base.
PG_HD uint32_t rev2bit32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __builtin_bitreverse32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
#else
    x = ((x & 0x33333333u) << 2) | ((x >> 2) & 0x33333333u);
    x = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);
    return __builtin_bswap32(x);
#endif
}
// Order of the m-mers = order of this hash of the canonical m-mer (not lexicographic: poly-A would win every window).
// A bijection of the 32-bit value, so distinct m-mers never tie; one xor and one 32-bit multiply -- it runs once per base
// of every read, and only the order matters (the partition id re-hashes the winning value, skm_partition).
PG_HD uint32_t mmer_hash(uint32_t canon) { return (canon ^ 0x5BD1E995u) * 0x9E3779B1u; }
PG_HD uint32_t mmer_value(const uint64_t* rd, int p, int m) {          // m <= 16: an m-mer is at most 32 bits
    const uint32_t fwd = (uint32_t)(bits_at(rd, 2 * p) >> (64 - 2 * m));
    const uint32_t rc = rev2bit32(fwd ^ 0xAAAAAAAAu) >> (32 - 2 * m);
    return mmer_hash(fwd < rc ? fwd : rc);
}
PG_HD uint32_t skm_partition(uint32_t minval, uint32_t part_mul) {
    uint32_t x = minval * 0x85EBCA6Bu;            // the minimum of a window is a small number: spread it over all bits
    x ^= x >> 15;                                 // (one multiply alone leaves the partitions visibly less even: the largest
    x *= 0xC2B2AE35u;                             //  held 1052 / 3465 distinct k-mers instead of 820 / 2535 on the K = 63 / 127 fixtures)
    // the hash scaled to [0, part_mul): with part_mul = 2^k this is its top k bits
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(x, part_mul);
#else
    return (uint32_t)(((uint64_t)x * part_mul) >> 32);
#endif
}

// ---- cutting a read into runs ------------------------------------------------------------------------------
// Calls emit(j0, n, partition) for every maximal run of consecutive k-mers [j0, j0 + n) with the same partition;
// runs are also cut at every position that is a multiple of nmax, so "k-mer j starts a run" is a local property:
//     starts(j) = j == 0 || partition(j) != partition(j - 1) || j % nmax == 0
// (the tiled GPU kernel evaluates exactly this per lane; this serial walk is the reference formulation).  Sliding minimum without a queue: remember the (rightmost) minimum and its
// position, rescan the window only when it slides out -- expected once per ~ (w + 1) / 2 steps.
template <typename Emit>
PG_HD void skm_split_read(const uint64_t* rd, int len, const SkmGeom& g, Emit&& emit) {
    const int nk = len - g.K + 1;
    uint32_t minval = ~0u;
    int minpos = -1;
    for (int p = 0; p < g.w; p++) {
        const uint32_t v = mmer_value(rd, p, g.m);
        if (v <= minval) { minval = v; minpos = p; }
    }
    uint32_t cur = skm_partition(minval, g.part_mul);
    int j0 = 0;
    for (int j = 1; j < nk; j++) {
        const int pnew = j + g.w - 1;
        const uint32_t v = mmer_value(rd, pnew, g.m);
        if (minpos < j) {                       // the minimum left the window
            minval = ~0u;
            for (int p = j; p <= pnew; p++) {
                const uint32_t u = mmer_value(rd, p, g.m);
                if (u <= minval) { minval = u; minpos = p; }
            }
        } else if (v <= minval) { minval = v; minpos = pnew; }
        const uint32_t pid = skm_partition(minval, g.part_mul);
        if (pid != cur || j % g.nmax == 0) {
            emit(j0, j - j0, cur);
            j0 = j;
            cur = pid;
        }
    }
    emit(j0, nk - j0, cur);
}

// ---- records ---------------------------------------------------------------------------------------------
// Fill `rec[0..rw)` for the run [j0, j0 + n) of a read of `len` bases whose first k-mer has ordinal ord0.
template <int PW>
PG_HD void skm_make_record(const uint64_t* rd, int len, int j0, int n, uint64_t ord0, const SkmGeom& g, uint64_t* rec) {
    const int has_left = j0 > 0, has_right = (j0 + n - 1 + g.K) < len;
    const int b0 = j0 - has_left;
    const int nb = n + g.K - 1 + has_left + has_right;
    rec[0] = skm_header(ord0 + (uint64_t)j0, n, has_left, has_right);
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int first = 32 * i;               // first base of this payload word, relative to b0
        uint64_t v = 0;
        if (first < nb) {
            v = bits_at(rd, 2 * (b0 + first));
            const int valid = nb - first;       // bases of this word that belong to the run
            if (valid < 32) v &= ~0ULL << (64 - 2 * valid);
        }
        rec[1 + i] = v;
    }
}
PG_HD int skm_record_bases(uint64_t header, int K) {
    return skm_n(header) + K - 1 + skm_has_left(header) + skm_has_right(header);
}

// ---- expanding a record back into k-mer occurrences ---------------------------------------------------------
// One pass over the record's bases with a rolling forward and reverse-complement k-mer (nextKmer / prevKmer of
// kmer.c:696-718), calling f(canonical key, left, right, ordinal) for each of its n k-mers with the flank rules of
// chopKmer4read (prlHashReads.c:198-257).  Equivalent to canonical_occurrence(rec + 1, has_left + t, nb, ...) for
// t = 0..n-1, at a few instructions per base instead of a window extraction per k-mer.
template <int NW, typename F>
PG_HD void skm_expand_record(const uint64_t* rec, int K, const Kmer<NW>& filter, F&& f) {
    const uint64_t h = rec[0];
    const int hl = skm_has_left(h), nb = skm_record_bases(h, K);
    const uint64_t ord0 = skm_ord(h);
    const int topw = NW - 1 - (2 * (K - 1)) / 64, tops = (2 * (K - 1)) % 64;     // where the first base of a k-mer sits
    Kmer<NW> fwd, rc;
#pragma unroll
    for (int i = 0; i < NW; i++) { fwd.w[i] = 0; rc.w[i] = 0; }
    uint64_t cur = 0;
    int left_prev = 4;
    auto emit = [&](int right, uint64_t ord) {
        if (kmer_less<NW>(fwd, rc)) f(fwd, left_prev, right, ord);
        else f(rc, right < 4 ? (right ^ 2) : 4, left_prev < 4 ? (left_prev ^ 2) : 4, ord);
    };
    const int first_end = K - 1 + hl;                       // position of the last base of the record's first k-mer
    for (int p = 0; p < nb; p++) {
        if ((p & 31) == 0) cur = rec[1 + (p >> 5)];
        const int b = (int)(cur >> 62);
        cur <<= 2;
        if (p - 1 >= first_end) emit(b, ord0 + (uint64_t)(p - 1 - first_end));     // the k-mer that ended at p - 1
        uint64_t topword = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) if (i == topw) topword = fwd.w[i];
        left_prev = p >= K ? (int)((topword >> tops) & 3) : 4;                        // base p - K
        fwd = kmer_next<NW>(fwd, b, filter);
#pragma unroll
        for (int i = NW - 1; i > 0; i--) rc.w[i] = (rc.w[i] >> 2) | (rc.w[i - 1] << 62);
        rc.w[0] >>= 2;
#pragma unroll
        for (int i = 0; i < NW; i++) if (i == topw) rc.w[i] |= (uint64_t)(b ^ 2) << tops;
    }
    if (!skm_has_right(h)) emit(4, ord0 + (uint64_t)(nb - 1 - first_end));     // with a right flank the last base ends no k-mer
}

// ---- 63-bit key words -------------------------------------------------------------------------------------
// The in-LDS set claims a slot word by word with 64-bit CAS from the all-ones "empty" pattern, so no key word
// may be all ones: the 64*NW-bit k-mer is re-cut into KW words of 63 bits (KW = 2 for NW = 2: 2K <= 126 bits;
// KW = 5 for NW = 4).
template <int NW> struct KeyWords { static constexpr int value = NW == 2 ? 2 : 5; };

template <int NW>
struct Key63 { uint64_t w[KeyWords<NW>::value]; };

template <int NW>
PG_HD Key63<NW> key63_from_kmer(const Kmer<NW>& k) {
    constexpr int KW = KeyWords<NW>::value;
    const uint64_t M63 = (1ULL << 63) - 1;
    Key63<NW> r;
    // word i = bits [63 i, 63 i + 63) of the value, i = 0 least significant
#pragma unroll
    for (int i = 0; i < KW; i++) {
        const int lo = 63 * i;                    // bit offset from the LSB of the whole value
        const int wi = lo >> 6, bo = lo & 63;     // word index from the least significant end
        uint64_t v = 0;
        if (wi < NW) {
            v = k.w[NW - 1 - wi] >> bo;
            if (bo > 1 && wi + 1 < NW) v |= k.w[NW - 2 - wi] << (64 - bo);
        }
        r.w[i] = v & M63;
    }
    return r;
}
template <int NW>
PG_HD Kmer<NW> kmer_from_key63(const Key63<NW>& r) {
    constexpr int KW = KeyWords<NW>::value;
    Kmer<NW> k;
#pragma unroll
    for (int i = 0; i < NW; i++) k.w[i] = 0;
#pragma unroll
    for (int i = 0; i < KW; i++) {
        const int lo = 63 * i;
        const int wi = lo >> 6, bo = lo & 63;
        if (wi < NW) {
            k.w[NW - 1 - wi] |= r.w[i] << bo;
            if (bo > 1 && wi + 1 < NW) k.w[NW - 2 - wi] |= r.w[i] >> (64 - bo);
        }
    }
    return k;
}

}  // namespace pg
