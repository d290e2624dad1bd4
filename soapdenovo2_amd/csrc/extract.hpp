// extract.hpp -- k-mer occurrences out of a 2-bit packed read (host + device).
// Restates chopKmer4read (standardPregraph/prlHashReads.c:163-259) for one position: the reference rolls the
// forward and reverse-complement k-mers along the read; here position j is rebuilt directly from the words that
// cover bases [j, j + K), so every lane of a wavefront can take its own position.
#pragma once
#include "kmer.hpp"

namespace pg {

// ---- read access ---------------------------------------------------------------------------------------
PG_HD int read_base(const uint64_t* rd, int i) {
    return (int)((rd[i >> 5] >> (62 - 2 * (i & 31))) & 3);
}

// right-aligned k-mer of bases [j, j+K) of a packed read (first base in the most significant bits)
template <int NW>
PG_HD Kmer<NW> read_kmer(const uint64_t* rd, int j, int K, const Kmer<NW>& filter) {
    const int s = 2 * j, e = s + 2 * K;
    const int a = s >> 6;
    Kmer<NW + 1> v;
#pragma unroll
    for (int i = 0; i <= NW; i++) v.w[i] = rd[a + i];          // buffer is padded, always readable
    v = kmer_shr<NW + 1>(v, 64 * (NW + 1) - (e - 64 * a));
    Kmer<NW> k;
#pragma unroll
    for (int i = 0; i < NW; i++) k.w[i] = v.w[i + 1] & filter.w[i];
    return k;
}

struct Occurrence { int left, right; };

// canonical k-mer + flanking bases in canonical orientation (prlHashReads.c:198-257)
template <int NW>
PG_HD Kmer<NW> canonical_occurrence(const uint64_t* rd, int j, int len, int K,
                                                         const Kmer<NW>& filter, Occurrence& occ) {
    Kmer<NW> word = read_kmer<NW>(rd, j, K, filter);
    Kmer<NW> bal = kmer_rc<NW>(word, K);
    const int prev = j > 0 ? read_base(rd, j - 1) : 4;
    const int next = j < len - K ? read_base(rd, j + K) : 4;
    if (kmer_less<NW>(word, bal)) {
        occ.left = prev; occ.right = next;
        return word;
    }
    occ.left = next < 4 ? (next ^ 2) : 4;
    occ.right = prev < 4 ? (prev ^ 2) : 4;
    return bal;
}

// advance a forward / reverse-complement pair by one base b (nextKmer / prevKmer, kmer.c:696-718)
template <int NW>
PG_HD void kmer_roll(Kmer<NW>& word, Kmer<NW>& bal, int b, int K, const Kmer<NW>& filter) {
    word = kmer_next<NW>(word, b, filter);
    const int topw = NW - 1 - (2 * (K - 1)) / 64, tops = (2 * (K - 1)) % 64;
#pragma unroll
    for (int i = NW - 1; i > 0; i--) bal.w[i] = (bal.w[i] >> 2) | (bal.w[i - 1] << 62);
    bal.w[0] >>= 2;
#pragma unroll
    for (int i = 0; i < NW; i++) if (i == topw) bal.w[i] |= (uint64_t)(b ^ 2) << tops;
}

}  // namespace pg
