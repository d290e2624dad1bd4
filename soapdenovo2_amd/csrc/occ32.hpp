// occ32.hpp -- one k-mer occurrence out of a super-k-mer record, in 32-bit arithmetic (host + device).
//
// Same result as canonical_occurrence (extract.hpp; chopKmer4read, standardPregraph/prlHashReads.c:198-257) for
// position p of a record: the canonical k-mer, and its left / right neighbour base in canonical orientation.  Written
// for the counting kernel (partition_kernels.hip, skm_count_kernel): the GPU's vector ALU is 32 bits wide, so the
// record's bases are kept as a string of dwords (first base in the top bits of dword 0) and everything is funnel
// shifts (v_alignbit_b32), bit reversals and selects -- no 64-bit shifts by run-time amounts, no 64-bit multiplies.
//
// One window of ND = 2 NW + 1 dwords ENDING behind the k-mer's right neighbour is cut from the string:
//
//      ... | prev | k-mer (2K bits) | next |            <- bit E = 2 (p + K + 1) of the string
//
// so that `next` and the k-mer sit at fixed bit positions of the window, `prev` at a position that depends on K alone
// (wave-uniform), and the reverse complement is the bit-reversed window moved down by a K-only amount.
#pragma once
#include "kmer.hpp"

namespace pg {

// ({hi, lo} >> sh) & 0xffffffff, sh taken modulo 32 (v_alignbit_b32)
PG_HD uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    sh &= 31u;
    return sh ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo;
#endif
}

// reverse the 16 bases of a dword and complement them (code ^ 2)
PG_HD uint32_t rc_dword(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t y = __builtin_bitreverse32(x);            // base order reversed, the two bits of a base swapped
    return (((y >> 1) & 0x55555555u) | ((y << 1) & 0xAAAAAAAAu)) ^ 0xAAAAAAAAu;
#else
    uint32_t y = ((x & 0x33333333u) << 2) | ((x >> 2) & 0x33333333u);
    y = ((y & 0x0F0F0F0Fu) << 4) | ((y >> 4) & 0x0F0F0F0Fu);
    return __builtin_bswap32(y) ^ 0xAAAAAAAAu;
#endif
}

// K-only constants of the extraction (wave-uniform; built once on the host)
struct OccConst {
    uint32_t flt[8];        // the 2K-bit filter as dwords, most significant first (2 NW of them used)
    int prev_q, prev_r;     // `prev` = (window dword [ND - 1 - prev_q] >> prev_r) & 3
    int rc_sd, rc_sb;       // the reversed register moves down by 32 rc_sd + rc_sb bits
};
constexpr OccConst occ_const(int K, int nw) {            // (constexpr: a kernel instantiated for one K folds the switches of occ_extract away)
    OccConst c{};
    int bits = 2 * K;
    for (int i = 2 * nw - 1; i >= 0; i--) { c.flt[i] = bits >= 32 ? 0xFFFFFFFFu : (bits > 0 ? ((1u << bits) - 1u) : 0u); bits -= 32; }
    for (int i = 2 * nw; i < 8; i++) c.flt[i] = 0;
    c.prev_q = (2 * K + 2) >> 5; c.prev_r = (2 * K + 2) & 31;
    const int sft = 64 * nw - 2 * K;
    c.rc_sd = sft >> 5; c.rc_sb = sft & 31;
    return c;
}

template <int N2, int SD>
PG_HD void occ_shr_dwords(const uint32_t (&R)[N2], uint32_t sb, uint32_t (&o)[N2]) {
#pragma unroll
    for (int i = 0; i < N2; i++) {
        const uint32_t hi = (i - SD - 1 >= 0) ? R[(i - SD - 1 >= 0) ? i - SD - 1 : 0] : 0u;
        const uint32_t lo = (i - SD >= 0) ? R[(i - SD >= 0) ? i - SD : 0] : 0u;
        o[i] = alignbit32(hi, lo, sb);
    }
}

// pay: the record's bases as dwords in string order; readable from pay[-(2 NW + 2)] to one dword past the last base.
// p = position of the k-mer's first base.  f / rc: forward k-mer and its reverse complement, right-aligned, dwords most
// significant first.  prev / next: the bases at p - 1 and p + K (garbage where the record has none: the caller knows).
template <int NW>
PG_HD void occ_extract(const uint32_t* pay, int p, int K, const OccConst& oc, uint32_t (&f)[2 * NW], uint32_t (&rc)[2 * NW],
                       uint32_t& prev, uint32_t& next) {
    constexpr int N2 = 2 * NW, ND = N2 + 1;
    const int E = 2 * (p + K) + 2;
    const int S = E - 32 * ND;                                  // first bit of the window (negative near the record's start)
    const int j = (S - 1) >> 5;                                 // window = bits [S, S + 32 ND) = dwords j .. j + ND at offset 1..32
    const uint32_t sh = (uint32_t)(32 - (S - 32 * j)) & 31u;
    uint32_t D[ND + 1];
#pragma unroll
    for (int i = 0; i <= ND; i++) D[i] = pay[j + i];
    uint32_t Y[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) Y[i] = alignbit32(D[i], D[i + 1], sh);
    next = Y[ND - 1] & 3u;
#pragma unroll
    for (int i = 0; i < N2; i++) f[i] = alignbit32(Y[i], Y[i + 1], 2u) & oc.flt[i];
    uint32_t pv = Y[0];
    switch (oc.prev_q) {                                          // wave-uniform
        case 0: pv = Y[ND - 1]; break;
        case 1: pv = Y[ND - 2]; break;
        case 2: pv = Y[ND - 3]; break;
        case 3: pv = Y[ND - 4]; break;
        case 4: pv = Y[ND - 5]; break;
        default:
            if (NW == 4) {
                switch (oc.prev_q) {
                    case 5: pv = Y[ND - 6 >= 0 ? ND - 6 : 0]; break;
                    case 6: pv = Y[ND - 7 >= 0 ? ND - 7 : 0]; break;
                    case 7: pv = Y[ND - 8 >= 0 ? ND - 8 : 0]; break;
                    default: pv = Y[0]; break;
                }
            }
            break;
    }
    prev = (pv >> oc.prev_r) & 3u;
    uint32_t R[N2];
#pragma unroll
    for (int i = 0; i < N2; i++) R[i] = rc_dword(f[N2 - 1 - i]);
    switch (oc.rc_sd) {                                           // wave-uniform
        case 0: occ_shr_dwords<N2, 0>(R, (uint32_t)oc.rc_sb, rc); break;
        case 1: occ_shr_dwords<N2, 1>(R, (uint32_t)oc.rc_sb, rc); break;
        case 2: occ_shr_dwords<N2, 2>(R, (uint32_t)oc.rc_sb, rc); break;
        case 3: occ_shr_dwords<N2, 3>(R, (uint32_t)oc.rc_sb, rc); break;
        case 4: occ_shr_dwords<N2, 4>(R, (uint32_t)oc.rc_sb, rc); break;
        case 5: occ_shr_dwords<N2, 5>(R, (uint32_t)oc.rc_sb, rc); break;
        case 6: occ_shr_dwords<N2, 6>(R, (uint32_t)oc.rc_sb, rc); break;
        default: occ_shr_dwords<N2, 7>(R, (uint32_t)oc.rc_sb, rc); break;
    }
}

// a < b, dwords most significant first (KmerSmaller, kmer.c:608-629)
template <int N2>
PG_HD bool occ_less(const uint32_t (&a)[N2], const uint32_t (&b)[N2]) {
    bool lt = false, eq = true;
#pragma unroll
    for (int i = 0; i < N2; i += 2) {
        const uint64_t x = ((uint64_t)a[i] << 32) | a[i + 1], y = ((uint64_t)b[i] << 32) | b[i + 1];
        lt = lt || (eq && x < y);
        eq = eq && x == y;
    }
    return lt;
}

// Slot hash of the canonical k-mer (free design: only this engine's LDS sets use it).  32-bit multiplies only.
template <int N2>
PG_HD uint32_t occ_hash(const uint32_t (&c)[N2]) {
    uint32_t x = c[N2 - 1], y = c[N2 - 2];
#pragma unroll
    for (int i = N2 - 3; i >= 0; i -= 2) {
        const uint32_t ra = (uint32_t)(5 * (N2 - i) + 3) & 31u, rb = (uint32_t)(7 * (N2 - i) + 2) & 31u;
        x ^= alignbit32(c[i], c[i], ra);                            // rotate
        if (i - 1 >= 0) y ^= alignbit32(c[i - 1 >= 0 ? i - 1 : 0], c[i - 1 >= 0 ? i - 1 : 0], rb);
    }
    uint32_t h = (x * 0x85EBCA6Bu) ^ (y * 0xC2B2AE35u);
    h ^= h >> 15;
    h *= 0x9E3779B1u;
    h ^= h >> 13;
    return h;
}

// the canonical k-mer as 63-bit key words for the LDS set (skm.hpp, key63_from_kmer), from its dwords
template <int NW>
PG_HD void occ_key63(const uint32_t (&c)[2 * NW], uint64_t (&kw)[NW == 2 ? 2 : 5]) {
    constexpr int N2 = 2 * NW, KW = NW == 2 ? 2 : 5;
    // bit b of the value lives in dword N2 - 1 - b / 32; word i = bits [63 i, 63 i + 63)
#pragma unroll
    for (int i = 0; i < KW; i++) {
        const int lo = 63 * i, d = lo >> 5, s = lo & 31;           // first dword (from the least significant end), bit offset
        auto dw = [&](int k) -> uint32_t { return (k >= 0 && k < N2) ? c[(k >= 0 && k < N2) ? N2 - 1 - k : 0] : 0u; };
        const uint32_t w0 = alignbit32(dw(d + 1), dw(d), (uint32_t)s);
        const uint32_t w1 = alignbit32(dw(d + 2), dw(d + 1), (uint32_t)s) & 0x7FFFFFFFu;
        kw[i] = ((uint64_t)w1 << 32) | w0;
    }
}

}  // namespace pg
