// kmer.hpp -- k-mer word arithmetic shared by the host pipeline and the HIP kernels.
//
// A k-mer is NW 64-bit words, w[0] most significant, first base in the most significant used bits and the
// last base in bits 1:0 of w[NW-1] -- the same value the reference keeps in Kmer{high,low} (NW = 2, the
// "63mer" binary) or Kmer{high1,low1,high2,low2} (NW = 4, the "127mer" binary); standardPregraph/inc/def.h:46-56.
// Base codes A0 C1 T2 G3, complement = code ^ 2 (inc/def.h:39-42).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PG_HD __host__ __device__ __forceinline__
#else
#define PG_HD inline
#endif

namespace pg {

template <int NW>
struct Kmer {
    uint64_t w[NW];
};

template <int NW>
PG_HD bool kmer_eq(const Kmer<NW>& a, const Kmer<NW>& b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < NW; i++) e = e && (a.w[i] == b.w[i]);
    return e;
}

// a < b, most significant word first (KmerSmaller, standardPregraph/kmer.c:608-629)
template <int NW>
PG_HD bool kmer_less(const Kmer<NW>& a, const Kmer<NW>& b) {
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (!decided && a.w[i] != b.w[i]) {
            lt = a.w[i] < b.w[i];
            decided = true;
        }
    }
    return lt;
}

// low 2K bits set (createFilter, kmer.c:738-758)
template <int NW>
PG_HD Kmer<NW> kmer_filter(int K) {
    Kmer<NW> f;
    int bits = 2 * K;
#pragma unroll
    for (int i = NW - 1; i >= 0; i--) {
        f.w[i] = bits >= 64 ? ~0ULL : (bits > 0 ? ((1ULL << bits) - 1) : 0ULL);
        bits -= 64;
    }
    return f;
}

// reverse the 32 two-bit groups of x
PG_HD uint64_t rev2bit(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __builtin_bitreverse64(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
#else
    x = ((x & 0x3333333333333333ULL) << 2) | ((x >> 2) & 0x3333333333333333ULL);
    x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL);
    return __builtin_bswap64(x);
#endif
}

// logical right shift of the NW-word value by d bits, 0 <= d < 64*NW, written without runtime-indexed
// arrays so the device version stays in registers
template <int NW>
PG_HD Kmer<NW> kmer_shr(Kmer<NW> a, int d) {
#pragma unroll
    for (int step = 0; step < NW - 1; step++) {
        if (d >= 64) {
#pragma unroll
            for (int i = NW - 1; i > 0; i--) a.w[i] = a.w[i - 1];
            a.w[0] = 0;
            d -= 64;
        }
    }
    if (d > 0) {
#pragma unroll
        for (int i = NW - 1; i > 0; i--) a.w[i] = (a.w[i] >> d) | (a.w[i - 1] << (64 - d));
        a.w[0] >>= d;
    }
    return a;
}

// reverse complement of a right-aligned `len`-mer (reverseComplement, kmer.c:819-855 / 532-591)
template <int NW>
PG_HD Kmer<NW> kmer_rc(const Kmer<NW>& a, int len) {
    Kmer<NW> r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.w[NW - 1 - i] = rev2bit(a.w[i] ^ 0xAAAAAAAAAAAAAAAAULL);
    return kmer_shr<NW>(r, 64 * NW - 2 * len);
}

// append base ch and drop the first base (nextKmer, kmer.c:696-702)
template <int NW>
PG_HD Kmer<NW> kmer_next(Kmer<NW> a, int ch, const Kmer<NW>& filter) {
#pragma unroll
    for (int i = 0; i < NW - 1; i++) a.w[i] = ((a.w[i] << 2) | (a.w[i + 1] >> 62)) & filter.w[i];
    a.w[NW - 1] = ((a.w[NW - 1] << 2) & filter.w[NW - 1]) | (uint64_t)ch;
    return a;
}

template <int NW>
PG_HD int kmer_last(const Kmer<NW>& a) { return (int)(a.w[NW - 1] & 3); }      // lastCharInKmer, kmer.c:720

template <int NW>
PG_HD int kmer_first(const Kmer<NW>& a, int K) {                                // firstCharInKmer, kmer.c:724
    int bit = 2 * (K - 1);
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < NW; i++)
        if (NW - 1 - bit / 64 == i) v = a.w[i];
    return (int)((v >> (bit % 64)) & 3);
}

// reverseComplement(word, K + 1) exactly as the reference computes it for the (K+1)-mer of a length-1 edge
// (node2edge.c:485, prlRead2path.c:694).  The 127-mer binary passes the length through a `char` (kmer.c:532): for
// K = 127 the length 128 wraps to -128, takes the `seq_size < 32` exit and only the lowest word is complemented and
// reversed (the shift count 64 - (-256) = 320 is applied modulo 64 by the hardware, i.e. not at all).
template <int NW>
PG_HD Kmer<NW> rc_plus(const Kmer<NW>& word, int K) {
    if (NW == 4 && K + 1 >= 128) {
        Kmer<NW> r = word;
        r.w[NW - 1] = rev2bit(word.w[NW - 1] ^ 0xAAAAAAAAAAAAAAAAULL);
        return r;
    }
    return kmer_rc<NW>(word, K + 1);
}
// KmerPlus (kmer.c:690-694): append one base without masking
template <int NW>
PG_HD Kmer<NW> kmer_plus(Kmer<NW> a, int ch) {
    for (int i = 0; i < NW - 1; i++) a.w[i] = (a.w[i] << 2) | (a.w[i + 1] >> 62);
    a.w[NW - 1] = (a.w[NW - 1] << 2) | (uint64_t)ch;
    return a;
}

// ---- node counter word: the reference's two 32-bit words of kmer_t (inc/newhash.h:77-102) as one u64,
// A in the low half, B in the high half.
//   A = l_links (4 x 6 bit, index = base code preceding the canonical k-mer) | covs << 24
//   B = r_links (4 x 6 bit) | linear << 24 | deleted << 25 | checked << 26 | single << 27 | twin << 28 | inEdge << 30
constexpr uint32_t B_LINEAR = 1u << 24;
constexpr uint32_t B_DELETED = 1u << 25;
// the reference's `checked` bit, never set by pregraph: scratch for the host tip scan ("this scan changed the node");
// must be 0 in every node outside Graph::tip_scan -- asserted before the nodes are uploaded or written out
constexpr uint32_t B_TOUCHED = 1u << 26;
constexpr uint32_t B_SINGLE = 1u << 27;
constexpr int B_TWIN_SHIFT = 28;
constexpr int B_INEDGE_SHIFT = 30;

// set_new_kmer (newhash.c:123-140): counters for the first occurrence; left/right = 4 means "none"
PG_HD uint64_t node_first(int left, int right) {
    uint32_t A = 1u << 24, B = B_SINGLE;
    if (left < 4) A |= 1u << (6 * left);
    if (right < 4) B |= 1u << (6 * right);
    return (uint64_t)A | ((uint64_t)B << 32);
}

// update_kmer + single = 0 (newhash.c:74-106, 510-511): saturating 6-bit arc counters, 8-bit total
PG_HD uint64_t node_update(uint64_t cnt, int left, int right) {
    uint32_t A = (uint32_t)cnt, B = (uint32_t)(cnt >> 32);
    if (left < 4 && ((A >> (6 * left)) & 63u) < 63u) A += 1u << (6 * left);
    if (right < 4 && ((B >> (6 * right)) & 63u) < 63u) B += 1u << (6 * right);
    if ((left < 4 || right < 4) && (A >> 24) < 255u) A += 1u << 24;
    B &= ~B_SINGLE;
    return (uint64_t)A | ((uint64_t)B << 32);
}

// ---- hash_kmer (hashFunction.c:123-158): CRC-32 (reflected 0xEDB88320, init 0, final xor) over the raw
// struct bytes, returned as `int` by the reference and therefore sign-extended before `% thrd_num`.
PG_HD uint32_t crc32_table_entry(uint32_t i) {
    uint32_t c = i;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    return c;
}

template <int NW, typename Table>
PG_HD uint32_t kmer_crc32(const Kmer<NW>& a, const Table& tab) {
    uint32_t crc = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
#pragma unroll
        for (int b = 0; b < 8; b++) crc = tab[(crc ^ (uint32_t)(a.w[i] >> (8 * b))) & 0xff] ^ (crc >> 8);
    }
    return crc ^ 0xffffffffu;
}

// The same CRC four bytes a step ("slicing by 4"): tab4[k][i] = CRC of byte i followed by k zero bytes, so a 32-bit chunk
// costs four independent look-ups instead of four dependent ones (the chain per k-mer shrinks from 8 NW to 2 NW steps).
// tab4 = [4][256], tab4[0] = the plain table.
PG_HD uint32_t crc32_slice_entry(int k, uint32_t i) {
    uint32_t c = crc32_table_entry(i);
    for (int q = 0; q < k; q++) c = (c >> 8) ^ crc32_table_entry(c & 0xff);
    return c;
}
template <int NW, typename Table4>
PG_HD uint32_t kmer_crc32_sliced(const Kmer<NW>& a, const Table4& t) {
    uint32_t crc = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t x = crc ^ (uint32_t)(a.w[i] >> (32 * h));
            crc = t[3 * 256 + (x & 0xff)] ^ t[2 * 256 + ((x >> 8) & 0xff)] ^ t[256 + ((x >> 16) & 0xff)] ^ t[x >> 24];
        }
    }
    return crc ^ 0xffffffffu;
}

// The same CRC with NO dependent look-ups (round 6).  With init 0 the CRC is linear over GF(2) in the message bits, so it is the xor of
// one table entry per NIBBLE of the key: nib[q][v] = the CRC of the message whose nibble q is v and whose other bits are 0 (q = 2 * byte
// position + (high nibble ? 1 : 0); bytes in the order kmer_crc32 walks them: word 0 first, low byte first).  16 NW tables of 16 entries
// = 1 KB a key word.  Against slicing by four: twice the look-ups, none of which waits for another (the sliced form is a chain of 2 NW
// rounds, each an LDS round trip), and a table's 16 entries lie in 16 different LDS banks, so the 64 lanes of a wave never collide (256
// entries of a sliced table do: conflict cycles were 0.77 of the LDS issue cycles of K2 at K = 127).  A nibble that is zero for every key
// (the bits above 2 K) contributes nothing and is skipped: `K` = the k-mer length (0: unknown, all nibbles).
PG_HD uint32_t crc32_nibble_entry(int n_bytes, int q, uint32_t v) {
    const int p = q >> 1;
    uint32_t c = crc32_table_entry((q & 1) ? v << 4 : v);
    for (int z = p + 1; z < n_bytes; z++) c = (c >> 8) ^ crc32_table_entry(c & 0xff);
    return c;
}
template <int NW, int K = 0, typename TableN>
PG_HD uint32_t kmer_crc32_nibbles(const Kmer<NW>& a, const TableN& t) {
    uint32_t crc = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        // word i holds the key's bits [64 (NW - 1 - i), 64 (NW - i)): of a k-mer of K bases only the lowest 2 K bits of the key are ever set
        const int live_bits = K ? (2 * K - 64 * (NW - 1 - i) > 64 ? 64 : 2 * K - 64 * (NW - 1 - i)) : 64;
        const uint32_t lo = (uint32_t)a.w[i], hi = (uint32_t)(a.w[i] >> 32);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (4 * j >= live_bits) continue;
            const uint32_t x = j < 8 ? lo : hi;
            crc ^= t[(16 * i + j) * 16 + ((x >> (4 * (j & 7))) & 15u)];
#if defined(__HIP_DEVICE_COMPILE__)
            // (eight look-ups in flight at a time, not all 16 NW: hoisted together they took 64 registers and the counting kernel spilled to scratch)
            if ((j & 7) == 7) asm volatile("" : "+v"(crc) :: "memory");
#endif
        }
    }
    return crc ^ 0xffffffffu;
}

// set picker: signext(crc) % P.  For a negative crc the 64-bit value is 2^64 - 2^32 + crc, so the result is
// ((2^64 - 2^32) % P + crc % P) % P; `bias` = (2^64 - 2^32) % P is precomputed by the caller.
PG_HD uint32_t set_of_crc(uint32_t crc, uint32_t P, uint32_t bias) {
    uint32_t r = crc % P;
    if (crc & 0x80000000u) {
        r += bias;
        if (r >= P) r -= P;
    }
    return r;
}
inline uint32_t set_bias(uint32_t P) { return (uint32_t)(0xFFFFFFFF00000000ULL % P); }

// device-table slot hash (free design; the reference layout is rebuilt by replay_layout / replay_streamed, host_graph.cpp)
template <int NW>
PG_HD uint64_t kmer_mix(const Kmer<NW>& a) {
    uint64_t h = 0x9E3779B97F4A7C15ULL;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        h ^= a.w[i];
        h *= 0xD6E8FEB86659FD93ULL;
        h ^= h >> 32;
    }
    h *= 0xD6E8FEB86659FD93ULL;
    h ^= h >> 29;
    return h;
}

}  // namespace pg
