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
#include "../soapdenovo2_amd/csrc/occ32.hpp"

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
            // the counting kernel's 32-bit extraction (occ32.hpp) on the record laid out as it is in LDS: dwords in string
            // order, readable a window's length before the payload (here: junk, as the previous record would be)
            constexpr int N2 = 2 * NW, PADD = 2 * NW + 4;
            std::vector<uint32_t> dws(PADD + 2 * (size_t)g.pw + 4, 0xDEADBEEFu);
            for (int w = 0; w < g.pw; w++) { dws[PADD + 2 * w] = (uint32_t)(rec[1 + w] >> 32); dws[PADD + 2 * w + 1] = (uint32_t)rec[1 + w]; }
            const OccConst oc = occ_const(K, NW);
            skm_expand_record<NW>(rec, K, filter, [&](const Kmer<NW>& key, int left, int right, uint64_t ord) {
                // the rolling expansion must agree with the window extraction of extract.hpp position by position
                Occurrence occ;
                Kmer<NW> ref = canonical_occurrence<NW>(rec + 1, hl + t, nb, K, filter, occ);
                if (!kmer_eq<NW>(ref, key) || occ.left != left || occ.right != right || ord != skm_ord(h) + (uint64_t)t) bad = 1;
                {
                    uint32_t f[N2], rc[N2], prev, next, c[N2];
                    occ_extract<NW>(dws.data() + PADD, hl + t, K, oc, f, rc, prev, next);
                    const bool lt = occ_less<N2>(f, rc);
                    const bool hasprev = hl + t > 0, hasnext = t < n - 1 + skm_has_right(h);
                    const int l32 = lt ? (hasprev ? (int)prev : 4) : (hasnext ? (int)(next ^ 2) : 4);
                    const int r32 = lt ? (hasnext ? (int)next : 4) : (hasprev ? (int)(prev ^ 2) : 4);
                    for (int q = 0; q < N2; q++) c[q] = lt ? f[q] : rc[q];
                    for (int q = 0; q < NW; q++) if (key.w[q] != (((uint64_t)c[2 * q] << 32) | c[2 * q + 1])) bad = 6;
                    if (l32 != left || r32 != right) bad = 7;
                    uint64_t kw[KeyWords<NW>::value];
                    occ_key63<NW>(c, kw);
                    Key63<NW> want = key63_from_kmer<NW>(key);
                    for (int q = 0; q < KeyWords<NW>::value; q++) if (kw[q] != want.w[q]) bad = 8;
                }
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

// ---- the tiled cutter (csrc/skm_tile.hpp) run serially: phases A, B, D, E over tiles of R reads, compared run for run
// and word for word with skm_split_read + skm_make_record.  Returns 0 or a negative code.
#include "../soapdenovo2_amd/csrc/skm_tile.hpp"

// word_off / lens = null: n_reads reads of `len` bases back to back; otherwise a ragged batch whose longest read has `len` bases (the
// rows of a tile are sized for it, a read's own length masks its segments: the RAGGED instantiation of skm_scatter_seg_kernel)
template <int NW, int S>
static int64_t tile_check(const uint64_t* packed, const uint64_t* word_off, const int32_t* lens, int64_t n_reads, int len, int K, int log2_parts, int R) {
    constexpr int PW = NW == 2 ? 5 : 7, RW = PW + 1;
    const SkmGeom g = skm_geometry(K, log2_parts, NW);
    const int wpr = (len + 31) / 32, kpr = len - K + 1, np = len - g.m + 1;
    const int wsd = (2 * wpr + 3) | 1, nseg = (kpr + S - 1) / S, nca = (np + 15) / 16, npad = (16 * nca) | 1;
    if (S > g.w && S != 7) return -100;
    int64_t checked = 0;
    uint64_t kb = 0;
    for (int64_t r0 = 0; r0 < n_reads; r0 += R) {
        const int nr = (int)std::min<int64_t>(R, n_reads - r0);
        std::vector<uint32_t> dw((size_t)nr * wsd, 0), v0((size_t)nr * npad, 0xABABABABu), pids((size_t)nr * kpr, 0), masks((size_t)nr * nseg, 0);
        std::vector<int> rlen(nr);
        std::vector<uint64_t> rkb(nr);
        std::vector<const uint64_t*> rd_of(nr);
        for (int r = 0; r < nr; r++) {
            rlen[r] = lens ? lens[r0 + r] : len;
            if (rlen[r] > len || rlen[r] < K + 1) return -107;
            rd_of[r] = packed + (word_off ? word_off[r0 + r] : (uint64_t)(r0 + r) * wpr);
            rkb[r] = kb; kb += (uint64_t)(rlen[r] - K + 1);
            for (int k = 0; k < wpr; k++) {
                const uint64_t wd = 32 * k < rlen[r] ? rd_of[r][k] : 0ULL;
                dw[(size_t)r * wsd + 2 * k] = (uint32_t)(wd >> 32); dw[(size_t)r * wsd + 2 * k + 1] = (uint32_t)wd;
            }
        }
        for (int t = 0; t < nr * nca; t++) {
            const int c = t / nr, r = t % nr;
            if (g.m == 16 && (t & 1)) tile_mmer_chunk<16>(dw.data() + (size_t)r * wsd, c, g.m, v0.data() + (size_t)r * npad);      // (both forms; every position is compared below)
            else tile_mmer_chunk(dw.data() + (size_t)r * wsd, c, g.m, v0.data() + (size_t)r * npad);
        }
        for (int r = 0; r < nr; r++)                                // phase A against the 64-bit formulation
            for (int p = 0; p < rlen[r] - g.m + 1; p++) if (v0[(size_t)r * npad + p] != mmer_value(rd_of[r], p, g.m)) return -101;
        for (int t = 0; t < nr * nseg; t++) {
            const int seg = t / nr, r = t % nr, j0 = seg * S, kpr_r = rlen[r] - K + 1, np_r = rlen[r] - g.m + 1, cnt = std::min(S, kpr_r - j0);
            if (cnt <= 0) { masks[(size_t)r * nseg + seg] = 0; continue; }
            uint32_t pid[S];
            masks[(size_t)r * nseg + seg] = tile_segment<S>(v0.data() + (size_t)r * npad, np_r, j0, cnt, g.w, g.nmax, g.part_mul, pid);
            for (int i = 0; i < cnt; i++) pids[(size_t)r * kpr + j0 + i] = pid[i];
            // the instantiations with the window length at compile time (what the kernel runs for K = 31 / 63 / 127): same bits, same ids
            uint32_t pid2[S];
            uint32_t mk2 = masks[(size_t)r * nseg + seg];
            bool have = true;
            if (g.w == 48 && S <= 48) mk2 = tile_segment<S, 48>(v0.data() + (size_t)r * npad, np_r, j0, cnt, g.w, g.nmax, g.part_mul, pid2);
            else if (g.w == 112 && S <= 112) mk2 = tile_segment<S, 112>(v0.data() + (size_t)r * npad, np_r, j0, cnt, g.w, g.nmax, g.part_mul, pid2);
            else if (g.w == 16 && S <= 16) mk2 = tile_segment<S, 16>(v0.data() + (size_t)r * npad, np_r, j0, cnt, g.w, g.nmax, g.part_mul, pid2);
            else have = false;
            if (have) {
                if (mk2 != masks[(size_t)r * nseg + seg]) return -105;
                for (int i = 0; i < cnt; i++) if (pid2[i] != pid[i]) return -106;
            }
        }
        for (int r = 0; r < nr; r++) {
            struct Run { int j0, n; uint32_t pid; };
            std::vector<Run> got, want;
            const int kpr_r = rlen[r] - K + 1;
            for (int seg = 0; seg < nseg; seg++) {
                uint32_t mk = masks[(size_t)r * nseg + seg];
                while (mk) {
                    const int i = __builtin_ffs((int)mk) - 1;
                    mk &= mk - 1;
                    const int j = seg * S + i, nxt = tile_next_start(masks.data() + (size_t)r * nseg, seg, nseg, S, i, kpr_r);
                    got.push_back(Run{j, nxt - j, pids[(size_t)r * kpr + j]});
                }
            }
            const uint64_t* rd = rd_of[r];
            skm_split_read(rd, rlen[r], g, [&](int j0, int n, uint32_t pid) { want.push_back(Run{j0, n, pid}); });
            if (got.size() != want.size()) return -102;
            for (size_t q = 0; q < got.size(); q++) {
                if (got[q].j0 != want[q].j0 || got[q].n != want[q].n || got[q].pid != want[q].pid) return -103;
                uint64_t a[RW], b[RW];
                const uint64_t ord0 = rkb[r] + 12345;
                tile_make_record<PW>(dw.data() + (size_t)r * wsd, rlen[r], got[q].j0, got[q].n, ord0, K, a);
                skm_make_record<PW>(rd, rlen[r], want[q].j0, want[q].n, ord0, g, b);
                for (int k = 0; k < RW; k++) if (a[k] != b[k]) return -104;
                checked++;
            }
        }
    }
    return checked;
}

extern "C" int64_t emu_tile_check(const uint64_t* packed, int64_t n_reads, int len, int K, int mer127, int log2_parts, int S, int R) {
#define TC(NWV, SV) case SV: return tile_check<NWV, SV>(packed, nullptr, nullptr, n_reads, len, K, log2_parts, R);
    if (mer127) switch (S) { TC(4, 7) TC(4, 9) TC(4, 11) TC(4, 13) TC(4, 15) default: return -99; }
    switch (S) { TC(2, 7) TC(2, 9) TC(2, 11) TC(2, 13) TC(2, 15) default: return -99; }
#undef TC
}
// a ragged batch: word_off[r] = first word of read r, lens[r] its bases, max_len >= every length
extern "C" int64_t emu_tile_check_ragged(const uint64_t* packed, const uint64_t* word_off, const int32_t* lens, int64_t n_reads, int max_len, int K, int mer127,
                                         int log2_parts, int S, int R) {
#define TC(NWV, SV) case SV: return tile_check<NWV, SV>(packed, word_off, lens, n_reads, max_len, K, log2_parts, R);
    if (mer127) switch (S) { TC(4, 7) TC(4, 9) TC(4, 11) TC(4, 13) TC(4, 15) default: return -99; }
    switch (S) { TC(2, 7) TC(2, 9) TC(2, 11) TC(2, 13) TC(2, 15) default: return -99; }
#undef TC
}
extern "C" int emu_tile_pick_segment(int kpr, int w) { return tile_pick_segment(kpr, w); }

// the sliced CRC of kmer.hpp against the byte-wise one
extern "C" int emu_crc_check(int n) {
    uint32_t t1[256], t4[1024];
    for (int i = 0; i < 256; i++) t1[i] = crc32_table_entry(i);
    for (int i = 0; i < 1024; i++) t4[i] = crc32_slice_entry(i >> 8, i & 255);
    uint64_t x = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < n; i++) {
        Kmer<2> a; Kmer<4> b;
        for (int q = 0; q < 2; q++) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; a.w[q] = x; }
        for (int q = 0; q < 4; q++) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; b.w[q] = x; }
        if (kmer_crc32<2>(a, t1) != kmer_crc32_sliced<2>(a, t4)) return 1;
        if (kmer_crc32<4>(b, t1) != kmer_crc32_sliced<4>(b, t4)) return 2;
    }
    // the nibble tables (no dependent look-ups), for every key and, with the k-mer length known, for keys of that length
    static uint32_t n2[32 * 16], n4[64 * 16];
    for (int q = 0; q < 32; q++) for (int v = 0; v < 16; v++) n2[q * 16 + v] = crc32_nibble_entry(16, q, (uint32_t)v);
    for (int q = 0; q < 64; q++) for (int v = 0; v < 16; v++) n4[q * 16 + v] = crc32_nibble_entry(32, q, (uint32_t)v);
    auto rnd = [&] { x = x * 6364136223846793005ULL + 1442695040888963407ULL; return x ^ (x >> 29); };
    for (int i = 0; i < n; i++) {
        Kmer<2> a; Kmer<4> b;
        for (int q = 0; q < 2; q++) a.w[q] = rnd();
        for (int q = 0; q < 4; q++) b.w[q] = rnd();
        if (kmer_crc32<2>(a, t1) != kmer_crc32_nibbles<2>(a, n2)) return 3;
        if (kmer_crc32<4>(b, t1) != kmer_crc32_nibbles<4>(b, n4)) return 4;
        Kmer<2> a31 = a, a63 = a, a25 = a; Kmer<4> b127 = b, b65 = b, b95 = b;
        a31.w[0] = 0; a31.w[1] &= (1ULL << 62) - 1;
        a25.w[0] = 0; a25.w[1] &= (1ULL << 50) - 1;
        a63.w[0] &= (1ULL << 62) - 1;
        b127.w[0] &= (1ULL << 62) - 1;
        b65.w[0] = b65.w[1] = 0; b65.w[2] &= 3;
        b95.w[0] = 0; b95.w[1] &= (1ULL << 62) - 1;
        if (kmer_crc32<2>(a31, t1) != kmer_crc32_nibbles<2, 31>(a31, n2)) return 5;
        if (kmer_crc32<2>(a25, t1) != kmer_crc32_nibbles<2, 25>(a25, n2)) return 6;
        if (kmer_crc32<2>(a63, t1) != kmer_crc32_nibbles<2, 63>(a63, n2)) return 7;
        if (kmer_crc32<4>(b127, t1) != kmer_crc32_nibbles<4, 127>(b127, n4)) return 8;
        if (kmer_crc32<4>(b65, t1) != kmer_crc32_nibbles<4, 65>(b65, n4)) return 9;
        if (kmer_crc32<4>(b95, t1) != kmer_crc32_nibbles<4, 95>(b95, n4)) return 10;
    }
    return 0;
}
