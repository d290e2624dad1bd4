// skm_tile.hpp -- the per-thread pieces of the tiled super-k-mer cutter (K1, partition_kernels.hip), host + device.
//
// skm_split_read (skm.hpp) is the reference formulation: one serial walk per read.  The tiled kernel computes the same
// runs for a tile of uniform-length reads with every thread doing a short serial piece of work on consecutive positions
// (no per-k-mer shuffles, no per-k-mer control flow):
//
//   A  tile_mmer_chunk      16 consecutive m-mer values of a read from two dwords of its base string
//   B  tile_segment<S>      the sliding-window minima of S consecutive k-mers (+ their predecessor) with w + S reads:
//                           all S + 1 windows share the core [first k-mer + S - 1, first k-mer - 1 + w), so
//                           min(window j) = min(suffix-min up to the core, core, prefix-min behind the core);
//                           then partition ids and the run-start bits of the S k-mers
//   D  tile_next_start      where the run that starts at a given bit ends (the next start bit, in this or a later segment)
//   E  tile_make_record<PW> the record of a run from the dword string
//
// Everything here is plain inline code shared with the CPU harness (tests/emu_skm.cpp), which runs the same phases serially
// and compares the records with skm_split_read + skm_make_record.
#pragma once
#include "skm.hpp"
#include "occ32.hpp"

namespace pg {

// 32 bits of a dword string (first base in the top bits of dword 0) starting at bit `bit` >= 0; d[bit/32 + 1] readable
PG_HD uint32_t dw_bits32(const uint32_t* d, int bit) {
    const int k = bit >> 5;
    const uint32_t o = (uint32_t)bit & 31u;
    const uint32_t x = alignbit32(d[k], d[k + 1], 32u - o);
    return o ? x : d[k];
}

// hashed canonical m-mer from its 16-base window x (m-mer in the top 2m bits); same value as mmer_value(rd, p, m)
PG_HD uint32_t mmer_from_window(uint32_t x, int m) {
    const uint32_t fwd = x >> (32 - 2 * m);
    uint32_t rc = rc_dword(x);
    if (m < 16) rc &= (1u << (2 * m)) - 1u;
    return mmer_hash(fwd < rc ? fwd : rc);
}

// A: m-mer values at positions 16 c .. 16 c + 15 of one read; row = its dword string (two readable dwords behind it),
// out = its value row, which has room for 16 * ceil(np / 16) values: positions >= np get junk nobody reads, and the
// sixteen stores need no bound test.  M: the m-mer length at compile time (0: `m`).
template <int M = 0>
PG_HD void tile_mmer_chunk(const uint32_t* row, int c, int m, uint32_t* out) {
    const uint32_t d0 = row[c], d1 = row[c + 1];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t x = i ? alignbit32(d0, d1, (uint32_t)(32 - 2 * i)) : d0;
        out[16 * c + i] = mmer_from_window(x, M ? M : m);
    }
}

// B: k-mers j0 .. j0 + cnt - 1 of a read (cnt <= S <= w), v = the read's m-mer values (np of them), w m-mers a k-mer.
// Returns the run-start bits (bit i = k-mer j0 + i starts a run: first k-mer, partition change, or a multiple of nmax)
// and the partition ids in pid_out[0 .. S) (those at cnt and above are junk).
// Straight-line on purpose: every load address is clamped into the row instead of being skipped, every "does this window
// exist" is a select -- a wave runs 64 segments in lockstep, and the branchy form spent more instructions on exec masks
// than on minima.  The three parts are cut at places that depend on the segment's FIRST k-mer alone (not on cnt): window q
// = k-mer j0 - 1 + q is  [j0 - 1 + q, J)  +  [J, j0 - 1 + w)  +  [j0 - 1 + w, j0 - 1 + q + w)  with J = j0 + S - 1, so the
// middle part has w - S positions for every lane and, with W = w at compile time, all three loops unroll into loads with
// immediate offsets.
template <int S, int W = 0>
PG_HD uint32_t tile_segment(const uint32_t* v, int np, int j0, int cnt, int w_rt, int nmax, uint32_t part_mul, uint32_t* pid_out) {
    // window q = k-mer j0 - 1 + q, q = 0 .. cnt: q = 0 is the predecessor of the segment's first k-mer (none when j0 = 0),
    // wanted only for the partition comparison.  All indices below are compile-time, so the arrays stay in registers.
    const int w = W ? W : w_rt;
    const int jl = j0 - 1, J = j0 + S - 1;
    uint32_t suf[S + 1], pre[S + 1];
    suf[S] = 0xFFFFFFFFu;                                          // suf[q] = min over positions [jl + q, J)   (J - 1 < np: S <= w)
#pragma unroll
    for (int q = S - 1; q >= 0; q--) {
        const int at = jl + q;
        const uint32_t x = v[at < 0 ? 0 : at];
        const uint32_t mn = x < suf[q + 1] ? x : suf[q + 1];
        suf[q] = at >= 0 ? mn : 0xFFFFFFFFu;                       // (at < 0: q = 0 of the read's first segment, a window nobody uses)
    }
    uint32_t core = 0xFFFFFFFFu;                                   // positions [J, jl + w): inside every window of the segment
    if (W) {
#pragma unroll
        for (int i = 0; i < (W ? W - S : 0); i++) { const uint32_t x = v[J + i]; core = x < core ? x : core; }
    } else {
        for (int p = J; p < jl + w; p++) { const uint32_t x = v[p]; core = x < core ? x : core; }
    }
    pre[0] = 0xFFFFFFFFu;                                          // pre[q] = min over positions [jl + w, jl + q + w)
#pragma unroll
    for (int q = 1; q <= S; q++) {
        const int at = jl + w + q - 1;
        const uint32_t x = v[at < np ? at : np - 1];
        const uint32_t xx = q <= cnt ? x : 0xFFFFFFFFu;
        pre[q] = xx < pre[q - 1] ? xx : pre[q - 1];
    }
    uint32_t mask = 0, prev_pid = 0;
    int next_cut = 0;                                              // first multiple of nmax >= j0
    while (next_cut < j0) next_cut += nmax;
#pragma unroll
    for (int q = 0; q <= S; q++) {
        uint32_t mv = suf[q] < core ? suf[q] : core;
        mv = pre[q] < mv ? pre[q] : mv;
        const uint32_t pid = skm_partition(mv, part_mul);        // (q = 0 without a predecessor, q > cnt: junk nobody uses)
        if (q >= 1) {
            const int j = j0 + q - 1;
            const bool cut = j == next_cut;
            const bool start = (j == 0 || pid != prev_pid || cut) && q <= cnt;
            next_cut += cut ? nmax : 0;
            mask |= start ? 1u << (q - 1) : 0u;
            pid_out[q - 1] = pid;
        }
        prev_pid = pid;
    }
    return mask;
}

// D: end of the run that starts at bit i of segment `seg`: the next start bit of the read (segments seg .. nseg - 1, S
// k-mers each), or kpr.  masks = the read's segment masks.
PG_HD int tile_next_start(const uint32_t* masks, int seg, int nseg, int S, int i, int kpr) {
    const uint32_t own = masks[seg];
    const uint32_t above = i < 31 ? (own >> (i + 1)) : 0u;
    if (above) return seg * S + i + __builtin_ffs((int)above);
    for (int sg = seg + 1; sg < nseg; sg++) {
        const uint32_t mk = masks[sg];
        if (mk) return sg * S + __builtin_ffs((int)mk) - 1;
    }
    return kpr;
}

// E: the record of the run [j0, j0 + n) of a read of `len` bases given as a dword string (two zero dwords behind it).
// Same words as skm_make_record (skm.hpp).
template <int PW>
PG_HD void tile_make_record(const uint32_t* row, int len, int j0, int n, uint64_t ord0, int K, uint64_t* rec) {
    const int has_left = j0 > 0, has_right = (j0 + n - 1 + K) < len;
    const int b0 = j0 - has_left;
    const int nb = n + K - 1 + has_left + has_right;
    rec[0] = skm_header(ord0 + (uint64_t)j0, n, has_left, has_right);
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int first = 32 * i;
        uint64_t v = 0;
        if (first < nb) {
            const int bit = 2 * (b0 + first);
            v = ((uint64_t)dw_bits32(row, bit) << 32) | dw_bits32(row, bit + 32);
            const int valid = nb - first;
            if (valid < 32) v &= ~0ULL << (64 - 2 * valid);
        }
        rec[1 + i] = v;
    }
}

// segment length for a read geometry: odd (the m-mer rows are read with stride S between lanes), <= w, least padding
inline int tile_pick_segment(int kpr, int w) {
    int best = 7;
    double best_eff = -1;
    for (int s = 7; s <= 15; s += 2) {
        if (s > w && s != 7) continue;
        const int nseg = (kpr + s - 1) / s;
        const double eff = (double)kpr / (double)(nseg * s) - 0.002 * (15 - s);   // ties: the longer segment (fewer reads per k-mer)
        if (eff > best_eff) { best_eff = eff; best = s; }
    }
    return best;
}

}  // namespace pg
