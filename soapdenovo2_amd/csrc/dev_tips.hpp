// dev_tips.hpp -- tip clipping decided on the device (removeSingleTips / removeMinorTips, cutTipPreGraph.c:363-488, with
// clipTipFromNode :43-346), same result as the reference's slot-order scan.  Written over a Backend (backend.hpp).
//
// The reference visits the nodes set by set, slot by slot; a visit of a live non-linear dead end c (in 0 / out 1 or in 1 /
// out 0; THIN: a frequency-one node) walks over linear nodes to the node u it stops at and, depending on u's counters,
// deletes c and unlinks / deletes u (cutTipPreGraph.c:271-342).  A visit therefore reads c, the linear interior and u, and
// writes c (deleted) and u.  What makes the scan sequential is u: several tips end on it and every clip changes what the next
// one finds there; a clipped u can become a dead end that the scan meets further down, or (MINOR) a linear node that later
// walks run through.  "Time" below = the global slot of the visited node (the scan order).
//
// Formulation: a fixed point over start decisions, every round fully parallel.
//   candidates   the nodes that start a tip at their time, with the counters they have then.  Round 0: every live dead end of
//                the state the scan begins with (S0).
//   walks        one lane per candidate that has no valid walk yet: S0's linear flags, plus the few nodes that became linear
//                earlier in this scan (table BL: node -> time, counters), which a later walk runs through.
//   arrivals     (u, c) pairs sorted by u, then by time c.
//   node step    one lane per stop node u replays what the scan does to u, in time order, from S0[u]: every arrival takes its
//                verdict from u's counters as they are then (tip_decide) and changes them (tip_apply); when time passes u
//                itself, u's own start is decided from its counters then (yes with these counters / no).  A u that became
//                linear sends its later arrivals back for a longer walk (BL).
//   Decisions feed back (a start that was a far node may not start, or start another way; a new dead end spawns a
//   candidate), so the rounds repeat until nothing changes.  Every dependency points from an earlier time to a later one:
//   the earliest wrong decision of a round is right in the next, and the fixed point is the sequential scan's outcome.
//   Only then the nodes are written.
// Rounds after the first touch few walks (the cache keeps a walk while its start's counters and the BL nodes it met are
// unchanged); sorting and replaying the arrivals is cheap.
#pragma once
#include <stdint.h>

#include "backend.hpp"
#include "graph_lookup.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

constexpr uint64_t TIP_NONE = ~0ULL;

PG_HD int ab_in(uint64_t ab) { return count_arcs24((uint32_t)ab); }
PG_HD int ab_out(uint64_t ab) { return count_arcs24((uint32_t)(ab >> 32)); }
PG_HD uint32_t ab_L(uint64_t ab, int i) { return ((uint32_t)ab >> (6 * i)) & 63u; }
PG_HD uint32_t ab_R(uint64_t ab, int i) { return ((uint32_t)(ab >> 32) >> (6 * i)) & 63u; }
PG_HD bool ab_flag(uint64_t ab, uint32_t bflag) { return ((uint32_t)(ab >> 32) & bflag) != 0; }
PG_HD bool ab_dead_end(uint64_t ab) {
    const int in = ab_in(ab), out = ab_out(ab);
    return (in == 0 && out == 1) || (in == 1 && out == 0);
}
// a node the scan may start a tip from (cutTipPreGraph.c:374-395, 428-455)
PG_HD bool ab_startable(uint64_t ab, bool thin) {
    const uint32_t B = (uint32_t)(ab >> 32);
    return !(B & (B_LINEAR | B_DELETED)) && (!thin || (B & B_SINGLE));
}
// dislink2prevUncertain (newhash.c:681-691)
PG_HD uint64_t ab_cut_prev(uint64_t ab, int ch, bool smaller) {
    return smaller ? (ab & ~(63ULL << (6 * ch))) : (ab & ~(63ULL << (32 + 6 * (ch ^ 2))));
}
// the verdict at the stop node (cutTipPreGraph.c:271-342): 0 keep, 1 both ends dead, 2 thin cut, 3 minority cut
PG_HD int tip_verdict(uint64_t far, int first, bool far_smaller, bool thin) {
    if (ab_in(far) + ab_out(far) == 1) return 1;
    if (thin) return 2;
    uint32_t strongest = 0;
    for (int c = 0; c < 4; c++) { const uint32_t x = far_smaller ? ab_L(far, c) : ab_R(far, c); strongest = x > strongest ? x : strongest; }
    const uint32_t mine = far_smaller ? ab_L(far, first) : ab_R(far, first ^ 2);
    return mine < strongest ? 3 : 0;
}
PG_HD uint64_t tip_apply_far(uint64_t far, int action, int first, bool far_smaller) {
    if (action == 1) return far | ((uint64_t)B_DELETED << 32);
    far = ab_cut_prev(far, first, far_smaller);
    if (action == 2) return far & ~((uint64_t)B_LINEAR << 32);
    if (ab_in(far) == 1 && ab_out(far) == 1) far |= (uint64_t)B_LINEAR << 32;
    return far;
}

// open addressing, key = global slot + 1 (0 = free), linear probing; built by CAS, read-only in the step after
struct SlotMap {
    unsigned long long* key;
    unsigned long long* val;
    uint64_t mask;
};
PG_HD uint64_t slot_hash(uint64_t g) { g *= 0x9E3779B97F4A7C15ULL; return g ^ (g >> 29); }
PG_HD bool slotmap_find(const SlotMap& m, uint64_t g, unsigned long long& v) {
    uint64_t h = slot_hash(g) & m.mask;
    for (;;) {
        const unsigned long long k = m.key[h];
        if (k == 0) return false;
        if (k == g + 1) { v = m.val[h]; return true; }
        h = (h + 1) & m.mask;
    }
}
// insert if absent; returns the slot of the entry (the caller writes val when it created it: *created)
PG_HD uint64_t slotmap_claim(const SlotMap& m, uint64_t g, bool& created) {
    uint64_t h = slot_hash(g) & m.mask;
    for (;;) {
        unsigned long long k = hd_atomic_load(&m.key[h]);
        if (k == 0) {
            k = hd_atomic_cas(&m.key[h], 0ULL, (unsigned long long)(g + 1));
            if (k == 0) { created = true; return h; }
        }
        if (k == g + 1) { created = false; return h; }
        h = (h + 1) & m.mask;
    }
}

// candidate flags (walk cache; written by the walk step and -- CF_REWALK -- by the node step of the candidate's stop node)
constexpr uint32_t CF_WALKED = 4, CF_VIA_BL = 8, CF_FAR_SMALLER = 16, CF_REWALK = 32;
// counters: 0 candidates, 1 changed decisions, 2 walks sent back, 3 clips, 4 errors (k-mer not found), 5 BL entries, 6 walks that
// ended elsewhere than their cached copy, 7 spawned, 8 digest of BL (order-independent sum)
struct TipState {
    SetsView view;
    int cut_len, thin;
    uint64_t sentinel;                   // arrival key of "no arrival" (above every slot)
    uint64_t n_initial;                  // candidates [0, n_initial) are S0's dead ends, the rest were spawned
    unsigned long long* c_slot;          // start node (time)
    unsigned long long* c_ab;            // its counters when it starts (this round)
    unsigned int* c_started;             // does it start (this round)?  written by the reset and by its own node's lane only
    unsigned long long* c_prev;          // counters it started with in the previous round; TIP_NONE = did not start
    unsigned long long* c_far;           // where the cached walk stopped (TIP_NONE: longer than the cut-off)
    unsigned int* c_flags;
    unsigned int* c_dir;                 // the arc the cached walk left by (tip_dir of the counters it was made with)
    unsigned int* c_first;               // first base of the last walk k-mer in front of the stop node
    unsigned int* c_action;              // this round's verdict on the candidate's clip
    SlotMap cmap;                        // start slot -> candidate index
    SlotMap bl;                          // node that became linear in this scan -> time << 24 | entry; counters in bl_ab[entry]
    unsigned long long* bl_ab;
    uint64_t bl_cap;
    unsigned long long* counters;
};
// which way a dead end's walk leaves: 0..3 forward by that base, 4..7 backward (the walk depends on the counters through this only)
PG_HD unsigned int tip_dir(uint64_t ab) {
    int ch;
    if (ab_in(ab) == 0) { for (ch = 0; ch < 4; ch++) if (ab_R(ab, ch)) break; return (unsigned int)(ch & 3); }
    for (ch = 0; ch < 4; ch++) if (ab_L(ab, ch)) break;
    return 4u + (unsigned int)(ch & 3);
}

// the walk of clipTipFromNode (cutTipPreGraph.c:65-269) from candidate i, seeing the nodes as they are at its time
template <int NW>
PG_HD void tip_walk_candidate(const TipState& t, uint64_t i, bool use_bl) {
    const uint64_t c = t.c_slot[i], ab0 = t.c_ab[i];
    const int K = t.view.K;
    const Kmer<NW> filter = kmer_filter<NW>(K);
    const uint64_t* nd0 = sv_node<NW>(t.view, c);
    Kmer<NW> seq;
#pragma unroll
    for (int k = 0; k < NW; k++) seq.w[k] = nd0[k];
    const unsigned int dir = tip_dir(ab0);
    const bool fwd = dir < 4;
    Kmer<NW> prev = fwd ? seq : kmer_rc<NW>(seq, K);
    const int ch = fwd ? (int)dir : (int)((dir & 3u) ^ 2u);
    uint32_t flags = CF_WALKED;
    t.c_dir[i] = dir;
    int count = 1;
    Kmer<NW> cur = kmer_next<NW>(prev, ch, filter);
    uint64_t slot;
    uint64_t* node;
    bool smaller;
    if (!sv_step<NW>(t.view, cur, slot, node, smaller)) { hd_atomic_add(&t.counters[4], 1ULL); t.c_far[i] = TIP_NONE; t.c_flags[i] = flags; return; }
    uint64_t ab = node[NW];
    bool reached = true;
    for (;;) {
        bool linear = ab_flag(ab, B_LINEAR);
        if (!linear && use_bl) {                                   // became linear earlier in this scan?
            unsigned long long e;
            if (slotmap_find(t.bl, slot, e) && (e >> 24) < c) { ab = t.bl_ab[(e & 0xFFFFFFu) % t.bl_cap]; linear = true; flags |= CF_VIA_BL; }
        }
        if (!linear) break;
        count++;
        if (t.thin && !ab_flag(ab, B_SINGLE)) break;
        if (count > t.cut_len) { reached = false; break; }
        prev = cur;
        cur = kmer_next<NW>(cur, linear_out_ab(ab, smaller), filter);
        if (!sv_step<NW>(t.view, cur, slot, node, smaller)) { hd_atomic_add(&t.counters[4], 1ULL); reached = false; break; }
        ab = node[NW];
    }
    const unsigned long long far = reached ? slot : TIP_NONE;
    const unsigned int first = (unsigned int)kmer_first<NW>(prev, K);
    if (reached && smaller) flags |= CF_FAR_SMALLER;
    const unsigned int old = t.c_flags[i];
    if (!(old & CF_WALKED) || t.c_far[i] != far || t.c_first[i] != first || ((old ^ flags) & CF_FAR_SMALLER)) hd_atomic_add(&t.counters[6], 1ULL);
    t.c_far[i] = far;
    t.c_first[i] = first;
    t.c_flags[i] = flags;
}

// what the scan does to one stop node, in time order.  ord[k] = the candidate of arrival k, arrivals sorted by (stop node, time).
template <int NW>
PG_HD void tip_node_step(const TipState& t, const unsigned int* ord, const unsigned long long* far_sorted, uint64_t a, uint64_t n_arr, bool commit,
                         unsigned long long* spawn, uint64_t spawn_cap) {
    if (a && far_sorted[a] == far_sorted[a - 1]) return;             // not the first arrival of its node
    const uint64_t u = far_sorted[a];
    if (u >= t.sentinel) return;
    uint64_t* nd = sv_node<NW>(t.view, u);
    uint64_t st = nd[NW];
    const bool thin = t.thin != 0;
    bool own_done = false;
    uint64_t t_lin = TIP_NONE;                                       // when u became linear in this scan (MINOR)
    auto own_start = [&]() {
        own_done = true;
        const bool yes = ab_startable(st, thin) && ab_dead_end(st);
        unsigned long long ci;
        if (slotmap_find(t.cmap, u, ci)) {
            t.c_started[ci] = yes ? 1u : 0u;
            if (yes) t.c_ab[ci] = st;
        } else if (yes && !commit) {                                 // a new dead end: a candidate from the next round on
            const unsigned long long at = hd_atomic_add(&t.counters[7], 1ULL);
            if (at < spawn_cap) spawn[at] = u;
        }
    };
    for (uint64_t k = a; k < n_arr && far_sorted[k] == u; k++) {
        const uint64_t ci = ord[k], c = t.c_slot[ci];
        if (!own_done && c >= u) own_start();
        if (t_lin != TIP_NONE) {                                     // u is a linear node by now: this walk goes on through it
            if (!commit) { t.c_flags[ci] |= CF_REWALK; hd_atomic_add(&t.counters[2], 1ULL); }
            t.c_action[ci] = 0;
            continue;
        }
        const int first = (int)t.c_first[ci];
        const bool far_smaller = (t.c_flags[ci] & CF_FAR_SMALLER) != 0;
        const int action = tip_verdict(st, first, far_smaller, thin);
        t.c_action[ci] = (unsigned int)action;
        if (!action) continue;
        st = tip_apply_far(st, action, first, far_smaller);
        if (!thin && action == 3 && ab_flag(st, B_LINEAR)) {
            t_lin = c;
            if (!commit) {
                bool created;
                const uint64_t h = slotmap_claim(t.bl, u, created);              // (one lane per node: always created)
                const unsigned long long e = hd_atomic_add(&t.counters[5], 1ULL) % t.bl_cap;     // (an overflow is reported by the caller)
                t.bl_ab[e] = st;
                t.bl.val[h] = ((unsigned long long)c << 24) | e;
                hd_atomic_add(&t.counters[8], (unsigned long long)(slot_hash(u) ^ slot_hash(c + 0x5555) ^ slot_hash(st)));
            }
        }
    }
    if (!own_done) own_start();
    if (commit) nd[NW] = st;
}

// geometry of the sets on the host side of the calls (the same numbers SetsView::geo holds for the lanes)
struct SetsGeo {
    int P;
    std::vector<uint64_t> first, size;
    std::vector<uint64_t*> base;
    uint64_t n_slots() const { return P ? first[P - 1] + size[P - 1] : 0; }
};

// f(node address, global slot) for every slot of every set: one launch per set, lane = slot, AT THE SET'S PLACE (the GPU that
// holds it: the reference gives every set one owner for its scans too, cutTipPreGraph.c:603-639).  The caller has waited for
// the lead's stream (be.sync()) and waits for the places afterwards (be.sync_places()).
template <class BE, int NW, class F>
void for_each_slot(BE& be, const SetsGeo& geo, F f) {
    for (int s = 0; s < geo.P; s++) {
        uint64_t* base = geo.base[s];
        const uint64_t first = geo.first[s];
        be.launch_at(be.place_of_set(s), geo.size[s], [=] PG_LAMBDA(uint64_t i) { f(base + i * (NW + 1), first + i); });
    }
}

// Mark1in1outNode (cutTipPreGraph.c:532-564): a live non-linear node with one arc each way becomes linear
template <class BE, int NW>
void remark_linear(BE& be, const SetsGeo& geo) {
    be.sync();
    for_each_slot<BE, NW>(be, geo, [=] PG_LAMBDA(uint64_t* nd, uint64_t) {
        if (nd[0] == SV_EMPTY) return;
        const uint64_t ab = nd[NW];
        if (ab_flag(ab, B_DELETED | B_LINEAR)) return;
        if (ab_in(ab) == 1 && ab_out(ab) == 1) nd[NW] = ab | ((uint64_t)B_LINEAR << 32);
    });
    be.sync_places();
}

// One scan of removeSingleTips (thin) / removeMinorTips over all sets.  Returns PG_OK or PG_E*; *removed = tips clipped,
// *rounds = fixed-point rounds it took.
template <class BE, int NW>
int tip_scan(BE& be, const SetsView& view, const SetsGeo& geo, int cut_len, bool thin, uint64_t* removed, int* rounds) {
    *removed = 0;
    *rounds = 0;
    const uint64_t n_slots = geo.n_slots();
    unsigned long long* counters = be.template alloc<unsigned long long>(12);
    be.fill(counters, 12, 0ULL);
    // ---- S0's dead ends: one scan over the sets lists them -- every set at its place, into a list and a counter of that place
    //      (room for one slot in eight, a second scan with the exact room should that ever be short); the lead gathers the lists
    unsigned long long h_cnt[12];
    for (int q = 0; q < 12; q++) h_cnt[q] = 0;
    unsigned long long* cand_list = nullptr;
    {
        const int NPL = be.n_places();
        std::vector<uint64_t> pl_slots(NPL, 0), pl_cap(NPL, 0), pl_n(NPL, 0);
        std::vector<unsigned long long*> pl_list(NPL, nullptr), pl_cnt(NPL, nullptr);
        for (int s = 0; s < geo.P; s++) pl_slots[be.place_of_set(s)] += geo.size[s];
        for (int pl = 0; pl < NPL; pl++) { pl_cap[pl] = pl_slots[pl] / 8 + 65536; pl_cnt[pl] = be.template alloc_at<unsigned long long>(pl, 1); }
        be.sync();                                                   // what the lead wrote into the sets so far is in place
        for (int attempt = 0; attempt < 2 && !be.error; attempt++) {
            bool short_of_room = false;
            for (int pl = 0; pl < NPL; pl++) {
                if (pl_list[pl]) continue;                               // (listed completely in the first attempt)
                pl_list[pl] = be.template alloc_at<unsigned long long>(pl, pl_cap[pl]);
                be.fill_at(pl, pl_cnt[pl], 1, 0ULL);
            }
            for (int s = 0; s < geo.P && !be.error; s++) {
                const int pl = be.place_of_set(s);
                if (pl_n[pl]) continue;
                uint64_t* base = geo.base[s];
                const uint64_t first = geo.first[s], cap_now = pl_cap[pl];
                unsigned long long* const list_now = pl_list[pl];
                unsigned long long* const cnt_now = pl_cnt[pl];
                be.append_at(pl, geo.size[s], [=] PG_LAMBDA(uint64_t i) -> unsigned long long {
                    const uint64_t* nd = base + i * (NW + 1);
                    if (nd[0] == SV_EMPTY) return ~0ULL;
                    const uint64_t ab = nd[NW];
                    if (!(ab_startable(ab, thin) && ab_dead_end(ab))) return ~0ULL;
                    return first + i;
                }, list_now, cnt_now, cap_now);
            }
            for (int pl = 0; pl < NPL && !be.error; pl++) {
                if (pl_n[pl]) continue;
                unsigned long long got = 0;
                be.to_host_at(pl, &got, pl_cnt[pl], 1);
                if (got > pl_cap[pl]) { be.release_at(pl, pl_list[pl]); pl_list[pl] = nullptr; pl_cap[pl] = got; short_of_room = true; }
                else pl_n[pl] = got ? got : ~0ULL;                      // (~0: listed, nothing found)
            }
            if (!short_of_room) break;
        }
        uint64_t total = 0;
        bool complete = !be.error;
        for (int pl = 0; pl < NPL; pl++) { if (!pl_n[pl]) complete = false; else if (pl_n[pl] != ~0ULL) total += pl_n[pl]; }
        if (complete && total) {
            cand_list = be.template alloc<unsigned long long>(total);
            uint64_t at = 0;
            for (int pl = 0; pl < NPL && cand_list; pl++) {
                if (pl_n[pl] == ~0ULL) continue;
                be.gather_at(pl, cand_list + at, pl_list[pl], pl_n[pl]);
                at += pl_n[pl];
            }
            be.sync();
        }
        for (int pl = 0; pl < NPL; pl++) { be.release_at(pl, pl_list[pl]); be.release_at(pl, pl_cnt[pl]); }
        if (be.error || !complete) { be.release(cand_list); be.release(counters); return be.error ? be.error : PG_ENOMEM; }
        h_cnt[0] = total;
    }
    const uint64_t n_init = h_cnt[0];
    if (!n_init || !cand_list) { be.release(cand_list); be.release(counters); return n_init ? PG_ENOMEM : PG_OK; }
    const uint64_t cap = n_init + n_init / 4 + 65536;
    if (cap >= 0xFFFFFFFFULL) { be.release(cand_list); be.release(counters); be.error_text = "tips: more than 2^32 dead ends"; return PG_EINVAL; }
    TipState t;
    t.view = view; t.cut_len = cut_len; t.thin = thin ? 1 : 0;
    int bits = 1;
    while (bits < 63 && (n_slots >> bits)) bits++;
    bits++;
    t.sentinel = (1ULL << bits) - 1;
    t.n_initial = n_init;
    t.c_slot = be.template alloc<unsigned long long>(cap);
    t.c_ab = be.template alloc<unsigned long long>(cap);
    t.c_started = be.template alloc<unsigned int>(cap);
    t.c_prev = be.template alloc<unsigned long long>(cap);
    t.c_far = be.template alloc<unsigned long long>(cap);
    t.c_flags = be.template alloc<unsigned int>(cap);
    t.c_dir = be.template alloc<unsigned int>(cap);
    t.c_first = be.template alloc<unsigned int>(cap);
    t.c_action = be.template alloc<unsigned int>(cap);
    uint64_t map_cap = 1024;
    while (map_cap < 2 * cap) map_cap <<= 1;
    t.cmap.key = be.template alloc<unsigned long long>(map_cap);
    t.cmap.val = be.template alloc<unsigned long long>(map_cap);
    t.cmap.mask = map_cap - 1;
    const uint64_t bl_cap = 1 << 20;                                  // nodes that turn linear during one scan: hundreds
    t.bl.key = be.template alloc<unsigned long long>(2 * bl_cap);
    t.bl.val = be.template alloc<unsigned long long>(2 * bl_cap);
    t.bl.mask = 2 * bl_cap - 1;
    t.bl_ab = be.template alloc<unsigned long long>(bl_cap);
    t.bl_cap = bl_cap;
    t.counters = counters;
    unsigned long long* k1 = be.template alloc<unsigned long long>(cap);   // sort keys (two buffers), values = candidate index
    unsigned long long* k2 = be.template alloc<unsigned long long>(cap);
    unsigned int* v1 = be.template alloc<unsigned int>(cap);
    unsigned int* v2 = be.template alloc<unsigned int>(cap);
    const uint64_t spawn_cap = cap - n_init;
    unsigned long long* spawn = be.template alloc<unsigned long long>(spawn_cap);
    int rc = PG_OK;
    uint64_t n_cand = n_init;
    const int NPL = be.n_places();
    std::vector<unsigned long long*> pl_counters(NPL, nullptr);
    if (NPL > 1) for (int pl = 0; pl < NPL; pl++) pl_counters[pl] = be.template alloc_at<unsigned long long>(pl, 12);
    auto cleanup = [&]() {
        for (int pl = 0; pl < NPL; pl++) be.release_at(pl, pl_counters[pl]);
        be.release(t.c_slot); be.release(t.c_ab); be.release(t.c_started); be.release(t.c_prev); be.release(t.c_far); be.release(t.c_flags);
        be.release(t.c_dir); be.release(t.c_first); be.release(t.c_action); be.release(t.cmap.key); be.release(t.cmap.val);
        be.release(t.bl.key); be.release(t.bl.val); be.release(t.bl_ab); be.release(k1); be.release(k2); be.release(v1); be.release(v2);
        be.release(spawn); be.release(counters); be.release(cand_list);
    };
    if (be.error) { cleanup(); return be.error; }
    be.fill(t.cmap.key, map_cap, 0ULL);
    be.fill(t.bl.key, 2 * bl_cap, 0ULL);
    be.fill(counters, 12, 0ULL);
    {   // the candidates (any order: the arrivals are sorted by time below) and their index
        const TipState tt = t;
        const unsigned long long* const list_now = cand_list;
        be.launch(n_init, [=] PG_LAMBDA(uint64_t i) {
            const uint64_t g = list_now[i], ab = sv_node<NW>(tt.view, g)[NW];
            tt.c_slot[i] = g; tt.c_ab[i] = ab; tt.c_started[i] = 1u; tt.c_prev[i] = ab; tt.c_flags[i] = 0; tt.c_action[i] = 0; tt.c_far[i] = TIP_NONE;
            bool created;
            const uint64_t h = slotmap_claim(tt.cmap, g, created);
            tt.cmap.val[h] = i;
        });
        be.fill(counters, 12, 0ULL);
    }
    be.sync();
    be.release(cand_list);
    cand_list = nullptr;
    bool any_bl = false;
    unsigned long long bl_digest_prev = 0;
    for (int round = 0;; round++) {
        if (round > 100000) { rc = PG_EINVAL; be.error_text = "tips: the fixed point did not settle"; break; }
        *rounds = round + 1;
        const TipState tt = t;
        const bool use_bl = any_bl;
        const uint64_t n = n_cand;
        be.fill(counters + 1, 11, 0ULL);
        // 1. walks that are missing or stale.  A walk is a lane a candidate and crosses sets by nature (the next k-mer hashes anywhere): with
        //    several places the candidates are dealt to them in equal shares -- every GPU of a sharded run walks its share against the
        //    peer-mapped sets, the state arrays stay the lead's (streamed through the peer mapping), every place counts into counters of its own
        //    (cutTipPreGraph.c:363-488 gives the scan to thrd_num workers the same way: each its own sets, the walks wherever they lead)
        if (NPL > 1) {
            be.sync();                                               // what the lead wrote for this round is in place
            for (int pl = 0; pl < NPL; pl++) {
                const uint64_t first = n * (uint64_t)pl / (uint64_t)NPL, count = n * (uint64_t)(pl + 1) / (uint64_t)NPL - first;
                TipState tp = tt;
                tp.view = be.view_at(pl, tt.view);
                tp.counters = pl_counters[pl];
                be.fill_at(pl, pl_counters[pl], 12, 0ULL);
                be.launch_walks_at(pl, count, [=] PG_LAMBDA(uint64_t q) {
                    const uint64_t i = first + q;
                    if (!tp.c_started[i]) return;
                    const unsigned int fl = tp.c_flags[i];
                    if ((fl & CF_WALKED) && !(fl & (CF_VIA_BL | CF_REWALK)) && tp.c_dir[i] == tip_dir(tp.c_ab[i])) return;
                    tip_walk_candidate<NW>(tp, i, use_bl);
                });
            }
            be.sync_places();
        } else
        be.launch(n, [=] PG_LAMBDA(uint64_t i) {
            if (!tt.c_started[i]) return;
            const unsigned int fl = tt.c_flags[i];
            if ((fl & CF_WALKED) && !(fl & (CF_VIA_BL | CF_REWALK)) && tt.c_dir[i] == tip_dir(tt.c_ab[i])) return;
            tip_walk_candidate<NW>(tt, i, use_bl);
        });
        // 2. the arrivals by (stop node, time): sort by time, then stably by stop node
        be.launch(n, [=] PG_LAMBDA(uint64_t i) { k1[i] = tt.c_slot[i]; v1[i] = (unsigned int)i; });
        be.sort_pairs(k1, k2, v1, v2, n, bits);
        be.launch(n, [=] PG_LAMBDA(uint64_t j) {
            const unsigned int i = v2[j];
            const unsigned long long far = tt.c_far[i];
            k1[j] = (tt.c_started[i] && far != TIP_NONE) ? far : tt.sentinel;
        });
        be.sort_pairs(k1, k2, v2, v1, n, bits);      // k2 = stop nodes, v1 = candidates
        // 3. defaults for the round: S0's dead ends start with S0's counters, spawned candidates do not -- unless their own
        //    node's lane says otherwise below
        be.launch(n, [=] PG_LAMBDA(uint64_t i) {
            tt.c_action[i] = 0;
            if (i < tt.n_initial) { tt.c_started[i] = 1u; tt.c_ab[i] = sv_node<NW>(tt.view, tt.c_slot[i])[NW]; }
            else tt.c_started[i] = 0u;
        });
        if (any_bl) be.fill(t.bl.key, 2 * bl_cap, 0ULL);
        // 4. every stop node replays its arrivals
        be.launch(n, [=] PG_LAMBDA(uint64_t a) { tip_node_step<NW>(tt, v1, k2, a, n, false, spawn, spawn_cap); });
        // 5. what changed?
        be.launch(n, [=] PG_LAMBDA(uint64_t i) {
            const unsigned long long cur = tt.c_started[i] ? tt.c_ab[i] : TIP_NONE;
            if (cur != tt.c_prev[i]) { hd_atomic_add(&tt.counters[1], 1ULL); tt.c_prev[i] = cur; }
            if (tt.c_action[i]) hd_atomic_add(&tt.counters[3], 1ULL);
        });
        be.to_host(h_cnt, counters, 12);
        for (int pl = 0; pl < NPL && NPL > 1 && !be.error; pl++) {       // what the places' walks counted: errors (4) and walks that ended elsewhere (6)
            unsigned long long pc[12];
            be.to_host_at(pl, pc, pl_counters[pl], 12);
            h_cnt[4] += pc[4]; h_cnt[6] += pc[6];
        }
        if (be.error) break;
        if (h_cnt[4]) { rc = PG_EINVAL; be.error_text = "Kmer is not found while clipping a tip."; break; }
        if (h_cnt[5] >= bl_cap) { rc = PG_ENOMEM; be.error_text = "tips: too many nodes turned linear in one scan"; break; }
        any_bl = any_bl || h_cnt[5] != 0;
        const uint64_t n_spawn = h_cnt[7];
        if (n_cand + n_spawn > cap) { rc = PG_ENOMEM; be.error_text = "tips: too many new dead ends in one scan"; break; }
        if (n_spawn) {
            const uint64_t base = n_cand;
            be.launch(n_spawn, [=] PG_LAMBDA(uint64_t q) {
                const uint64_t i = base + q, g = spawn[q];
                tt.c_slot[i] = g; tt.c_ab[i] = 0; tt.c_started[i] = 0u; tt.c_prev[i] = TIP_NONE; tt.c_flags[i] = 0; tt.c_action[i] = 0; tt.c_far[i] = TIP_NONE;
                bool created;
                const uint64_t h = slotmap_claim(tt.cmap, g, created);
                tt.cmap.val[h] = i;
            });
            n_cand += n_spawn;
        }
        const bool bl_same = h_cnt[8] == bl_digest_prev;
        bl_digest_prev = h_cnt[8];
        if (h_cnt[1] == 0 && h_cnt[2] == 0 && n_spawn == 0 && bl_same && (round == 0 || h_cnt[6] == 0)) {
            // settled: the same replay once more, this time into the nodes; then the starts of the clipped tips die
            *removed = h_cnt[3];
            be.launch(n, [=] PG_LAMBDA(uint64_t a) { tip_node_step<NW>(tt, v1, k2, a, n, true, spawn, spawn_cap); });
            be.launch(n, [=] PG_LAMBDA(uint64_t i) {
                if (tt.c_action[i]) hd_atomic_or((unsigned long long*)(sv_node<NW>(tt.view, tt.c_slot[i]) + NW), (unsigned long long)B_DELETED << 32);
            });
            be.sync();
            break;
        }
    }
    if (be.error && rc == PG_OK) rc = be.error;
    cleanup();
    return rc;
}

// removeSingleTips + removeMinorTips as call_pregraph runs them (pregraph.c:106-120); counts as the reference prints them
struct TipTotals { uint64_t single = 0, minor = 0; int minor_cycles = 0, rounds = 0; std::vector<uint64_t> per_cycle; };
template <class BE, int NW>
int clip_tips(BE& be, const SetsView& view, const SetsGeo& geo, bool cut_single, TipTotals& out) {
    const int cut = 2 * view.K;
    int rounds = 0;
    if (cut_single) {
        int rc = tip_scan<BE, NW>(be, view, geo, cut, true, &out.single, &rounds);
        if (rc) return rc;
        out.rounds += rounds;
        remark_linear<BE, NW>(be, geo);
    }
    for (;;) {
        uint64_t removed = 0;
        int rc = tip_scan<BE, NW>(be, view, geo, cut, false, &removed, &rounds);
        if (rc) return rc;
        out.rounds += rounds;
        out.minor += removed;
        out.per_cycle.push_back(removed);
        out.minor_cycles++;
        if (!removed) break;
    }
    remark_linear<BE, NW>(be, geo);
    be.sync();
    return be.error;
}

}  // namespace pg
