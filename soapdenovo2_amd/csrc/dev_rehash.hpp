// dev_rehash.hpp -- the layout of a GROWABLE k-mer set (-a 0) on the device: put_kmerset with encap_kmerset's in-place
// rehash (newhash.c:340-528), the slot every k-mer ends up in.  Written over a Backend (backend.hpp), like dev_graph.hpp.
//
// A growable set lives through a handful of sizes (ref_sizes.hpp: grow_schedule).  At one size it is first-come-first-served
// linear probing -- dev_graph.hpp shows that this is one sweep per probe cluster in which a slot takes the earliest-arrived
// pending key -- and the only question is what "arrived" means for the keys that were already there when the set grew:
//
//   encap_kmerset walks the OLD slots in index order; the element of slot i, if it has not moved yet, is taken out and
//   inserted at (key mod new size), probing over slots that hold already-MOVED elements only; if it comes to rest on a slot
//   whose old element has not moved yet, it takes the slot and that element is inserted next, and so on (newhash.c:403-452).
//
// So every element has an insertion TIME (origin slot of its chain, depth in the chain): (j, 0) for the element of old slot j
// if the walk finds it in place, (i, d + 1) if the element that was inserted at time (i, d) came to rest on slot j first.  The
// keys that arrive after the growth come later than all of them, in arrival order.  With the times known the new layout is the
// same sweep as for a static pool.  The times depend on the layout (who rests on slot j?) and the layout on the times -- but
// only forwards: an element can only be kicked by one that was inserted EARLIER.  Hence a fixed point from above: start with
// T(e_j) = (j, 0), sweep, look at every old slot j -- if its new occupant y is another element with T(y) < (j, 0), then e_j was
// kicked: T(e_j) = T(y) + one level -- and sweep again the probe clusters that hold an element whose time changed, until none
// does.  Why it ends at the reference's layout: (1) no time is ever EARLIER than the true one -- a slot of a
// first-come-first-served table fills no earlier when every key arrives no earlier, so the occupant of slot j in a round comes no
// earlier than the true one, and an element that truly stays in place is never taken for kicked; (2) by (1) the elements with the
// k earliest true times, once they carry them, make the first k steps of a round's sweep the true ones, so the (k+1)-th finds its
// true kicker (or nobody) on its slot and has its true time from the next round on.  Both need a round's times to be read as a
// whole: the sweep only READS times and lists the changes it finds (old element, new time); they are applied between the rounds.
// (A first version updated the times in place while other lanes read them and livelocked on the host backend's four threads.)
// A round costs what changed: the sweep runs over a list of the clusters that hold an element whose time changed, and the only old
// elements whose time can change are those whose old slot lies in such a cluster -- the sweep meets them on its way.  Chains are a
// dozen levels deep (a fifth of the moves are kicks), a set settles in 20 - 30 rounds a size, all but the first few tiny.
// Which slots are occupied, the clusters and the wrap-around frame depend on the homes alone and are computed once per size.
// What a size costs is random memory accesses (profiles/r05x_growable_layout_ab.json): its preparation gathers the times in sorted order and
// finds the slot -> element map of the previous size where that size's sweeps left it (RhSweep::elem_at), and the first round of a large size
// runs over a list of the cluster starts, every lane of a wave a walk.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <vector>

#include "backend.hpp"
#include "env.hpp"
#include "graph_lookup.hpp"
#include "ref_sizes.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace pg {

constexpr int RH_DEPTH_BITS = 22;                  // time = origin << 22 | depth
constexpr uint32_t RH_NONE = 0xFFFFFFFFu;
constexpr unsigned long long RH_NO_TIME = ~0ULL;

// one probe cluster: the slots of frame [hs[j], ...) take the pending element with the smallest time; the old element of every
// slot met on the way is given the time the slot's new occupant implies -- if that is not the time it has, onto the change list
struct RhSweep {
    const uint64_t* hs;                 // homes in the frame, sorted
    const uint32_t* is;                 // elements in the same order
    const long long* m;                 // prefix maximum of hs[j] - j
    const unsigned long long* Ts;       // time of every element, in the sorted order (read only here)
    const unsigned long long* T_old;    // per table slot j < n_prev: the time of the element that sat there before the growth (RH_NO_TIME: nobody)
    unsigned int* dirty;                // per sorted position: the cluster that starts here is on the list
    const uint32_t* list;               // cluster starts to sweep (null: every cluster, lane = sorted position)
    unsigned long long* heap_t;
    uint32_t* heap_e;
    unsigned long long* slot_new;       // table slot of every element
    const uint32_t* elem_prev;          // per table slot j < n_prev: which element that is
    uint32_t* elem_at;                  // per table slot of THIS size: the element the sweep puts there (what the next size finds there; null: no next size)
    uint32_t* chg_e;                    // changes (element, time): a cluster's go into ITS stretch of these arrays (it places as many
    unsigned long long* chg_t;          //   elements as it has, and every placement changes at most one old element), chg_n[start] of them --
    uint32_t* chg_n;                    //   one counter for all of them would be a returned atomic on one address inside the loop
    unsigned long long* n_chg;          // their number over the round
    uint64_t n, S, origin, n_prev;
    int only_dirty;                     // (list == null) skip the clusters whose flag is down
    const unsigned long long* list_n;   // (list != null) entries of the list, where only the device knows the number (null: every lane has one)
    // A lane's walk is a chain of loads that depend on each other through the loop (is the next key's home at or before this slot? then its time,
    // its element; the old time of the slot): issued where they are needed, every one of them is a full memory latency on the lane's critical
    // path, and the sweep is bound by exactly that (more lanes in flight helped, several clusters a lane made it slower:
    // profiles/r05x_growable_layout_ab.json).  So the next key (home, time, element) is loaded one key AHEAD, and the slot's old time and old
    // element at the top of the iteration, before the pending set is touched -- an iteration then waits for memory once.
    PG_HD void operator()(uint64_t lane) const {
        uint64_t j = lane;
        if (list) { if (list_n && lane >= *list_n) return; j = list[lane]; }
        else if ((j && m[j] <= m[j - 1]) || (only_dirty && !dirty[j])) return;     // not the first key of a cluster / nothing changed in it
        dirty[j] = 0;
        uint32_t n_found = 0;
        // the pending elements: the RF earliest wait in registers (sorted; all-ones = free), the rest in a heap whose storage is
        // the cluster's own stretch of the heap arrays -- most clusters never touch it, and a heap in memory is a chain of
        // dependent round trips per element
        constexpr int RF = 6;
        unsigned long long rt[RF];
        uint32_t re[RF];
#pragma unroll
        for (int i = 0; i < RF; i++) { rt[i] = RH_NO_TIME; re[i] = 0; }
        unsigned long long* ht = heap_t + j;
        uint32_t* he = heap_e + j;
        uint64_t hn = 0, nxt = j, p = hs[j];
        uint64_t h_nx = p;                                          // the key at sorted position nxt, loaded ahead (valid while nxt < n)
        unsigned long long t_nx = Ts[j];
        uint32_t e_nx = is[j];
        for (;;) {
            uint64_t slot = p + origin;
            if (slot >= S) slot -= S;
            const bool had_owner = slot < n_prev;
            const unsigned long long told = had_owner ? T_old[slot] : RH_NO_TIME;      // (read-only during a sweep: the changes are applied between the rounds)
            const uint32_t e_old = had_owner ? elem_prev[slot] : RH_NONE;
            while (nxt < n && h_nx <= p) {
                unsigned long long t = t_nx;
                uint32_t e = e_nx;
                nxt++;
                if (nxt < n) { h_nx = hs[nxt]; t_nx = Ts[nxt]; e_nx = is[nxt]; }
#pragma unroll
                for (int i = 0; i < RF; i++)                         // into the sorted registers; what falls out at the end is the latest of them
                    if (t < rt[i]) { const unsigned long long xt = rt[i]; const uint32_t xe = re[i]; rt[i] = t; re[i] = e; t = xt; e = xe; }
                if (t != RH_NO_TIME) {
                    uint64_t c = hn++;
                    while (c) { const uint64_t par = (c - 1) >> 1; if (ht[par] <= t) break; ht[c] = ht[par]; he[c] = he[par]; c = par; }
                    ht[c] = t; he[c] = e;
                }
            }
            uint32_t first;
            unsigned long long t_first;
            if (hn && ht[0] < rt[0]) {                              // (an element that fell out of full registers can be earlier than one that came later)
                first = he[0];
                t_first = ht[0];
                hn--;
                const unsigned long long lt = ht[hn];
                const uint32_t le = he[hn];
                if (hn) {
                    uint64_t c = 0;
                    for (;;) {
                        uint64_t ch = 2 * c + 1;
                        if (ch >= hn) break;
                        if (ch + 1 < hn && ht[ch + 1] < ht[ch]) ch++;
                        if (ht[ch] >= lt) break;
                        ht[c] = ht[ch]; he[c] = he[ch]; c = ch;
                    }
                    ht[c] = lt; he[c] = le;
                }
            } else {
                if (rt[0] == RH_NO_TIME) break;                     // nothing pending: the next key starts its own cluster
                first = re[0];
                t_first = rt[0];
#pragma unroll
                for (int i = 0; i + 1 < RF; i++) { rt[i] = rt[i + 1]; re[i] = re[i + 1]; }
                rt[RF - 1] = RH_NO_TIME;
            }
            slot_new[first] = slot;
            if (elem_at) elem_at[slot] = first;                     // (a cluster's slots are the same in every round; its last sweep is the one that counts)
            // (times are unique, so "the slot's old element came to rest on it itself" is t_first == told; everything the
            //  sweep reads lies in the order it walks: sorted position or slot)
            if (told != RH_NO_TIME) {
                const unsigned long long natural = (unsigned long long)slot << RH_DEPTH_BITS;
                unsigned long long t = natural;
                if (t_first != told && t_first < natural) t = t_first + 1;   // `first` came to rest here before the walk reached the slot: the old element went next
                if (t != told) {
                    chg_e[j + n_found] = e_old; chg_t[j + n_found] = t;
                    n_found++;
                }
            }
            p++;
        }
        chg_n[j] = n_found;
        if (n_found) hd_atomic_add(n_chg, (unsigned long long)n_found);
    }
};

// between two sweeps: the changes a cluster's sweep found are applied -- the times where the next sweep reads them (by sorted
// position, by old slot) -- and the clusters of the changed elements flagged; with want_list they are listed as well, by an atomic
// each (few changes; many: the caller makes the list from the flags)
struct RhApply {
    const uint32_t* list;               // clusters swept this round (null: all positions)
    uint32_t* chg_n;
    const uint32_t* chg_e;
    const unsigned long long* chg_t;
    const uint32_t* pos_of;
    const long long* cs;
    const unsigned long long* slot_prev;
    unsigned long long* Ts;
    unsigned long long* T_old;
    unsigned int* dirty;
    uint32_t* list_next;
    unsigned long long* n_dirty;
    const unsigned long long* n_chg;    // the sweep's count of time changes: a list of the clusters to sweep again is made here while it is short (the host
    unsigned long long list_max;        // reads both counters in ONE go afterwards and decides the same way)
    const unsigned long long* list_n;   // as RhSweep's
    PG_HD void operator()(uint64_t lane) const {
        if (list && list_n && lane >= *list_n) return;
        const uint64_t j = list ? list[lane] : lane;
        const uint32_t k = chg_n[j];
        if (!k) return;
        const bool want_list = *n_chg <= list_max;
        chg_n[j] = 0;
        uint32_t fresh = 0;
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t e = chg_e[j + i];
            const unsigned long long t = chg_t[j + i];
            const uint32_t pos = pos_of[e];
            Ts[pos] = t;
            T_old[slot_prev[e]] = t;
            const uint32_t start = (uint32_t)cs[pos];
            if (hd_atomic_exch(&dirty[start], 1u) == 0u) {
                if (want_list) list_next[hd_atomic_add(n_dirty, 1ULL)] = start;
                else fresh++;
            }
        }
        if (fresh) hd_atomic_add(n_dirty, (unsigned long long)fresh);
    }
};

// scratch for one set at a time, sized for the largest
template <class BE>
struct RhWork {
    uint64_t *hk = nullptr, *hs = nullptr, *hr = nullptr;
    uint32_t *iv = nullptr, *is = nullptr, *ir = nullptr, *pos_of = nullptr, *heap_e = nullptr, *elem_prev = nullptr, *elem_next = nullptr, *chg_e = nullptr, *chg_n = nullptr, *list_a = nullptr, *list_b = nullptr;
    long long *v = nullptr, *m = nullptr, *cs = nullptr;
    unsigned long long *Ts = nullptr, *T_old = nullptr, *chg_t = nullptr, *heap_t = nullptr, *slot_prev = nullptr, *scal = nullptr;
    unsigned int* dirty = nullptr;
    uint64_t cap = 0, owner_cap = 0;
    bool reserve(BE& be, uint64_t n, uint64_t n_owner) {
        n = std::max<uint64_t>(n, 1);
        cap = n; owner_cap = std::max<uint64_t>(n_owner, 1);
        hk = be.template alloc<uint64_t>(n); hs = be.template alloc<uint64_t>(n); hr = be.template alloc<uint64_t>(n);
        iv = be.template alloc<uint32_t>(n); is = be.template alloc<uint32_t>(n); ir = be.template alloc<uint32_t>(n);
        v = be.template alloc<long long>(n); m = be.template alloc<long long>(n);
        cs = be.template alloc<long long>(n);                          // cluster start (sorted position) of every sorted position
        pos_of = be.template alloc<uint32_t>(n);                       // sorted position of every element
        Ts = be.template alloc<unsigned long long>(n);
        chg_e = be.template alloc<uint32_t>(n); chg_t = be.template alloc<unsigned long long>(n); chg_n = be.template alloc<uint32_t>(n);
        list_a = be.template alloc<uint32_t>(n); list_b = be.template alloc<uint32_t>(n);             // cluster starts to sweep, this round's and the next's
        dirty = be.template alloc<unsigned int>(n);
        heap_t = be.template alloc<unsigned long long>(n); heap_e = be.template alloc<uint32_t>(n);
        slot_prev = be.template alloc<unsigned long long>(n);
        elem_prev = be.template alloc<uint32_t>(owner_cap); elem_next = be.template alloc<uint32_t>(owner_cap); T_old = be.template alloc<unsigned long long>(owner_cap);
        scal = be.template alloc<unsigned long long>(4);
        return !be.error;
    }
    void release(BE& be) {
        be.release(hk); be.release(hs); be.release(hr); be.release(iv); be.release(is); be.release(ir); be.release(v); be.release(m); be.release(cs);
        be.release(pos_of); be.release(Ts); be.release(T_old); be.release(chg_e); be.release(chg_t); be.release(chg_n); be.release(list_a); be.release(list_b); be.release(dirty); be.release(heap_t); be.release(heap_e);
        be.release(slot_prev); be.release(elem_prev); be.release(elem_next); be.release(scal);
        *this = RhWork();
    }
    // bytes a key / an old slot (for the caller's memory planning)
    static constexpr uint64_t bytes_per_key = 8 * 3 + 4 * 3 + 8 * 3 + 4 + 8 + 16 + 8 + 4 + 8 + 4 + 8;
};
// the largest old size a schedule's growths walk over
inline uint64_t grow_owner_slots(const std::vector<GrowEpoch>& sched) {
    uint64_t c = 0;
    for (size_t e = 1; e < sched.size(); e++) c = std::max(c, sched[e - 1].size);
    return c;
}

// the slots of a growable set's n keys (arrival order = record order) after put_kmerset's whole history (sched: grow_schedule
// of the set).  slots_out: backend memory, n entries.  *rounds_out += fixed-point rounds.  wk: reserved for >= n keys and the
// schedule's grow_owner_slots.
template <class BE, int NW>
int layout_growable(BE& be, RhWork<BE>& wk, const uint64_t* rec, uint64_t n, const std::vector<GrowEpoch>& sched, unsigned long long* slots_out,
                    int* rounds_out) {
    constexpr int RW = NW + 2;
    if (!n) return PG_OK;
    if (n >= 0xFFFFFFF0ULL) { be.error_text = "layout_growable: more than 2^32 keys in a set"; return PG_EINVAL; }
    if (n > wk.cap || grow_owner_slots(sched) > wk.owner_cap) { be.error_text = "layout_growable: scratch too small"; return PG_EINVAL; }
    uint64_t *hk = wk.hk, *hs = wk.hs, *hr = wk.hr;
    uint32_t *iv = wk.iv, *is = wk.is, *ir = wk.ir, *pos_of = wk.pos_of, *heap_e = wk.heap_e, *elem_prev = wk.elem_prev, *elem_next = wk.elem_next, *chg_e = wk.chg_e, *chg_n = wk.chg_n;
    long long *v = wk.v, *m = wk.m, *cs = wk.cs;
    unsigned long long *Ts = wk.Ts, *T_old = wk.T_old, *chg_t = wk.chg_t, *heap_t = wk.heap_t, *slot_prev = wk.slot_prev, *scal = wk.scal;
    unsigned int* dirty = wk.dirty;
    unsigned long long* slot_new = slots_out;
    int rc = PG_OK;
    const bool rh_debug = pg::env_measure("PG_RH_DEBUG") != nullptr;
    auto rh_now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    for (size_t ei = 0; ei < sched.size() && !be.error && rc == PG_OK; ei++) {
        const GrowEpoch ep = sched[ei];
        const double t_epoch = rh_debug ? rh_now() : 0.0;
        const uint64_t S = ep.size, M = ep.n_end, n_old = ep.n_old;
        const uint64_t s_prev = ei ? sched[ei - 1].size : 0;
        if (!M) continue;
        if ((s_prev + M) >> (63 - RH_DEPTH_BITS)) { rc = PG_EINVAL; be.error_text = "layout_growable: set too large for the time encoding"; break; }
        int bits = 1;
        while (bits < 64 && (S >> bits)) bits++;
        // ---- homes at this size, sorted; occupied slots, clusters, the frame in which nothing wraps
        const ModConst mc = make_modconst(S);
        be.launch(M, [=] PG_LAMBDA(uint64_t i) {
            Kmer<NW> k;
#pragma unroll
            for (int w = 0; w < NW; w++) k.w[w] = rec[i * RW + w];
            hk[i] = home_slot<NW>(k, mc);
            iv[i] = (uint32_t)i;
        });
        be.sort_pairs(hk, hs, iv, is, M, bits);
        be.launch(M, [=] PG_LAMBDA(uint64_t j) { v[j] = (long long)hs[j] - (long long)j; });
        be.inclusive_max(v, m, M);
        long long m_last = 0;
        be.to_host(&m_last, m + (M - 1), 1);
        if (be.error) break;
        const uint64_t q_last = (uint64_t)((long long)(M - 1) + m_last);
        const uint64_t* hs_use = hs;
        const uint32_t* is_use = is;
        uint64_t origin = 0;
        if (q_last >= S) {                                           // the last cluster wraps (dev_graph.hpp, step 3)
            const uint64_t W = q_last - (S - 1);
            be.launch(1, [=] PG_LAMBDA(uint64_t) {
                uint64_t seen = 0, prev_end = 0, e = 0;
                bool found = false;
                for (uint64_t j = 0; j < M && !found; j++) {
                    const uint64_t q = (uint64_t)((long long)j + m[j]);
                    if (q >= S) break;
                    const uint64_t gap = q - prev_end;
                    if (seen + gap >= W + 1) { e = prev_end + (W + 1 - seen) - 1; found = true; }
                    seen += gap;
                    prev_end = q + 1;
                }
                if (!found) e = prev_end + (W + 1 - seen) - 1;
                const uint64_t o = e + 1 == S ? 0 : e + 1;
                uint64_t lo = 0, hi = M;
                while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (hs[mid] >= o) hi = mid; else lo = mid + 1; }
                scal[0] = o; scal[1] = lo;
            });
            unsigned long long orot[2] = {0, 0};
            be.to_host(orot, scal, 2);
            if (be.error) break;
            origin = orot[0];
            const uint64_t r = orot[1] % M, o = origin;
            be.launch(M, [=] PG_LAMBDA(uint64_t j) {
                const uint64_t src = j + r >= M ? j + r - M : j + r;
                const uint64_t h = hs[src];
                hr[j] = h >= o ? h - o : h + S - o;
                ir[j] = is[src];
                v[j] = (long long)hr[j] - (long long)j;
            });
            be.inclusive_max(v, m, M);
            be.to_host(&m_last, m + (M - 1), 1);
            if (be.error) break;
            if ((uint64_t)((long long)(M - 1) + m_last) >= S) { rc = PG_EINVAL; be.error_text = "layout_growable: the rotated frame still wraps"; break; }
            hs_use = hr; is_use = ir;
        }
        // ---- where every element sits in the sorted order, and the cluster it belongs to; times; everything to be swept.
        //      The times are GATHERED in sorted order (an old element's: the slot it sat in; a new key's: its arrival rank, no memory at all), and which
        //      element sat in which old slot is what the previous size's sweeps left behind (elem_at, written slot by slot as they walk): setting both up by
        //      scatters -- Ts[pos_of[i]], elem_prev[slot_prev[i]], T_old[slot_prev[i]] -- was 60 % of the random accesses of a size's preparation.
        uint32_t* const elem_at = ei + 1 < sched.size() ? elem_next : nullptr;
        {
            const uint32_t* isu = is_use;
            const long long* mm = m;
            const unsigned long long* sp = slot_prev;
            be.launch(M, [=] PG_LAMBDA(uint64_t j) {
                const uint64_t i = isu[j];
                pos_of[i] = (uint32_t)j;
                v[j] = (j == 0 || mm[j] > mm[j - 1]) ? (long long)j : 0;
                dirty[j] = 0u;
                chg_n[j] = 0u;
                Ts[j] = (i < n_old ? sp[i] : (unsigned long long)(s_prev + (i - n_old))) << RH_DEPTH_BITS;
            });
            be.inclusive_max(v, cs, M);
            if (n_old) {
                const uint32_t* ep = elem_prev;
                be.launch(s_prev, [=] PG_LAMBDA(uint64_t slot) { T_old[slot] = ep[slot] != RH_NONE ? (unsigned long long)slot << RH_DEPTH_BITS : RH_NO_TIME; });
            }
            if (elem_at) be.fill(elem_at, (size_t)S, RH_NONE);
        }
        if (rh_debug) {
            unsigned long long* longest = scal + 3;
            const long long* cc = cs;
            be.fill(longest, 1, 0ULL);
            be.launch(M, [=] PG_LAMBDA(uint64_t j) { if (j + 1 == M || cc[j + 1] != cc[j]) hd_atomic_max(longest, (unsigned long long)(j + 1 - (uint64_t)cc[j])); });
            unsigned long long lg = 0;
            be.to_host(&lg, longest, 1);
            fprintf(stderr, "rh size %llu keys %llu (%llu there before): sort, clusters, times ready %.3f ms since the size began; longest cluster %llu keys\n", (unsigned long long)S,
                    (unsigned long long)M, (unsigned long long)n_old, 1e3 * (rh_now() - t_epoch), lg);
        }
        // ---- the fixed point: sweep (every cluster first, then the listed ones), apply the changes it found, list their clusters
        const uint32_t* list_cur = nullptr;                           // null: the sweep runs over all positions
        uint32_t* list_next = wk.list_a;
        uint64_t n_list = 0;
        const char* const shift_env = pg::env_test("PG_RH_LIST_SHIFT");
        const int list_shift = shift_env ? std::max(0, std::min(40, atoi(shift_env))) : 5;
        // more changes than this: the list is made from the flags by a prefix sum (three passes over all M positions), not by an append a cluster.  The
        // append is ONE returned atomic a wave (the compiler folds a wave's adds on one address into one), so it wins until a round changes a few
        // per cent of the keys: M / 1024 -> M / 32 took 5 % off the layout at 60 M reads, M / 8 put 13 % on (profiles/r05x_growable_layout_ab.json)
        const uint64_t list_max = std::max<uint64_t>(1024, M >> list_shift);
        // Small sizes are nothing but rounds, and a round's work there is microseconds against the ~0.6 ms its read-back costs (a drained
        // stream and a copy): eight rounds at a time are launched blind -- full sweeps that skip the clusters whose flag is down, no lists --
        // and the host looks at the last one's change count; the rounds behind the fixed point find nothing to do.
        const char* const blind_env = pg::env_test("PG_RH_BLIND_MAX");              // (0: every round read back, for A/B runs and tests)
        const uint64_t blind_max = blind_env ? (uint64_t)atoll(blind_env) : (uint64_t)1 << 18;
        const bool blind = n_old && M <= blind_max;
        for (int round = 0; blind;) {
            if (round > 100000) { rc = PG_EINVAL; be.error_text = "layout_growable: the fixed point did not settle"; break; }
            for (int q = 0; q < 8; q++, round++) {
                if (rounds_out) (*rounds_out)++;
                be.fill(scal, 2, 0ULL);
                be.launch(M, RhSweep{hs_use, is_use, m, Ts, T_old, dirty, nullptr, heap_t, heap_e, slot_new, elem_prev, elem_at, chg_e, chg_t, chg_n, scal, M, S, origin, s_prev, round > 0, nullptr});
                be.launch(M, RhApply{nullptr, chg_n, chg_e, chg_t, pos_of, cs, slot_prev, Ts, T_old, dirty, list_next, scal + 1, scal, 0ULL, nullptr});
            }
            unsigned long long last_chg = 0;
            be.to_host(&last_chg, scal, 1);
            if (be.error || !last_chg) break;
        }
        // The first round of a large size sweeps every cluster.  With a lane a sorted position two lanes in three leave at once (they are not
        // the first key of a cluster) and a wave's 64 lanes hold some twenty walks; over a LIST of the cluster starts every lane of a wave
        // walks -- a third of the waves for the same walks.  The list is a prefix sum over the start flags; its length stays on the
        // device (scal[2]: no read-back), the launch has a lane a position and the lanes behind the list's end leave.
        const char* const dense_env = pg::env_test("PG_RH_DENSE_MIN");             // (keys from which the first round runs over a list; 0: never)
        const uint64_t dense_min = dense_env ? (uint64_t)atoll(dense_env) : (uint64_t)1 << 18;
        const unsigned long long* list_n_dev = nullptr;
        if (!blind && dense_min && M >= dense_min) {
            unsigned long long* f64 = (unsigned long long*)hk;          // (the unsorted homes are done with)
            unsigned long long* at64 = (unsigned long long*)v;          // (so are the scans' inputs)
            const long long* mm = m;
            uint32_t* ln = wk.list_a;
            unsigned long long* nl = scal + 2;
            be.launch(M, [=] PG_LAMBDA(uint64_t j) { f64[j] = (j == 0 || mm[j] > mm[j - 1]) ? 1ULL : 0ULL; });
            be.exclusive_sum(f64, at64, M);
            be.launch(M, [=] PG_LAMBDA(uint64_t j) {
                if (f64[j]) ln[at64[j]] = (uint32_t)j;
                if (j + 1 == M) *nl = at64[j] + f64[j];
            });
            list_cur = wk.list_a;
            list_next = wk.list_b;
            n_list = M;
            list_n_dev = scal + 2;
        }
        for (int round = 0; !blind; round++) {
            if (round > 100000) { rc = PG_EINVAL; be.error_text = "layout_growable: the fixed point did not settle"; break; }
            if (rounds_out) (*rounds_out)++;
            be.fill(scal, 2, 0ULL);
            const unsigned long long* const bound = round == 0 ? list_n_dev : nullptr;      // (later lists: the host knows their length)
            be.launch(list_cur ? n_list : M, RhSweep{hs_use, is_use, m, Ts, T_old, dirty, list_cur, heap_t, heap_e, slot_new, elem_prev, elem_at, chg_e, chg_t, chg_n, scal, M, S, origin,
                                                     n_old ? s_prev : 0, round > 0, bound});
            if (!n_old) break;                                        // nobody was there before: arrival order is all there is
            // (the changes are applied before the host knows whether there were any: one read-back a round instead of two -- a read-back is a
            //  drained stream and a copy, ~0.6 ms, and the small sizes of a set are nothing but rounds)
            be.launch(list_cur ? n_list : M, RhApply{list_cur, chg_n, chg_e, chg_t, pos_of, cs, slot_prev, Ts, T_old, dirty, list_next, scal + 1, scal, (unsigned long long)list_max, bound});
            unsigned long long both[2] = {0, 0};
            be.to_host(both, scal, 2);
            const unsigned long long n_chg = both[0], n_dirty_h = both[1];
            if (be.error || !n_chg) break;
            const bool want_list = n_chg <= list_max;
            if (rh_debug) fprintf(stderr, "rh size %llu keys %llu round %d: %llu time changes, %llu clusters to sweep again, %.3f ms since the size began\n", (unsigned long long)S,
                                  (unsigned long long)M, round, n_chg, n_dirty_h, 1e3 * (rh_now() - t_epoch));
            if (!want_list) {
                // many clusters: the list by a prefix sum over the flags (a sweep over all positions that looks at the flags would
                // run with a few live lanes a wave; listed clusters fill the waves)
                unsigned long long* f64 = (unsigned long long*)hk;      // (the unsorted homes are done with)
                unsigned long long* at64 = (unsigned long long*)v;      // (so are the scans' inputs)
                unsigned int* dd = dirty;
                uint32_t* ln = list_next;
                be.launch(M, [=] PG_LAMBDA(uint64_t j) { f64[j] = dd[j] ? 1ULL : 0ULL; });
                be.exclusive_sum(f64, at64, M);
                be.launch(M, [=] PG_LAMBDA(uint64_t j) { if (dd[j]) ln[at64[j]] = (uint32_t)j; });
            }
            n_list = n_dirty_h;
            list_cur = list_next;
            list_next = list_cur == wk.list_a ? wk.list_b : wk.list_a;
        }
        if (rh_debug) { be.sync(); fprintf(stderr, "rh size %llu keys %llu: settled %.3f ms since the size began\n", (unsigned long long)S, (unsigned long long)M, 1e3 * (rh_now() - t_epoch)); }
        if (ei + 1 < sched.size()) { be.copy(slot_prev, slot_new, M); std::swap(elem_prev, elem_next); }
    }
    be.sync();
    if (be.error) return be.error;
    return rc;
}

// scratch bytes one call of layout_growable_sets takes for sets of at most n_max keys that grow out of at most owner_max slots
inline uint64_t growable_scratch_bytes(uint64_t n_max, uint64_t owner_max) { return (RhWork<int>::bytes_per_key + 8) * n_max + 16 * owner_max + (1u << 20); }

// P growable sets: records sorted by (set, ordinal) in backend memory; per_set_count, trailing (a duplicate put arrived after
// the set's last new key), set_first_slot (the set's slot 0 in `nodes`, in slots) on the host.  nodes (optional): the image,
// every slot's first word preset to SV_EMPTY -- a key's record words 0..NW go to its slot.  slots_all (optional, backend memory,
// one entry a record): the slot within the set.
template <class BE, int NW>
int layout_growable_sets(BE& be, const uint64_t* records, const uint64_t* per_set_count, const unsigned char* trailing, int P, uint64_t init_size,
                         const uint64_t* set_first_slot, uint64_t* nodes, unsigned long long* slots_all, uint64_t* rounds_out, int s_begin = 0, int s_step = 1) {
    // (s_begin, s_step: this call lays out the sets s_begin, s_begin + s_step, ... -- a round of the fixed point is a handful of
    //  small dependent launches, so the caller runs several calls side by side, each on its own backend / stream)
    constexpr int RW = NW + 2;
    uint64_t n_max = 0, owner_max = 0;
    std::vector<std::vector<GrowEpoch>> sched(P);
    std::vector<uint64_t> first_of((size_t)P + 1, 0);
    for (int s = 0; s < P; s++) {
        first_of[s + 1] = first_of[s] + per_set_count[s];
        if (s < s_begin || (s - s_begin) % s_step) continue;
        sched[s] = grow_schedule(per_set_count[s], trailing && trailing[s], init_size);
        n_max = std::max(n_max, per_set_count[s]);
        owner_max = std::max(owner_max, grow_owner_slots(sched[s]));
    }
    if (!n_max) return PG_OK;
    RhWork<BE> wk;
    unsigned long long* slots_tmp = slots_all ? nullptr : be.template alloc<unsigned long long>(n_max);
    int rc = wk.reserve(be, n_max, owner_max) ? PG_OK : PG_ENOMEM;
    for (int s = s_begin; s < P && rc == PG_OK && !be.error; s += s_step) {
        const uint64_t n = per_set_count[s], first = first_of[s];
        const uint64_t* rec = records + first * RW;
        unsigned long long* slots = slots_all ? slots_all + first : slots_tmp;
        if (!n) continue;
        int rounds = 0;
        rc = layout_growable<BE, NW>(be, wk, rec, n, sched[s], slots, &rounds);
        if (rounds_out) rounds_out[s] = (uint64_t)rounds;
        if (rc != PG_OK || !nodes) continue;
        uint64_t* set_nodes = nodes + set_first_slot[s] * (uint64_t)(NW + 1);
        be.launch(n, [=] PG_LAMBDA(uint64_t i) {
            const uint64_t* r = rec + i * RW;
            uint64_t* nd = set_nodes + slots[i] * (uint64_t)(NW + 1);
#pragma unroll
            for (int w = 0; w <= NW; w++) nd[w] = r[w];
        });
    }
    be.sync();
    wk.release(be);
    be.release(slots_tmp);
    if (be.error) return be.error;
    return rc;
}

}  // namespace pg
