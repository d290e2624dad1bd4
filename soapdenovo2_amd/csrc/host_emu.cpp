// host_emu.cpp -- the device graph stages (dev_graph.hpp, dev_tips.hpp) instantiated over the HostBackend: the same function
// objects the HIP kernels run, driven by host threads.  TEST HOOKS ONLY (pg_host_emu_*): the `-m "not gpu"` tests compare them
// with the sequential host stages and with a plain model, so that the logic of the kernels is pinned before it ever meets a
// GPU.  Nothing in the product path calls these; call_pregraph runs the HipBackend instantiations (graph_kernels.hip).
#include <stdint.h>

#include <string>
#include <vector>

#include "arena_list.hpp"
#include "backend.hpp"
#include "dev_graph.hpp"
#include "../../include/soapdenovo2_amd.h"

void pg_set_error(const std::string& s);

extern "C" int pg_host_emu_layout_static(const uint64_t* records, const uint64_t* per_set_count, int n_sets, uint64_t set_size, int mer127,
                                         int n_threads, uint64_t* nodes_out) {
    if (!records || !per_set_count || !nodes_out || n_sets < 1 || set_size < 2) { pg_set_error("pg_host_emu_layout_static: bad argument"); return PG_EINVAL; }
    const int nw1 = (mer127 ? 4 : 2) + 1;
    for (uint64_t i = 0; i < (uint64_t)n_sets * set_size; i++) {
        nodes_out[i * nw1] = pg::SV_EMPTY;
        for (int w = 1; w < nw1; w++) nodes_out[i * nw1 + w] = 0;
    }
    pg::HostBackend be(n_threads);
    const int rc = mer127 ? pg::layout_static<pg::HostBackend, 4>(be, records, per_set_count, n_sets, set_size, nodes_out)
                          : pg::layout_static<pg::HostBackend, 2>(be, records, per_set_count, n_sets, set_size, nodes_out);
    if (rc < 0) pg_set_error(be.error_text.empty() ? "pg_host_emu_layout_static failed" : be.error_text);
    return rc;
}

// key mod size with the precomputed reciprocal (graph_lookup.hpp: ModConst, rem128, home_slot), as every lookup of the graph stages and
// both layouts compute it; keys = n x (mer127 ? 4 : 2) words, out[n]
extern "C" int pg_host_emu_home_slots(const uint64_t* keys, uint64_t n, int mer127, uint64_t size, uint64_t* out) {
    if (!keys || !out || size < 1 || (size >> 63)) { pg_set_error("pg_host_emu_home_slots: bad argument"); return PG_EINVAL; }
    const pg::ModConst mc = pg::make_modconst(size);
    for (uint64_t i = 0; i < n; i++) {
        if (mer127) { pg::Kmer<4> k; for (int w = 0; w < 4; w++) k.w[w] = keys[4 * i + w]; out[i] = pg::home_slot<4>(k, mc); }
        else { pg::Kmer<2> k; for (int w = 0; w < 2; w++) k.w[w] = keys[2 * i + w]; out[i] = pg::home_slot<2>(k, mc); }
    }
    return PG_OK;
}

// The device arena's block list (arena_list.hpp) on a random sequence of cuts and returns over a range of `size` bytes: every block lies inside the
// range, at its alignment, disjoint from every live block; the books (bytes in use, holes + blocks = the range) balance after every step; when the
// last block is back the list is ONE hole again.  Returns 0, or the step (1-based, negative) at which an invariant broke.
// out[0] = cuts that succeeded, out[1] = cuts refused for lack of a hole, out[2] = peak bytes in use, out[3] = largest number of holes seen.
extern "C" long long pg_host_emu_arena_blocks(uint64_t seed, uint64_t n_ops, uint64_t size, uint64_t max_block, uint64_t out[4]) {
    pg::BlockList bl;
    bl.reset((size_t)size);
    std::vector<std::pair<size_t, size_t>> live;                       // (offset, bytes held)
    uint64_t x = seed * 0x9E3779B97F4A7C15ULL + 1, ok = 0, refused = 0, max_holes = 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto check = [&]() -> bool {
        uint64_t sum = 0, prev_end = 0;
        bool first = true;
        for (const auto& h : bl.free_) {                                // holes: sorted, inside the range, never adjacent (they would have merged)
            if (h.second == 0 || h.first + h.second > size) return false;
            if (!first && h.first <= prev_end) return false;
            first = false; prev_end = h.first + h.second; sum += h.second;
        }
        uint64_t used_sum = 0;
        for (const auto& b : live) used_sum += b.second;
        max_holes = std::max<uint64_t>(max_holes, bl.free_.size());
        return used_sum == bl.in_use && sum + used_sum == size && bl.used.size() == live.size();
    };
    for (uint64_t step = 1; step <= n_ops; step++) {
        const bool cut = live.empty() || (rnd() % 100) < 50;
        if (cut) {
            uint64_t want = rnd() % 7 == 0 ? (rnd() % max_block) + 1 : (rnd() % (max_block / 64 + 1)) + 1;       // mostly small, now and then large
            if (rnd() % 50 == 0) want = 0;
            size_t off = 0, need = 0;
            if (!bl.cut((size_t)want, &off, &need)) { refused++; continue; }
            ok++;
            const size_t al = need >= ((size_t)1 << 20) ? 4096 : pg::BlockList::ALIGN;
            if (need < std::max<uint64_t>(want, 1) || need % pg::BlockList::ALIGN || off % al || off + need > size) return -(long long)step;
            for (const auto& b : live) if (off < b.first + b.second && b.first < off + need) return -(long long)step;          // overlaps a live block
            live.emplace_back(off, need);
        } else if (rnd() % 5 == 0) {
            // a block is cut back to a part of itself (arena_shrink: the export array behind the counting pass): the rest is a hole at once, merged with the hole behind it
            const size_t i = (size_t)(rnd() % live.size());
            const size_t keep = (size_t)(rnd() % (live[i].second + 1));
            const size_t had = live[i].second;
            const bool did = bl.shrink(live[i].first, keep);
            const size_t want = pg::BlockList::round_up(keep ? keep : 1, 4096);
            if (did != (want < had)) return -(long long)step;
            if (did) live[i].second = want;
            if (bl.shrink(live[i].first + 1, 1)) return -(long long)step;                                                       // (not a block's start: refused)
        } else {
            const size_t i = (size_t)(rnd() % live.size());
            if (!bl.give_back(live[i].first)) return -(long long)step;
            if (bl.give_back(live[i].first)) return -(long long)step;                                                       // (twice is refused)
            live[i] = live.back();
            live.pop_back();
        }
        if (!check()) return -(long long)step;
    }
    while (!live.empty()) { if (!bl.give_back(live.back().first)) return -(long long)(n_ops + 1); live.pop_back(); }
    if (!bl.empty() || bl.free_.size() != 1 || bl.free_.begin()->first != 0 || bl.free_.begin()->second != size) return -(long long)(n_ops + 2);
    if (out) { out[0] = ok; out[1] = refused; out[2] = bl.peak; out[3] = max_holes; }
    return 0;
}
