// arena_list.hpp -- the block bookkeeping of the device arena (arena.cpp), free of HIP: a range [0, size) cut first-fit into blocks, holes
// coalesced when blocks come back.  Its own header so that the `-m "not gpu"` tests can run it on random sequences (pg_host_emu_arena_blocks).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <map>
#include <unordered_map>

namespace pg {

struct BlockList {
    static constexpr size_t ALIGN = 256;            // what hipMalloc promises at least; blocks of a megabyte and more start on 4 KiB
    std::map<size_t, size_t> free_;                 // offset -> bytes, coalesced
    std::unordered_map<size_t, size_t> used;        // offset -> bytes
    uint64_t in_use = 0, peak = 0, n_cut = 0, n_back = 0;

    static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
    void reset(size_t size) { free_.clear(); used.clear(); free_[0] = size; in_use = 0; }
    // first fit; false = no hole takes it.  *need_out = the bytes the block really holds (the request rounded up)
    bool cut(size_t bytes, size_t* off_out, size_t* need_out) {
        const size_t need = round_up(bytes ? bytes : 1, ALIGN);
        const size_t al = need >= ((size_t)1 << 20) ? 4096 : ALIGN;
        auto it = free_.begin();
        size_t pad = 0;
        for (; it != free_.end(); ++it) {
            pad = round_up(it->first, al) - it->first;
            if (it->second >= need + pad) break;
        }
        if (it == free_.end()) return false;
        const size_t hole_off = it->first, hole = it->second, off = hole_off + pad;
        free_.erase(it);
        if (pad) free_[hole_off] = pad;
        if (hole > pad + need) free_[off + need] = hole - pad - need;
        used[off] = need;
        in_use += need;
        if (in_use > peak) peak = in_use;
        n_cut++;
        *off_out = off; *need_out = need;
        return true;
    }
    // the block at `off` goes back (false: no such block); the hole merges with its neighbours
    bool give_back(size_t off) {
        auto it = used.find(off);
        if (it == used.end()) return false;
        size_t bytes = it->second;
        used.erase(it);
        in_use -= bytes;
        n_back++;
        auto nx = free_.lower_bound(off);
        if (nx != free_.end() && off + bytes == nx->first) { bytes += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { pv->second += bytes; return true; }
        }
        free_[off] = bytes;
        return true;
    }
    // the block at `off` keeps its first `bytes` bytes (rounded up), the rest becomes a hole; false: no such block, or it is not larger
    bool shrink(size_t off, size_t bytes) {
        auto it = used.find(off);
        if (it == used.end()) return false;
        const size_t keep = round_up(bytes ? bytes : 1, 4096), had = it->second;
        if (keep >= had) return false;
        it->second = keep;
        in_use -= had - keep;
        size_t hole_off = off + keep, hole = had - keep;
        auto nx = free_.lower_bound(hole_off);
        if (nx != free_.end() && hole_off + hole == nx->first) { hole += nx->second; free_.erase(nx); }
        free_[hole_off] = hole;
        return true;
    }
    bool empty() const { return used.empty(); }
};

}  // namespace pg
