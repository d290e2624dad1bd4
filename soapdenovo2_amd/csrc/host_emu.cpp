// host_emu.cpp -- the device graph stages (dev_graph.hpp, dev_tips.hpp) instantiated over the HostBackend: the same function
// objects the HIP kernels run, driven by host threads.  TEST HOOKS ONLY (pg_host_emu_*): the `-m "not gpu"` tests compare them
// with the sequential host stages and with a plain model, so that the logic of the kernels is pinned before it ever meets a
// GPU.  Nothing in the product path calls these; call_pregraph runs the HipBackend instantiations (graph_kernels.hip).
#include <stdint.h>

#include <string>
#include <vector>

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
