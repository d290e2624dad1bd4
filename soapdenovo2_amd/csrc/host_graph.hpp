// host_graph.hpp -- small host helpers shared between the host translation units.
#pragma once
#include <stdint.h>
#include <string>

void pg_set_error(const std::string& s);

namespace pg {
const uint32_t* host_crc_table();
// size of a fresh reference k-mer set: next "prime" >= 1024, or the -a derived size (prlHashReads.c:369-390)
uint64_t ref_initial_set_size(int a_gb, int n_sets, int mer127);
}  // namespace pg
