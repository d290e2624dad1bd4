// graph_dev.hpp -- seam between the host graph (host_graph.cpp, g++) and the device stages (graph_kernels.hip, hipcc).
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace pg {

struct P2Device;

// the k-mer sets as the host stages left them: per set a slot array of (NW + 1) 64-bit words a slot (key words, then
// A | B << 32), an empty slot has all-ones in its first word
struct P2Sets {
    const void* nodes[255];
    uint64_t size[255];
};
struct P2Arc { uint32_t from, to, mult; uint64_t first; };       // first = smallest (read ordinal << 16 | position)
struct P2Result {
    std::vector<P2Arc> arcs;
    std::vector<unsigned int> marker;                             // per edge id, unsaturated (-R only)
    long long reads_deleted = 0, markers = 0;
    int lanes = 1;                                                // > 1: a (from, to) pair may come from several lanes -- merge by sum and minimum
    // folded on the device (round 6): from, to, multiplicity of every pre-arc in FILE order -- source edges ascending, a source's targets latest
    // first-met first, the lanes' entries of one pair merged (prlRead2path.c:388-403, 426-476) -- and `arcs` stays empty; the host only prints
    bool folded = false;
    std::vector<uint32_t> folded3;
};

// one edge record as the device built it, in the reference's order (the host formats output_1edge's text from it)
struct P2EdgeRec {
    uint32_t length, bal;
    unsigned long long sum;              // sum of the interior nodes' left-arc counters
    unsigned long long text_off;         // where its `length` bases start in P2Edges::text
    uint64_t first_kmer[4], last_kmer[4];
};
struct P2Edges {
    std::vector<P2EdgeRec> recs;
    std::string text;
    long long n_ids = 0, n_len1 = 0;
};

// step by step: upload the sets; then either the host's (K+1)-mer table (p2_set_patch) or the edges built on the device
// (p2_build_edges: tags the device copy of the sets and fills the device (K+1)-mer table itself); then p2_begin_reads
// upload of the host's sets; set_device (may be null: everything on `device`) puts set s on HIP device set_device[s] of this
// process, the kernels run on `device` and reach the others' sets through peer mappings
P2Device* p2_open(int device, int K, int nw, int n_sets, const P2Sets& sets, int max_nk, const int* set_device = nullptr);
// SURVEY.md App. C "K6": static (-a) pools laid out on the device from the records as they lie there sorted by (set, ordinal)
// (every set has set_size slots); *unsuited = true (and nullptr) when a set fills its pool or holds >= 2^32 keys -- the
// caller replays on the host then.  p2_download_set: one set's slot array into host memory.  p2_fetch_words: device -> host
// copies from several host threads at once (n_words = 0: the calling thread is done).
P2Device* p2_open_layout(int device, int K, int nw, int n_sets, const uint64_t* d_records, const uint64_t* per_set_count, uint64_t set_size,
                         int max_nk, bool* unsuited);
int p2_download_set(P2Device* d, int set, void* dst);
// the sharded form of the same: p2_layout_rank lays out the sets ONE rank owns (n_own sets of set_size slots, back to back in
// a fresh allocation on its device; 1 = unsuited), p2_adopt makes the graph over sets that already lie in device memory
// (set s: set_size[s] slots at set_ptr[s] on set_device[s]; the allocations in `owned` change hands)
int p2_layout_rank(int device, int nw, int n_own, const uint64_t* d_records, const uint64_t* own_counts, uint64_t set_size, uint64_t** d_nodes_out,
                   void** alloc_out = nullptr);      // *alloc_out = what to hipFree in the end (the image may sit inside a block taken over)
// growable (-a 0) sets: set i of the rank ends with sizes_out[i] slots (the reference's growth schedule), the sets lie back to
// back; own_trailing[i]: a duplicate put arrived after the set's last new key (it still runs the growth test)
int p2_layout_rank_growable(int device, int nw, int n_own, const uint64_t* d_records, const uint64_t* own_counts, const unsigned char* own_trailing,
                            uint64_t init_size, uint64_t* sizes_out, uint64_t** d_nodes_out, void** alloc_out = nullptr);
P2Device* p2_adopt(int lead_device, int K, int nw, int n_sets, const uint64_t* set_size, const int* set_device, uint64_t* const* set_ptr,
                   const std::vector<std::pair<int, void*>>& owned, int max_nk);
int p2_fetch_words(int device, const uint64_t* d_src, uint64_t n_words, uint64_t* dst);
void pg_device_free_on(int device, void* d_ptr);      // hipFree on that device
void pg_device_release_layout(int device, void* d_ptr);   // a layout's allocation given up: re-offered if it was a taken-over block, freed otherwise
int p2_set_patch(P2Device* d, const uint64_t* patch_keys, const uint32_t* patch_val, uint64_t patch_cap);
int p2_build_edges(P2Device* d, P2Edges& out);
int p2_begin_reads(P2Device* d, uint32_t num_ed, bool reps);

// one dead-end start and where its walk over linear nodes stopped (global slots = set base + slot; far = ~0: the walk
// was longer than the cut-off)
struct P2TipWalk { unsigned long long pos, far; uint32_t first, far_smaller; };
// the tip walks of one scan in slot order; afterwards the host sends back the nodes it changed and re-marks
int p2_tip_walks(P2Device* d, int cut_len, bool thin, std::vector<P2TipWalk>& out);
// removeSingleTips / removeMinorTips decided on the device (dev_tips.hpp); the counts the reference prints
struct P2TipTotals { unsigned long long single = 0, minor = 0; int cycles = 0, rounds = 0; std::vector<unsigned long long> per_cycle; };
int p2_clip_tips(P2Device* d, bool cut_single, P2TipTotals& out);
// the vertices -- live non-linear nodes -- in slot order, nw key words each (what output_vertex prints)
int p2_list_vertices(P2Device* d, std::vector<uint64_t>& keys);
int p2_mirror_nodes(P2Device* d, const uint64_t* slots, const uint64_t* ab, uint64_t n);
int p2_remark_linear(P2Device* d);

// patch table: open addressing over `patch_cap` (a power of two) entries, NW key words + (id, twin) an entry, id 0 =
// empty, slot = kmer_mix(key) & (cap - 1), linear probing
P2Device* p2_create(int device, int K, int nw, int n_sets, const P2Sets& sets, const uint64_t* patch_keys, const uint32_t* patch_val,
                    uint64_t patch_cap, uint32_t num_ed, int max_nk, bool reps);
void p2_destroy(P2Device* d);
// one batch of 2-bit packed reads (pg_pack_read), read i at words + word_off[i]; with reps the walks come back as rows of
// max_nk ids (walks_out) with their lengths (walk_len_out, 0 = the read has no recorded walk)
int p2_add_packed(P2Device* d, const uint64_t* words, const uint64_t* word_off, const int32_t* lens, uint64_t n_reads, uint64_t n_words,
                  uint32_t* walks_out, uint16_t* walk_len_out);
int p2_add_packed_device(P2Device* d, const uint64_t* d_words, uint64_t n_reads, int read_len, int device);   // reads already on a lane's device, one length, back to back
int p2_add_packed_device_segments(P2Device* d, const uint64_t* const* d_segs, const uint64_t* seg_reads, int n_segs, int read_len, int device);   // ... all of pass 1's batches at once, threaded in genome order
int p2_add_packed_device_ragged(P2Device* d, const uint64_t* d_words, const uint64_t* d_word_off, const uint64_t* d_kmer_base, uint64_t n_reads, uint64_t n_kmers,
                                int device);                                                                    // ... any mix of lengths, with pass 1's index arrays
// the ranks of a sharded run (lane 0 = the lead's device): the per-set scans run on the owner's lane (set s -> lane s mod n_lanes),
// pass 2 deals its read batches to the lanes in turn, each with a pre-arc table of its own
int p2_use_lanes(P2Device* d, const int* lane_devices, int n_lanes);
int p2_finish(P2Device* d, P2Result& out);

}  // namespace pg
