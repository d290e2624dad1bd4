// pass2_kernels.hip -- pass 2 of pregraph on the device: every read is threaded through the finished edges and the
// pre-arcs are accumulated in HBM (prlRead2edge, standardPregraph/prlRead2path.c:786-1370).
//
// What the reference does with a 100 M-k-mer buffer and thrd_num threads per batch (chopKmer4read, searchKmer,
// parse1read, search1kmerPlus, thread_add1preArc, recordPathBin) is one kernel here, one lane per read:
//   * the k-mer sets come over exactly as the host stages left them (slot arrays in the reference's own layout,
//     an empty slot carries an impossible key), so a lookup is the reference's: set = signext(crc32) % thrd_num,
//     slot = key mod size, linear probing (newhash.c:277-318);
//   * parse1read's little state machine (prlRead2path.c:598-745) runs in registers; the (K+1)-mers of branch-to-branch
//     steps are resolved on the spot in a device copy of KmerSetsPatch;
//   * a pre-arc list is ordered by the first time each target was met (new targets go to the head,
//     prlRead2path.c:388-403), so the device keeps, per (from, to), the multiplicity and the smallest sequence number
//     (read ordinal, position) in an open-addressing table; the host only sorts and prints;
//   * -R: the walk of every read goes to a row of a staging matrix (the host writes .path), the per-edge marker
//     counts are atomic adds (saturated to 255 when written).
// All integer work, HBM-latency bound (one random 24/40-byte probe per k-mer); nothing here is shaped for MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/soapdenovo2_amd.h"
#include "device_ctx.hpp"
#include "extract.hpp"
#include "kmer.hpp"
#include "pass2.hpp"

namespace pg {

#define P2_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            pg_set_error(std::string("pass 2: ") + #call + ": " + hipGetErrorString(e_));                   \
            return PG_ENODEV;                                                                                  \
        }                                                                                                    \
    } while (0)

constexpr uint64_t P2_EMPTY = ~0ULL;
constexpr int P2_MAX_SETS = 255;

struct P2Params {
    const uint64_t* nodes;              // all sets back to back, (NW + 1) words a slot: key words, then A | B << 32
    const uint64_t* set_base;           // first slot of each set (device array, P entries)
    const uint64_t* set_size;
    uint32_t P, bias;
    int K;
    // (K+1)-mer patch table
    const uint64_t* patch_keys;         // NW words an entry
    const uint32_t* patch_val;          // id, twin an entry; id 0 = empty
    uint64_t patch_mask;
    // pre-arc table
    unsigned long long* arc_key;        // from << 32 | to, 0 = empty
    unsigned int* arc_cnt;
    unsigned long long* arc_first;
    uint64_t arc_mask;
    // -R
    uint32_t* stage;                    // [read in batch][max_nk]
    uint16_t* walk_len;                 // valid entries per read (0 when the walk does not qualify)
    unsigned int* marker;               // per edge
    int max_nk;
    uint32_t id_end;                    // num_ed + 1
    // counters: 0 reads without any usable item, 1 lookups that found nothing, 2 arc table overflow, 3 distinct arcs,
    // 4 markers, 5 edge id out of range
    unsigned long long* counters;
};

// ((r << 32) | chunk) mod d for r < d, exact for any d < 2^63
__device__ inline uint64_t mod_step32(uint64_t r, uint32_t chunk, uint64_t d) {
    if (d <= 0x100000000ULL) return ((r << 32) | chunk) % d;
    for (int b = 31; b >= 0; b--) {
        r = (r << 1) | ((chunk >> b) & 1u);
        if (r >= d) r -= d;
    }
    return r;
}
template <int NW>
__device__ inline uint64_t home_slot(const Kmer<NW>& k, uint64_t size) {
    if (NW == 2) {                                   // exact 128-bit modulus (newhash.c:36-57, 63-mer build)
        uint64_t r = k.w[0] % size;
        r = mod_step32(r, (uint32_t)(k.w[1] >> 32), size);
        return mod_step32(r, (uint32_t)k.w[1], size);
    }
    uint64_t t = k.w[0] % size;                      // the 127-mer build folds 32-bit chunks in 64-bit arithmetic
#pragma unroll
    for (int i = 1; i < NW; i++) {
        t = (t << 32 | (k.w[i] >> 32)) % size;
        t = (t << 32 | (k.w[i] & 0xffffffffULL)) % size;
    }
    return t;
}

// search_kmerset: the node's two counter words, or false
template <int NW>
__device__ inline bool find_node(const P2Params& p, const Kmer<NW>& key, const uint32_t* crc_tab, const uint64_t* set_geo, uint64_t& ab) {
    const uint32_t set = set_of_crc(kmer_crc32<NW>(key, crc_tab), p.P, p.bias);
    const uint64_t size = set_geo[2 * set + 1];
    const uint64_t* base = p.nodes + set_geo[2 * set] * (NW + 1);
    uint64_t hc = home_slot<NW>(key, size);
    for (uint64_t step = 0; step < size; step++) {
        const uint64_t* nd = base + hc * (NW + 1);
        const uint64_t w0 = nd[0];
        if (w0 == P2_EMPTY) return false;
        bool eq = w0 == key.w[0];
#pragma unroll
        for (int i = 1; i < NW; i++) eq = eq && nd[i] == key.w[i];
        if (eq) { ab = nd[NW]; return true; }
        if (++hc == size) hc = 0;
    }
    return false;
}

template <int NW>
__device__ inline uint32_t find_patch(const P2Params& p, const Kmer<NW>& key, bool smaller) {
    uint64_t h = kmer_mix<NW>(key) & p.patch_mask;
    for (;;) {
        const uint32_t id = p.patch_val[2 * h];
        if (!id) return 0;
        const uint64_t* k = p.patch_keys + h * NW;
        bool eq = true;
#pragma unroll
        for (int i = 0; i < NW; i++) eq = eq && k[i] == key.w[i];
        if (eq) return smaller ? id : id + p.patch_val[2 * h + 1] - 1;
        h = (h + 1) & p.patch_mask;
    }
}

__device__ inline void add_prearc(const P2Params& p, uint32_t from, uint32_t to, unsigned long long seq) {
    const unsigned long long key = ((unsigned long long)from << 32) | to;
    uint64_t h = (key * 0x9E3779B97F4A7C15ULL) >> 20;
    h &= p.arc_mask;
    for (uint64_t step = 0; step <= p.arc_mask; step++) {
        unsigned long long cur = __hip_atomic_load(&p.arc_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            unsigned long long expected = 0;
            if (__hip_atomic_compare_exchange_strong(&p.arc_key[h], &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicAdd(&p.counters[3], 1ULL);
                cur = key;
            } else cur = expected;
        }
        if (cur == key) {
            atomicAdd(&p.arc_cnt[h], 1u);
            atomicMin(&p.arc_first[h], seq);
            return;
        }
        h = (h + 1) & p.arc_mask;
    }
    atomicAdd(&p.counters[2], 1ULL);
}

// one lane = one read (chopKmer4read + searchKmer + parse1read + search1kmerPlus + thread_add1preArc + recordPathBin)
template <int NW>
__global__ __launch_bounds__(256) void p2_thread_kernel(P2Params p, const uint64_t* __restrict__ words, const uint64_t* __restrict__ word_off,
                                                        const int32_t* __restrict__ lens, uint64_t n_reads, uint64_t first_ordinal) {
    __shared__ uint32_t crc_tab[256];
    __shared__ uint64_t set_geo[2 * P2_MAX_SETS];                        // (first slot, size) of every set
    crc_tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
    if (threadIdx.x < p.P) { set_geo[2 * threadIdx.x] = p.set_base[threadIdx.x]; set_geo[2 * threadIdx.x + 1] = p.set_size[threadIdx.x]; }
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int K = p.K;
    const int len = lens[r];
    if (p.walk_len) p.walk_len[r] = 0;
    if (len < K + 1) return;                                             // prlRead2path.c:1103
    const uint64_t* rd = words + word_off[r];
    const Kmer<NW> filter = kmer_filter<NW>(K);
    const int nk = len - K + 1;
    uint32_t* row = p.stage ? p.stage + r * (uint64_t)p.max_nk : nullptr;
    const unsigned long long seq0 = (first_ordinal + r) << 16;

    Kmer<NW> word = read_kmer<NW>(rd, 0, K, filter);
    Kmer<NW> bal = kmer_rc<NW>(word, K);
    unsigned retain = 0;
    bool is_prev = false;
    Kmer<NW> prev_k;
#pragma unroll
    for (int i = 0; i < NW; i++) prev_k.w[i] = 0;
    int n_items = 0;             // items pushed since the last restart
    int n_valid = -1;            // index of the first unresolved item (id 0), -1 while there is none
    uint32_t last_id = 0;        // id of the latest item
    for (int j = 0; j < nk; j++) {
        if (j) kmer_roll<NW>(word, bal, read_base(rd, j + K - 1), K, filter);
        const bool smaller = kmer_less<NW>(word, bal);
        uint64_t ab;
        if (!find_node<NW>(p, smaller ? word : bal, crc_tab, set_geo, ab)) { atomicAdd(&p.counters[1], 1ULL); return; }
        const uint32_t A = (uint32_t)ab, B = (uint32_t)(ab >> 32);
        const bool linear = B & B_LINEAR, in_edge = (B >> B_INEDGE_SHIFT) & 3;
        if ((B & B_DELETED) || (linear && !in_edge)) {                   // deleted, or on a floating loop
            if (retain < 2) { retain = 0; n_items = 0; n_valid = -1; continue; }
            break;
        }
        uint32_t id = 0;
        bool push = false;
        if (linear) {
            const uint32_t twin = (B >> B_TWIN_SHIFT) & 3;
            id = smaller ? A : A + twin - 1;
            if (retain == 0 || is_prev) { push = true; is_prev = false; }
            else if (id != last_id) push = true;
        } else {
            if (is_prev) {                                               // branch node after branch node: a length-1 edge
                const Kmer<NW> plus = kmer_plus<NW>(prev_k, kmer_last<NW>(word));
                const Kmer<NW> bal_plus = rc_plus<NW>(plus, K);
                const bool sm = kmer_less<NW>(plus, bal_plus);
                id = find_patch<NW>(p, sm ? plus : bal_plus, sm);
                push = true;
            }
            is_prev = true;
            prev_k = word;
        }
        if (!push) continue;
        retain++;
        // thread_add1preArc walks the items pairwise up to the first unresolved one (prlRead2path.c:405-424); an item is
        // never taken back once two are retained, so the pair can go out as soon as its second half is known
        if (n_items >= 1 && n_valid < 0 && id != 0) {
            if (last_id >= p.id_end || id >= p.id_end) atomicAdd(&p.counters[5], 1ULL);
            else add_prearc(p, last_id, id, seq0 | (unsigned)(n_items - 1));
        }
        if (id == 0 && n_valid < 0) n_valid = n_items;
        if (row && n_items < p.max_nk) row[n_items] = id;
        last_id = id;
        n_items++;
    }
    if (retain < 1) atomicAdd(&p.counters[0], 1ULL);
    if (retain < 2 || !row) return;
    // recordPathBin (prlRead2path.c:478-543): the walk up to the first unresolved entry, if its first three are resolved
    const int upto = n_valid < 0 ? n_items : n_valid;
    if (upto < 3) return;
    p.walk_len[r] = (uint16_t)upto;
    for (int i = 0; i < upto; i++) {
        const uint32_t e = row[i];
        if (e < p.id_end) atomicAdd(&p.marker[e], 1u); else atomicAdd(&p.counters[5], 1ULL);
    }
    atomicAdd(&p.counters[4], (unsigned long long)upto);
}

// the occupied slots of the pre-arc table, densely (order does not matter: the host sorts)
__global__ void p2_compact_arcs(const unsigned long long* key, const unsigned int* cnt, const unsigned long long* first, uint64_t cap,
                                P2Arc* out, unsigned long long* n_out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = key[i];
        if (!k) continue;
        const unsigned long long at = atomicAdd(n_out, 1ULL);
        out[at] = P2Arc{(uint32_t)(k >> 32), (uint32_t)k, cnt[i], first[i]};
    }
}

__global__ void p2_fill_u64(unsigned long long* a, uint64_t n, unsigned long long v) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
struct P2Device {
    int device = 0, K = 0, nw = 2, P = 1, max_nk = 0;
    bool reps = false;
    uint32_t num_ed = 0;
    P2Params prm;
    uint64_t* d_nodes = nullptr;
    uint64_t* d_geo = nullptr;          // set_base[P] then set_size[P]
    uint64_t* d_patch_keys = nullptr;
    uint32_t* d_patch_val = nullptr;
    unsigned long long* d_arc_key = nullptr;
    unsigned int* d_arc_cnt = nullptr;
    unsigned long long* d_arc_first = nullptr;
    unsigned long long* d_counters = nullptr;
    unsigned int* d_marker = nullptr;
    // batch buffers (grown on demand)
    uint64_t* d_words = nullptr; size_t cap_words = 0;
    uint64_t* d_off = nullptr; int32_t* d_lens = nullptr; size_t cap_reads = 0;
    uint32_t* d_stage = nullptr; uint16_t* d_walk_len = nullptr; size_t cap_stage_reads = 0;
    uint64_t ordinal = 0;
    hipStream_t stream = nullptr;
};

static void p2_free(P2Device* d) {
    if (!d) return;
    hipSetDevice(d->device);
    hipFree(d->d_nodes); hipFree(d->d_geo); hipFree(d->d_patch_keys); hipFree(d->d_patch_val);
    hipFree(d->d_arc_key); hipFree(d->d_arc_cnt); hipFree(d->d_arc_first);
    hipFree(d->d_counters); hipFree(d->d_marker);
    hipFree(d->d_words); hipFree(d->d_off); hipFree(d->d_lens); hipFree(d->d_stage); hipFree(d->d_walk_len);
    if (d->stream) hipStreamDestroy(d->stream);
    delete d;
}

static int p2_create_impl(P2Device* d, const P2Sets& sets, const uint64_t* patch_keys, const uint32_t* patch_val, uint64_t patch_cap) {
    P2_HIP(hipSetDevice(d->device));
    P2_HIP(hipStreamCreate(&d->stream));
    memset(&d->prm, 0, sizeof(d->prm));
    const int NW1 = d->nw + 1;
    uint64_t total = 0;
    std::vector<uint64_t> geo(2 * (size_t)d->P);
    for (int s = 0; s < d->P; s++) { geo[s] = total; geo[d->P + s] = sets.size[s]; total += sets.size[s]; }
    P2_HIP(hipMalloc((void**)&d->d_geo, geo.size() * sizeof(uint64_t)));
    P2_HIP(hipMemcpy(d->d_geo, geo.data(), geo.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    P2_HIP(hipMalloc((void**)&d->d_nodes, std::max<uint64_t>(total, 1) * NW1 * sizeof(uint64_t)));
    for (int s = 0; s < d->P; s++)
        if (sets.size[s])
            P2_HIP(hipMemcpyAsync(d->d_nodes + geo[s] * NW1, sets.nodes[s], sets.size[s] * NW1 * sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
    P2_HIP(hipMalloc((void**)&d->d_patch_keys, patch_cap * d->nw * sizeof(uint64_t)));
    P2_HIP(hipMalloc((void**)&d->d_patch_val, patch_cap * 2 * sizeof(uint32_t)));
    P2_HIP(hipMemcpyAsync(d->d_patch_keys, patch_keys, patch_cap * d->nw * sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
    P2_HIP(hipMemcpyAsync(d->d_patch_val, patch_val, patch_cap * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
    // every edge has a handful of successors: eight slots an edge id keep the load low; the kernel counts overflows
    uint64_t arc_cap = 1 << 16;
    while (arc_cap < (uint64_t)d->num_ed * 8) arc_cap <<= 1;
    P2_HIP(hipMalloc((void**)&d->d_arc_key, arc_cap * sizeof(unsigned long long)));
    P2_HIP(hipMalloc((void**)&d->d_arc_cnt, arc_cap * sizeof(unsigned int)));
    P2_HIP(hipMalloc((void**)&d->d_arc_first, arc_cap * sizeof(unsigned long long)));
    P2_HIP(hipMemsetAsync(d->d_arc_key, 0, arc_cap * sizeof(unsigned long long), d->stream));
    P2_HIP(hipMemsetAsync(d->d_arc_cnt, 0, arc_cap * sizeof(unsigned int), d->stream));
    hipLaunchKernelGGL(p2_fill_u64, dim3(1024), dim3(256), 0, d->stream, d->d_arc_first, arc_cap, ~0ULL);
    P2_HIP(hipMalloc((void**)&d->d_counters, 8 * sizeof(unsigned long long)));
    P2_HIP(hipMemsetAsync(d->d_counters, 0, 8 * sizeof(unsigned long long), d->stream));
    if (d->reps) {
        P2_HIP(hipMalloc((void**)&d->d_marker, ((size_t)d->num_ed + 1) * sizeof(unsigned int)));
        P2_HIP(hipMemsetAsync(d->d_marker, 0, ((size_t)d->num_ed + 1) * sizeof(unsigned int), d->stream));
    }
    P2_HIP(hipStreamSynchronize(d->stream));
    P2Params& p = d->prm;
    p.nodes = d->d_nodes;
    p.set_base = d->d_geo; p.set_size = d->d_geo + d->P;
    p.P = (uint32_t)d->P; p.bias = set_bias((uint32_t)d->P); p.K = d->K;
    p.patch_keys = d->d_patch_keys; p.patch_val = d->d_patch_val; p.patch_mask = patch_cap - 1;
    p.arc_key = d->d_arc_key; p.arc_cnt = d->d_arc_cnt; p.arc_first = d->d_arc_first; p.arc_mask = arc_cap - 1;
    p.marker = d->d_marker; p.max_nk = d->max_nk; p.id_end = d->num_ed + 1;
    p.counters = d->d_counters;
    return PG_OK;
}

P2Device* p2_create(int device, int K, int nw, int n_sets, const P2Sets& sets, const uint64_t* patch_keys, const uint32_t* patch_val,
                    uint64_t patch_cap, uint32_t num_ed, int max_nk, bool reps) {
    if (n_sets < 1 || n_sets > P2_MAX_SETS || (nw != 2 && nw != 4) || (patch_cap & (patch_cap - 1)) || !patch_cap) {
        pg_set_error("pass 2: bad arguments");
        return nullptr;
    }
    P2Device* d = new P2Device();
    d->device = device; d->K = K; d->nw = nw; d->P = n_sets; d->num_ed = num_ed; d->max_nk = std::max(max_nk, 1); d->reps = reps;
    if (p2_create_impl(d, sets, patch_keys, patch_val, patch_cap) != PG_OK) { p2_free(d); return nullptr; }
    return d;
}

void p2_destroy(P2Device* d) { p2_free(d); }

int p2_add_packed(P2Device* d, const uint64_t* words, const uint64_t* word_off, const int32_t* lens, uint64_t n_reads, uint64_t n_words,
                  uint32_t* walks_out, uint16_t* walk_len_out) {
    if (!n_reads) return PG_OK;
    P2_HIP(hipSetDevice(d->device));
    if (n_words + 8 > d->cap_words) {
        hipFree(d->d_words);
        d->cap_words = (n_words + 8) * 5 / 4;
        P2_HIP(hipMalloc((void**)&d->d_words, d->cap_words * sizeof(uint64_t)));
    }
    if (n_reads > d->cap_reads) {
        hipFree(d->d_off); hipFree(d->d_lens);
        d->cap_reads = n_reads * 5 / 4;
        P2_HIP(hipMalloc((void**)&d->d_off, d->cap_reads * sizeof(uint64_t)));
        P2_HIP(hipMalloc((void**)&d->d_lens, d->cap_reads * sizeof(int32_t)));
    }
    if (d->reps && n_reads > d->cap_stage_reads) {
        hipFree(d->d_stage); hipFree(d->d_walk_len);
        d->cap_stage_reads = n_reads;
        P2_HIP(hipMalloc((void**)&d->d_stage, d->cap_stage_reads * (size_t)d->max_nk * sizeof(uint32_t)));
        P2_HIP(hipMalloc((void**)&d->d_walk_len, d->cap_stage_reads * sizeof(uint16_t)));
    }
    P2_HIP(hipMemcpyAsync(d->d_words, words, n_words * sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
    P2_HIP(hipMemsetAsync(d->d_words + n_words, 0, 8 * sizeof(uint64_t), d->stream));      // readable padding for the window loads
    P2_HIP(hipMemcpyAsync(d->d_off, word_off, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, d->stream));
    P2_HIP(hipMemcpyAsync(d->d_lens, lens, n_reads * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    P2Params p = d->prm;
    p.stage = d->reps ? d->d_stage : nullptr;
    p.walk_len = d->reps ? d->d_walk_len : nullptr;
    const dim3 grid((unsigned)((n_reads + 255) / 256)), block(256);
    if (d->nw == 2) hipLaunchKernelGGL(p2_thread_kernel<2>, grid, block, 0, d->stream, p, d->d_words, d->d_off, d->d_lens, n_reads, d->ordinal);
    else hipLaunchKernelGGL(p2_thread_kernel<4>, grid, block, 0, d->stream, p, d->d_words, d->d_off, d->d_lens, n_reads, d->ordinal);
    P2_HIP(hipGetLastError());
    if (d->reps && walks_out && walk_len_out) {
        P2_HIP(hipMemcpyAsync(walks_out, d->d_stage, n_reads * (size_t)d->max_nk * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
        P2_HIP(hipMemcpyAsync(walk_len_out, d->d_walk_len, n_reads * sizeof(uint16_t), hipMemcpyDeviceToHost, d->stream));
    }
    P2_HIP(hipStreamSynchronize(d->stream));
    d->ordinal += n_reads;
    return PG_OK;
}

int p2_finish(P2Device* d, P2Result& out) {
    P2_HIP(hipSetDevice(d->device));
    unsigned long long c[8];
    P2_HIP(hipMemcpy(c, d->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    if (c[1]) { pg_set_error("pass 2: " + std::to_string(c[1]) + " k-mer(s) of the reads are not in the sets"); return PG_EINVAL; }
    if (c[2]) { pg_set_error("pass 2: pre-arc table overflow"); return PG_ENOMEM; }
    if (c[5]) { pg_set_error("pass 2: edge id out of range"); return PG_EINVAL; }
    out.reads_deleted = (long long)c[0];
    out.markers = (long long)c[4];
    const uint64_t cap = d->prm.arc_mask + 1;
    const size_t n_arcs = (size_t)c[3];
    out.arcs.assign(n_arcs, P2Arc{0, 0, 0, 0});
    if (n_arcs) {
        P2Arc* d_arcs = nullptr;
        P2_HIP(hipMalloc((void**)&d_arcs, n_arcs * sizeof(P2Arc)));
        P2_HIP(hipMemsetAsync(d->d_counters + 6, 0, sizeof(unsigned long long), d->stream));
        hipLaunchKernelGGL(p2_compact_arcs, dim3(2048), dim3(256), 0, d->stream, d->d_arc_key, d->d_arc_cnt, d->d_arc_first, cap, d_arcs, d->d_counters + 6);
        P2_HIP(hipMemcpyAsync(out.arcs.data(), d_arcs, n_arcs * sizeof(P2Arc), hipMemcpyDeviceToHost, d->stream));
        unsigned long long got = 0;
        P2_HIP(hipMemcpyAsync(&got, d->d_counters + 6, sizeof(got), hipMemcpyDeviceToHost, d->stream));
        P2_HIP(hipStreamSynchronize(d->stream));
        hipFree(d_arcs);
        if (got != n_arcs) { pg_set_error("pass 2: pre-arc table is inconsistent"); return PG_EINVAL; }
    }
    out.marker.clear();
    if (d->reps) {
        out.marker.resize((size_t)d->num_ed + 1);
        P2_HIP(hipMemcpy(out.marker.data(), d->d_marker, out.marker.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
    }
    return PG_OK;
}

}  // namespace pg
