// sort_records.hip -- exported node records ordered by (k-mer set, first-occurrence ordinal) on the device.
// The host's layout replay (host_graph.cpp: put_kmerset / encap_kmerset order, newhash.c:340-528) inserts every set's
// k-mers in the order the reference first met them; the record's last word is set << 56 | ordinal, so one radix sort of
// that word puts the records exactly in replay order and the host neither buckets nor sorts.  rocPRIM's device radix
// sort with 64-bit sizes (plain library plumbing, not a hot kernel), so there is no 2^31 limit: beyond 2^32 - 1 records
// the permutation is kept as 64-bit indices.  Only the bits that vary are sorted: the ordinals of an input use far fewer
// than 56 bits, so the key is packed as set << bits(max ordinal) | ordinal first.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>

#include "../../include/soapdenovo2_amd.h"
#include "arena.hpp"
#include "env.hpp"
#include "device_ctx.hpp"

namespace pg {

template <int RW, typename Idx>
__global__ void sr_tags(const uint64_t* __restrict__ rec, uint64_t n, int ord_bits, uint64_t* __restrict__ tag, Idx* __restrict__ idx) {
    const uint64_t ord_mask = ord_bits >= 56 ? PG_ORD_MASK : ((1ULL << ord_bits) - 1);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = rec[i * RW + RW - 1];
        tag[i] = ((t >> PG_ORD_BITS) << ord_bits) | (t & ord_mask);
        idx[i] = (Idx)i;
    }
}
template <int RW>
__global__ void sr_max_ord(const uint64_t* __restrict__ rec, uint64_t n, unsigned long long* __restrict__ out) {
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long o = rec[i * RW + RW - 1] & PG_ORD_MASK;
        m = o > m ? o : m;
    }
    for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
template <int RW, typename Idx>
__global__ void sr_gather(const uint64_t* __restrict__ rec, const Idx* __restrict__ idx, uint64_t n, uint64_t* __restrict__ out) {
    // 16 bytes a lane: consecutive lanes write consecutive pieces of the output; the reads are whole records
    constexpr int PIECES = RW / 2;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * PIECES; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / PIECES, w = i % PIECES;
        ((ulonglong2*)out)[i] = ((const ulonglong2*)(rec + (uint64_t)idx[r] * RW))[w];
    }
}

// order-independent digest of a record array: per-column sums mod 2^64, the sum of the coverage fields and the number of
// saturated ones -- what a caller compares between two passes over the same reads (any batching, any engine) and against
// the k-mer occurrences that went in (bench.py's conservation check)
__global__ __launch_bounds__(256) void sr_checksum(const uint64_t* __restrict__ rec, uint64_t n, int rw, unsigned long long* __restrict__ out) {
    unsigned long long col[6] = {0, 0, 0, 0, 0, 0}, cov = 0, sat = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t* r = rec + i * rw;
        for (int w = 0; w < rw; w++) col[w] += r[w];
        const uint32_t c = (uint32_t)(r[rw - 2] >> 24) & 0xFFu;
        cov += c;
        sat += c == 255u;
    }
    for (int w = 0; w < 8; w++) {
        unsigned long long v = w < 6 ? col[w] : (w == 6 ? cov : sat);
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[w], v);
    }
}

#define SR_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { pg_set_error(std::string("pg_sort_records: ") + #call + ": " + hipGetErrorString(e_)); rc = (e_ == hipErrorOutOfMemory) ? PG_ENOMEM : PG_ENODEV; goto done; } \
    } while (0)

template <int RW, typename Idx>
static int sort_impl(uint64_t* d_records, uint64_t n, void* d_ws, size_t ws_bytes, hipStream_t stream) {
    int rc = PG_OK;
    const bool verbose = pg::env_measure("PG_SORT_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) { if (verbose) { (void)hipStreamSynchronize(stream); const double t = now(); fprintf(stderr, "[sort] %s: %.3fs\n", what, t - t0); t0 = t; } };
    // one allocation for the keys and indices (double-buffered) and the sort's scratch; one for the permuted records
    unsigned char* work = nullptr;
    uint64_t* sorted = nullptr;
    unsigned long long* d_max = nullptr;
    unsigned long long h_max = 0;
    size_t tmp_bytes = 0;
    int ord_bits = 1;
    bool own_work = true, own_sorted = true;
    size_t work_bytes = 0, sorted_bytes = 0;
    const size_t a16 = 255;
    const size_t key_bytes = (n * sizeof(uint64_t) + a16) & ~a16, idx_bytes = (n * sizeof(Idx) + a16) & ~a16;
    uint64_t *tag_in, *tag_out;
    Idx *idx_in, *idx_out;
    SR_HIP(pg::arena_malloc((void**)&d_max, sizeof(unsigned long long)));
    SR_HIP(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(sr_max_ord<RW>, dim3(2048), dim3(256), 0, stream, d_records, n, d_max);
    SR_HIP(hipMemcpyAsync(&h_max, d_max, sizeof h_max, hipMemcpyDeviceToHost, stream));
    SR_HIP(hipStreamSynchronize(stream));
    while (ord_bits < 56 && (h_max >> ord_bits)) ord_bits++;
    SR_HIP((rocprim::radix_sort_pairs<rocprim::default_config, uint64_t*, uint64_t*, Idx*, Idx*, size_t>(
        nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0u, (unsigned)(ord_bits + 8), stream)));
    // the caller's workspace (call_pregraph hands over the record pool pass 1 is done with) saves two large hipMalloc /
    // hipFree pairs, which take seconds right after a big hipFree
    work_bytes = (2 * key_bytes + 2 * idx_bytes + tmp_bytes + 511) & ~(size_t)255;
    sorted_bytes = n * RW * sizeof(uint64_t);
    own_work = !d_ws || ws_bytes < work_bytes + 256;
    own_sorted = own_work || ws_bytes < work_bytes + sorted_bytes + 512;
    if (own_work) SR_HIP(pg::arena_malloc((void**)&work, work_bytes));
    else work = (unsigned char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
    tag_in = (uint64_t*)work; tag_out = (uint64_t*)(work + key_bytes);
    idx_in = (Idx*)(work + 2 * key_bytes); idx_out = (Idx*)(work + 2 * key_bytes + idx_bytes);
    lap("max ordinal + work allocation");
    hipLaunchKernelGGL((sr_tags<RW, Idx>), dim3(4096), dim3(256), 0, stream, d_records, n, ord_bits, tag_in, idx_in);
    SR_HIP(hipGetLastError());
    SR_HIP((rocprim::radix_sort_pairs<rocprim::default_config, uint64_t*, uint64_t*, Idx*, Idx*, size_t>(
        work + 2 * key_bytes + 2 * idx_bytes, tmp_bytes, tag_in, tag_out, idx_in, idx_out, (size_t)n, 0u, (unsigned)(ord_bits + 8), stream)));
    lap("radix sort of the tags");
    if (own_sorted) SR_HIP(pg::arena_malloc((void**)&sorted, sorted_bytes));
    else sorted = (uint64_t*)(work + work_bytes);
    hipLaunchKernelGGL((sr_gather<RW, Idx>), dim3(8192), dim3(256), 0, stream, d_records, idx_out, n, sorted);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(d_records, sorted, n * RW * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    SR_HIP(hipStreamSynchronize(stream));
    lap("gather + copy back");
done:
    if (own_work) (void)pg::arena_free(work);
    if (own_sorted) (void)pg::arena_free(sorted);
    (void)pg::arena_free(d_max);
    lap("free");
    return rc;
}

}  // namespace pg

extern "C" int pg_sort_records_ws(uint64_t* d_records, uint64_t n, int mer127, void* d_workspace, uint64_t workspace_bytes, void* stream_v) {
    using namespace pg;
    if (!n) return PG_OK;
    if (!d_records) { pg_set_error("pg_sort_records: null records"); return PG_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_v;
    // 32-bit record indices while they suffice; PG_SORT_WIDE=1 forces the 64-bit flavour (tests)
    bool wide = n > 0xFFFFFFFFULL;
    if (const char* e = pg::env_test("PG_SORT_WIDE")) wide = wide || atoi(e) != 0;
    if (mer127) return wide ? sort_impl<6, uint64_t>(d_records, n, d_workspace, workspace_bytes, stream) : sort_impl<6, uint32_t>(d_records, n, d_workspace, workspace_bytes, stream);
    return wide ? sort_impl<4, uint64_t>(d_records, n, d_workspace, workspace_bytes, stream) : sort_impl<4, uint32_t>(d_records, n, d_workspace, workspace_bytes, stream);
}

extern "C" int pg_sort_records(uint64_t* d_records, uint64_t n, int mer127, void* stream_v) {
    return pg_sort_records_ws(d_records, n, mer127, nullptr, 0, stream_v);
}

extern "C" int pg_records_checksum(const uint64_t* d_records, uint64_t n, int rec_words, uint64_t out[8], void* stream_v) {
    using namespace pg;
    if (!out || (!d_records && n) || rec_words < 3 || rec_words > 6) { pg_set_error("pg_records_checksum: bad argument"); return PG_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_v;
    int rc = PG_OK;
    unsigned long long* d = nullptr;
    SR_HIP(pg::arena_malloc((void**)&d, 8 * sizeof(unsigned long long)));
    SR_HIP(hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), stream));
    if (n) hipLaunchKernelGGL(sr_checksum, dim3(4096), dim3(256), 0, stream, d_records, n, rec_words, d);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(out, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SR_HIP(hipStreamSynchronize(stream));
done:
    (void)pg::arena_free(d);
    return rc;
}
