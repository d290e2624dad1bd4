// sort_records.hip -- exported node records ordered by (k-mer set, first-occurrence ordinal) on the device.
// The host's layout replay (host_graph.cpp: put_kmerset / encap_kmerset order, newhash.c:340-528) inserts every set's
// k-mers in the order the reference first met them; the record's last word is set << 56 | ordinal, so one 64-bit radix
// sort of that word (rocPRIM through hipCUB -- plain library plumbing, not a hot kernel) puts the records exactly in
// replay order and the host neither buckets nor sorts.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <string>

#include "../../include/soapdenovo2_amd.h"
#include "device_ctx.hpp"

namespace pg {

template <int RW>
__global__ void sr_tags(const uint64_t* __restrict__ rec, uint64_t n, uint64_t* __restrict__ tag, uint32_t* __restrict__ idx) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        tag[i] = rec[i * RW + RW - 1];
        idx[i] = (uint32_t)i;
    }
}
template <int RW>
__global__ void sr_gather(const uint64_t* __restrict__ rec, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * RW; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / RW, w = i % RW;
        out[i] = rec[(uint64_t)idx[r] * RW + w];
    }
}

#define SR_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { pg_set_error(std::string("pg_sort_records: ") + #call + ": " + hipGetErrorString(e_)); rc = PG_ENODEV; goto done; } \
    } while (0)

}  // namespace pg

extern "C" int pg_sort_records(uint64_t* d_records, uint64_t n, int mer127, void* stream_v) {
    using namespace pg;
    if (!n) return PG_OK;
    if (!d_records) { pg_set_error("pg_sort_records: null records"); return PG_EINVAL; }
    if (n >= 0xFFFFFFFFULL) { pg_set_error("pg_sort_records: more than 2^32 - 1 records in one call"); return PG_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_v;
    const int RW = mer127 ? 6 : 4;
    int rc = PG_OK;
    uint64_t *tag_in = nullptr, *tag_out = nullptr, *sorted = nullptr;
    uint32_t *idx_in = nullptr, *idx_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    SR_HIP(hipMalloc((void**)&tag_in, n * sizeof(uint64_t)));
    SR_HIP(hipMalloc((void**)&tag_out, n * sizeof(uint64_t)));
    SR_HIP(hipMalloc((void**)&idx_in, n * sizeof(uint32_t)));
    SR_HIP(hipMalloc((void**)&idx_out, n * sizeof(uint32_t)));
    if (mer127) hipLaunchKernelGGL(sr_tags<6>, dim3(2048), dim3(256), 0, stream, d_records, n, tag_in, idx_in);
    else hipLaunchKernelGGL(sr_tags<4>, dim3(2048), dim3(256), 0, stream, d_records, n, tag_in, idx_in);
    SR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, tag_in, tag_out, idx_in, idx_out, (int)n, 0, 64, stream));
    SR_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
    SR_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, tag_in, tag_out, idx_in, idx_out, (int)n, 0, 64, stream));
    SR_HIP(hipMalloc((void**)&sorted, n * RW * sizeof(uint64_t)));
    if (mer127) hipLaunchKernelGGL(sr_gather<6>, dim3(4096), dim3(256), 0, stream, d_records, idx_out, n, sorted);
    else hipLaunchKernelGGL(sr_gather<4>, dim3(4096), dim3(256), 0, stream, d_records, idx_out, n, sorted);
    SR_HIP(hipMemcpyAsync(d_records, sorted, n * RW * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    SR_HIP(hipStreamSynchronize(stream));
done:
    hipFree(tag_in); hipFree(tag_out); hipFree(idx_in); hipFree(idx_out); hipFree(tmp); hipFree(sorted);
    return rc;
}
