// exchange.hip -- multi-GPU pass 1: the all-to-all that brings every k-mer occurrence to the GPU owning it.
//
// The reference fans k-mers out to its `thrd_num` sets through shared memory (every worker scans the whole buffer and keeps
// `hashBanBuffer[i] % thrd_num == id`, standardPregraph/prlHashReads.c:79-90).  Here the unit that travels is the
// super-k-mer record (~19 k-mers in 48 bytes) and owner(record) = minimizer partition mod n_ranks, so all occurrences of a
// k-mer meet on one GPU and each rank counts its own partitions with no further communication (partition_kernels.hip).
//
// Two transports behind one interface:
//   PG_COMM_RCCL  ncclSend / ncclRecv inside one group = a direct all-to-all over xGMI (every pair talks over its own
//                 link; nothing is relayed through a ring).  One rank per process (bench.py under torch.distributed.run,
//                 the id travels through its store) or one rank per host thread of one process (call_pregraph with
//                 SOAPDENOVO2_AMD_DEVICES=0,1,...).  librccl is loaded with dlopen at first use, so single-GPU users and
//                 the CPU-only ABI tests never need it.
//   PG_COMM_P2P   one process only: ranks meet at a barrier, publish their send buffers and pull their share with
//                 peer-to-peer copies (SDMA over xGMI).  Also works when several ranks share one device, which RCCL
//                 refuses -- that is how the N-rank path is exercised on a 1-GPU box.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device_ctx.hpp"
#include "arena.hpp"
#include "env.hpp"
#include "../../include/soapdenovo2_amd.h"

namespace {

// Memory a communicator owns for its exchanges (count words, send / receive regions).  With the RCCL transport it comes straight from hipMalloc: these
// buffers are handed to ncclSend / ncclRecv, they are allocated once per communicator (nothing for the arena to save), and RCCL has only ever met
// plain device allocations here -- one unknown less on the first multi-GPU run.  (pg::arena_free gives either kind back.)
static hipError_t comm_malloc(int transport, void** p, size_t bytes) {
    if (transport == PG_COMM_RCCL) return hipMalloc(p, bytes ? bytes : 1);
    return pg::arena_malloc(p, bytes);
}
#define X_TRY(expr)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pg_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            return (e_ == hipErrorOutOfMemory) ? PG_ENOMEM : PG_ENODEV;                        \
        }                                                                                      \
    } while (0)

// ---- librccl, resolved at run time -----------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool load() {
        if (lib) return true;
        const char* names[] = {pg::env_user("SOAPDENOVO2_AMD_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
            why = dlerror();
        }
        if (!lib) return false;
        auto sym = [&](const char* s) { void* p = dlsym(lib, s); if (!p) why = std::string("missing symbol ") + s; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !AllReduce || !GetErrorString) {
            dlclose(lib); lib = nullptr; return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
bool rccl_ready() { std::lock_guard<std::mutex> lk(g_rccl_mu); return g_rccl.load(); }

#define N_TRY(expr)                                                                            \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            pg_set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));               \
            return PG_ENODEV;                                                                  \
        }                                                                                      \
    } while (0)

// ---- in-process group: a reusable barrier and one mailbox per rank ------------------------------------------------
struct LocalGroup {
    int n = 0, refs = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t gen = 0;
    std::vector<const void*> a, b;         // published pointers (device memory of the publishing rank)
    std::vector<uint64_t> v;               // published scalar
    std::vector<int> device;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

}  // namespace

struct pg_comm {
    int n = 1, rank = 0, device = 0, transport = PG_COMM_P2P;
    ncclComm_t nccl = nullptr;
    LocalGroup* grp = nullptr;
    // working buffers of pg_count_reads_sharded / pg_exchange_gather_records (device memory on `device`)
    uint64_t* d_send_recs = nullptr;
    uint32_t* d_send_parts = nullptr;
    uint64_t cap = 0;                      // records per owner region
    int rw = 0;
    uint64_t* d_counts = nullptr;          // [n] records for owner o / [n] received from rank q
    uint64_t* d_rcounts = nullptr;
    uint64_t* d_recv_recs = nullptr;
    uint32_t* d_recv_parts = nullptr;
    uint64_t recv_cap = 0;
    std::vector<uint64_t> h_counts, h_rcounts;
    uint64_t sent_records = 0, recv_records = 0, rounds = 0;
    uint64_t regrouped_from = 0, regrouped_in = 0;   // distinct k-mers before / after pg_exchange_regroup_by_set
    // PG_COMM_HOST: the caller's all-to-all over host buffers (bench.py over gloo: ranks in several processes that share a GPU)
    pg_host_alltoallv_fn host_a2a = nullptr;
    void* host_user = nullptr;
    std::vector<char> h_stage_s, h_stage_r;
    // ---- the pipeline of pg_count_reads_sharded: two slots (send region + receive region each), the exchange of round i on
    // its own stream while the caller's stream cuts round i + 1 and appends round i - 1
    struct Slot {
        uint64_t* d_send_recs = nullptr; uint32_t* d_send_parts = nullptr;
        uint64_t* d_recv_recs = nullptr; uint32_t* d_recv_parts = nullptr;
        hipEvent_t routed = nullptr, exchanged = nullptr, t0 = nullptr, t1 = nullptr;
        uint64_t total_in = 0;
        bool pending = false;                 // exchanged (or on its way), not yet appended to the partition streams
        bool timed = false;                   // t0 / t1 hold an exchange that has not been added to exchange_ms yet
        bool used = false;                    // `exchanged` has been recorded at least once
    } slot[2];
    uint64_t pipe_cap = 0;                    // records per owner region (both slots)
    int pipe_rw = 0;
    hipStream_t xstream = nullptr;            // the exchange's stream
    uint64_t* d_pairs = nullptr;              // [2 n] what this rank tells rank o: (records for o, this rank's largest region | error flag)
    uint64_t* d_rpairs = nullptr;             // [2 n] the same from every rank
    uint64_t* h_pairs = nullptr;              // pinned: [4 n] = own pairs, received pairs
    uint64_t round_no = 0;
    double exchange_ms = 0;                   // device time of the record exchanges (events on the exchange stream)
    uint64_t host_syncs = 0, repeats = 0, bytes_sent = 0;
    pg_ctx* pipe_ctx = nullptr;
};

namespace {

int comm_alloc_small(pg_comm* c) {
    X_TRY(hipSetDevice(c->device));
    X_TRY(comm_malloc(c->transport, (void**)&c->d_counts, sizeof(uint64_t) * (size_t)c->n));
    X_TRY(comm_malloc(c->transport, (void**)&c->d_rcounts, sizeof(uint64_t) * (size_t)c->n));
    c->h_counts.assign(c->n, 0);
    c->h_rcounts.assign(c->n, 0);
    return PG_OK;
}

// Error discipline of everything collective below: a rank that fails locally still takes part in every barrier / group of the
// call (with nothing to give), remembers its first error and returns it at the end -- a rank that left early would leave its
// peers waiting forever in LocalGroup::barrier() or ncclRecv.  Calls that move data in several steps exchange an error word
// first (agree()), so that all ranks skip the later steps together.
struct FirstError {
    int rc = PG_OK;
    std::string why;
    void hip(hipError_t e, const char* what) {
        if (e == hipSuccess || rc) return;
        rc = e == hipErrorOutOfMemory ? PG_ENOMEM : PG_ENODEV;
        why = std::string(what) + ": " + hipGetErrorString(e);
    }
    void nccl(ncclResult_t r, const char* what) {
        if (r == ncclSuccess || rc) return;
        rc = PG_ENODEV;
        why = std::string(what) + ": " + g_rccl.GetErrorString(r);
    }
    void set(int code, const std::string& text) { if (!rc) { rc = code; why = text; } }
    int done() const { if (rc) pg_set_error(why); return rc; }
};

// every rank sends k words to every rank: d_send[o * k ..] goes to rank o, d_recv[q * k ..] comes from rank q
int alltoall_words(pg_comm* c, const uint64_t* d_send, uint64_t* d_recv, uint64_t k, hipStream_t st) {
    FirstError err;
    if (c->n == 1) { err.hip(hipMemcpyAsync(d_recv, d_send, k * 8, hipMemcpyDeviceToDevice, st), "copy"); return err.done(); }
    if (c->transport == PG_COMM_RCCL) {
        err.hip(hipMemcpyAsync(d_recv + (uint64_t)c->rank * k, d_send + (uint64_t)c->rank * k, k * 8, hipMemcpyDeviceToDevice, st), "copy");
        err.nccl(g_rccl.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < c->n; p++) {
            if (p == c->rank) continue;
            err.nccl(g_rccl.Send(d_send + (uint64_t)p * k, k, ncclUint64, p, c->nccl, st), "ncclSend");
            err.nccl(g_rccl.Recv(d_recv + (uint64_t)p * k, k, ncclUint64, p, c->nccl, st), "ncclRecv");
        }
        err.nccl(g_rccl.GroupEnd(), "ncclGroupEnd");
        return err.done();
    }
    LocalGroup* g = c->grp;
    err.hip(hipStreamSynchronize(st), "sync");                  // what we publish must be complete
    g->a[c->rank] = d_send;
    g->barrier();
    for (int q = 0; q < c->n; q++)
        err.hip(hipMemcpyAsync(d_recv + (uint64_t)q * k, (const uint64_t*)g->a[q] + (uint64_t)c->rank * k, k * 8, hipMemcpyDefault, st), "peer copy");
    err.hip(hipStreamSynchronize(st), "sync");
    g->barrier();                                               // everybody has pulled: the send buffers may change again
    return err.done();
}

// collective: the largest word any rank brings (an error code negated, a capacity, ...); scratch = the communicator's count arrays
int agree_max(pg_comm* c, uint64_t mine, uint64_t* out, hipStream_t st) {
    FirstError err;
    std::vector<uint64_t> h(c->n, mine);
    err.hip(hipMemcpyAsync(c->d_counts, h.data(), sizeof(uint64_t) * (size_t)c->n, hipMemcpyHostToDevice, st), "copy");
    const int rc = alltoall_words(c, c->d_counts, c->d_rcounts, 1, st);
    if (rc) err.set(rc, pg_last_error());
    err.hip(hipMemcpyAsync(h.data(), c->d_rcounts, sizeof(uint64_t) * (size_t)c->n, hipMemcpyDeviceToHost, st), "copy");
    err.hip(hipStreamSynchronize(st), "sync");
    uint64_t m = mine;
    for (int q = 0; q < c->n; q++) m = std::max(m, h[q]);
    *out = err.rc ? std::max<uint64_t>(m, 1) : m;
    return err.done();
}

// variable all-to-all of bytes: what goes to rank p lies at d_send + send_off[p] (send_cnt[p] bytes); what comes from rank q
// lands at d_recv + recv_off[q] (recv_cnt[q] bytes).  Host arrays of n entries.
int alltoallv_bytes(pg_comm* c, const void* d_send, const uint64_t* send_off, const uint64_t* send_cnt, void* d_recv, const uint64_t* recv_off,
                    const uint64_t* recv_cnt, hipStream_t st) {
    FirstError err;
    const int me = c->rank;
    const char* src = (const char*)d_send;
    char* dst = (char*)d_recv;
    if (c->transport == PG_COMM_RCCL || c->n == 1) {
        if (send_cnt[me]) err.hip(hipMemcpyAsync(dst + recv_off[me], src + send_off[me], send_cnt[me], hipMemcpyDeviceToDevice, st), "copy");
        if (c->n == 1) return err.done();
        err.nccl(g_rccl.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < c->n; p++) {
            if (p == me) continue;
            if (send_cnt[p]) err.nccl(g_rccl.Send(src + send_off[p], send_cnt[p], ncclUint8, p, c->nccl, st), "ncclSend");
            if (recv_cnt[p]) err.nccl(g_rccl.Recv(dst + recv_off[p], recv_cnt[p], ncclUint8, p, c->nccl, st), "ncclRecv");
        }
        err.nccl(g_rccl.GroupEnd(), "ncclGroupEnd");
        return err.done();
    }
    LocalGroup* g = c->grp;
    err.hip(hipStreamSynchronize(st), "sync");
    g->a[me] = d_send; g->b[me] = send_off;
    g->barrier();
    for (int q = 0; q < c->n; q++)
        if (recv_cnt[q]) err.hip(hipMemcpyAsync(dst + recv_off[q], (const char*)g->a[q] + ((const uint64_t*)g->b[q])[me], recv_cnt[q], hipMemcpyDefault, st), "peer copy");
    err.hip(hipStreamSynchronize(st), "sync");
    g->barrier();
    return err.done();
}

}  // namespace

extern "C" int pg_comm_unique_id(uint8_t id[128]) {
    if (!id) { pg_set_error("null argument"); return PG_EINVAL; }
    if (!rccl_ready()) { pg_set_error("librccl could not be loaded: " + g_rccl.why); return PG_ENODEV; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    N_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, 128);
    return PG_OK;
}

extern "C" pg_comm* pg_comm_create(int n_ranks, int rank, int device, const uint8_t id[128]) {
    if (n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks || !id) { pg_set_error("pg_comm_create: bad arguments"); return nullptr; }
    if (!rccl_ready()) { pg_set_error("librccl could not be loaded: " + g_rccl.why); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { pg_set_error("pg_comm_create: hipSetDevice failed"); return nullptr; }
    pg_comm* c = new pg_comm();
    c->n = n_ranks; c->rank = rank; c->device = device; c->transport = PG_COMM_RCCL;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, n_ranks, u, rank);
    if (r != ncclSuccess) { pg_set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete c; return nullptr; }
    if (comm_alloc_small(c) != PG_OK) { pg_comm_destroy(c); return nullptr; }
    return c;
}

extern "C" int pg_comm_create_local(int n_ranks, const int* devices, int transport, pg_comm** out) {
    if (n_ranks < 1 || n_ranks > 256 || !devices || !out) { pg_set_error("pg_comm_create_local: bad arguments"); return PG_EINVAL; }
    bool distinct = true;
    for (int i = 0; i < n_ranks; i++) for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    if (transport < 0) {
        transport = (distinct && n_ranks > 1) ? PG_COMM_RCCL : PG_COMM_P2P;
        if (const char* e = pg::env_user("SOAPDENOVO2_AMD_EXCHANGE")) {
            if (!strcmp(e, "p2p")) transport = PG_COMM_P2P;
            else if (!strcmp(e, "rccl")) transport = PG_COMM_RCCL;
        }
    }
    if (transport == PG_COMM_RCCL && !distinct && n_ranks > 1) { pg_set_error("RCCL needs one device per rank; ranks sharing a device use the p2p transport"); return PG_EINVAL; }
    for (int i = 0; i < n_ranks; i++) out[i] = nullptr;
    if (transport == PG_COMM_RCCL) {
        uint8_t id[128];
        int rc = pg_comm_unique_id(id);
        if (rc) return rc;
        std::vector<std::string> errs(n_ranks);
        std::vector<std::thread> th;
        for (int i = 0; i < n_ranks; i++)                       // ncclCommInitRank is collective: all ranks at once
            th.emplace_back([&, i] { out[i] = pg_comm_create(n_ranks, i, devices[i], id); if (!out[i]) errs[i] = pg_last_error(); });
        for (auto& t : th) t.join();
        for (int i = 0; i < n_ranks; i++)
            if (!out[i]) {
                pg_set_error(errs[i]);
                for (int j = 0; j < n_ranks; j++) { pg_comm_destroy(out[j]); out[j] = nullptr; }
                return PG_ENODEV;
            }
        return PG_OK;
    }
    LocalGroup* g = new LocalGroup();
    g->n = n_ranks; g->refs = n_ranks;
    g->a.assign(n_ranks, nullptr); g->b.assign(n_ranks, nullptr); g->v.assign(n_ranks, 0);
    g->device.assign(devices, devices + n_ranks);
    for (int i = 0; i < n_ranks; i++) {
        pg_comm* c = new pg_comm();
        c->n = n_ranks; c->rank = i; c->device = devices[i]; c->transport = PG_COMM_P2P; c->grp = g;
        out[i] = c;
        if (comm_alloc_small(c) != PG_OK) {
            const std::string why = pg_last_error();
            for (int j = 0; j <= i; j++) { pg_comm_destroy(out[j]); out[j] = nullptr; }
            for (int j = i + 1; j < n_ranks; j++) if (--g->refs == 0) delete g;
            pg_set_error(why);
            return PG_ENOMEM;
        }
        for (int j = 0; j < i; j++)                              // direct peer copies where the fabric allows them
            if (devices[j] != devices[i]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[i]); (void)hipDeviceEnablePeerAccess(devices[j], 0);
                    (void)hipSetDevice(devices[j]); (void)hipDeviceEnablePeerAccess(devices[i], 0);
                    (void)hipGetLastError();
                }
            }
    }
    return PG_OK;
}

namespace { void pipe_free(pg_comm* c); }
extern "C" void pg_comm_destroy(pg_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl) g_rccl.CommDestroy(c->nccl);
    for (void* p : {(void*)c->d_send_recs, (void*)c->d_send_parts, (void*)c->d_counts, (void*)c->d_rcounts, (void*)c->d_recv_recs, (void*)c->d_recv_parts})
        if (p) (void)pg::arena_free(p);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->pipe_ctx && c->pipe_ctx->pending_user == c) { c->pipe_ctx->pending_drain = nullptr; c->pipe_ctx->pending_detach = nullptr; c->pipe_ctx->pending_user = nullptr; }
    pipe_free(c);
    for (auto& sl : c->slot) for (hipEvent_t e : {sl.routed, sl.exchanged, sl.t0, sl.t1}) if (e) (void)hipEventDestroy(e);
    if (c->d_pairs) (void)pg::arena_free(c->d_pairs);
    if (c->d_rpairs) (void)pg::arena_free(c->d_rpairs);
    if (c->h_pairs) (void)hipHostFree(c->h_pairs);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->grp) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->grp->m); last = --c->grp->refs == 0; }
        if (last) delete c->grp;
    }
    delete c;
}

extern "C" int pg_comm_rank(const pg_comm* c) { return c ? c->rank : -1; }
extern "C" int pg_comm_size(const pg_comm* c) { return c ? c->n : 0; }
extern "C" int pg_comm_transport(const pg_comm* c) { return c ? c->transport : -1; }

extern "C" int pg_exchange_counts(pg_comm* c, const uint64_t* d_send_counts, uint64_t* d_recv_counts, void* stream) {
    if (!c || !d_send_counts || !d_recv_counts) { pg_set_error("null argument"); return PG_EINVAL; }
    X_TRY(hipSetDevice(c->device));
    return alltoall_words(c, d_send_counts, d_recv_counts, 1, (hipStream_t)stream);
}

extern "C" int pg_exchange_allreduce_u64(pg_comm* c, uint64_t* d_buf, uint64_t n, void* stream) {
    if (!c || (!d_buf && n)) { pg_set_error("null argument"); return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    X_TRY(hipSetDevice(c->device));
    if (c->n == 1 || n == 0) return PG_OK;
    if (c->transport == PG_COMM_RCCL) { N_TRY(g_rccl.AllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, c->nccl, st)); return PG_OK; }
    LocalGroup* g = c->grp;
    X_TRY(hipStreamSynchronize(st));
    g->a[c->rank] = d_buf;
    g->barrier();
    std::vector<uint64_t> sum(n, 0), tmp(n);
    for (int q = 0; q < c->n; q++) {
        X_TRY(hipMemcpy(tmp.data(), g->a[q], n * 8, hipMemcpyDefault));
        for (uint64_t i = 0; i < n; i++) sum[i] += tmp[i];
    }
    g->barrier();                                               // every rank has read every buffer
    X_TRY(hipMemcpy(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice));
    return PG_OK;
}

// Variable all-to-all of super-k-mer records and their partition ids.  Sender side: owner o's records lie at
// d_send_recs + o * cap * rw, its ids at d_send_parts + o * cap, send_counts[o] of them.  Receiver side: what rank q sent
// lands behind what ranks < q sent (recv_counts[] as exchanged by pg_exchange_counts), densely.
extern "C" int pg_exchange_records(pg_comm* c, const uint64_t* d_send_recs, const uint32_t* d_send_parts, uint64_t cap, int rw,
                                   const uint64_t* send_counts, const uint64_t* recv_counts, uint64_t* d_recv_recs, uint32_t* d_recv_parts,
                                   void* stream) {
    if (!c || !send_counts || !recv_counts) { pg_set_error("null argument"); return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    X_TRY(hipSetDevice(c->device));
    const int n = c->n;
    std::vector<uint64_t> so(n), sc(n), ro(n), rcn(n), so2(n), sc2(n), ro2(n), rc2(n);
    uint64_t at = 0;
    for (int q = 0; q < n; q++) {
        so[q] = (uint64_t)q * cap * rw * 8; sc[q] = send_counts[q] * (uint64_t)rw * 8; ro[q] = at * rw * 8; rcn[q] = recv_counts[q] * (uint64_t)rw * 8;
        so2[q] = (uint64_t)q * cap * 4; sc2[q] = send_counts[q] * 4; ro2[q] = at * 4; rc2[q] = recv_counts[q] * 4;
        at += recv_counts[q];
    }
    // both steps are entered by every rank whatever the first one returned
    const int e1 = alltoallv_bytes(c, d_send_recs, so.data(), sc.data(), d_recv_recs, ro.data(), rcn.data(), st);
    const std::string w1 = e1 ? pg_last_error() : "";
    const int e2 = alltoallv_bytes(c, d_send_parts, so2.data(), sc2.data(), d_recv_parts, ro2.data(), rc2.data(), st);
    if (e1) { pg_set_error(w1); return e1; }
    return e2;
}

// ---- pass 1 on n ranks, pipelined -------------------------------------------------------------------------------------------
// One call = one round (collective: every rank calls it once per round, with n_reads = 0 when it has nothing).  Round i:
//   caller's stream:   cut batch i into records grouped by owner (send region of slot i & 1); pack (count, largest) pairs
//   exchange stream:   [behind the records of round i - 1]  pairs all-to-all (2 words a peer)           --> ONE host wait a round
//   host:              every rank's counts and flags: an error anywhere ends the round for all; a region that overflowed
//                      anywhere makes all ranks grow alike and cut again
//   caller's stream:   append round i - 1 (its records arrived while batch i was cut)
//   exchange stream:   records + partition ids of round i, variable all-to-all, no host wait
// so the records of a round travel while the next batch is cut and the previous one is appended; the last round is appended
// by whatever consumes the partition streams next (pg_finalize) or by pg_comm_flush.  The reference's fan-out through shared
// memory has no serial section either (prlHashReads.c:79-90).  Receive regions hold n x cap records -- what n full send
// regions could bring -- so a round never has to size them; they grow with the send regions, all ranks alike.
// A rank that fails still goes through the round's collective steps (with nothing to give) and says so in its flag word.
namespace {

constexpr uint64_t PIPE_ERR = 1ULL << 62;

__global__ void pack_pairs_kernel(const uint64_t* counts, int n, uint64_t floor, uint64_t flag_or, uint64_t* pairs) {
    uint64_t largest = floor;
    for (int q = 0; q < n; q++) largest = counts[q] > largest ? counts[q] : largest;
    for (int o = threadIdx.x; o < n; o += blockDim.x) { pairs[2 * o] = counts[o]; pairs[2 * o + 1] = largest | flag_or; }
}

void pipe_free(pg_comm* c) {
    for (auto& sl : c->slot) {
        for (void* p : {(void*)sl.d_send_recs, (void*)sl.d_send_parts, (void*)sl.d_recv_recs, (void*)sl.d_recv_parts}) if (p) (void)pg::arena_free(p);
        sl.d_send_recs = nullptr; sl.d_send_parts = nullptr; sl.d_recv_recs = nullptr; sl.d_recv_parts = nullptr;
    }
    c->pipe_cap = 0;
}
int pipe_alloc(pg_comm* c, uint64_t cap, int rw, FirstError& err) {
    pipe_free(c);
    const uint64_t n = (uint64_t)c->n;
    for (auto& sl : c->slot) {
        err.hip(comm_malloc(c->transport, (void**)&sl.d_send_recs, cap * n * rw * 8), "hipMalloc (send region)");
        err.hip(comm_malloc(c->transport, (void**)&sl.d_send_parts, cap * n * 4), "hipMalloc (send region)");
        err.hip(comm_malloc(c->transport, (void**)&sl.d_recv_recs, cap * n * rw * 8), "hipMalloc (receive region)");
        err.hip(comm_malloc(c->transport, (void**)&sl.d_recv_parts, cap * n * 4), "hipMalloc (receive region)");
    }
    if (err.rc) { pipe_free(c); return err.rc; }
    c->pipe_cap = cap; c->pipe_rw = rw;
    return PG_OK;
}
int pipe_init(pg_comm* c, FirstError& err) {
    if (c->xstream) return PG_OK;
    err.hip(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking), "hipStreamCreate");
    for (auto& sl : c->slot) {
        err.hip(hipEventCreateWithFlags(&sl.routed, hipEventDisableTiming), "hipEventCreate");
        err.hip(hipEventCreateWithFlags(&sl.exchanged, hipEventDisableTiming), "hipEventCreate");
        err.hip(hipEventCreate(&sl.t0), "hipEventCreate");
        err.hip(hipEventCreate(&sl.t1), "hipEventCreate");
    }
    err.hip(comm_malloc(c->transport, (void**)&c->d_pairs, sizeof(uint64_t) * 2 * (size_t)c->n), "hipMalloc");
    err.hip(comm_malloc(c->transport, (void**)&c->d_rpairs, sizeof(uint64_t) * 2 * (size_t)c->n), "hipMalloc");
    err.hip(hipHostMalloc((void**)&c->h_pairs, sizeof(uint64_t) * 4 * (size_t)c->n, hipHostMallocPortable), "hipHostMalloc");
    return err.rc;
}
// the exchange of a slot has been timed: add it up (its events are complete: the caller waited for something behind them)
void pipe_collect_time(pg_comm* c, pg_comm::Slot& sl) {
    if (!sl.timed) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, sl.t0, sl.t1) == hipSuccess) c->exchange_ms += (double)ms;
    sl.timed = false;
}
// append what a slot received to the partition streams (caller's stream, behind the slot's exchange)
int pipe_ingest(pg_comm* c, pg_ctx* ctx, pg_comm::Slot& sl, hipStream_t st) {
    if (!sl.pending) return PG_OK;
    sl.pending = false;
    if (hipStreamWaitEvent(st, sl.exchanged, 0) != hipSuccess) { pg_set_error("hipStreamWaitEvent failed"); return PG_ENODEV; }
    const int rc = pg::e2_ingest(ctx, sl.d_recv_recs, sl.d_recv_parts, sl.total_in, st);
    if (rc) return rc;
    c->recv_records += sl.total_in;
    ctx->batches++;
    return PG_OK;
}
// everything in flight lands and is appended; the streams are idle afterwards
int pipe_drain(pg_comm* c, pg_ctx* ctx, hipStream_t st) {
    FirstError err;
    err.hip(hipSetDevice(c->device), "hipSetDevice");
    for (int k = 0; k < 2; k++) {
        pg_comm::Slot& sl = c->slot[(c->round_no + (uint64_t)k) & 1];       // the older round first
        if (sl.pending && ctx) { const int rc = pipe_ingest(c, ctx, sl, st); if (rc) err.set(rc, pg_last_error()); }
        sl.pending = false;
    }
    if (c->xstream) err.hip(hipStreamSynchronize(c->xstream), "sync");
    err.hip(hipStreamSynchronize(st), "sync");
    for (auto& sl : c->slot) pipe_collect_time(c, sl);
    return err.done();
}
int pipe_drain_hook(pg_ctx* ctx, void* user, hipStream_t st) { return pipe_drain((pg_comm*)user, ctx, st); }
void pipe_detach_hook(pg_ctx* ctx, void* user) { pg_comm* c = (pg_comm*)user; if (c && c->pipe_ctx == ctx) c->pipe_ctx = nullptr; }

// the pairs of this round: every rank's (records for me, its largest region | flags) into h_pairs[2n ..], own pairs into h_pairs[0 ..]
int pipe_exchange_pairs(pg_comm* c, hipStream_t st, pg_comm::Slot& sl, FirstError& err) {
    const int n = c->n;
    const size_t bytes = sizeof(uint64_t) * 2 * (size_t)n;
    if (c->transport == PG_COMM_RCCL || n == 1) {
        // on the exchange stream, behind the previous round's records and behind this round's cut
        err.hip(hipStreamWaitEvent(c->xstream, sl.routed, 0), "hipStreamWaitEvent");
        if (n == 1) err.hip(hipMemcpyAsync(c->d_rpairs, c->d_pairs, bytes, hipMemcpyDeviceToDevice, c->xstream), "copy");
        else { const int rc = alltoall_words(c, c->d_pairs, c->d_rpairs, 2, c->xstream); if (rc) err.set(rc, pg_last_error()); }
        err.hip(hipMemcpyAsync(c->h_pairs, c->d_pairs, bytes, hipMemcpyDeviceToHost, c->xstream), "copy");
        err.hip(hipMemcpyAsync(c->h_pairs + 2 * n, c->d_rpairs, bytes, hipMemcpyDeviceToHost, c->xstream), "copy");
        err.hip(hipStreamSynchronize(c->xstream), "sync");
        c->host_syncs++;
        return err.rc;
    }
    // one process (mailbox of the group) or the caller's all-to-all: the pairs travel through host memory
    err.hip(hipMemcpyAsync(c->h_pairs, c->d_pairs, bytes, hipMemcpyDeviceToHost, st), "copy");
    err.hip(hipStreamSynchronize(st), "sync");
    c->host_syncs++;
    if (c->transport == PG_COMM_HOST) {
        std::vector<uint64_t> off(n), cnt(n, 16);
        for (int q = 0; q < n; q++) off[q] = 16 * (uint64_t)q;
        if (c->host_a2a(c->host_user, c->h_pairs, off.data(), cnt.data(), c->h_pairs + 2 * n, off.data(), cnt.data()) != 0) err.set(PG_ENODEV, "the caller's all-to-all failed");
        return err.rc;
    }
    LocalGroup* g = c->grp;
    g->a[c->rank] = c->h_pairs;
    g->barrier();
    for (int q = 0; q < n; q++) { const uint64_t* theirs = (const uint64_t*)g->a[q]; c->h_pairs[2 * n + 2 * q] = theirs[2 * c->rank]; c->h_pairs[2 * n + 2 * q + 1] = theirs[2 * c->rank + 1]; }
    g->barrier();
    return err.rc;
}

// records + partition ids of the slot, variable all-to-all on the exchange stream; nobody waits for it here
int pipe_exchange_records(pg_comm* c, pg_comm::Slot& sl, int rw, hipStream_t st, FirstError& err) {
    const int n = c->n, me = c->rank;
    const uint64_t cap = c->pipe_cap;
    std::vector<uint64_t> so(n), sc(n), ro(n), rcn(n), so2(n), sc2(n), ro2(n), rc2(n);
    uint64_t at = 0;
    for (int q = 0; q < n; q++) {
        so[q] = (uint64_t)q * cap * rw * 8; sc[q] = c->h_counts[q] * (uint64_t)rw * 8; ro[q] = at * rw * 8; rcn[q] = c->h_rcounts[q] * (uint64_t)rw * 8;
        so2[q] = (uint64_t)q * cap * 4; sc2[q] = c->h_counts[q] * 4; ro2[q] = at * 4; rc2[q] = c->h_rcounts[q] * 4;
        at += c->h_rcounts[q];
    }
    sl.total_in = at;
    hipStream_t xs = c->xstream;
    // behind this round's cut -- and with it behind the append of two rounds ago, which read this slot's receive region
    err.hip(hipStreamWaitEvent(xs, sl.routed, 0), "hipStreamWaitEvent");
    if (c->transport == PG_COMM_RCCL || n == 1) {
        err.hip(hipEventRecord(sl.t0, xs), "hipEventRecord");
        int rc = alltoallv_bytes(c, sl.d_send_recs, so.data(), sc.data(), sl.d_recv_recs, ro.data(), rcn.data(), xs);
        if (rc) err.set(rc, pg_last_error());
        rc = alltoallv_bytes(c, sl.d_send_parts, so2.data(), sc2.data(), sl.d_recv_parts, ro2.data(), rc2.data(), xs);
        if (rc) err.set(rc, pg_last_error());
        err.hip(hipEventRecord(sl.t1, xs), "hipEventRecord");
    } else if (c->transport == PG_COMM_HOST) {
        // staged through the host and the caller's all-to-all (test set-ups): device -> host, exchange, host -> device; blocking
        uint64_t s_bytes = 0, r_bytes = 0;
        std::vector<uint64_t> hso(n), hro(n), hsc(n), hrc(n);
        for (int q = 0; q < n; q++) { hso[q] = s_bytes; hsc[q] = sc[q] + sc2[q]; s_bytes += hsc[q]; hro[q] = r_bytes; hrc[q] = rcn[q] + rc2[q]; r_bytes += hrc[q]; }
        c->h_stage_s.resize(s_bytes + 8); c->h_stage_r.resize(r_bytes + 8);
        err.hip(hipStreamWaitEvent(xs, sl.routed, 0), "hipStreamWaitEvent");
        err.hip(hipEventRecord(sl.t0, xs), "hipEventRecord");
        for (int q = 0; q < n; q++) {
            if (sc[q]) err.hip(hipMemcpyAsync(c->h_stage_s.data() + hso[q], (const char*)sl.d_send_recs + so[q], sc[q], hipMemcpyDeviceToHost, xs), "copy");
            if (sc2[q]) err.hip(hipMemcpyAsync(c->h_stage_s.data() + hso[q] + sc[q], (const char*)sl.d_send_parts + so2[q], sc2[q], hipMemcpyDeviceToHost, xs), "copy");
        }
        err.hip(hipStreamSynchronize(xs), "sync");
        if (c->host_a2a(c->host_user, c->h_stage_s.data(), hso.data(), hsc.data(), c->h_stage_r.data(), hro.data(), hrc.data()) != 0) err.set(PG_ENODEV, "the caller's all-to-all failed");
        for (int q = 0; q < n; q++) {
            if (rcn[q]) err.hip(hipMemcpyAsync((char*)sl.d_recv_recs + ro[q], c->h_stage_r.data() + hro[q], rcn[q], hipMemcpyHostToDevice, xs), "copy");
            if (rc2[q]) err.hip(hipMemcpyAsync((char*)sl.d_recv_parts + ro2[q], c->h_stage_r.data() + hro[q] + rcn[q], rc2[q], hipMemcpyHostToDevice, xs), "copy");
        }
        err.hip(hipEventRecord(sl.t1, xs), "hipEventRecord");
        err.hip(hipStreamSynchronize(xs), "sync");                  // (the staging buffers are reused by the next round)
    } else {
        // one process: every rank has cut (the pairs' barrier came after each rank waited for its own cut); publish the regions and
        // pull -- asynchronously, on the exchange stream
        LocalGroup* g = c->grp;
        struct Pub { const void* recs; const void* parts; uint64_t cap; };
        Pub mine{sl.d_send_recs, sl.d_send_parts, cap};
        g->a[me] = &mine;
        g->b[me] = c->h_counts.data();
        g->barrier();
        err.hip(hipEventRecord(sl.t0, xs), "hipEventRecord");
        for (int q = 0; q < n; q++) {
            const Pub* theirs = (const Pub*)g->a[q];
            if (rcn[q]) err.hip(hipMemcpyAsync((char*)sl.d_recv_recs + ro[q], (const char*)theirs->recs + (uint64_t)me * theirs->cap * rw * 8, rcn[q], hipMemcpyDefault, xs), "peer copy");
            if (rc2[q]) err.hip(hipMemcpyAsync((char*)sl.d_recv_parts + ro2[q], (const char*)theirs->parts + (uint64_t)me * theirs->cap * 4, rc2[q], hipMemcpyDefault, xs), "peer copy");
        }
        err.hip(hipEventRecord(sl.t1, xs), "hipEventRecord");
        g->barrier();                                               // (`mine` is on this stack)
    }
    err.hip(hipEventRecord(sl.exchanged, xs), "hipEventRecord");
    sl.used = true; sl.timed = true; sl.pending = true;
    for (int q = 0; q < n; q++) { c->sent_records += c->h_counts[q]; if (q != me) c->bytes_sent += sc[q] + sc2[q]; }
    (void)st;
    return err.rc;
}

}  // namespace

extern "C" int pg_count_reads_sharded(pg_ctx* ctx, pg_comm* c, const uint64_t* d_packed, const uint64_t* d_word_off, const uint64_t* d_kmer_base,
                                      uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, uint64_t ord_base, void* stream) {
    if (!ctx || !c || (n_reads && !d_packed)) { pg_set_error("null argument"); return PG_EINVAL; }
    if (ctx->engine != 2) { pg_set_error("pg_count_reads_sharded needs the partition engine"); return PG_ESTATE; }
    if (ctx->finalized) { pg_set_error("pg_count_reads_sharded after pg_finalize"); return PG_ESTATE; }
    if (ctx->device != c->device) { pg_set_error("context and communicator live on different devices"); return PG_EINVAL; }
    if (ctx->n_owners != 1 && ctx->n_owners != c->n) { pg_set_error("the context shares its partitions between " + std::to_string(ctx->n_owners) + " owner(s) (pg_expect), the communicator has " + std::to_string(c->n) + " rank(s)"); return PG_EINVAL; }
    if (c->pipe_ctx && c->pipe_ctx != ctx) { pg_set_error("the communicator is in use by another context (pg_comm_flush it first)"); return PG_ESTATE; }
    hipStream_t st = (hipStream_t)stream;
    X_TRY(hipSetDevice(c->device));
    const int n = c->n, rw = ctx->e2.g.rw;
    if (uniform_len) n_kmers = n_reads * (uint64_t)(uniform_len - ctx->K + 1);
    FirstError err;                                             // this rank's first failure; the round is finished regardless
    pipe_init(c, err);
    c->pipe_ctx = ctx;
    ctx->pending_drain = &pipe_drain_hook; ctx->pending_detach = &pipe_detach_hook; ctx->pending_user = c;
    // a read of k k-mers makes about 2k / (w + 1) + 1 records; twice that, spread over n owners, plus slack
    const uint64_t est = 2 * n_kmers / (uint64_t)(ctx->e2.g.w + 1) + n_reads;
    uint64_t want_cap = n_reads ? 2 * est / (uint64_t)n + 4096 : 0;
    if (const char* e = pg::env_test("PG_ROUTE_CAP")) { const long v = atol(e); if (v > 0 && n_reads) want_cap = std::max<uint64_t>(c->pipe_cap * 4 / 5, (uint64_t)v); }   // tests: start small, exercise the repeat
    pg_comm::Slot& sl = c->slot[c->round_no & 1];
    pg_comm::Slot& prev = c->slot[(c->round_no & 1) ^ 1];
    const bool one_process = c->transport == PG_COMM_P2P && n > 1;
    if (c->pipe_cap && rw != c->pipe_rw) {                          // (a communicator that served the other k-mer width before)
        (void)pipe_drain(c, ctx, st);
        pipe_free(c);
        for (auto& q : c->slot) q.used = false;
    }
    // The owner regions have ONE size on all ranks (a receive region holds what n full send regions can bring), and it only ever
    // changes by agreement: every rank packs the largest count of its cut -- or the size it would like, when its batch asks for
    // more than the regions hold (the first round: there are none yet) -- into its flag word, and when the largest word of the
    // round exceeds the regions, all ranks drain what is in flight, grow to the same size and cut again (the batch is still
    // resident).  A batch dominated by one minimizer (low-complexity reads, adapter dimers, high-copy repeats) sends one owner far
    // more than its even share: same path.
    for (int attempt = 0;; attempt++) {
        // the slot's send region was last read by the exchange of two rounds ago
        if (sl.used) err.hip(hipStreamWaitEvent(st, sl.exchanged, 0), "hipStreamWaitEvent");
        if (one_process && sl.used) {                             // ... by the PEERS' pulls of two rounds ago: all of them are done?
            err.hip(hipEventSynchronize(sl.exchanged), "hipEventSynchronize");
            c->grp->barrier();
        }
        pipe_collect_time(c, sl);
        const bool can_cut = !err.rc && c->pipe_cap != 0;
        if (can_cut) {
            const int rc = pg::e2_route(ctx, d_packed, d_word_off, d_kmer_base, n_reads, uniform_len, ord_base, n, sl.d_send_recs, sl.d_send_parts, c->pipe_cap,
                                        c->d_counts, st);
            if (rc) err.set(rc, pg_last_error());
        }
        if (!can_cut || err.rc) (void)hipMemsetAsync(c->d_counts, 0, sizeof(uint64_t) * (size_t)n, st);   // nothing to give, but still in the round
        const uint64_t wish = (!err.rc && want_cap > c->pipe_cap) ? want_cap : 0;
        hipLaunchKernelGGL(pack_pairs_kernel, dim3(1), dim3(64), 0, st, c->d_counts, n, wish, err.rc ? PIPE_ERR : 0ULL, c->d_pairs);
        err.hip(hipGetLastError(), "pack");
        err.hip(hipEventRecord(sl.routed, st), "hipEventRecord");
        pipe_exchange_pairs(c, st, sl, err);
        // ---- what everybody said
        uint64_t verdict = 0, largest_all = 0;
        for (int q = 0; q < n; q++) {
            c->h_counts[q] = c->h_pairs[2 * q];
            c->h_rcounts[q] = c->h_pairs[2 * n + 2 * q];
            const uint64_t f = c->h_pairs[2 * n + 2 * q + 1];
            verdict |= f & PIPE_ERR;
            largest_all = std::max(largest_all, f & ~PIPE_ERR);
        }
        if (verdict || err.rc) {                                    // some rank failed: nobody exchanges records this round
            err.set(PG_ENODEV, "pg_count_reads_sharded: another rank failed in this round");
            (void)pg::e2_clear_route_overflow(ctx, st);
            (void)pipe_drain(c, ctx, st);
            return err.done();
        }
        if (largest_all <= c->pipe_cap) break;                      // every owner region held what it was given (and nobody wants more)
        if (attempt >= 3) { err.set(PG_ENOMEM, "pg_count_reads_sharded: an owner's send region overflowed three times"); (void)pipe_drain(c, ctx, st); return err.done(); }
        // all ranks grow alike and cut again: first everything in flight lands (nobody pulls from or sends into the old regions)
        c->repeats += c->pipe_cap ? 1 : 0;
        { const int rc = pipe_drain(c, ctx, st); if (rc) err.set(rc, pg_last_error()); }
        if (one_process) c->grp->barrier();
        for (auto& q : c->slot) q.used = false;
        if (c->pipe_cap) { const int crc = pg::e2_clear_route_overflow(ctx, st); if (crc) err.set(crc, pg_last_error()); }
        pipe_alloc(c, largest_all + largest_all / 4 + 4096, rw, err);   // (a function of the agreed word alone: the same on every rank)
    }
    // ---- the previous round's records have arrived while this batch was cut: append them; then send this round's
    { const int rc = pipe_ingest(c, ctx, prev, st); if (rc) err.set(rc, pg_last_error()); }
    pipe_exchange_records(c, sl, rw, st, err);
    c->rounds++;
    c->round_no++;
    if (pg::env_measure("PG_PIPE_SERIAL")) (void)pipe_drain(c, ctx, st);     // A/B: no overlap, every round complete when the call returns
    return err.done();
}

// everything the sharded pass 1 still has in flight is appended to the partition streams of `ctx` (pg_finalize does this by itself)
extern "C" int pg_comm_flush(pg_ctx* ctx, pg_comm* c, void* stream) {
    if (!ctx || !c) { pg_set_error("null argument"); return PG_EINVAL; }
    const int rc = pipe_drain(c, ctx, (hipStream_t)stream);
    c->pipe_ctx = nullptr;
    if (ctx->pending_user == c) { ctx->pending_drain = nullptr; ctx->pending_detach = nullptr; ctx->pending_user = nullptr; }
    return rc;
}

// the caller brings the all-to-all (host buffers): ranks in several processes without RCCL between them -- e.g. sharing one GPU
extern "C" pg_comm* pg_comm_create_host(int n_ranks, int rank, int device, pg_host_alltoallv_fn fn, void* user) {
    if (n_ranks < 1 || n_ranks > 256 || rank < 0 || rank >= n_ranks || !fn) { pg_set_error("pg_comm_create_host: bad arguments"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { pg_set_error("pg_comm_create_host: hipSetDevice failed"); return nullptr; }
    pg_comm* c = new pg_comm();
    c->n = n_ranks; c->rank = rank; c->device = device; c->transport = PG_COMM_HOST;
    c->host_a2a = fn; c->host_user = user;
    if (comm_alloc_small(c) != PG_OK) { pg_comm_destroy(c); return nullptr; }
    return c;
}

// ---- the regroup after pass 1 ("reference set id -> GPU") ----------------------------------------------------------
// Pass 1 leaves every distinct k-mer on the rank that owns its minimizer partition.  Everything after it works on the
// reference's k-mer sets (KmerSets[s], s = hash_kmer % thrd_num, prlHashReads.c:79-90): the layout replay is per set, and
// the scans walk the sets in order (node2edge.c:383-406, output_pregraph.c:60-75).  So the distinct k-mers -- and only
// they: 32 / 48 bytes per k-mer, once -- move a second time, to owner(set s) = s mod n_ranks; from then on a set lives
// whole on one GPU and no rank holds more than its share of sets.
__global__ __launch_bounds__(256) void rg_count(const uint64_t* rec, uint64_t n, int rw, int n_ranks, unsigned long long* counts) {
    __shared__ unsigned int local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&local[(unsigned)(rec[i * rw + rw - 1] >> PG_ORD_BITS) % (unsigned)n_ranks], 1u);
    __syncthreads();
    if ((int)threadIdx.x < n_ranks && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}
// records grouped by destination: block-aggregated cursors (one returned atomic per block and destination)
__global__ __launch_bounds__(256) void rg_scatter(const uint64_t* rec, uint64_t n, int rw, int n_ranks, unsigned long long* cursor, uint64_t* out) {
    __shared__ unsigned int local[256];
    __shared__ unsigned long long base[256];
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
        local[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t i = i0 + threadIdx.x;
        unsigned dest = 0, rank_in = 0;
        if (i < n) { dest = (unsigned)(rec[i * rw + rw - 1] >> PG_ORD_BITS) % (unsigned)n_ranks; rank_in = atomicAdd(&local[dest], 1u); }
        __syncthreads();
        if ((int)threadIdx.x < n_ranks && local[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)local[threadIdx.x]);
        __syncthreads();
        if (i < n) {
            uint64_t* o = out + (base[dest] + rank_in) * rw;
            for (int w = 0; w < rw; w++) o[w] = rec[i * rw + w];
        }
        __syncthreads();
    }
}

// With a workspace (the record pool pass 1 is done with, pg_export_take_ws): the send buffer is the workspace's front, the
// regrouped records its tail when both fit (*out_in_workspace = 1: not to be freed on their own), and d_records stays the
// caller's -- no allocation and no release of that size, both of which cost a fresh process seconds.  Without: as before (the
// input array is released as soon as it has been copied, the result is an allocation of its own).
extern "C" int pg_exchange_regroup_by_set_ws(pg_comm* c, uint64_t* d_records, uint64_t n_local, int rec_words, void* d_workspace, uint64_t workspace_bytes,
                                             uint64_t** d_out, uint64_t* n_out, int* out_in_workspace, void* stream) {
    if (!c || !d_out || !n_out || (n_local && !d_records) || rec_words < 3 || rec_words > 6) { pg_set_error("bad argument"); return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    *d_out = nullptr; *n_out = 0;
    if (out_in_workspace) *out_in_workspace = 0;
    FirstError err;
    err.hip(hipSetDevice(c->device), "hipSetDevice");
    const int n = c->n;
    uint64_t* d_send = nullptr;
    unsigned long long* d_cur = nullptr;
    const uint64_t send_bytes = (n_local * (uint64_t)rec_words * 8 + 255) & ~255ULL;
    const bool send_in_ws = d_workspace && out_in_workspace && send_bytes <= workspace_bytes;
    std::vector<uint64_t> scnt(n, 0), soff(n, 0), rcnt(n, 0), roff(n, 0);
    if (!err.rc) err.hip(pg::arena_malloc((void**)&d_cur, sizeof(unsigned long long) * (size_t)n), "hipMalloc");
    if (send_in_ws) d_send = (uint64_t*)d_workspace;
    else if (!err.rc && n_local) err.hip(pg::arena_malloc((void**)&d_send, n_local * (uint64_t)rec_words * 8), "hipMalloc (regroup send buffer)");
    if (!err.rc) {
        err.hip(hipMemsetAsync(c->d_counts, 0, sizeof(uint64_t) * (size_t)n, st), "memset");
        if (n_local) hipLaunchKernelGGL(rg_count, dim3((unsigned)std::min<uint64_t>((n_local + 255) / 256, 4096)), dim3(256), 0, st, d_records, n_local, rec_words, n,
                                        (unsigned long long*)c->d_counts);
        err.hip(hipGetLastError(), "rg_count");
        err.hip(hipMemcpyAsync(scnt.data(), c->d_counts, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, st), "copy");
        err.hip(hipStreamSynchronize(st), "sync");
    }
    if (!err.rc) {
        for (int q = 1; q < n; q++) soff[q] = soff[q - 1] + scnt[q - 1];
        err.hip(hipMemcpyAsync(d_cur, soff.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, st), "copy");
        if (n_local) hipLaunchKernelGGL(rg_scatter, dim3((unsigned)std::min<uint64_t>((n_local + 255) / 256, 8192)), dim3(256), 0, st, d_records, n_local, rec_words, n, d_cur, d_send);
        err.hip(hipGetLastError(), "rg_scatter");
        err.hip(hipStreamSynchronize(st), "sync");
    }
    if (d_records && !(d_workspace && out_in_workspace)) (void)pg::arena_free(d_records);   // the caller's array has been regrouped into d_send: halve the peak
    if (err.rc) (void)hipMemsetAsync(c->d_counts, 0, sizeof(uint64_t) * (size_t)n, st);
    {   // counts (zeros from a rank that failed)
        const int rc = alltoall_words(c, c->d_counts, c->d_rcounts, 1, st);
        if (rc) err.set(rc, pg_last_error());
        err.hip(hipMemcpyAsync(rcnt.data(), c->d_rcounts, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, st), "copy");
        err.hip(hipStreamSynchronize(st), "sync");
    }
    uint64_t total = 0;
    for (int q = 0; q < n; q++) { roff[q] = total; total += rcnt[q]; }
    uint64_t* out = nullptr;
    bool out_in_ws = false;
    if (send_in_ws && total) {
        const uint64_t out_bytes = total * (uint64_t)rec_words * 8;
        if (out_bytes + 256 <= workspace_bytes) {
            const uint64_t off = (workspace_bytes - out_bytes) & ~255ULL;
            if (off >= send_bytes) { out = (uint64_t*)((char*)d_workspace + off); out_in_ws = true; }
        }
    }
    if (!err.rc && total && !out) err.hip(pg::arena_malloc((void**)&out, total * (uint64_t)rec_words * 8), "hipMalloc (regrouped records)");
    uint64_t verdict = 0;
    {
        const int arc = agree_max(c, err.rc ? 1 : 0, &verdict, st);
        if (arc) err.set(arc, pg_last_error());
    }
    if (!verdict && !err.rc) {
        const uint64_t B = (uint64_t)rec_words * 8;
        for (int q = 0; q < n; q++) { scnt[q] *= B; soff[q] *= B; rcnt[q] *= B; roff[q] *= B; }
        const int rc = alltoallv_bytes(c, d_send, soff.data(), scnt.data(), out, roff.data(), rcnt.data(), st);
        if (rc) err.set(rc, pg_last_error());
        err.hip(hipStreamSynchronize(st), "sync");
    } else err.set(PG_ENODEV, "pg_exchange_regroup_by_set: another rank failed");
    if (d_send && !send_in_ws) (void)pg::arena_free(d_send);
    if (d_cur) (void)pg::arena_free(d_cur);
    if (err.rc) { if (out && !out_in_ws) (void)pg::arena_free(out); return err.done(); }
    *d_out = out; *n_out = total;
    if (out_in_workspace) *out_in_workspace = out_in_ws ? 1 : 0;
    c->regrouped_in = total; c->regrouped_from = n_local;
    return PG_OK;
}
extern "C" int pg_exchange_regroup_by_set(pg_comm* c, uint64_t* d_records, uint64_t n_local, int rec_words, uint64_t** d_out, uint64_t* n_out,
                                          void* stream) {
    return pg_exchange_regroup_by_set_ws(c, d_records, n_local, rec_words, nullptr, 0, d_out, n_out, nullptr, stream);
}

// All ranks' exported records on rank `root` (rank order), for the stages that work on the whole graph.
extern "C" int pg_exchange_gather_records(pg_comm* c, const uint64_t* d_records, uint64_t n_local, int rec_words, int root, uint64_t* d_out,
                                          uint64_t capacity, uint64_t* n_out, void* stream) {
    if (!c || root < 0 || root >= c->n || (n_local && !d_records)) { pg_set_error("bad argument"); return PG_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    X_TRY(hipSetDevice(c->device));
    std::vector<uint64_t> mine(c->n, n_local);
    X_TRY(hipMemcpyAsync(c->d_counts, mine.data(), sizeof(uint64_t) * (size_t)c->n, hipMemcpyHostToDevice, st));
    int rc = alltoall_words(c, c->d_counts, c->d_rcounts, 1, st);
    if (rc) return rc;
    X_TRY(hipMemcpyAsync(c->h_rcounts.data(), c->d_rcounts, sizeof(uint64_t) * (size_t)c->n, hipMemcpyDeviceToHost, st));
    X_TRY(hipStreamSynchronize(st));
    std::vector<uint64_t> off(c->n + 1, 0);
    for (int q = 0; q < c->n; q++) off[q + 1] = off[q] + c->h_rcounts[q];
    if (n_out) *n_out = off[c->n];
    const bool is_root = c->rank == root;
    int err = PG_OK;
    if (is_root && (off[c->n] > capacity || (off[c->n] && !d_out))) { pg_set_error("pg_exchange_gather_records: capacity too small"); err = PG_EINVAL; }
    if (c->transport == PG_COMM_RCCL || c->n == 1) {
        if (is_root && !err && n_local) X_TRY(hipMemcpyAsync(d_out + off[root] * rec_words, d_records, n_local * (uint64_t)rec_words * 8, hipMemcpyDeviceToDevice, st));
        if (c->n > 1) {
            // a root without room still has to take what is sent to it: it receives into nothing only if nothing is sent, so
            // on error it drains into a scratch allocation
            uint64_t* sink = d_out;
            if (is_root && err) { X_TRY(pg::arena_malloc((void**)&sink, std::max<uint64_t>(off[c->n], 1) * (uint64_t)rec_words * 8)); }
            N_TRY(g_rccl.GroupStart());
            if (!is_root && n_local) N_TRY(g_rccl.Send(d_records, n_local * (uint64_t)rec_words, ncclUint64, root, c->nccl, st));
            if (is_root)
                for (int q = 0; q < c->n; q++)
                    if (q != root && c->h_rcounts[q]) N_TRY(g_rccl.Recv(sink + off[q] * rec_words, c->h_rcounts[q] * (uint64_t)rec_words, ncclUint64, q, c->nccl, st));
            N_TRY(g_rccl.GroupEnd());
            X_TRY(hipStreamSynchronize(st));
            if (sink != d_out) (void)pg::arena_free(sink);
        }
        return err;
    }
    LocalGroup* g = c->grp;
    X_TRY(hipStreamSynchronize(st));
    g->a[c->rank] = d_records;
    g->barrier();
    if (is_root && !err)
        for (int q = 0; q < c->n; q++)
            if (c->h_rcounts[q]) X_TRY(hipMemcpyAsync(d_out + off[q] * rec_words, g->a[q], c->h_rcounts[q] * (uint64_t)rec_words * 8, hipMemcpyDefault, st));
    X_TRY(hipStreamSynchronize(st));
    g->barrier();
    return err;
}

extern "C" int pg_comm_stats(const pg_comm* c, uint64_t out[4]) {
    if (!c || !out) { pg_set_error("null argument"); return PG_EINVAL; }
    out[0] = c->rounds; out[1] = c->sent_records; out[2] = c->recv_records; out[3] = c->pipe_cap ? c->pipe_cap : c->cap;
    return PG_OK;
}
// out: 0 device microseconds of the record exchanges (events on the exchange stream), 1 bytes sent to other ranks, 2 host waits
// (one a round), 3 repeated cuts, 4 rounds, 5 records per owner region
extern "C" int pg_comm_pipeline_stats(const pg_comm* c, uint64_t out[8]) {
    if (!c || !out) { pg_set_error("null argument"); return PG_EINVAL; }
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = (uint64_t)(c->exchange_ms * 1000.0); out[1] = c->bytes_sent; out[2] = c->host_syncs; out[3] = c->repeats; out[4] = c->rounds; out[5] = c->pipe_cap;
    return PG_OK;
}
extern "C" int pg_comm_regroup_stats(const pg_comm* c, uint64_t out[2]) {
    if (!c || !out) { pg_set_error("null argument"); return PG_EINVAL; }
    out[0] = c->regrouped_from; out[1] = c->regrouped_in;
    return PG_OK;
}
