// call_pregraph.cpp -- the drop-in boundary: `pregraph -s config -o prefix [-K k -p sets -a GB -d cut -R]`.
//
// Mirrors standardPregraph/pregraph.c:62-220 (call_pregraph, initenv): same getopt string, same K clamp, same
// phase order and stderr phase lines, same output files.  Pass 1 runs on the GPU through the pg_* device
// operators (there is no CPU fallback); the k-mer-set layout replay runs on the host threads, the tip walks, the
// edge construction and pass 2 on the GPU (host_graph.cpp drives graph_kernels.hip), the writers on the host.
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "host_graph.hpp"
#include "arena.hpp"
#include "cmd_plan.hpp"
#include "env.hpp"
#include "host_reads.hpp"
#include "../../include/soapdenovo2_amd.h"

extern "C" size_t pg_packed_words(uint32_t len) { return (len + 31) / 32; }

extern "C" void pg_pack_read(const uint8_t* codes, uint32_t len, uint64_t* out) {
    const size_t nw = pg_packed_words(len);
    for (size_t w = 0; w < nw; w++) {
        uint64_t v = 0;
        const uint32_t lo = (uint32_t)w * 32, hi = std::min(len, lo + 32);
        for (uint32_t i = lo; i < hi; i++) v |= (uint64_t)(codes[i] & 3) << (62 - 2 * (i - lo));
        out[w] = v;
    }
}

// Host-only view of the ingestion stage (a1 + a2): every accepted read in the reference's order.
extern "C" int pg_host_read_all(const char* config, int K, uint8_t* codes_out, int32_t* lens_out, uint64_t capacity_reads,
                                uint64_t stride, uint64_t* n_records, uint64_t* n_accepted, int* max_read_len_out) {
    if (!config) { pg_set_error("null config path"); return PG_EINVAL; }
    struct Collect : pg::ReadSink {
        int K; uint8_t* codes; int32_t* lens; uint64_t cap, stride, n = 0; bool overflow = false;
        void on_read(const uint8_t* c, int len) override {
            if (len < K + 1) return;
            if (codes) {
                if (n >= cap || (uint64_t)len > stride) { overflow = true; n++; return; }
                memcpy(codes + n * stride, c, (size_t)len);
                if (lens) lens[n] = len;
            }
            n++;
        }
    } sink;
    sink.K = K; sink.codes = codes_out; sink.lens = lens_out; sink.cap = capacity_reads; sink.stride = stride;
    pg::LibConfig cfg = pg::parse_lib_config(config);
    const int mrl = cfg.max_rd_len ? cfg.max_rd_len : 100;
    long long rec = 0;
    for (pg::InputFile f : pg::input_order(cfg, mrl)) { f.keep_len = K + 1; rec += pg::stream_reads(f, sink); }
    if (n_records) *n_records = (uint64_t)rec;
    if (n_accepted) *n_accepted = sink.n;
    if (max_read_len_out) *max_read_len_out = mrl;
    if (sink.overflow) { pg_set_error("pg_host_read_all: output buffers too small"); return PG_ENOMEM; }
    return PG_OK;
}

// the BAM reader's pairing state survives files, passes and calls, as the reference's static does (readseq1by1.c:44): a
// caller that wants the reads of one pass on its own sets it first (-3 = a fresh process) and may read it afterwards
extern "C" int pg_host_bam_pair_state(int set, int value) { return pg::bam_pair_state(set != 0, value); }

namespace {

struct Options {
    std::string config, prefix;
    int K = 23, sets = 8, delow = 0, a_gb = 0;       // defaults: inc/global.h:59,79
    bool reps = false;
};

void usage(bool mer127) {      // display_pregraph_usage, pregraph.c:222-236
    fprintf(stderr, "\npregraph -s configFile -o outputGraph [-R] [-K kmer -p n_cpu -a initMemoryAssumption -d KmerFreqCutoff]\n");
    fprintf(stderr, "  -s <string>      configFile: the config file of solexa reads\n");
    fprintf(stderr, "  -o <string>      outputGraph: prefix of output graph file name\n");
    fprintf(stderr, "  -K <int>         kmer(min 13, max %d): kmer size, [23]\n", mer127 ? 127 : 63);
    fprintf(stderr, "  -p <int>         n_cpu: number of cpu for use, [8]\n");
    fprintf(stderr, "  -a <int>         initMemoryAssumption: memory assumption initialized to avoid further reallocation, unit GB, [0]\n");
    fprintf(stderr, "  -R (optional)    output extra information for resolving repeats in contig step, [NO]\n");
    fprintf(stderr, "  -d <int>         KmerFreqCutoff: kmers with frequency no larger than KmerFreqCutoff will be deleted, [0]\n");
}

Options parse_args(int argc, char** argv, bool mer127) {      // initenv, pregraph.c:142-220
    Options o;
    bool in = false, out = false;
    optind = 1;
    fprintf(stderr, "Parameters: pregraph ");
    int c;
    while ((c = getopt(argc, argv, "a:s:o:K:p:d:R")) != EOF) {
        switch (c) {
            case 's': fprintf(stderr, "-s %s ", optarg); in = true; o.config = optarg; break;
            case 'o': fprintf(stderr, "-o %s ", optarg); out = true; o.prefix = optarg; break;
            case 'K': fprintf(stderr, "-K %s ", optarg); o.K = atoi(optarg); break;
            case 'p': fprintf(stderr, "-p %s ", optarg); o.sets = atoi(optarg); break;
            case 'R': o.reps = true; fprintf(stderr, "-R "); break;
            case 'd': fprintf(stderr, "-d %s ", optarg); o.delow = atoi(optarg) >= 0 ? atoi(optarg) : 0; break;
            case 'a': fprintf(stderr, "-a %s ", optarg); o.a_gb = atoi(optarg); break;
            default:
                if (!in || !out) { usage(mer127); exit(-1); }
        }
    }
    fprintf(stderr, "\n\n");
    if (!in || !out) { usage(mer127); exit(-1); }
    return o;
}

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, pg_last_error());
    exit(-1);
}
#define HIP_OK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); exit(-1); } \
    } while (0)

// Pass 1 driver: reads -> 2-bit packed pinned batches -> hipMemcpyAsync -> pg_count_reads.  Two batches in
// flight so parsing overlaps the copy + kernel of the previous batch.
// The accepted reads, 2 bits a base, as pass 1 saw them: pass 2 threads the same reads again (prlRead2edge) and a second
// go at pass 1 (global-set engine) replays them.  Blocks of 256 MiB on huge pages, whole reads per block, nothing is ever
// moved or value-initialised.
struct KeptReads {
    struct Block {
        uint64_t* words = nullptr; size_t cap = 0, used = 0, bytes = 0;
        std::vector<int32_t> lens;
    };
    static constexpr size_t BLOCK_WORDS = (size_t)1 << 25;
    std::vector<Block> blocks;
    size_t total_bytes = 0;
    KeptReads() {}
    KeptReads(const KeptReads&) = delete;
    KeptReads& operator=(const KeptReads&) = delete;
    ~KeptReads() { clear(); }
    void clear() {
        for (Block& b : blocks) if (b.words) munmap(b.words, b.bytes);
        blocks.clear();
        total_bytes = 0;
    }
    void swap(KeptReads& o) { blocks.swap(o.blocks); std::swap(total_bytes, o.total_bytes); }
    size_t reads() const { size_t n = 0; for (const Block& b : blocks) n += b.lens.size(); return n; }
    bool new_block(size_t min_words) {
        Block b;
        b.cap = std::max(BLOCK_WORDS, min_words);
        b.bytes = b.cap * sizeof(uint64_t);
        void* q = mmap(nullptr, b.bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        madvise(q, b.bytes, MADV_HUGEPAGE);
        b.words = (uint64_t*)q;
        b.lens.reserve(b.cap / 4);
        blocks.push_back(std::move(b));
        return true;
    }
    // n whole reads, nw words in all, back to back; false = out of memory
    bool append(const uint64_t* w, size_t nw, const int32_t* lens, size_t n) {
        size_t r = 0, at = 0;
        while (r < n) {
            if (blocks.empty() || blocks.back().used == blocks.back().cap) { if (!new_block(0)) return false; }
            Block& b = blocks.back();
            size_t take = 0, take_words = 0;
            if (b.used + (nw - at) <= b.cap) { take = n - r; take_words = nw - at; }            // the rest fits
            else {
                while (r + take < n) {
                    const size_t rw = ((size_t)lens[r + take] + 31) / 32;
                    if (b.used + take_words + rw > b.cap) break;
                    take_words += rw; take++;
                }
                if (take == 0) {                               // not even one read: this block is done
                    const size_t rw = ((size_t)lens[r] + 31) / 32;
                    if (!new_block(rw)) return false;
                    continue;
                }
            }
            memcpy(b.words + b.used, w + at, take_words * sizeof(uint64_t));
            b.lens.insert(b.lens.end(), lens + r, lens + r + take);
            b.used += take_words;
            total_bytes += take_words * sizeof(uint64_t) + take * sizeof(int32_t);
            r += take; at += take_words;
        }
        return true;
    }
};

// Reads kept for pass 2 in DEVICE memory: the batches pass 1 has uploaded anyway are copied on (device to device) into chunks of
// 1 GiB, one segment a batch; pass 2 threads them where they are (pg_graph_add_packed_device[_ragged]) -- no copy back.  A batch of one
// read length is its words alone; a ragged batch (round 6: any mix of lengths stays on the device, as the reference threads reads of
// any length the same way, prlRead2path.c:1056-1110) keeps the two index arrays pass 1 took behind its words.  Only a budget that
// runs out moves everything into the host store (KeptReads), which then goes on as before.
struct DevKept {
    // len > 0: n_reads reads of len bases back to back.  len = 0: n_words words of reads, 8 words of padding, word_off[n_reads],
    // kmer_base[n_reads + 1] (all 64-bit); max_len = the longest read.
    struct Seg {
        uint64_t* d; uint64_t n_reads; int len; uint64_t n_words, n_kmers; int max_len;
        const uint64_t* d_off() const { return d + n_words + 8; }
        const uint64_t* d_base() const { return d + n_words + 8 + n_reads; }
    };
    static constexpr size_t CHUNK_WORDS = (size_t)1 << 27;
    std::vector<void*> chunks;
    std::vector<Seg> segs;
    size_t used_in_last = 0, total_bytes = 0;
    int device = 0;
    DevKept() {}
    DevKept(const DevKept&) = delete;
    DevKept& operator=(const DevKept&) = delete;
    ~DevKept() { clear(); }
    void clear() {
        for (void* c : chunks) (void)pg::arena_free(c);
        chunks.clear(); segs.clear(); used_in_last = 0; total_bytes = 0;
    }
    void swap(DevKept& o) { chunks.swap(o.chunks); segs.swap(o.segs); std::swap(used_in_last, o.used_in_last); std::swap(total_bytes, o.total_bytes); std::swap(device, o.device); }
    // a segment's reads into the host store's form: words back to back + a length a read; false = the copy failed
    static bool fetch(const Seg& sg, int K, std::vector<uint64_t>& words, std::vector<int32_t>& lens) {
        const size_t nw = sg.len ? sg.n_reads * (((size_t)sg.len + 31) / 32) : sg.n_words;
        words.resize(nw);
        if (hipMemcpy(words.data(), sg.d, nw * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (sg.len) { lens.assign(sg.n_reads, (int32_t)sg.len); return true; }
        std::vector<uint64_t> base(sg.n_reads + 1);
        if (hipMemcpy(base.data(), sg.d_base(), base.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
        lens.resize(sg.n_reads);
        for (uint64_t r = 0; r < sg.n_reads; r++) lens[r] = (int32_t)(base[r + 1] - base[r]) + K - 1;
        return true;
    }
    // room for n_words words (<= CHUNK_WORDS); nullptr = no device memory
    uint64_t* take(size_t n_words) {
        if (n_words > CHUNK_WORDS) return nullptr;
        if (chunks.empty() || used_in_last + n_words > CHUNK_WORDS) {
            void* c = nullptr;
            if (pg::arena_malloc(&c, CHUNK_WORDS * sizeof(uint64_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            chunks.push_back(c);
            used_in_last = 0;
        }
        uint64_t* p = (uint64_t*)chunks.back() + used_in_last;
        used_in_last += n_words;
        total_bytes += n_words * sizeof(uint64_t);
        return p;
    }
};


// Pass 1 driver, common part: accepted reads -> 2-bit packed batches in pinned host buffers.  What happens to a full
// batch is the backend's business (submit): one GPU takes batch after batch (Pass1), several GPUs take a round of one
// batch each and exchange (ShardedPass1).
class BatchFiller : public pg::ReadSink {
public:
    struct Buf {
        uint64_t *h_words = nullptr, *h_off = nullptr, *h_base = nullptr;
        size_t n_reads = 0, n_words = 0;
        uint64_t n_kmers = 0, ord_base = 0;
        int first_len = 0, max_len = 0;
        bool uniform = true;
    };
    BatchFiller(int K, size_t max_words, size_t max_reads, int n_bufs) : K_(K), max_words_(max_words), max_reads_(max_reads), buf_(n_bufs) {
        const bool trace = pg::env_user("PG_STARTUP_TRACE") && atoi(pg::env_user("PG_STARTUP_TRACE"));
        const auto t0 = std::chrono::steady_clock::now();
        for (Buf& b : buf_) {
            HIP_OK(hipHostMalloc((void**)&b.h_words, (max_words_ + 8) * sizeof(uint64_t), hipHostMallocDefault));
            if (trace && &b == &buf_[0]) fprintf(stderr, "[cli]   first page-locked buffer (with this thread's share of the runtime start-up): %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            HIP_OK(hipHostMalloc((void**)&b.h_off, max_reads_ * sizeof(uint64_t), hipHostMallocDefault));
            HIP_OK(hipHostMalloc((void**)&b.h_base, (max_reads_ + 1) * sizeof(uint64_t), hipHostMallocDefault));
        }
        if (trace) fprintf(stderr, "[cli]   all page-locked batch buffers: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        reset(buf_[0]);
    }
    ~BatchFiller() override {
        for (Buf& b : buf_) { hipHostFree(b.h_words); hipHostFree(b.h_off); hipHostFree(b.h_base); }
    }
    void on_read(const uint8_t* codes, int len) override {
        if (len < K_ + 1) return;                               // prlHashReads.c:642
        const size_t nw = pg_packed_words((uint32_t)len);
        Buf* b = &buf_[cur_];
        if (b->n_reads == max_reads_ || b->n_words + nw > max_words_) { submit(); b = &buf_[cur_]; }
        pg_pack_read(codes, (uint32_t)len, b->h_words + b->n_words);
        if (keep_ && !dev_keep_) { const int32_t l32 = len; keep_append(b->h_words + b->n_words, nw, &l32, 1); }   // pass 2 threads the same reads again (dev_keep_: the batch stays on the device, submit)
        b->h_off[b->n_reads] = b->n_words;
        b->h_base[b->n_reads] = b->n_kmers;
        note_length(b, len);
        b->n_words += nw;
        b->n_kmers += (uint64_t)(len - K_ + 1);
        b->n_reads++;
        accepted_++;
    }
    // runs from the multi-threaded reader: the words are copied as they are
    void on_packed(const uint64_t* words, const int32_t* lens, size_t n, int min_len, int max_len) override {
        size_t at = 0;
        if (min_len >= K_ + 1 && min_len == max_len) {          // every read accepted and of one length: block copies
            const int len = min_len;
            const size_t wpr = pg_packed_words((uint32_t)len);
            const uint64_t kpr = (uint64_t)(len - K_ + 1);
            size_t r = 0;
            while (r < n) {
                Buf* b = &buf_[cur_];
                size_t take = std::min(n - r, max_reads_ - b->n_reads);
                take = std::min(take, (max_words_ - b->n_words) / wpr);
                if (take == 0) { submit(); continue; }
                memcpy(b->h_words + b->n_words, words + at, take * wpr * sizeof(uint64_t));
                note_length(b, len);
                if (!b->uniform)                                    // (a uniform batch goes to the device without index arrays)
                    for (size_t i = 0; i < take; i++) {
                        b->h_off[b->n_reads + i] = b->n_words + i * wpr;
                        b->h_base[b->n_reads + i] = b->n_kmers + i * kpr;
                    }
                if (keep_ && !dev_keep_) keep_append(words + at, take * wpr, lens + r, take);      // (dev_keep_: kept when the batch is on the device, submit)
                b->n_words += take * wpr; b->n_kmers += take * kpr; b->n_reads += take;
                accepted_ += (long long)take;
                at += take * wpr; r += take;
            }
            return;
        }
        for (size_t r = 0; r < n; r++) {
            const int len = lens[r];
            const size_t nw = pg_packed_words((uint32_t)len);
            if (len >= K_ + 1) {                                    // prlHashReads.c:642
                Buf* b = &buf_[cur_];
                if (b->n_reads == max_reads_ || b->n_words + nw > max_words_) { submit(); b = &buf_[cur_]; }
                memcpy(b->h_words + b->n_words, words + at, nw * sizeof(uint64_t));
                if (keep_ && !dev_keep_) keep_append(words + at, nw, &lens[r], 1);
                b->h_off[b->n_reads] = b->n_words;
                b->h_base[b->n_reads] = b->n_kmers;
                note_length(b, len);
                b->n_words += nw;
                b->n_kmers += (uint64_t)(len - K_ + 1);
                b->n_reads++;
                accepted_++;
            }
            at += nw;
        }
    }
    // first read of a batch, or a read of another length: from then on the batch needs its index arrays, and the block
    // copies above did not fill them while every read had the same length
    void note_length(Buf* b, int len) {
        if (len > b->max_len) b->max_len = len;
        if (b->n_reads == 0) { b->first_len = len; return; }
        if (!b->uniform || len == b->first_len) return;
        const size_t wpr = pg_packed_words((uint32_t)b->first_len);
        const uint64_t kpr = (uint64_t)(b->first_len - K_ + 1);
        for (size_t i = 0; i < b->n_reads; i++) { b->h_off[i] = i * wpr; b->h_base[i] = i * kpr; }
        b->uniform = false;
    }
    virtual bool finish_ok() = 0;
    uint64_t total_kmers() const { return ord_; }
    // the packed reads kept for pass 2, or nothing when they outgrew the budget (then the files are parsed again)
    void keep_reads(size_t budget_bytes) { keep_ = budget_bytes > 0; keep_budget_ = budget_bytes; }
    // ... on the device while they are of one length (single-GPU pass 1 only; see DevKept)
    void keep_reads_on_device(size_t budget_bytes) { dev_keep_ = keep_ && budget_bytes > 0; dev_budget_ = budget_bytes; }
    bool take_dev_kept(DevKept& out) {
        if (!dev_keep_) return false;
        out.swap(dev_kept_);
        return !out.segs.empty();
    }
    // everything kept on the device so far, and the reads waiting in the batch being filled, into the host store; from here on
    // the host store is the only one
    virtual void dev_keep_to_host() {
        if (!dev_keep_) return;
        dev_keep_ = false;
        std::vector<uint64_t> tmp;
        std::vector<int32_t> ls;
        (void)hipDeviceSynchronize();
        for (const DevKept::Seg& sg : dev_kept_.segs) {
            if (!DevKept::fetch(sg, K_, tmp, ls)) { keep_ = false; kept_.clear(); break; }
            if (keep_) keep_append(tmp.data(), tmp.size(), ls.data(), sg.n_reads);
        }
        dev_kept_.clear();
        const Buf& b = buf_[cur_];
        if (keep_ && b.n_reads) {                                 // (no host copy was made of these while the device kept the batches)
            batch_lens(b, ls);
            keep_append(b.h_words, b.n_words, ls.data(), b.n_reads);
        }
    }
    // the lengths of a batch's reads (a batch that is still being filled has no h_base[n_reads] yet)
    void batch_lens(const Buf& b, std::vector<int32_t>& ls) const {
        if (b.uniform) { ls.assign(b.n_reads, (int32_t)b.first_len); return; }
        ls.resize(b.n_reads);
        for (size_t r = 0; r < b.n_reads; r++) ls[r] = (int32_t)((r + 1 < b.n_reads ? b.h_base[r + 1] : b.n_kmers) - b.h_base[r]) + K_ - 1;
    }
    bool take_kept(KeptReads& out) {
        if (!keep_) return false;
        out.swap(kept_);
        return true;
    }

protected:
    // the batch in buf_[cur_] is complete: take it, and leave cur_ on a buffer that may be filled
    virtual void submit() = 0;
    void seal(Buf& b) {                                             // the last touches before a batch leaves the host
        for (int i = 0; i < 8; i++) b.h_words[b.n_words + i] = 0;   // readable padding for the window loads
        b.h_base[b.n_reads] = b.n_kmers;
        b.ord_base = ord_;
        ord_ += b.n_kmers;
    }
    void reset(Buf& b) { b.n_reads = 0; b.n_words = 0; b.n_kmers = 0; b.first_len = 0; b.max_len = 0; b.uniform = true; }
    int K_;
    size_t max_words_, max_reads_;
    std::vector<Buf> buf_;
    int cur_ = 0;
    uint64_t ord_ = 0;
    long long accepted_ = 0;

private:
    void keep_append(const uint64_t* w, size_t nw, const int32_t* lens, size_t n) {
        if (kept_.total_bytes + nw * sizeof(uint64_t) + n * sizeof(int32_t) > keep_budget_ || !kept_.append(w, nw, lens, n)) {
            keep_ = false;                                       // over the budget: pass 2 parses the files again
            kept_.clear();
        }
    }
    bool keep_ = false;
    size_t keep_budget_ = 0;
    KeptReads kept_;

protected:
    void keep_host(const uint64_t* w, size_t nw, const int32_t* lens, size_t n) { if (keep_) keep_append(w, nw, lens, n); }
    bool dev_keep_ = false;
    size_t dev_budget_ = 0;
    DevKept dev_kept_;
};

// One GPU: hipMemcpyAsync + pg_count_reads, two batches in flight so parsing overlaps the copy + kernel of the previous one.
class Pass1 : public BatchFiller {
public:
    Pass1(pg_ctx* ctx, int K, size_t max_words, size_t max_reads) : BatchFiller(K, max_words, max_reads, 2), ctx_(ctx) {
        const bool trace = pg::env_user("PG_STARTUP_TRACE") && atoi(pg::env_user("PG_STARTUP_TRACE"));
        auto t0 = std::chrono::steady_clock::now();
        auto step = [&](const char* what) {
            if (!trace) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[cli]   %s: %.3f s\n", what, std::chrono::duration<double>(t - t0).count());
            t0 = t;
        };
        HIP_OK(hipStreamCreate(&stream_));
        step("pass 1's stream");
        for (int i = 0; i < 2; i++) {
            Dev& d = dev_[i];
            HIP_OK(pg::arena_malloc((void**)&d.d_words, (max_words_ + 8) * sizeof(uint64_t)));
            HIP_OK(pg::arena_malloc((void**)&d.d_off, max_reads_ * sizeof(uint64_t)));
            HIP_OK(pg::arena_malloc((void**)&d.d_base, (max_reads_ + 1) * sizeof(uint64_t)));
            HIP_OK(hipEventCreateWithFlags(&d.done, hipEventDisableTiming));
            d.busy = false;
            step("a batch's device buffers (96 MiB) + event");
        }
    }
    ~Pass1() override {
        for (int i = 0; i < 2; i++) {
            Dev& d = dev_[i];
            pg::arena_free(d.d_words); pg::arena_free(d.d_off); pg::arena_free(d.d_base);
            hipEventDestroy(d.done);
        }
        hipStreamDestroy(stream_);
    }
    bool finish_ok() override { submit(); HIP_OK(hipStreamSynchronize(stream_)); return !failed_; }
    void set_ctx(pg_ctx* ctx) { ctx_ = ctx; }                       // (the buffers are made while the context's pools are still being allocated)

private:
    struct Dev { uint64_t *d_words, *d_off, *d_base; hipEvent_t done; bool busy; };
    void submit() override {
        Buf& b = buf_[cur_];
        Dev& d = dev_[cur_];
        if (b.n_reads) {
            seal(b);
            HIP_OK(hipMemcpyAsync(d.d_words, b.h_words, (b.n_words + 8) * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
            if (!b.uniform) {
                HIP_OK(hipMemcpyAsync(d.d_off, b.h_off, b.n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
                HIP_OK(hipMemcpyAsync(d.d_base, b.h_base, (b.n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
            }
            if (!b.uniform && !failed_ && pg_set_read_len_bound(ctx_, (uint32_t)b.max_len) != PG_OK) failed_ = true;   // (the tiles of a ragged batch are sized for its longest read)
            if (!failed_ && pg_count_reads(ctx_, d.d_words, b.uniform ? nullptr : d.d_off, b.uniform ? nullptr : d.d_base, b.n_reads,
                                           b.uniform ? (uint32_t)b.first_len : 0u, b.n_kmers, b.ord_base, stream_) != PG_OK)
                failed_ = true;                                     // the caller decides (finish_ok)
            if (dev_keep_) {                                        // the batch stays on the device for pass 2 (a ragged one with its index arrays)
                const size_t extra = b.uniform ? 0 : 2 * b.n_reads + 1, need = b.n_words + 8 + extra;
                uint64_t* dst = dev_kept_.total_bytes + need * sizeof(uint64_t) <= dev_budget_ ? dev_kept_.take(need) : nullptr;
                bool ok = dst && hipMemcpyAsync(dst, d.d_words, (b.n_words + 8) * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream_) == hipSuccess;
                if (ok && !b.uniform)
                    ok = hipMemcpyAsync(dst + b.n_words + 8, d.d_off, b.n_reads * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream_) == hipSuccess &&
                         hipMemcpyAsync(dst + b.n_words + 8 + b.n_reads, d.d_base, (b.n_reads + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream_) == hipSuccess;
                if (ok) dev_kept_.segs.push_back(DevKept::Seg{dst, b.n_reads, b.uniform ? b.first_len : 0, b.n_words, b.n_kmers, b.max_len});
                else {                                              // no room: this batch and all before it go to the host store
                    (void)hipGetLastError();
                    Buf hold = b;                                   // (dev_keep_to_host looks at buf_[cur_]: make it see an empty batch, this one is appended below)
                    b.n_reads = 0;
                    dev_keep_to_host();
                    b = hold;
                    std::vector<int32_t> ls;
                    batch_lens(b, ls);
                    keep_host(b.h_words, b.n_words, ls.data(), b.n_reads);
                }
            }
            HIP_OK(hipEventRecord(d.done, stream_));
            d.busy = true;
        }
        cur_ ^= 1;
        if (dev_[cur_].busy) { HIP_OK(hipEventSynchronize(dev_[cur_].done)); dev_[cur_].busy = false; }
        reset(buf_[cur_]);
    }
    bool failed_ = false;
    pg_ctx* ctx_;
    Dev dev_[2];
    hipStream_t stream_;
};

// Several GPUs (SOAPDENOVO2_AMD_DEVICES=0,1,...): batches are dealt to the ranks in turn, a round = one batch per rank.
// One host thread per rank copies its batch to its GPU and runs pg_count_reads_sharded -- cut, all-to-all, append
// (exchange.hip) -- while this thread fills the next round's buffers.  Every k-mer keeps the ordinal it has in the
// reference's read order, so nothing downstream depends on which GPU counted it.
class ShardedPass1 : public BatchFiller {
public:
    ShardedPass1(const std::vector<pg_ctx*>& ctxs, const std::vector<pg_comm*>& comms, const std::vector<int>& devices, int K, size_t max_words,
                 size_t max_reads)
        : BatchFiller(K, max_words, max_reads, 2 * (int)ctxs.size()), n_((int)ctxs.size()), ctx_(ctxs), comm_(comms), device_(devices) {
        posted_.assign(2, 0);
        for (int r = 0; r < n_; r++) workers_.emplace_back([this, r] { worker(r); });
    }
    ~ShardedPass1() override { stop(); }
    bool finish_ok() override {
        if (buf_[cur_].n_reads) submit();
        if (cur_ % n_ != 0) {                                      // a partly dealt round: the other ranks take part empty-handed
            while (cur_ % n_ != 0) { seal(buf_[cur_]); cur_++; }
            post_round((cur_ - 1) / n_);
        }
        wait_all();
        stop();
        return !failed_.load();
    }

private:
    void submit() override {
        Buf& b = buf_[cur_];
        if (!b.n_reads) return;                                    // nothing to hand over: keep filling this buffer
        seal(b);
        const int set = cur_ / n_;
        if (cur_ % n_ == n_ - 1) post_round(set);
        cur_ = (cur_ + 1) % (2 * n_);
        if (cur_ % n_ == 0) wait_set_free(cur_ / n_);             // the set we are about to fill: its last round must be done
        reset(buf_[cur_]);
    }
    void post_round(int set) {
        std::lock_guard<std::mutex> lk(mu_);
        rounds_posted_++;
        posted_[set] = rounds_posted_;                              // round number (1-based) now waiting in this set
        cv_.notify_all();
    }
    void wait_set_free(int set) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return posted_[set] == 0 || rounds_done_ >= posted_[set]; });
        for (int r = 0; r < n_; r++) reset(buf_[set * n_ + r]);
    }
    void wait_all() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return rounds_done_ >= rounds_posted_; });
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
        for (auto& t : workers_) if (t.joinable()) t.join();
        workers_.clear();
    }
    void worker(int r) {
        HIP_OK(hipSetDevice(device_[r]));
        hipStream_t st;
        HIP_OK(hipStreamCreate(&st));
        uint64_t *d_words, *d_off, *d_base;
        HIP_OK(pg::arena_malloc((void**)&d_words, (max_words_ + 8) * sizeof(uint64_t)));
        HIP_OK(pg::arena_malloc((void**)&d_off, max_reads_ * sizeof(uint64_t)));
        HIP_OK(pg::arena_malloc((void**)&d_base, (max_reads_ + 1) * sizeof(uint64_t)));
        for (uint64_t round = 1;; round++) {
            int set = -1;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || rounds_posted_ >= round; });
                if (rounds_posted_ < round) break;                  // stopped with nothing left
                set = (int)((round - 1) & 1);                       // rounds alternate between the two buffer sets
            }
            const Buf& b = buf_[set * n_ + r];
            if (b.n_reads) {
                HIP_OK(hipMemcpyAsync(d_words, b.h_words, (b.n_words + 8) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                if (!b.uniform) {
                    HIP_OK(hipMemcpyAsync(d_off, b.h_off, b.n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                    HIP_OK(hipMemcpyAsync(d_base, b.h_base, (b.n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                }
            }
            if (!b.uniform && b.n_reads) (void)pg_set_read_len_bound(ctx_[r], (uint32_t)b.max_len);      // (the tiles of a ragged batch are sized for its longest read)
            if (pg_count_reads_sharded(ctx_[r], comm_[r], d_words, b.uniform ? nullptr : d_off, b.uniform ? nullptr : d_base, b.n_reads,
                                       (b.uniform && b.n_reads) ? (uint32_t)b.first_len : 0u, b.n_kmers, b.ord_base, st) != PG_OK) {
                fprintf(stderr, "rank %d: pg_count_reads_sharded: %s\n", r, pg_last_error());
                failed_.store(true);
            }
            HIP_OK(hipStreamSynchronize(st));
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (++done_in_round_ == n_) { done_in_round_ = 0; rounds_done_++; cv_.notify_all(); }
            }
        }
        pg::arena_free(d_words); pg::arena_free(d_off); pg::arena_free(d_base);
        hipStreamDestroy(st);
    }
    int n_;
    std::vector<pg_ctx*> ctx_;
    std::vector<pg_comm*> comm_;
    std::vector<int> device_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<uint64_t> posted_;            // per buffer set: the round waiting in it (0 = none yet)
    uint64_t rounds_posted_ = 0, rounds_done_ = 0;
    int done_in_round_ = 0;
    bool stop_ = false;
    std::atomic<bool> failed_{false};
};

int run(int argc, char** argv, bool mer127) {
    const time_t t_start = time(nullptr);
    const bool verbose = pg::env_user("PG_HOST_VERBOSE") != nullptr;
    auto nowf = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    double tv = nowf();
    auto lap = [&](const char* what) { if (verbose) { const double t = nowf(); fprintf(stderr, "[cli] %s: %.2fs\n", what, t - tv); tv = t; } };
    const double t_begin = tv;
    auto mark = [&](const char* what) { if (verbose) fprintf(stderr, "[cli]   at %.2fs: %s\n", nowf() - t_begin, what); };
    fprintf(stderr, "\n********************\nPregraph\n********************\n\n");
    Options o = parse_args(argc, argv, mer127);
    int K = o.K;                                                   // pregraph.c:71-97
    if (K % 2 == 0) { K++; fprintf(stderr, "K should be an odd number.\n"); }
    if (K < 13) { K = 13; fprintf(stderr, "K should not be less than 13.\n"); }
    else if (K > (mer127 ? 127 : 63)) { K = mer127 ? 127 : 63; fprintf(stderr, "K should not be greater than %d.\n", K); }
    if (o.sets < 1 || o.sets > 255) { fprintf(stderr, "-p must be within 1..255 (k-mer sets).\n"); exit(-1); }

    // ---- pass 1 (prlRead2HashTable, prlHashReads.c:304-760)
    time_t t0 = time(nullptr);
    pg::LibConfig cfg = pg::parse_lib_config(o.config.c_str());
    const int max_read_len = cfg.max_rd_len ? cfg.max_rd_len : 100;          // prlHashReads.c:325-328
    fprintf(stderr, "In %s, %d lib(s), maximum read length %d, maximum name length %d.\n\n", o.config.c_str(),
            (int)cfg.libs.size(), max_read_len, 256);
    std::vector<pg::InputFile> files = pg::input_order(cfg, max_read_len);
    for (pg::InputFile& f : files) f.keep_len = K + 1;                        // (BAM: prlHashReads.c:437 keeps reads of K + 1 bases and more)

    // SOAPDENOVO2_AMD_DEVICES=0,1,2,...: pass 1 sharded over these GPUs (one rank each; an ordinal may repeat, which puts
    // several ranks on one GPU -- how the N-rank path is tested on a 1-GPU box); everything after pass 1 runs on the first.
    int device = 0;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_DEVICE")) device = atoi(e);
    std::vector<int> devices;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_DEVICES")) {
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
            if (end == q && *q) break;
        }
        if (!devices.empty()) device = devices[0];
    }
    const int n_ranks = devices.size() > 1 ? (int)devices.size() : 1;
    // After pass 1 a reference k-mer set lives whole on one GPU (set s on rank s mod N: the layout and the scans are per set), so
    // -p is also the number of GPUs the graph stages can use: with fewer sets than ranks the other ranks only help in pass 1 and pass 2's threading.
    if (n_ranks > 1 && o.sets < n_ranks)
        fprintf(stderr, "warning: -p %d gives %d k-mer set(s) for %d GPU ranks: %d rank(s) will hold no set after pass 1 (use -p >= %d, or a multiple of it, to spread the graph stages).\n",
                o.sets, o.sets, n_ranks, n_ranks - o.sets, n_ranks);
    // every device of the command keeps its arena's physical memory until the command ends (csrc/arena.hpp)
    std::vector<std::unique_ptr<pg::ArenaPin>> arena_pins;
    {
        std::vector<int> seen;
        for (int d : n_ranks > 1 ? devices : std::vector<int>{device})
            if (std::find(seen.begin(), seen.end(), d) == seen.end()) { seen.push_back(d); arena_pins.emplace_back(new pg::ArenaPin(d)); }
    }
    // how much is coming: bases ~ half the bytes of a FASTQ file, all of a FASTA file (x4 behind gzip)
    uint64_t est_kmers = 0, est_reads = 0;
    for (const pg::InputFile& f : files)
        for (const std::string* path : {&f.path1, &f.path2}) {
            struct stat st;
            if (path->empty() || stat(path->c_str(), &st) != 0) continue;
            double bases = (double)st.st_size * ((f.type == 2 || f.type == 6) ? 0.5 : 1.0);
            if (path->size() > 3 && path->compare(path->size() - 3, 3, ".gz") == 0) bases *= 4.0;
            if (f.type == 4) bases *= 3.0;                              // BAM: 4-bit bases + qualities, deflated
            // a read of l <= max_rd_len bases has l - K + 1 k-mers: at most (max_rd_len - K + 1) / max_rd_len of its bases
            // (what is sized from this -- the record pool, the partition count, the export array -- is handed out cleared by
            //  the driver, by the gigabyte: an estimate twice too large cost the start of the command up to two seconds)
            if (max_read_len > 0) est_reads += (uint64_t)(bases / (double)max_read_len);       // (reads of max_rd_len bases: shorter reads make more of them, and fewer k-mers each)
            if (max_read_len > K) bases *= (double)(max_read_len - K + 1) / (double)max_read_len;
            est_kmers += (uint64_t)bases;
        }
    // The device export array: room for one distinct k-mer per 6 occurrences (it is enlarged and the partitions are
    // counted again when that is too little), but never more than a third of the device memory -- the record pool needs
    // the rest.  -a only presizes the reference's HOST k-mer sets (prlHashReads.c:369-390) and is used for exactly that
    // in the layout replay; it does not size anything on the device.
    int log2_slots = 24;
    uint64_t export_records = 0;                                   // distinct k-mers a rank's export array is made for (cmd_plan.hpp)
    {
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipSetDevice(device));
        HIP_OK(pg::arena_mem_info(&free_b, &total_b));
        log2_slots = pg::cmd_log2_slots(est_kmers, mer127, total_b);
        export_records = pg::cmd_export_records(est_kmers, mer127, n_ranks, total_b);
    }
    // a pass-1 batch: 64 MiB of packed reads / 2 M reads (SOAPDENOVO2_AMD_BATCH_READS: smaller batches, for tests)
    size_t batch_words = (size_t)pg::CMD_BATCH_WORDS, batch_reads = (size_t)pg::CMD_BATCH_READS;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_BATCH_READS")) { const long v = atol(e); if (v > 0) { batch_reads = (size_t)v; batch_words = std::min(batch_words, batch_reads * 160 + 64); } }
    size_t keep_budget = (size_t)sysconf(_SC_PHYS_PAGES) * (size_t)sysconf(_SC_PAGE_SIZE) / 4;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_KEEP_READS_GB")) keep_budget = (size_t)(atof(e) * 1073741824.0);
    KeptReads kept;
    bool have_kept = false;
    // ... on the device instead, while the reads are of one length and pass 2 runs there without -R (SOAPDENOVO2_AMD_KEEP_ON_HOST=1: never)
    DevKept devkept;
    size_t dev_keep_budget = 0;
    {
        const char* e2 = pg::env_user("SOAPDENOVO2_AMD_PASS2");
        const char* e1 = pg::env_user("SOAPDENOVO2_AMD_EDGES");
        const bool host_side = (e2 && !strcmp(e2, "host")) || (e1 && !strcmp(e1, "host"));
        size_t free_b = 0, total_b = 0;
        if (!o.reps && !host_side && !pg::env_test("SOAPDENOVO2_AMD_KEEP_ON_HOST") && keep_budget > 0 && pg::arena_mem_info(&free_b, &total_b) == hipSuccess)
            dev_keep_budget = (size_t)pg::cmd_dev_keep_budget(total_b);
    }
    long long n_records = 0;
    uint64_t total_kmers = 0;
    uint64_t hist[256];
    std::vector<uint64_t> set_last(o.sets, 0);
    uint64_t n_distinct = 0;
    pg_ctx* ctx = nullptr;
    uint64_t* d_rec = nullptr;                                     // the distinct k-mers of pass 1, on `device`
    std::vector<uint64_t*> sh_rec;                                 // sharded run: per rank, the k-mers of the sets it owns (replay order), on its GPU
    std::vector<uint64_t> sh_n, sh_per_set;
    // ... the record pool each rank's pass 1 is done with (regroup, sort and, as an offered block, the k-mer sets' layout work inside
    // it: no allocation of that size behind a release of that size), how much of its front is free, what is to be released in the end
    std::vector<void*> sh_ws, sh_mine;
    std::vector<uint64_t> sh_ws_front;
    std::vector<char> sh_in_ws, sh_offered;
    int engine_used = 2;
    if (n_ranks > 1) {
        // ---- pass 1 on n_ranks GPUs: cut + all-to-all + append per batch, then every rank counts its own partitions
        std::vector<pg_ctx*> ctxs(n_ranks, nullptr);
        std::vector<pg_comm*> comms(n_ranks, nullptr);
        mark("input files sized");
        if (pg_comm_create_local(n_ranks, devices.data(), -1, comms.data()) != PG_OK) die("pg_comm_create_local");
        mark("communicator created");
        fprintf(stderr, "%d k-mer set(s), pass 1 on %d rank(s) (HIP devices", o.sets, n_ranks);
        for (int r = 0; r < n_ranks; r++) fprintf(stderr, " %d", devices[r]);
        fprintf(stderr, "), records exchanged by %s.\n", pg_comm_transport(comms[0]) == PG_COMM_RCCL ? "RCCL all-to-all" : "peer copies");
        int shared = 1;                                            // ranks on one device share its memory
        for (int r = 0; r < n_ranks; r++) { int c = 0; for (int q = 0; q < n_ranks; q++) c += devices[q] == devices[r]; shared = std::max(shared, c); }
        int ls = log2_slots;
        for (int sft = n_ranks * shared; sft > 1 && ls > 20; sft >>= 1) ls--;
        for (int r = 0; r < n_ranks; r++) {
            // the partition count follows the WHOLE input (every rank cuts with the same geometry, and a partition holds what all
            // ranks send to it); the export array follows this rank's share
            // (round 6: a rank STORES the partitions it owns and no others -- cursors, chunk table and pool follow 1 / n_ranks of the job)
            ctxs[r] = pg_create_planned(devices[r], K, mer127 ? 1 : 0, o.sets, ls, 2, est_kmers + 1, est_reads, export_records, n_ranks);
            if (!ctxs[r]) die("pg_create");
        }
        mark("device context created (HIP start-up, record pools, export arrays of all ranks)");
        {
            ShardedPass1 p1(ctxs, comms, devices, K, batch_words, batch_reads);
            mark("pinned batch buffers allocated");
            p1.keep_reads(keep_budget);
            for (const pg::InputFile& f : files) {
                fprintf(stderr, "Import reads from file:\n %s\n", f.path1.c_str());
                if (!f.path2.empty()) fprintf(stderr, "Import reads from file:\n %s\n", f.path2.c_str());
                n_records += pg::stream_reads(f, p1);
            }
            mark("files read");
            if (!p1.finish_ok()) { fprintf(stderr, "pass 1 failed on a rank\n"); exit(-1); }
            mark("last round cut and sent on every rank");
            total_kmers = p1.total_kmers();
            have_kept = p1.take_kept(kept);
        }
        lap("parse + scatter + exchange (pass 1)");
        // count: all ranks at once; the coverage histogram and the per-set counts are summed over the ranks (all-reduce), the
        // per-set last put is the latest over the ranks.  Then the distinct k-mers move to the rank that owns their reference
        // set (set s -> rank s mod n_ranks) and are put in replay order there: from here on a k-mer set lives whole
        // on one GPU and nothing is gathered anywhere.
        const int rw1 = (mer127 ? 4 : 2) + 2;
        std::vector<std::vector<uint64_t>> last(n_ranks, std::vector<uint64_t>(o.sets, 0));
        std::vector<uint64_t> n_before(n_ranks, 0);
        std::vector<std::string> err(n_ranks);
        sh_rec.assign(n_ranks, nullptr);
        sh_n.assign(n_ranks, 0);
        sh_ws.assign(n_ranks, nullptr); sh_mine.assign(n_ranks, nullptr); sh_ws_front.assign(n_ranks, 0); sh_in_ws.assign(n_ranks, 0); sh_offered.assign(n_ranks, 0);
        sh_per_set.assign(o.sets, 0);
        {
            std::vector<std::thread> th;
            for (int r = 0; r < n_ranks; r++)
                th.emplace_back([&, r] {
                    uint64_t h[512];                                  // coverage histogram | distinct k-mers per set
                    uint64_t* d_h = nullptr;
                    auto fail = [&](const char* what) { if (err[r].empty()) err[r] = std::string(what) + ": " + pg_last_error(); };
                    bool ok = pg_finalize(ctxs[r], o.delow, h, nullptr, nullptr) == PG_OK;
                    if (!ok) fail("pg_finalize");
                    if (ok && pg_set_counts(ctxs[r], h + 256, nullptr) != PG_OK) { ok = false; fail("pg_set_counts"); }
                    if (!ok) memset(h, 0, sizeof h);
                    // the collectives below are entered by every rank, failed or not
                    if (hipSetDevice(devices[r]) != hipSuccess || pg::arena_malloc((void**)&d_h, sizeof h) != hipSuccess ||
                        hipMemcpy(d_h, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "rank %d: no device memory\n", r); exit(-1); }
                    if (pg_exchange_allreduce_u64(comms[r], d_h, 512, nullptr) != PG_OK && ok) { ok = false; fail("pg_exchange_allreduce_u64"); }
                    HIP_OK(hipMemcpy(h, d_h, sizeof h, hipMemcpyDeviceToHost));
                    if (r == 0) { memcpy(hist, h, 256 * sizeof(uint64_t)); for (int sidx = 0; sidx < o.sets; sidx++) sh_per_set[sidx] = h[256 + sidx]; }
                    pg::arena_free(d_h);
                    // every rank sees the same totals, so all of them take the same decision about the last put
                    if (ok && pg_host_last_put_matters(h + 256, o.sets, o.a_gb, mer127 ? 1 : 0) && pg_last_put(ctxs[r], last[r].data(), nullptr) != PG_OK) { ok = false; fail("pg_last_put"); }
                    uint64_t* d_mine = nullptr;
                    uint64_t n_mine = 0;
                    void* d_ws = nullptr;
                    uint64_t ws_bytes = 0;
                    if (ok && pg_export_take_ws(ctxs[r], &d_mine, &n_mine, &d_ws, &ws_bytes) != PG_OK) { ok = false; fail("pg_export_take"); }
                    if (!ok) { n_mine = 0; d_mine = nullptr; }
                    n_before[r] = n_mine;
                    pg_destroy(ctxs[r]);                              // the tables; the record pool stays as the workspace of what follows
                    uint64_t* d_g = nullptr;
                    uint64_t n_g = 0;
                    int in_ws = 0;
                    if (pg_exchange_regroup_by_set_ws(comms[r], d_mine, n_mine, rw1, d_ws, ws_bytes, &d_g, &n_g, &in_ws, nullptr) != PG_OK) { ok = false; fail("pg_exchange_regroup_by_set"); }
                    const uint64_t front = in_ws ? (uint64_t)((char*)d_g - (char*)d_ws) : ws_bytes;
                    if (ok && n_g && (hipSetDevice(devices[r]) != hipSuccess || pg_sort_records_ws(d_g, n_g, mer127 ? 1 : 0, d_ws, front, nullptr) != PG_OK)) { ok = false; fail("pg_sort_records"); }
                    sh_rec[r] = d_g; sh_n[r] = n_g;
                    sh_ws[r] = d_ws; sh_ws_front[r] = front; sh_in_ws[r] = (char)in_ws; sh_mine[r] = d_ws ? (void*)d_mine : nullptr;   // (without a workspace the regroup has released d_mine)
                });
            for (auto& t : th) t.join();
        }
        for (int r = 0; r < n_ranks; r++) if (!err[r].empty()) { fprintf(stderr, "rank %d: %s\n", r, err[r].c_str()); exit(-1); }
        uint64_t n_all = 0;
        for (int r = 0; r < n_ranks; r++) n_all += sh_n[r];
        for (int r = 0; r < n_ranks; r++) {
            for (int sidx = 0; sidx < o.sets; sidx++) set_last[sidx] = std::max(set_last[sidx], last[r][sidx]);
            if (verbose) {
                uint64_t cs[4];
                pg_comm_stats(comms[r], cs);
                fprintf(stderr, "[cli] rank %d (device %d): %llu distinct k-mers after pass 1, %llu of %llu after the regroup by set; %llu rounds, %llu records sent, %llu received\n",
                        r, devices[r], (unsigned long long)n_before[r], (unsigned long long)sh_n[r], (unsigned long long)n_all, (unsigned long long)cs[0],
                        (unsigned long long)cs[1], (unsigned long long)cs[2]);
                uint64_t ps[8];
                pg_comm_pipeline_stats(comms[r], ps);
                fprintf(stderr, "[cli] rank %d exchange: %.1f ms of record exchanges on their own stream (%.2f GB to other ranks), %llu host wait(s) in %llu round(s), %llu repeated cut(s), "
                                "%llu records an owner region\n", r, (double)ps[0] / 1000.0, (double)ps[1] / 1e9, (unsigned long long)ps[2], (unsigned long long)ps[4],
                        (unsigned long long)ps[3], (unsigned long long)ps[5]);
            }
            pg_comm_destroy(comms[r]);
        }
        n_distinct = n_all;
        HIP_OK(hipSetDevice(device));
    } else {
    // Pass 1 on the partition engine; should one partition outgrow its chunk list (one minimizer owning a huge share of
    // the input) the reads go through the global-set engine instead -- slower, indifferent to skew, same result.
    int engine = 2;
    if (const char* e = pg::env_user("PG_ENGINE")) engine = atoi(e);
    for (int attempt = 0;; attempt++) {
        mark("input files sized");
        // the context (HIP start-up, the record pool: tens of gigabytes the driver hands out cleared) on a thread of its own while this one
        // page-locks the batch buffers (which takes half a second when the page cache is full of a file somebody has just written)
        std::string ctx_err;
        std::thread ctx_thread([&] {
            (void)hipSetDevice(device);
            ctx = pg_create_planned(device, K, mer127 ? 1 : 0, o.sets, log2_slots, engine, est_kmers, est_reads, engine == 2 ? export_records : 0, 1);
            if (!ctx) ctx_err = pg_last_error();
        });
        if (pg::env_measure("PG_CTX_THREAD") && atoi(pg::env_measure("PG_CTX_THREAD")) == 0) ctx_thread.join();      // (A/B: one after the other)
        bool ok = true;
        {
            Pass1 p1(nullptr, K, batch_words, batch_reads);
            mark("pinned batch buffers allocated");
            if (ctx_thread.joinable()) ctx_thread.join();
            if (!ctx) { fprintf(stderr, "%s\n", ctx_err.c_str()); die("pg_create"); }
            p1.set_ctx(ctx);
            mark("device context created (HIP start-up, record pool; the export array follows beside pass 1)");
            if (attempt == 0) fprintf(stderr, "%d k-mer set(s) on HIP device %d.\n", o.sets, device);
            if (attempt == 0) {
                p1.keep_reads(keep_budget);
                p1.keep_reads_on_device(dev_keep_budget);
                for (const pg::InputFile& f : files) {
                    fprintf(stderr, "Import reads from file:\n %s\n", f.path1.c_str());
                    if (!f.path2.empty()) fprintf(stderr, "Import reads from file:\n %s\n", f.path2.c_str());
                    n_records += pg::stream_reads(f, p1);
                }
            } else if (have_kept) {
                for (const KeptReads::Block& kb : kept.blocks) {
                    int mn = 0x7fffffff, mx = 0;
                    for (int32_t l : kb.lens) { mn = std::min(mn, (int)l); mx = std::max(mx, (int)l); }
                    if (!kb.lens.empty()) p1.on_packed(kb.words, kb.lens.data(), kb.lens.size(), mn, mx);
                }
            } else {
                for (const pg::InputFile& f : files) pg::stream_reads(f, p1);
            }
            mark("files read");
            ok = p1.finish_ok();
            mark("last batch cut on the device");
            total_kmers = p1.total_kmers();
            if (attempt == 0) {
                (void)p1.take_dev_kept(devkept);                   // (the batch still being filled went out with finish_ok)
                devkept.device = device;
                have_kept = p1.take_kept(kept);
            }
        }
        lap("parse + scatter (pass 1)");
        // ---- -d filter, linear marking, .kmerFreq (deLowCov / Mark1in1outNode / freqStat); with the partition engine
        // this is also where the partitions are counted, so the node count is known only afterwards
        if (ok && pg_finalize(ctx, o.delow, hist, engine == 2 ? nullptr : set_last.data(), nullptr) != PG_OK) ok = false;
        if (ok && pg_distinct(ctx, &n_distinct, nullptr) != PG_OK) ok = false;
        if (ok && engine == 2) {
            // the per-set last put costs a second expansion of every record and matters only when a set ends exactly at a
            // growth threshold of the reference's size schedule (newhash.c:477): ask first
            uint64_t per[256];
            if (pg_set_counts(ctx, per, nullptr) != PG_OK) ok = false;
            else if (pg_host_last_put_matters(per, o.sets, o.a_gb, mer127 ? 1 : 0) && pg_last_put(ctx, set_last.data(), nullptr) != PG_OK) ok = false;
        }
        if (ok) { engine_used = engine; break; }
        if (engine != 2 || attempt > 0) die("pass 1");
        fprintf(stderr, "Partition engine gave up (%s); counting again with the global k-mer set.\n", pg_last_error());
        pg_destroy(ctx);
        engine = 1;
        if (!devkept.segs.empty()) {                               // the second attempt is fed from the host store
            std::vector<uint64_t> tmp;
            std::vector<int32_t> ls;
            for (const DevKept::Seg& sg : devkept.segs)
                if (!DevKept::fetch(sg, K, tmp, ls) || !kept.append(tmp.data(), tmp.size(), ls.data(), sg.n_reads)) { have_kept = false; kept.clear(); break; }
            devkept.clear();
        }
    }
    }
    lap("count partitions (finalize)");
    time_t t1 = time(nullptr);
    fprintf(stderr, "Time spent on hashing reads: %ds, %lld read(s) processed.\n", (int)(t1 - t0), n_records);
    fprintf(stderr, "%llu node(s) allocated, %llu kmer(s) in reads, %llu kmer(s) processed.\n", (unsigned long long)n_distinct,
            (unsigned long long)total_kmers, (unsigned long long)total_kmers);
    fprintf(stderr, "done hashing nodes\n");
    t0 = time(nullptr);
    if (pg_host_write_kmerfreq(hist, o.prefix.c_str()) != PG_OK) die("kmerFreq");
    fprintf(stderr, "Time spent on marking linear nodes: %ds.\n", (int)(time(nullptr) - t0));
    fprintf(stderr, "Time spent on pre-graph construction: %ds.\n\n", (int)(time(nullptr) - t_start));

    // ---- export the distinct k-mers and hand over to the host stages
    const int rw = (mer127 ? 4 : 2) + 2;
    // the host copy of the records: plain anonymous memory on huge pages, never value-initialised (several GB)
    struct HostRecords {
        uint64_t* p = nullptr; size_t bytes = 0, words = 0;
        void alloc(size_t n_words) {
            words = n_words;
            const size_t HP = (size_t)2 << 20;
            bytes = (n_words * sizeof(uint64_t) + HP - 1) / HP * HP + HP;
            void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) { fprintf(stderr, "out of memory (%zu bytes for the node records)\n", bytes); exit(-1); }
            madvise(q, bytes, MADV_HUGEPAGE);
            p = (uint64_t*)q;
        }
        void release() { if (p) munmap(p, bytes); p = nullptr; bytes = words = 0; }
        uint64_t* data() { return p; }
        size_t size() const { return words; }
        ~HostRecords() { release(); }
    } records;
    // Records in replay order stay on the device and are pulled set by set, chunk by chunk, while the layout is rebuilt
    // (pg_graph_begin_streamed); with SOAPDENOVO2_AMD_STREAM_RECORDS=0 they are downloaded whole.
    bool stream_records = n_distinct > 0;
    if (const char* e = pg::env_test("SOAPDENOVO2_AMD_STREAM_RECORDS")) stream_records = stream_records && atoi(e) != 0;
    std::vector<uint64_t> per_set(o.sets, 0);
    void* d_ws = nullptr;
    uint64_t ws_bytes = 0;
    if (n_distinct && n_ranks == 1) {
        if (ctx) {
            // the partition engine hands its export array over as it is and frees its streams (no second copy in HBM);
            // the global-set engine compacts its table into a fresh array
            uint64_t got = 0;
            if (engine_used == 2) { if (pg_export_take_ws(ctx, &d_rec, &got, &d_ws, &ws_bytes) != PG_OK) die("pg_export_take"); }
            else {
                HIP_OK(pg::arena_malloc((void**)&d_rec, (size_t)n_distinct * rw * sizeof(uint64_t)));
                if (pg_export(ctx, d_rec, n_distinct, &got, nullptr) != PG_OK) die("pg_export");
            }
            if (got != n_distinct) { fprintf(stderr, "export count mismatch\n"); exit(-1); }
        }
        // replay order (set, first occurrence) on the device, in the memory pass 1 is done with -- or, when that cannot hold the sort's work space, without it:
        // the pool goes back to the arena first and the sort cuts what it needs out of the hole (cmd_plan.hpp)
        if (d_ws && ws_bytes < pg::cmd_sort_ws_bytes(n_distinct, mer127)) { (void)pg::arena_free(d_ws); d_ws = nullptr; ws_bytes = 0; }
        if (pg_sort_records_ws(d_rec, n_distinct, mer127 ? 1 : 0, d_ws, ws_bytes, nullptr) != PG_OK) die("pg_sort_records");
        if (stream_records) {
            // where every set starts: binary search over the sorted tags (a few hundred 8-byte copies)
            auto set_of_record = [&](uint64_t i) { uint64_t tag = 0; HIP_OK(hipMemcpy(&tag, d_rec + i * rw + rw - 1, sizeof tag, hipMemcpyDeviceToHost)); return tag >> 56; };
            std::vector<uint64_t> start(o.sets + 1, n_distinct);
            start[0] = 0;
            for (int sidx = 1; sidx < o.sets; sidx++) {               // first record whose set id is >= sidx
                uint64_t lo = start[sidx - 1], hi = n_distinct;
                while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (set_of_record(mid) >= (uint64_t)sidx) hi = mid; else lo = mid + 1; }
                start[sidx] = lo;
            }
            for (int sidx = 0; sidx < o.sets; sidx++) per_set[sidx] = start[sidx + 1] - start[sidx];
        } else {
            records.alloc((size_t)n_distinct * rw + 1);
            HIP_OK(hipMemcpy(records.data(), d_rec, (size_t)n_distinct * rw * sizeof(uint64_t), hipMemcpyDeviceToHost));
            pg::arena_free(d_rec);
            d_rec = nullptr;
        }
    }
    if (ctx) pg_destroy(ctx);
    // pass 1's record pool: the k-mer sets are laid out inside it (no allocation of that size behind a free of that size:
    // seconds); when the records go to the host it goes now -- the host replay gives the driver the time
    bool ws_offered = false;
    if (d_ws && stream_records && pg_device_scratch_offer(device, d_ws, ws_bytes) == PG_OK) ws_offered = true;
    else if (d_ws) { (void)pg::arena_free(d_ws); d_ws = nullptr; }
    lap("export + download records");

    // ---- tips + edges (removeSingleTips / removeMinorTips / kmer2edges), graph kept for pass 2
    t0 = time(nullptr);
    // edges and pass 2 on the device; SOAPDENOVO2_AMD_EDGES=host / SOAPDENOVO2_AMD_PASS2=host keep them on the host threads
    bool host_edges = false, host_pass2 = false;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_EDGES")) host_edges = strcmp(e, "host") == 0;
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_PASS2")) host_pass2 = strcmp(e, "host") == 0;
    if (host_pass2) host_edges = true;                               // host pass 2 needs the host copy of the sets tagged
    // records still on the device: with -a the layout is made there (K6), otherwise the replay's workers pull their stretches
    (void)pg_host_edge_file_in_background(pg::env_test("SOAPDENOVO2_AMD_EDGE_FILE_INLINE") ? 0 : 1);    // <o>.edge.gz is formatted and deflated beside pass 2
    for (int r = 0; r < (int)sh_ws.size(); r++)                      // one block a device (ranks that share a GPU in test set-ups: the first one's)
        if (sh_ws[r] && sh_ws_front[r] && pg_device_scratch_offer(devices[r], sh_ws[r], sh_ws_front[r]) == PG_OK) sh_offered[r] = 1;
    pg_graph* graph = n_ranks > 1
        ? pg_graph_begin_sharded(n_ranks, devices.data(), sh_rec.data(), sh_n.data(), sh_per_set.data(), set_last.data(), K, mer127 ? 1 : 0, o.sets, o.delow == 0,
                                 o.a_gb, max_read_len, 0, o.prefix.c_str())
        : stream_records
        ? pg_graph_begin_device(d_rec, device, n_distinct, per_set.data(), set_last.data(), K, mer127 ? 1 : 0, o.sets, o.delow == 0, o.a_gb,
                                max_read_len, 0, o.prefix.c_str(), host_edges ? -1 : device)
        : pg_graph_begin(records.data(), n_distinct, set_last.data(), K, mer127 ? 1 : 0, o.sets, o.delow == 0, o.a_gb,
                         max_read_len, 0, o.prefix.c_str(), host_edges ? -1 : device);
    (void)pg_host_edge_file_in_background(0);                         // (the decision was taken while the edges were built; other callers in this process keep the default)
    if (ws_offered) { if (void* back = pg_device_scratch_withdraw(device)) (void)pg::arena_free(back); d_ws = nullptr; }
    if (d_rec) { pg::arena_free(d_rec); d_rec = nullptr; }
    for (int r = 0; r < (int)sh_rec.size(); r++) {
        (void)hipSetDevice(devices[r]);
        // the pool: offered and still there -> ours to release; offered and gone -> a layout laid its k-mer sets out in it and the
        // graph owns it now (the records in its tail are done with either way)
        if (sh_offered[r]) { if (void* back = pg_device_scratch_withdraw(devices[r])) pg::arena_free(back); }
        else if (sh_ws[r]) pg::arena_free(sh_ws[r]);
        if (sh_rec[r] && !sh_in_ws[r]) pg::arena_free(sh_rec[r]);
        sh_rec[r] = nullptr;
        if (sh_mine[r]) pg::arena_free(sh_mine[r]);
    }
    (void)hipSetDevice(device);
    if (!graph) die("pg_host_graph_begin");
    if (o.reps && pg_host_graph_resolve_repeats(graph, 1) != PG_OK) die("pg_host_graph_resolve_repeats");
    if (!host_pass2 && pg_graph_use_device(graph, device) != PG_OK) die("pg_graph_use_device");
    records.release();
    fprintf(stderr, "Time spent on removing tips and constructing edges: %ds.\n\n", (int)(time(nullptr) - t0));
    lap("layout + tips + edges");

    // ---- pass 2 (prlRead2edge): the reads again, in the same order, threaded through the edges -> .preArc
    t0 = time(nullptr);
    if (have_kept && !devkept.segs.empty() && !host_pass2) {
        // the reads of pass 1 are still on the device (one segment a batch): threaded where they are
        fprintf(stderr, "In file: %s, max seq len %d, max name len %d.\n", o.config.c_str(), max_read_len, 256);
        bool one_length = true;
        for (const DevKept::Seg& sg : devkept.segs) one_length = one_length && sg.len != 0 && sg.len == devkept.segs[0].len;
        if (one_length) {                                              // all of them at once, threaded in genome order (graph_kernels.hip: p2_add_packed_device_segments)
            std::vector<const uint64_t*> ptrs;
            std::vector<uint64_t> counts;
            for (const DevKept::Seg& sg : devkept.segs) { ptrs.push_back(sg.d); counts.push_back(sg.n_reads); }
            if (pg_graph_add_packed_device_segments(graph, ptrs.data(), counts.data(), (int)ptrs.size(), devkept.segs[0].len, devkept.device) != PG_OK) die("pg_graph_add_packed_device_segments");
        } else for (const DevKept::Seg& sg : devkept.segs) {
            if (sg.len) { if (pg_graph_add_packed_device(graph, sg.d, sg.n_reads, sg.len, devkept.device) != PG_OK) die("pg_graph_add_packed_device"); }
            else if (pg_graph_add_packed_device_ragged(graph, sg.d, sg.d_off(), sg.d_base(), sg.n_reads, sg.n_kmers, sg.max_len, devkept.device) != PG_OK) die("pg_graph_add_packed_device_ragged");
        }
        devkept.clear();
        fprintf(stderr, "%lld read(s) processed.\n", n_records);
    } else if (have_kept) {
        fprintf(stderr, "In file: %s, max seq len %d, max name len %d.\n", o.config.c_str(), max_read_len, 256);
        // a pass-2 batch: what a pass-1 batch was (a sharded graph deals them to its lanes in turn, graph_kernels.hip: P2Lane)
        const uint64_t step = std::min<uint64_t>((uint64_t)1 << 22, (uint64_t)batch_reads);
        for (const KeptReads::Block& kb : kept.blocks) {
            const uint64_t total = kb.lens.size();
            uint64_t word_at = 0;
            for (uint64_t lo = 0; lo < total; lo += step) {
                const uint64_t n = std::min(step, total - lo);
                if (pg_host_graph_add_packed(graph, kb.words + word_at, kb.lens.data() + lo, n, 0) != PG_OK) die("pg_host_graph_add_packed");
                for (uint64_t r = lo; r < lo + n; r++) word_at += pg_packed_words((uint32_t)kb.lens[r]);
            }
        }
        fprintf(stderr, "%lld read(s) processed.\n", n_records);
    } else {
        struct Pass2 : pg::ReadSink {
            pg_graph* g; int K; size_t stride, cap, n = 0;
            std::vector<uint8_t> codes; std::vector<int32_t> lens;
            void flush() {
                if (!n) return;
                if (pg_host_graph_add_reads(g, codes.data(), lens.data(), n, stride, 0) != PG_OK) die("pg_host_graph_add_reads");
                n = 0;
            }
            void on_read(const uint8_t* c, int len) override {
                if (len < K + 1) return;                              // prlRead2path.c:1103
                memcpy(codes.data() + n * stride, c, (size_t)len);
                lens[n++] = len;
                if (n == cap) flush();
            }
        } p2;
        p2.g = graph; p2.K = K; p2.stride = (size_t)max_read_len; p2.cap = (size_t)1 << 20;
        p2.codes.resize(p2.stride * p2.cap); p2.lens.resize(p2.cap);
        fprintf(stderr, "In file: %s, max seq len %d, max name len %d.\n", o.config.c_str(), max_read_len, 256);
        long long n2 = 0;
        for (const pg::InputFile& f : files) {
            fprintf(stderr, "Import reads from file:\n %s\n", f.path1.c_str());
            n2 += pg::stream_reads(f, p2);
        }
        p2.flush();
        fprintf(stderr, "%lld read(s) processed.\n", n2);
    }
    lap("pass 2 batches");
    int num_vt = 0, num_ed = 0;
    long long num_arc = 0;
    if (pg_host_graph_finish(graph, &num_vt, &num_ed, &num_arc) != PG_OK) die("pg_host_graph_finish");
    lap("pass 2 finish (preArc, vertex)");
    fprintf(stderr, "Time spent on aligning reads: %ds.\n\n", (int)(time(nullptr) - t0));
    fprintf(stderr, "Overall time spent on constructing pre-graph: %dm.\n\n", (int)(time(nullptr) - t_start) / 60);
    return 0;
}

}  // namespace

extern "C" int call_pregraph(int argc, char** argv) { return run(argc, argv, false); }
extern "C" int call_pregraph_127mer(int argc, char** argv) { return run(argc, argv, true); }
