/*
 * soapdenovo2_amd.h -- C ABI of libsoapdenovo2_amd.so: the MI355X-native `pregraph` hot path of SOAPdenovo2.
 *
 * Plain C, plain pointers and sizes.  Three layers, outermost first:
 *
 *  1. call_pregraph()      the drop-in boundary.  Same signature, argv grammar, output files and exit
 *                          behaviour as the reference's `int call_pregraph(int argc, char **argv)`
 *                          (standardPregraph/pregraph.c:62; reached from main, standardPregraph/main.c:72-75).
 *  2. pg_host_*()          host-side stages that need no GPU: k-mer-set layout replay, tip clipping, edge
 *                          construction and the writers (replaces newhash.c:340-455, cutTipPreGraph.c,
 *                          node2edge.c, output_pregraph.c).
 *  3. pg_* (device)        the HIP operators of pass 1 on caller-owned device buffers: k-mer extraction +
 *                          hash-set insert, finalize (low-coverage filter, linear marking, k-mer frequency
 *                          histogram), export (replaces prlHashReads.c:163-259 chopKmer4read,
 *                          hashFunction.c:155 hash_kmer, newhash.c:473-528 put_kmerset,
 *                          prlHashReads.c:953-1132 thread_delow/thread_mark/freqStat).
 *
 * Every function returns 0 on success and a negative PG_E* code on failure unless stated otherwise;
 * pg_last_error() returns a thread-local message.  There is no CPU fallback: device functions fail with
 * PG_ENODEV when no HIP device is usable.
 */
#ifndef SOAPDENOVO2_AMD_H
#define SOAPDENOVO2_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_EINVAL (-1)   /* bad argument */
#define PG_ENODEV (-2)   /* no usable HIP device / HIP runtime error */
#define PG_ENOMEM (-3)   /* host or device allocation failed, or table full and cannot grow */
#define PG_EIO (-4)      /* file error */
#define PG_ESTATE (-5)   /* call order violated */

const char *pg_last_error(void);
const char *pg_version(void);

/* ------------------------------------------------------------------------------------------------
 * 1. Drop-in boundary.
 *
 * argv grammar = the reference's getopt string "a:s:o:K:p:d:R" (pregraph.c:142-220):
 *   pregraph -s configFile -o outputGraph [-K kmer -p n_sets -a initMemoryAssumption -d KmerFreqCutoff -R]
 * -p is the number of k-mer sets ("threads" in the reference); it fixes the order of .vertex/.edge.gz and
 * is honoured as such regardless of how many GPUs or host threads do the work.
 * -p must be within 1..255 (the reference keeps thread ids in an `unsigned char`, prlHashReads.c:66-126; larger
 * values misbehave there and are refused here).
 * Writes <o>.kmerFreq <o>.preGraphBasic <o>.vertex <o>.edge.gz <o>.preArc, and with -R also <o>.path and
 * <o>.markOnEdge (recordPathBin / output_arcs, prlRead2path.c:435-543).
 * Returns 0; fatal errors print to stderr and exit(), as the reference does (check.c:31,96).
 * call_pregraph        = behaviour of the SOAPdenovo-63mer binary (K <= 63, two hex words per k-mer)
 * call_pregraph_127mer = behaviour of the SOAPdenovo-127mer binary (K <= 127, four hex words)
 * Extra environment knobs (not in the reference): SOAPDENOVO2_AMD_DEVICE (HIP device ordinal, default 0).
 * ------------------------------------------------------------------------------------------------ */
int call_pregraph(int argc, char **argv);
int call_pregraph_127mer(int argc, char **argv);

/* ------------------------------------------------------------------------------------------------
 * Common record type: one distinct canonical k-mer after pass 1.
 *   key[0..nw)  k-mer words, most significant first (nw = 2 for the 63-mer flavour, 4 for the 127-mer one)
 *   cnt         low 32 bit = word A (l_links | covs<<24), high 32 bit = word B (r_links | flag bits),
 *               exactly the two 32-bit words of the reference's kmer_t (inc/newhash.h:77-102)
 *   ord         bits 55:0 = global ordinal of the k-mer's first occurrence (running k-mer index over all
 *               accepted reads, prlHashReads.c:647-648); bits 63:56 = k-mer set id = hash_kmer % n_sets
 * A record is (nw + 2) uint64_t; arrays of records are dense.
 * ------------------------------------------------------------------------------------------------ */
#define PG_ORD_BITS 56
#define PG_ORD_MASK ((1ULL << PG_ORD_BITS) - 1)

/* ------------------------------------------------------------------------------------------------
 * 2. Host stages (no GPU needed).
 * ------------------------------------------------------------------------------------------------ */

/* Pack base codes (one byte per base, values 0..3 = A C T G) into the device read format: 2 bits per base,
 * first base in the most significant bits of a 64-bit word, every read starting on a word boundary.
 * words_out must hold pg_packed_words(len) words.  Replaces the seqBuffer layout of prlHashReads.c:354-361. */
size_t pg_packed_words(uint32_t len);
void pg_pack_read(const uint8_t *codes, uint32_t len, uint64_t *words_out);

/* The ingestion stage alone (scan_libInfo lib.c:130-506, openNextFile/AIORead prlHashReads.c:771-951, readseqInLib
 * readseq1by1.c:927-1035): parse `config`, visit the input files in the reference's order and deliver every read it
 * would hand to pass 1 (length >= K + 1 after truncation to max_rd_len / rd_len_cutoff, reverse_seq applied) as base
 * codes, read i at codes_out + i * stride with its length in lens_out[i].  codes_out = NULL only counts.
 * n_records = records parsed (the reference's "read(s) processed"), n_accepted = reads delivered. */
int pg_host_read_all(const char *config, int K, uint8_t *codes_out, int32_t *lens_out, uint64_t capacity_reads,
                     uint64_t stride, uint64_t *n_records, uint64_t *n_accepted, int *max_read_len_out);

/* BAM inputs (b=): the reader pairs records up two by two and takes pairs with a QC-fail mate back (readseq1by1.c:449-592);
 * the pairing state is a static of the reference (readseq1by1.c:44) which it puts back to -3 at every end of file
 * (readseq1by1.c:584-587) -- mirrored here, so every file and both passes start pairing afresh.
 * set != 0 stores `value` (-3 = as in a fresh process); returns the state. */
int pg_host_bam_pair_state(int set, int value);

/* Build the reference's k-mer-set layout from the distinct k-mers of pass 1 and run everything after it:
 * [-d] is assumed already applied to `records` by pg_finalize; this replays put_kmerset/encap_kmerset slot
 * placement (newhash.c:340-528) for n_sets sets, then removeSingleTips/removeMinorTips, kmer2edges and
 * output_vertex (pregraph.c:106-131), writing <prefix>.vertex .edge.gz .preGraphBasic.
 *   records        n_records records as described above, any order
 *   set_last_put   per set: 1 + ordinal of the last k-mer occurrence routed to that set (0 = none); needed
 *                  to decide whether a duplicate put arrived after the set's last distinct key (the growth
 *                  check of newhash.c:477 runs before the probe)
 *   cut_single     non-zero = run removeSingleTips first (the reference does when -d 0, pregraph.c:106)
 *   a_gb           the -a value (0 = growable sets starting at 1031 slots)
 *   n_threads      host threads for the per-set replay (0 = hardware concurrency)
 * out_num_vertex / out_num_edge (may be NULL) receive the counts written to .preGraphBasic. */
int pg_host_build_graph(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put,
                        int K, int mer127, int n_sets, int cut_single, int a_gb, int max_read_len,
                        int n_threads, const char *prefix, int *out_num_vertex, int *out_num_edge);

/* The same host stages kept alive for pass 2 (prlRead2edge, prlRead2path.c:786-1370: read -> edge threading, pre-arcs):
 *   pg_host_graph_begin      replay + tips + edges (writes <prefix>.edge.gz), keeps the k-mer sets and the (K+1)-mer
 *                            patch table of the length-1 edges (KmerSetsPatch, node2edge.c:481-542) in memory
 *   pg_host_graph_add_reads  threads a batch of reads (base codes, read i at codes + i * stride, lens[i] bases or
 *                            `stride` bases when lens is NULL) through the graph and accumulates the pre-arcs; batches
 *                            must come in the reference's read order (the order of a pre-arc list is first-encounter order)
 *   pg_host_graph_add_packed the same for reads packed with pg_pack_read and laid back to back (read i starts
 *                            pg_packed_words(lens[0]) + ... + pg_packed_words(lens[i-1]) words in): lets a caller that kept
 *                            pass 1's packed reads in memory skip the second parse of the input files
 *   pg_host_graph_resolve_repeats  the reference's -R (pregraph.c:181-184): call with on != 0 right after _begin to also
 *                            record every read's edge walk in <prefix>.path (recordPathBin, prlRead2path.c:478-543) and the
 *                            per-edge marker counts in <prefix>.markOnEdge (output_arcs, prlRead2path.c:435-449)
 *   pg_graph_use_device      call right after _begin (and _resolve_repeats) to run pass 2 on HIP device `device` instead of the
 *                            host threads: the k-mer sets and the (K+1)-mer table are copied to HBM once, every batch is
 *                            threaded by one kernel (a lane a read) and the pre-arcs are accumulated in a device table
 *                            (multiplicity + first-met order per (from, to)); same files, byte for byte
 *   pg_host_graph_finish     writes <prefix>.preArc, <prefix>.vertex, <prefix>.preGraphBasic and frees the handle */
typedef struct pg_graph pg_graph;
/* pg_graph_begin: as pg_host_graph_begin, but with device >= 0 the edges are built on that HIP device (make_edge /
 * stringBeads / merge_linearV2, node2edge.c:61-649): after the layout replay and the tips the k-mer sets are copied to HBM,
 * every (vertex, arc) is walked by a lane, the walk that comes first in slot order keeps its chain, ids follow from a sort
 * and a prefix sum, interior nodes are tagged and length-1 edges entered into KmerSetsPatch on the device; the host only
 * formats <prefix>.edge.gz.  Pass 2 then runs on the same device (pg_graph_use_device is implied).  device = -1: host. */
pg_graph *pg_graph_begin(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put, int K, int mer127,
                         int n_sets, int cut_single, int a_gb, int max_read_len, int n_threads, const char *prefix, int device);
/* pg_graph_begin for records that are still in device memory, in replay order (pg_sort_records): per_set_count[s] records
 * of set s follow one another; fetch(user, first_record, n_records, dst) copies a stretch of them to host memory (a
 * hipMemcpy in the caller's hands, called from several threads) and returns PG_OK; a call with n_records = 0 tells the
 * caller that the calling thread will not ask again (e.g. to un-register its destination buffer).  Every set's worker pulls and inserts
 * chunk by chunk, so the download overlaps the layout replay and no host copy of all records is needed. */
pg_graph *pg_graph_begin_streamed(int (*fetch)(void *user, uint64_t first_record, uint64_t n_records, uint64_t *dst), void *user,
                                  uint64_t n_records, const uint64_t *per_set_count, const uint64_t *set_last_put, int K,
                                  int mer127, int n_sets, int cut_single, int a_gb, int max_read_len, int n_threads,
                                  const char *prefix, int device);
/* pg_graph_begin for records that lie in the memory of HIP device `records_device`, in replay order (pg_sort_records), with the
 * copies in the library's hands.  With -a (a_gb != 0: static pools, which never grow or rehash, newhash.c:353-366) and
 * device == records_device the k-mer-set layout itself is made on the device (SURVEY.md App. C "K6": first-come-first-served
 * probing as a sort by home slot + one sweep per probe cluster) and only downloaded; growable sets (a_gb == 0) are replayed
 * by the host threads, which pull their stretch of records chunk by chunk.  SOAPDENOVO2_AMD_LAYOUT=host keeps the host
 * replay for -a too (A/B runs). */
pg_graph *pg_graph_begin_device(const uint64_t *d_records, int records_device, uint64_t n_records, const uint64_t *per_set_count,
                                const uint64_t *set_last_put, int K, int mer127, int n_sets, int cut_single, int a_gb,
                                int max_read_len, int n_threads, const char *prefix, int device);
/* The sharded form ("reference set id -> GPU"): rank r of n_ranks holds, in the memory of HIP device devices[r], the
 * records of the k-mer sets s with s mod n_ranks == r (what pg_exchange_regroup_by_set leaves there), sorted by (set, ordinal)
 * (pg_sort_records); n_records[r] of them; per_set_count[s] over all sets.  Every set's layout is made where its records are --
 * -a: on its rank's GPU (K6); growable sets: by the host threads, pulling from that GPU -- and the set then lives on that GPU
 * only: no rank ever holds more than its share of the sets.  The tip, edge and pass-2 kernels run on devices[0] and reach the
 * other ranks' sets through peer mappings (all ranks belong to this process). */
pg_graph *pg_graph_begin_sharded(int n_ranks, const int *devices, const uint64_t *const *d_records, const uint64_t *n_records,
                                 const uint64_t *per_set_count, const uint64_t *set_last_put, int K, int mer127, int n_sets,
                                 int cut_single, int a_gb, int max_read_len, int n_threads, const char *prefix);
/* <prefix>.edge.gz written beside pass 2 (on != 0): the graph stages that build edges on the device return once the device has
 * handed the edges back, a background thread formats and deflates the text, and pg_host_graph_finish (or destroying the graph)
 * waits for it and reports its error.  A choice of the calling thread for the graphs it begins (graphs of other threads are not
 * touched: the flag is thread-local -- set it on the thread that calls pg_graph_begin*, a Python worker thread included; api.edge_file_in_background
 * says the same); off by default, so a caller that reads the file right after pg_graph_begin* finds it complete.
 * call_pregraph turns it on (output_1edge, node2edge.c:88-110, has no reader before the process ends). */
int pg_host_edge_file_in_background(int on);
/* Device memory the caller is done with, offered to the graph stages for reuse (one block per device): with -a, pg_graph_begin_device
 * lays the k-mer sets out inside it when it is large enough instead of allocating -- a hipMalloc of tens of gigabytes right behind
 * a hipFree of as much takes seconds.  After pg_graph_begin_* the caller withdraws the offer: a non-null result is still the
 * caller's to free, null means the block was taken over (and is freed with the graph). */
int pg_device_scratch_offer(int device, void *d_ptr, uint64_t bytes);
void *pg_device_scratch_withdraw(int device);
pg_graph *pg_host_graph_begin(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put, int K, int mer127,
                              int n_sets, int cut_single, int a_gb, int max_read_len, int n_threads, const char *prefix);
int pg_host_graph_resolve_repeats(pg_graph *g, int on);
int pg_graph_use_device(pg_graph *g, int device);
int pg_host_graph_add_packed(pg_graph *g, const uint64_t *words, const int32_t *lens, uint64_t n_reads, int n_threads);
/* the same for reads that are already in the memory of HIP device `device` (the graph's own: pg_graph_use_device), all of
 * read_len bases, packed back to back, 8 readable words behind the last -- what pass 1 leaves there when it keeps its batches;
 * no copy, no index arrays.  Not for -R runs (the walks of a read come back through the host path). */
int pg_graph_add_packed_device(pg_graph *g, const uint64_t *d_words, uint64_t n_reads, int read_len, int device);
/* ... all the batches pass 1 left there at once (n_segs segments of seg_reads[0] reads each, the last may hold fewer; host arrays of device
 * pointers / counts): one launch threads them sorted by their smallest hashed 16-mer -- reads of one place of the genome side by side, whose
 * lookups the L2 then serves -- instead of in file order.  The order reads are threaded in changes nothing (prlRead2path.c:388-403: a pre-arc
 * list is ordered by the ordinal of the read that met it first, which every read carries).  Measured in round 6 and NOT the default
 * (SOAPDENOVO2_AMD_P2_SORT=1 asks for it; profiles/r06_p2_genome_order_ab.json): without it, with several lanes or with -R the segments are
 * threaded one by one in file order, as by pg_graph_add_packed_device. */
int pg_graph_add_packed_device_segments(pg_graph *g, const uint64_t *const *d_segs, const uint64_t *seg_reads, int n_segs, int read_len,
                                        int device);
/* ... and for a ragged batch left there with the index arrays pg_count_reads took (d_word_off[n_reads], d_kmer_base[n_reads + 1];
 * every read has >= K + 1 bases, none more than max_len; n_kmers = d_kmer_base[n_reads]): prlRead2path.c:1056-1110 threads reads of
 * any length, lengths from lenBuffer.  Same restrictions. */
int pg_graph_add_packed_device_ragged(pg_graph *g, const uint64_t *d_words, const uint64_t *d_word_off, const uint64_t *d_kmer_base,
                                      uint64_t n_reads, uint64_t n_kmers, int max_len, int device);
int pg_host_graph_add_reads(pg_graph *g, const uint8_t *codes, const int32_t *lens, uint64_t n_reads, uint64_t stride,
                            int n_threads);
int pg_host_graph_finish(pg_graph *g, int *out_num_vertex, int *out_num_edge, long long *out_num_prearc);
/* A caller whose process ends right after call_pregraph (the stand-alone executable) says so: the k-mer sets -- tens of
 * gigabytes at scale, seconds to unmap -- are then left to the kernel's exit path instead of being released page by page.
 * Default 0: everything is released before pg_host_graph_finish / call_pregraph return (the `all` pipeline goes on). */
void pg_process_exits_after_this(int yes);

/* The layout replay alone (init_kmerset / put_kmerset / encap_kmerset, newhash.c:200-233,340-528): for every
 * record the slot it occupies in its reference k-mer set, and per set the final table size. */
int pg_host_replay_layout(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put, int mer127,
                          int n_sets, int a_gb, uint64_t *out_slot /* [n_records] */,
                          uint64_t *out_set_size /* [n_sets], may be NULL */);

/* Host twins of the multi-GPU routing step (no GPU; tests and tools): cut a uniform-length batch of packed reads into
 * super-k-mer records exactly as pg_skm_route does -- records_out[i * W ..] (W = 6 / 8 words), tags_out[i] = partition << 8
 * | owner with owner = partition mod n_owners -- and expand records back into k-mer occurrences, rows of
 * (key words..., left, right, ordinal); left / right = base code or 4 for none.  Return the count, or -1. */
int64_t pg_host_skm_cut(const uint64_t *packed, uint64_t n_reads, uint32_t read_len, int K, int mer127, int log2_parts, uint64_t ord_base,
                        int n_owners, uint64_t *records_out, uint64_t *tags_out, uint64_t capacity);
int64_t pg_host_skm_expand(const uint64_t *records, uint64_t n_records, int K, int mer127, uint64_t *out, uint64_t capacity);

/* Host twin of the grouping step of pg_exchange_regroup_by_set (no GPU; tests): counts_out[q] = records whose reference set is
 * owned by rank q (set s -> rank s mod n_ranks), grouped_out = the records grouped by that rank, in their order. */
int pg_host_regroup_plan(const uint64_t *records, uint64_t n_records, int rec_words, int n_ranks, uint64_t *counts_out, uint64_t *grouped_out);

/* Write <prefix>.kmerFreq from the 256-bin coverage histogram (freqStat, prlHashReads.c:1104-1132). */
int pg_host_write_kmerfreq(const uint64_t hist[256], const char *prefix);

/* ------------------------------------------------------------------------------------------------
 * 3. Device operators (HIP, gfx950).  All pointers named d_* are device pointers owned by the caller;
 *    `stream` is a hipStream_t (NULL = default stream).  Launches are asynchronous on that stream unless
 *    stated otherwise.
 * ------------------------------------------------------------------------------------------------ */
typedef struct pg_ctx pg_ctx;

/* Create a counting context on HIP device `device`.
 *   K        k-mer size after the reference's clamp (odd, 13..63 or 13..127)
 *   mer127   0 = two-word k-mers (63-mer binary), 1 = four-word k-mers (127-mer binary)
 *   n_sets   the reference's -p (1..255)
 *   log2_slots  initial capacity of the device hash set (it grows by rehash when load would exceed 70 %) */
pg_ctx *pg_create(int device, int K, int mer127, int n_sets, int log2_slots);
/* Same, choosing the pass-1 formulation explicitly (pg_create uses 2 unless PG_ENGINE says otherwise):
 *   1 = one DRAM-resident open-addressed set, one atomic insert per k-mer occurrence; supports pg_count_records
 *   2 = reads are cut into super-k-mers routed by minimizer partition, every partition is counted in LDS by one
 *       workgroup during pg_finalize; pg_distinct / pg_export are valid only after pg_finalize.
 * Both give bit-identical records. */
pg_ctx *pg_create_engine(int device, int K, int mer127, int n_sets, int log2_slots, int engine);
void pg_destroy(pg_ctx *ctx);
/* Empty the set (keeps its current capacity); asynchronous on `stream`. */
int pg_reset(pg_ctx *ctx, void *stream);
/* on = 0: never grow (and never synchronise) in pg_count_*: the caller guarantees the capacity; if the set
 * fills up anyway the lost occurrences are reported as PG_ENOMEM by pg_distinct / pg_finalize / pg_export. */
int pg_set_autogrow(pg_ctx *ctx, int on);

/* Pass 1 over one batch of packed reads already resident in HBM (chopKmer4read + put_kmerset).
 *   d_packed     packed reads (pg_pack_read layout), padded with >= 4 readable words after the last read
 *   n_reads      reads in the batch (all have length >= K + 1)
 *   uniform_len  if non-zero every read has this length, read r starts at word r * pg_packed_words(len) and
 *                its first k-mer has ordinal ord_base + r * (len - K + 1); d_word_off / d_kmer_base unused
 *   d_word_off   [n_reads]      start word of each read in d_packed          (ragged batches)
 *   d_kmer_base  [n_reads + 1]  exclusive prefix sum of (len - K + 1), so read r has
 *                               d_kmer_base[r+1] - d_kmer_base[r] k-mers      (ragged batches)
 *   ord_base     ordinal of the batch's first k-mer
 * The set grows first if this batch could push the load past 70 % (synchronises the stream in that case). */
int pg_count_reads(pg_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_word_off,
                   const uint64_t *d_kmer_base, uint64_t n_reads, uint32_t uniform_len,
                   uint64_t n_kmers, uint64_t ord_base, void *stream);

/* Ragged batches (uniform_len = 0) of any mix of lengths are cut by the same tiled kernel as uniform ones
 * (prlHashReads.c:163-259,642-648: the reference chops every read of >= K + 1 bases the same way, lengths from lenBuffer); the
 * tiles are sized for the longest read.  max_len = an upper bound on the read lengths of the ragged batches to come (the
 * reference's maxReadLen); 0 = none known: every ragged batch then costs a device reduction and one host wait.  A read
 * longer than the bound fails the pass (reported by pg_finalize). */
int pg_set_read_len_bound(pg_ctx *ctx, uint32_t max_len);

/* Multi-GPU path, step 1: extract the batch's k-mer occurrences as routed records instead of inserting.
 * Record = (nw + 1) uint64_t: key words, then meta = ordinal << 6 | left << 3 | right (left/right = base code
 * or 4 for none).  Records are written grouped by owner = set id % n_owners into d_out at the offsets
 * d_owner_off[n_owners + 1] computed by pg_route_count (exclusive prefix sum of its counts). */
int pg_route_count(pg_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_word_off, const uint64_t *d_kmer_base,
                   uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, int n_owners,
                   uint64_t *d_counts /* [n_owners], zeroed by the call */, void *stream);
int pg_route_scatter(pg_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_word_off, const uint64_t *d_kmer_base,
                     uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, uint64_t ord_base, int n_owners,
                     const uint64_t *d_owner_off, uint64_t *d_cursor /* [n_owners] scratch */, uint64_t *d_out,
                     void *stream);
/* Multi-GPU path, step 2 (after the all-to-all): insert routed records into this rank's set. */
int pg_count_records(pg_ctx *ctx, const uint64_t *d_records, uint64_t n_records, void *stream);

/* Multi-GPU path of the partition engine (engine 2): what travels between GPUs is super-k-mer records (about one per
 * 20 k-mers) instead of k-mer records.  owner(partition) = partition mod n_owners, so a rank owns whole partitions and
 * counts them locally afterwards (pg_finalize); no other exchange is needed.
 *   pg_skm_route   cut a uniform-length batch and write its records grouped by owner: owner o's records at
 *                  d_send_records + o * capacity_per_owner * W words (W = pg_stats out[3] / 8) and their partition ids at
 *                  d_send_parts + o * capacity_per_owner; d_counts[o] (zeroed by the call) = records for owner o.
 *                  An owner region that overflows is reported as PG_ENOMEM by the next pg_finalize.
 *   pg_skm_ingest  append records received from other ranks (and the rank's own share) to the local partition streams. */
int pg_skm_route(pg_ctx *ctx, const uint64_t *d_packed, uint64_t n_reads, uint32_t uniform_len, uint64_t ord_base, int n_owners,
                 uint64_t *d_send_records, uint32_t *d_send_parts, uint64_t capacity_per_owner, uint64_t *d_counts, void *stream);
int pg_skm_ingest(pg_ctx *ctx, const uint64_t *d_records, const uint32_t *d_parts, uint64_t n_records, void *stream);

/* Number of distinct k-mers stored so far (synchronises the stream). */
int pg_distinct(pg_ctx *ctx, uint64_t *out, void *stream);
/* Current capacity (slots) and bytes per slot of the device set. */
int pg_table_info(pg_ctx *ctx, uint64_t *slots, uint32_t *slot_bytes);

/* Tell the context how many k-mer occurrences the whole input will bring (an estimate is fine).  The partition engine
 * sizes its partition count from it -- about 8 k occurrences a partition, so that one partition is counted in one LDS
 * pass -- instead of from log2_slots, which sizes the export array for the DISTINCT k-mers.  Before the first batch. */
int pg_expect_kmers(pg_ctx *ctx, uint64_t total_kmers);
/* The same with the other things a caller may know (round 6; pg_expect_kmers(ctx, t) = pg_expect(ctx, t, 0, 0, 1)):
 *   total_reads    the job's reads (0: unknown): every read makes at least one super-k-mer record, which is most of the records
 *                  where a read has few k-mers (K = 127 from 150-base reads); sizes the record pool;
 *   distinct_here  distinct k-mers this context's export array should hold (0: what log2_slots says, 0.7 x 2^log2_slots); too
 *                  few is not an error: pg_finalize then counts the partitions again into an array of the true size;
 *   n_owners       ranks that share the job's partition ids (pg_count_reads_sharded over a communicator of that size): the
 *                  context then stores only the partitions it owns -- id mod n_owners == its rank, at id / n_owners -- and its
 *                  cursors, chunk table and record pool are sized for 1 / n_owners of the job.  What the reference does with
 *                  shared memory (prlHashReads.c:79-90: worker t keeps hash % thrd_num == t) as per-rank storage.  Such a
 *                  context takes its batches through pg_count_reads_sharded only.
 * total_kmers is the WHOLE job's count (all ranks).  Before the first batch. */
int pg_expect(pg_ctx *ctx, uint64_t total_kmers, uint64_t total_reads, uint64_t distinct_here, int n_owners);
/* pg_create_engine with that estimate known up front (no second allocation); expected_kmers = 0: unknown. */
pg_ctx *pg_create_sized(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers);
/* ... and with pg_expect's other figures (expected_reads = 0, distinct_here = 0, n_owners = 1: pg_create_sized). */
pg_ctx *pg_create_planned(int device, int K, int mer127, int n_sets, int log2_slots, int engine, uint64_t expected_kmers,
                          uint64_t expected_reads, uint64_t distinct_here, int n_owners);

/* Counters for reporting (synchronise first): out[0] engine, out[1] distinct k-mers, out[2] super-k-mer records,
 * out[3] bytes per record (engine 2) / per slot (engine 1), out[4] pool chunks handed out, out[5] pool chunks,
 * out[6] partitions (engine 2) / slots (engine 1), out[7] export capacity. */
int pg_stats(pg_ctx *ctx, uint64_t out[8]);

/* After the last batch: apply the -d filter (thread_delow), mark linear nodes and build the coverage
 * histogram (thread_mark, freqStat) in one scan over the set.  hist_out[256] (host) receives the histogram;
 * set_last_put_out[n_sets] (host, may be NULL) receives the per-set last-put values.  Synchronises. */
int pg_finalize(pg_ctx *ctx, int delow, uint64_t hist_out[256], uint64_t *set_last_put_out, void *stream);

/* Compact the stored k-mers into records (layout above) in device memory, d_records holding room for
 * pg_distinct() records; order is unspecified.  *n_out (host) receives the count.  Synchronises. */
int pg_export(pg_ctx *ctx, uint64_t *d_records, uint64_t capacity, uint64_t *n_out, void *stream);

/* Partition engine, after pg_finalize(..., set_last_put_out = NULL, ...): the number of distinct k-mers per reference set
 * (out[256]), and the per-set last put computed on demand.  The layout replay needs the last put only when a set's count sits
 * exactly at a growth threshold of the reference's size schedule (pg_host_last_put_matters on the counts, summed over the
 * ranks in a multi-GPU run); otherwise that second expansion of every record is skipped and zeros are handed over. */
int pg_set_counts(pg_ctx *ctx, uint64_t out[256], void *stream);
int pg_last_put(pg_ctx *ctx, uint64_t *set_last_put_out, void *stream);
/* 1 when, for some set, a duplicate put arriving after its last new key would grow the reference's table (newhash.c:477). */
int pg_host_last_put_matters(const uint64_t *set_counts, int n_sets, int a_gb, int mer127);

/* The device memory one rank of a `pregraph` command takes, stage by stage, computed on the host from the sizing functions the
 * command and the partition engine allocate by (csrc/cmd_plan.hpp, e2_plan.hpp, ref_sizes.hpp) -- no GPU is touched.  Replaces no
 * reference function: the reference sizes its host tables from -a (prlHashReads.c:369-390) and grows them otherwise; this is the
 * check that a configuration fits a 288 GB GPU before a terabyte of reads has been parsed.
 *   reads_total, read_len   the whole job's reads;  fastq_bytes  the input files' size (0: reads x (2 read_len + 16))
 *   distinct_total          distinct k-mers of the job (an estimate: genome + about 35 error k-mers an erroneous base at K = 63)
 *   n_sets, a_gb            the reference's -p and -a;  n_ranks  GPUs (one rank each);  device_bytes  a GPU's memory
 * out[24] (bytes unless said): 0 peak, 1 its stage (1 pass 1 + count, 2 hand-over, 3 layout, 4 graph + pass 2), 2 cursors + chunk
 * table, 3 record pool, 4 export array as allocated, 5 ... after the count, 6 reads kept on the device, 7 batch buffers + exchange
 * regions, 8 k-mer sets, 9 layout arrays, 10 the sort's copy outside the pool, 11 log2 partition ids of the job, 12 log2
 * partitions a rank stores, 13 chunks a partition at computed addresses, 14 records the pool holds, 15 fits (peak <= 0.97 x
 * device_bytes), 16 - 19 the four stages, 20 k-mer occurrences estimated, 21 export records allocated for, 22 slots a set,
 * 23 log2_slots (+ 2^32 when the export array is made for fewer k-mers than distinct_total says: one more counting pass).  PG_OK, PG_EINVAL, or PG_ENOMEM when the partition engine's own checks refuse the size. */
int pg_host_plan_memory(uint64_t reads_total, uint32_t read_len, uint64_t fastq_bytes, uint64_t distinct_total, int K, int mer127,
                        int n_sets, int a_gb, int n_ranks, uint64_t device_bytes, uint64_t out[24]);

/* Partition engine: the export array itself instead of a copy of it (the caller owns *d_records_out and frees it with
 * pg_device_free -- the block comes out of the library's device arena, csrc/arena.hpp, not out of hipMalloc); everything else the context holds on the device is released, the context can only be destroyed afterwards. */
int pg_export_take(pg_ctx *ctx, uint64_t **d_records_out, uint64_t *n_out);
/* The same, with the record pool handed over too (*d_workspace_out, pg_device_free it when done): scratch memory for
 * pg_sort_records_ws, so that the hand-over needs no large allocation (hipMalloc right after a big hipFree takes seconds). */
int pg_export_take_ws(pg_ctx *ctx, uint64_t **d_records_out, uint64_t *n_out, void **d_workspace_out, uint64_t *workspace_bytes_out);
void pg_device_free(void *d_ptr);   /* gives a block the library handed over back to its device arena (a pointer from hipMalloc: hipFree) */
/* Everything the library allocates on a GPU is cut from one arena per device: a reserved virtual range whose physical memory is created as its
 * high-water mark grows and goes back to the driver only when the arena is empty and unpinned -- the driver clears released memory, and a
 * hipMalloc behind a large hipFree waits for that.  call_pregraph pins its devices for the command; a library caller that creates and destroys
 * contexts in a loop may pin a device around the loop.  SOAPDENOVO2_AMD_ARENA=0: plain hipMalloc / hipFree. */
void pg_device_arena_pin(int device);
void pg_device_arena_unpin(int device);
/* out[0..7] = active (1 = arena, 0 = plain hipMalloc), bytes reserved, mapped, in use, peak in use, blocks cut, physical pieces created,
 * microseconds spent creating + mapping them */
void pg_device_arena_stats(int device, uint64_t out[8]);
/* TEST HOOK (no GPU): the arena's block list on a random sequence of n_ops cuts and returns over `size` bytes with its invariants checked after
 * every step (csrc/host_emu.cpp); 0 = all held.  out[0..3] = cuts granted, cuts refused, peak bytes in use, most holes at once. */
long long pg_host_emu_arena_blocks(uint64_t seed, uint64_t n_ops, uint64_t size, uint64_t max_block, uint64_t out[4]);
/* A read-only look at the partition engine's export array after pg_finalize (no copy; valid until the next pg_reset,
 * pg_export_take or pg_destroy). */
int pg_export_peek(pg_ctx *ctx, const uint64_t **d_records_out, uint64_t *n_out);
/* Order-independent digest of a device array of records (rec_words = nw + 2 words each): out[0 .. rec_words) the column
 * sums mod 2^64, out[6] the sum of the coverage fields (bits 31:24 of cnt: min(puts, 255), newhash.c:95-103), out[7] the
 * number of saturated ones.  Equal digests from two passes over the same reads -- whatever the batching, the engine, the
 * number of GPUs -- and out[6] == the k-mer occurrences that went in when out[7] == 0: the conservation check bench.py
 * prints with its timed result.  Synchronises. */
int pg_records_checksum(const uint64_t *d_records, uint64_t n_records, int rec_words, uint64_t out[8], void *stream);

/* Order exported records (device memory, on the current device) by their last word, i.e. by k-mer set and then by
 * first-occurrence ordinal: the order in which the layout replay of pg_host_build_graph / pg_host_graph_begin inserts
 * them (put_kmerset order, newhash.c:473-528).  Records handed over in this order are inserted as they lie; otherwise
 * the host buckets and sorts them itself.  Any n_records that fits: rocPRIM's radix sort with 64-bit sizes, the
 * permutation kept as 64-bit indices beyond 2^32 - 1 records (PG_SORT_WIDE=1 forces that flavour); needs about 2.5x the
 * records' bytes of free device memory.  Synchronises. */
int pg_sort_records(uint64_t *d_records, uint64_t n_records, int mer127, void *stream);
/* The same on caller-provided scratch memory (used when it holds the keys, indices and the permuted copy: about 2.5x the
 * records' bytes; otherwise the call allocates what is missing). */
int pg_sort_records_ws(uint64_t *d_records, uint64_t n_records, int mer127, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * 4. Multi-GPU pass 1.  The reference hands every k-mer to the set it hashes to through shared memory
 *    (each of its `thrd_num` workers scans the whole buffer for `hashBanBuffer[i] % thrd_num == id`,
 *    standardPregraph/prlHashReads.c:79-90).  Across GPUs the unit that travels is the super-k-mer record and
 *    owner(record) = minimizer partition mod n_ranks; a rank counts the partitions it owns with no further exchange.
 *    A communicator has one rank per GPU: one per process (PG_COMM_RCCL: the 128-byte id made by rank 0 travels out of
 *    band, e.g. through the torch.distributed store) or several in one process, one host thread each
 *    (pg_comm_create_local; PG_COMM_RCCL when every rank has its own device, PG_COMM_P2P -- peer copies between the
 *    ranks' buffers -- when ranks share a device or on request).  Every call below is collective: all ranks make it,
 *    in the same order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct pg_comm pg_comm;
#define PG_COMM_RCCL 0   /* ncclSend / ncclRecv in one group: a direct all-to-all over xGMI */
#define PG_COMM_P2P 1    /* one process: barrier + peer-to-peer copies */
#define PG_COMM_HOST 2   /* the caller's all-to-all over host buffers (pg_comm_create_host): ranks in several processes without RCCL
                            between them, e.g. sharing one GPU under torch.distributed / gloo; staged through the host, for tests */
int pg_comm_unique_id(uint8_t id[128]);
pg_comm *pg_comm_create(int n_ranks, int rank, int device, const uint8_t id[128]);
/* transport: PG_COMM_RCCL, PG_COMM_P2P, or -1 = RCCL when the devices are pairwise distinct (SOAPDENOVO2_AMD_EXCHANGE=p2p|rccl
 * overrides), else P2P.  out[n_ranks] receives the ranks' communicators. */
int pg_comm_create_local(int n_ranks, const int *devices, int transport, pg_comm **out);
/* COLLECTIVE for the ranks of one process (PG_COMM_P2P): the peers pull out of THIS rank's send regions on their own streams, and a rank's
 * destroy (or pg_comm_flush) waits for its own streams only -- destroy the communicators of a group after every rank has finalized or
 * flushed (call_pregraph and api.LocalGroup do), never one rank's while another rank is still counting. */
void pg_comm_destroy(pg_comm *comm);
int pg_comm_rank(const pg_comm *comm);
int pg_comm_size(const pg_comm *comm);
int pg_comm_transport(const pg_comm *comm);
/* out[0] rounds, out[1] records sent, out[2] records received, out[3] records per owner region */
int pg_comm_stats(const pg_comm *comm, uint64_t out[4]);
/* the pipeline of pg_count_reads_sharded: out[0] device microseconds spent in the record exchanges (events on the exchange stream),
 * [1] bytes sent to other ranks, [2] host waits (one a round), [3] cuts repeated with larger owner regions, [4] rounds, [5] records per
 * owner region */
int pg_comm_pipeline_stats(const pg_comm *comm, uint64_t out[8]);
/* A variable all-to-all over HOST buffers brought by the caller: what goes to rank p lies at send + send_off[p] (send_cnt[p] bytes), what
 * comes from rank q lands at recv + recv_off[q] (recv_cnt[q] bytes); arrays of n_ranks entries; returns 0 when done.  Collective. */
typedef int (*pg_host_alltoallv_fn)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_cnt, void *recv,
                                    const uint64_t *recv_off, const uint64_t *recv_cnt);
pg_comm *pg_comm_create_host(int n_ranks, int rank, int device, pg_host_alltoallv_fn fn, void *user);
/* Appends what pg_count_reads_sharded still has in flight (its last round travels when the call returns) to the partition streams of
 * ctx and waits for it.  pg_finalize / pg_reset / pg_destroy do this by themselves; callers that look at the streams otherwise use it.
 * Like pg_comm_destroy it waits for this rank's streams: with PG_COMM_P2P every rank of the group flushes before any send region is reused
 * or released.  (A rank whose pipeline could not be set up -- stream / event creation failed -- still takes part in the round's collective
 * steps with nothing to give and raises the error flag of the round: all ranks return the error together.) */
int pg_comm_flush(pg_ctx *ctx, pg_comm *comm, void *stream);

/* d_send_counts[o] goes to rank o, d_recv_counts[q] comes from rank q (device memory, n_ranks words each). */
int pg_exchange_counts(pg_comm *comm, const uint64_t *d_send_counts, uint64_t *d_recv_counts, void *stream);
/* Variable all-to-all of the records pg_skm_route wrote (owner o's at d_send_records + o * capacity_per_owner * rec_words,
 * ids at d_send_parts + o * capacity_per_owner).  send_counts / recv_counts are host arrays (what pg_exchange_counts moved);
 * what rank q sent lands behind what ranks < q sent. */
int pg_exchange_records(pg_comm *comm, const uint64_t *d_send_records, const uint32_t *d_send_parts, uint64_t capacity_per_owner,
                        int rec_words, const uint64_t *send_counts, const uint64_t *recv_counts, uint64_t *d_recv_records,
                        uint32_t *d_recv_parts, void *stream);
/* In-place sum over the ranks (the 256-bin coverage histogram behind .kmerFreq, prlHashReads.c:1104-1132). */
int pg_exchange_allreduce_u64(pg_comm *comm, uint64_t *d_buf, uint64_t n, void *stream);
/* Every rank's exported records on rank `root`, rank after rank (d_out / capacity / n_out matter on the root only). */
int pg_exchange_gather_records(pg_comm *comm, const uint64_t *d_records, uint64_t n_local, int rec_words, int root, uint64_t *d_out,
                               uint64_t capacity, uint64_t *n_out, void *stream);
/* After pass 1 (pg_finalize + pg_export_take on every rank): the distinct k-mers move once more, to owner(set s) = s mod
 * n_ranks, so that every k-mer set of the reference (KmerSets[s], s = hash_kmer % thrd_num, prlHashReads.c:79-90) lies whole
 * on one GPU for the layout replay and the set-by-set scans.  Takes ownership of d_records (freed); the
 * regrouped records come back in a fresh device allocation *d_out (pg_device_free).  Collective; a rank that fails makes
 * every rank return an error instead of leaving the others waiting. */
int pg_exchange_regroup_by_set(pg_comm *comm, uint64_t *d_records, uint64_t n_local, int rec_words, uint64_t **d_out, uint64_t *n_out,
                               void *stream);
/* the same inside a workspace (the record pool pass 1 is done with, pg_export_take_ws): the send buffer is its front and, when both
 * fit, the regrouped records its tail (*out_in_workspace = 1: they are freed with the workspace); d_records stays the caller's.
 * Nothing of that size is allocated or released, which costs a fresh process seconds. */
int pg_exchange_regroup_by_set_ws(pg_comm *comm, uint64_t *d_records, uint64_t n_local, int rec_words, void *d_workspace,
                                  uint64_t workspace_bytes, uint64_t **d_out, uint64_t *n_out, int *out_in_workspace, void *stream);
/* out[0] = distinct k-mers this rank held before the regroup, out[1] = after it */
int pg_comm_regroup_stats(const pg_comm *comm, uint64_t out[2]);
/* One batch of pass 1 on all ranks: pg_skm_route (ragged batches too: d_word_off / d_kmer_base as in pg_count_reads), the count
 * and record exchange, pg_skm_ingest.  A rank with nothing to contribute in a round calls it with n_reads = 0.  The send and
 * receive regions live in the communicator and grow as needed; a batch that sends one owner more than its region holds (one
 * minimizer dominating the batch) is cut again with larger regions, all ranks alike.  A rank that fails still finishes the
 * round's collectives, and every rank returns an error.  After the last round: pg_finalize on every rank. */
int pg_count_reads_sharded(pg_ctx *ctx, pg_comm *comm, const uint64_t *d_packed, const uint64_t *d_word_off, const uint64_t *d_kmer_base,
                           uint64_t n_reads, uint32_t uniform_len, uint64_t n_kmers, uint64_t ord_base, void *stream);

/* ------------------------------------------------------------------------------------------------
 * 5. Test hooks.  The device graph stages (csrc/dev_graph.hpp, csrc/dev_tips.hpp) are written once over a backend; these
 *    entry points run the SAME function objects on host threads instead of HIP lanes, so that the CPU-only tests can
 *    compare them with the sequential host stages.  Not a fallback: nothing in the product path calls them.
 * ------------------------------------------------------------------------------------------------ */
/* layout of static (-a) k-mer sets, SURVEY.md App. C "K6" (put_kmerset into a table that never grows, newhash.c:353-366,
 * 473-528): records sorted by (set, ordinal), per_set_count[s] of them per set; nodes_out = n_sets * set_size slots of
 * nw + 1 words (key words, cnt), empty slots with all-ones in their first word.  Returns PG_OK, 1 = unsuited (a set that
 * fills its pool or holds >= 2^32 keys: the caller replays on the host), or PG_E*. */
int pg_host_emu_layout_static(const uint64_t *records, const uint64_t *per_set_count, int n_sets, uint64_t set_size, int mer127,
                              int n_threads, uint64_t *nodes_out);

/* layout of growable (-a 0) k-mer sets as the device computes it (csrc/dev_rehash.hpp: encap_kmerset's in-place rehash as a fixed
 * point over insertion times, newhash.c:340-528): records sorted by (set, ordinal); out_slot[i] = slot of record i in its set,
 * out_set_size[s] = final size, out_rounds[s] = fixed-point rounds; out_nodes (optional, nodes_cap_slots slots of 3 / 5 words): the
 * sets' image back to back, word 0 of an empty slot all ones.  To be compared with pg_host_replay_layout. */
int pg_host_emu_layout_growable(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put, int mer127, int n_sets,
                                int n_threads, uint64_t *out_slot, uint64_t *out_set_size, uint64_t *out_rounds, uint64_t *out_nodes,
                                uint64_t nodes_cap_slots);
/* tip clipping (removeSingleTips / removeMinorTips, cutTipPreGraph.c:363-488) as the device decides it (csrc/dev_tips.hpp: a
 * fixed point over start decisions, one lane per stop node) next to the sequential host scan, on two copies of the layout
 * replayed from `records`: out[0], out[1] = tips the sequential scan removed (single, minor); out[2], out[3] = the same from
 * the emulated device stage; out[4] = nodes whose counter words differ afterwards (must be 0); out[5] = fixed-point rounds;
 * out[6] = minor cycles. */
int pg_host_emu_clip_tips(const uint64_t *records, uint64_t n_records, const uint64_t *set_last_put, int K, int mer127, int n_sets,
                          int cut_single, int a_gb, int n_threads, uint64_t out[8]);

/* key mod size (newhash.c:36-57: the exact 128-bit modulus of the 63-mer build, the 32-bit chunks folded in 64-bit arithmetic of
 * the 127-mer build) as the graph stages' lookups and both device layouts compute it: by a precomputed reciprocal of the size
 * instead of the compiler's 64-bit `%` (csrc/graph_lookup.hpp: ModConst, rem128).  keys = n x (mer127 ? 4 : 2) words. */
int pg_host_emu_home_slots(const uint64_t *keys, uint64_t n, int mer127, uint64_t size, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* SOAPDENOVO2_AMD_H */
