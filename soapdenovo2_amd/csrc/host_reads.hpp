// host_reads.hpp -- library config + read ingestion for the pregraph stage (host side).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace pg {

// one [LIB] block of the config file (scan_libInfo, standardPregraph/lib.c:130-506)
struct LibInfo {
    int avg_ins = 0, min_ins = 0, max_ins = 0;
    int asm_flag = 3, rank = 0, pair_num_cut = 0, rd_len_cutoff = 0, map_len = 0, reverse = 0;
    std::vector<std::string> f1, f2, q1, q2, f, q, p, b;
};

struct LibConfig {
    int max_rd_len = 0;                // "max_rd_len" given before the first [LIB]; 0 = absent
    std::vector<LibInfo> libs;         // sorted by avg_ins (stable)
};

// Parses the config exactly as the reference tokenises it (splitColumn, lib.c:70-108).  Exits the process
// with the reference's messages on a malformed file (no [LIB], f1/f2 count mismatch, ...).
LibConfig parse_lib_config(const char* path);

// One input stream in the order the reference visits them (nextValidIndex, readseq1by1.c:595-674, with
// pairs = 0 and asm_ctg = 1): libs by avg_ins; per lib f1/f2 pairs, q1/q2 pairs, p, b, f, q.
struct InputFile {
    int lib;
    int type;            // 1 f1/f2, 2 q1/q2, 3 p, 4 b, 5 f, 6 q  (lib_array[].curr_type)
    std::string path1, path2;
    int max_read_len;    // min(rd_len_cutoff, max_rd_len) or max_rd_len (openNextFile, prlHashReads.c:921-928)
    int reverse;
    int asm_flag = 3;    // BAM only: 1 = QC-fail records (flag 0x200) are skipped one by one, else pairs with one are taken back
    int keep_len = 0;    // BAM only: the length from which the caller keeps a read (K + 1, prlHashReads.c:437): a pair that is
                         // taken back takes the last KEPT read with it (prlHashReads.c:414-426), so the reader has to know
};
std::vector<InputFile> input_order(const LibConfig& cfg, int max_read_len_all);

// Sink for accepted reads: base codes 0..3, one byte per base.
struct ReadSink {
    virtual void on_read(const uint8_t* codes, int len) = 0;
    // A run of reads already packed 2 bits a base (32 bases a word, first base in the top bits, every read starting
    // on a word: the layout of pg_pack_read), handed over by the multi-threaded reader.  min_len / max_len bound the
    // read lengths of the run.  The default unpacks and calls on_read.
    virtual void on_packed(const uint64_t* words, const int32_t* lens, size_t n, int min_len, int max_len);
    virtual ~ReadSink() {}
};

// host threads for the parallel stages: `n_threads` if positive, else SOAPDENOVO2_AMD_HOST_THREADS, else the hardware
// threads capped by the container's CPU quota
int host_threads(int n_threads);

// Streams every read of `in` to `sink` in the reference's order, reproducing its 32 KiB chunking
// (AIORead, prlHashReads.c:771-901) and record parsers (readseqfq / readseqInBuf, readseq1by1.c:138-360).
// Returns the number of records parsed ("read(s) processed" in the reference's log counts these).
// Reads are delivered untruncated by K; the caller drops reads shorter than K + 1 (prlHashReads.c:642).
long long stream_reads(const InputFile& in, ReadSink& sink);
// the BAM reader's pairing state (the reference's static `state`, readseq1by1.c:44); set = true stores `value` first
int bam_pair_state(bool set, int value);

}  // namespace pg
