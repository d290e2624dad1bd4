// host_reads.cpp -- library config parser and read ingestion (plain C++, no HIP).
//
// Replaces, for the pregraph stage: scan_libInfo (standardPregraph/lib.c:130-506), nextValidIndex /
// openFileInLib (readseq1by1.c:595-786), AIORead (prlHashReads.c:771-901), readseqfq / readseqInBuf /
// readseqInLib (readseq1by1.c:138-360, 927-1035).  The point of following the reference's chunking is that
// the SET and ORDER of reads it hands to pass 1 defines the first-occurrence order of every k-mer and with
// it the bytes of .vertex / .edge.gz, including its corner cases (reads cut at max_rd_len, records that
// straddle a 32 KiB chunk, a file whose size is a multiple of 32768).
#include "host_reads.hpp"
#include "env.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace pg {

// ---- config ----------------------------------------------------------------------------------------------
// splitColumn (lib.c:70-108): the first two maximal runs of printable characters (32..126) other than '='
static bool split_column(const char* line, std::string tok[2]) {
    const int len = (int)strlen(line);
    int i = 0, n = 0;
    auto ok = [](char c) { return c >= 32 && c <= 126 && c != '='; };
    while (i < len) {
        if (ok(line[i])) {
            tok[n].clear();
            while (i < len && ok(line[i])) tok[n].push_back(line[i++]);
            if (++n == 2) return true;
        }
        i++;
    }
    return false;
}

LibConfig parse_lib_config(const char* path) {
    FILE* fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "Cannot open %s. Now exit to system...\n", path); exit(-1); }
    LibConfig cfg;
    char line[1024];
    std::string tok[2];
    int cur = -1;
    std::vector<bool> pe;
    while (fgets(line, sizeof line, fp)) {
        if (strncmp(line, "[LIB]", 5) == 0) {
            cfg.libs.emplace_back();
            pe.push_back(false);
            cur++;
            continue;
        }
        if (!split_column(line, tok)) continue;
        if (cur < 0) {                                     // only max_rd_len is looked at before the first [LIB]
            if (tok[0] == "max_rd_len") cfg.max_rd_len = atoi(tok[1].c_str());
            continue;
        }
        LibInfo& L = cfg.libs[cur];
        const std::string& k = tok[0];
        const std::string& v = tok[1];
        if (k == "f1") { L.f1.push_back(v); pe[cur] = true; }
        else if (k == "q1") { L.q1.push_back(v); pe[cur] = true; }
        else if (k == "f2") { L.f2.push_back(v); pe[cur] = true; }
        else if (k == "q2") { L.q2.push_back(v); pe[cur] = true; }
        else if (k == "f") L.f.push_back(v);
        else if (k == "q") L.q.push_back(v);
        else if (k == "p") { L.p.push_back(v); pe[cur] = true; }
        else if (k == "b") { L.b.push_back(v); pe[cur] = true; }
        else if (k == "min_ins") L.min_ins = atoi(v.c_str());
        else if (k == "max_ins") L.max_ins = atoi(v.c_str());
        else if (k == "avg_ins") L.avg_ins = atoi(v.c_str());
        else if (k == "rd_len_cutoff") L.rd_len_cutoff = atoi(v.c_str());
        else if (k == "reverse_seq") L.reverse = atoi(v.c_str());
        else if (k == "asm_flags") L.asm_flag = atoi(v.c_str());
        else if (k == "rank") L.rank = atoi(v.c_str());
        else if (k == "pair_num_cutoff") L.pair_num_cut = atoi(v.c_str());
        else if (k == "map_len") L.map_len = atoi(v.c_str());
    }
    fclose(fp);
    if (cfg.libs.empty()) { fprintf(stderr, "Config file error: no [LIB] in file\n"); exit(-1); }
    for (size_t i = 0; i < cfg.libs.size(); i++) {
        const LibInfo& L = cfg.libs[i];
        if (L.f1.size() != L.f2.size()) { fprintf(stderr, "Config file error: the number of mark \"f1\" is not the same as \"f2\"!\n"); exit(-1); }
        if (L.q1.size() != L.q2.size()) { fprintf(stderr, "Config file error: the number of mark \"q1\" is not the same as \"q2\"!\n"); exit(-1); }
        for (size_t j = 0; j < L.f1.size(); j++)
            if (L.f1[j] == L.f2[j]) { fprintf(stderr, "Config file error: f2 file is the same as f1 file\nf1=%s\nf2=%s\n", L.f1[j].c_str(), L.f2[j].c_str()); exit(-1); }
        for (size_t j = 0; j < L.q1.size(); j++)
            if (L.q1[j] == L.q2[j]) { fprintf(stderr, "Config file error: q2 file is the same as q1 file\nq1=%s\nq2=%s\n", L.q1[j].c_str(), L.q2[j].c_str()); exit(-1); }
        if (pe[i] && L.avg_ins == 0) { fprintf(stderr, "Config file error: PE reads need avg_ins in [LIB] %d\n", (int)i + 1); exit(-1); }
    }
    std::stable_sort(cfg.libs.begin(), cfg.libs.end(), [](const LibInfo& a, const LibInfo& b) { return a.avg_ins < b.avg_ins; });
    return cfg;
}

std::vector<InputFile> input_order(const LibConfig& cfg, int max_all) {
    std::vector<InputFile> out;
    for (size_t i = 0; i < cfg.libs.size(); i++) {
        const LibInfo& L = cfg.libs[i];
        if (L.asm_flag != 1 && L.asm_flag != 3) continue;              // asm_ctg == 1 (prlHashReads.c:308,406)
        const int mrl = L.rd_len_cutoff > 0 ? std::min(L.rd_len_cutoff, max_all) : max_all;
        auto add = [&](int type, const std::string& a, const std::string& b) {
            out.push_back(InputFile{(int)i, type, a, b, mrl, L.reverse, L.asm_flag, 0});
        };
        for (size_t j = 0; j < L.f1.size(); j++) add(1, L.f1[j], L.f2[j]);
        for (size_t j = 0; j < L.q1.size(); j++) add(2, L.q1[j], L.q2[j]);
        for (size_t j = 0; j < L.p.size(); j++) add(3, L.p[j], "");
        for (size_t j = 0; j < L.b.size(); j++) add(4, L.b[j], "");
        for (size_t j = 0; j < L.f.size(); j++) add(5, L.f[j], "");
        for (size_t j = 0; j < L.q.size(); j++) add(6, L.q[j], "");
    }
    return out;
}

// ---- chunked file access -----------------------------------------------------------------------------------
namespace {

constexpr size_t CHUNK = 32768;                          // maxAIOSize, prlHashReads.c:345

struct Source {
    FILE* fp = nullptr;
    bool piped = false;
    void open(std::string name) {                          // openFile4read, readseq1by1.c:676-714
        while (!name.empty() && name.back() == ' ') name.pop_back();
        if (name.size() > 3 && name.compare(name.size() - 3, 3, ".gz") == 0) {
            fp = popen(("gzip -dc " + name).c_str(), "r");
            piped = true;
        } else fp = fopen(name.c_str(), "r");
        if (!fp) { fprintf(stderr, "Cannot open %s. Now exit to system...\n", name.c_str()); exit(-1); }
        // a FIFO / process substitution cannot be read with pread(): take it as it comes, like a pipe
        struct stat st;
        if (!piped && fstat(fileno(fp), &st) == 0 && !S_ISREG(st.st_mode)) unseekable = true;
    }
    bool unseekable = false;
    bool sequential() const { return piped || unseekable; }
    void close() { if (fp) { if (piped) pclose(fp); else fclose(fp); fp = nullptr; } }
};

// Where to cut a full 32 KiB FASTQ chunk so the buffer ends on a record boundary: the last "\n@", moved one
// record back when that '@' opens a quality line (the heuristic of prlHashReads.c:820-855, restated).
size_t fastq_cut(const char* t, size_t get) {
    long i = (long)get - 1;
    int newlines = 0;
    while (i > 0 && !(t[i] == '@' && t[i - 1] == '\n')) { if (t[i] == '\n') newlines++; i--; }
    if (i <= 0) return 0;
    if (newlines <= 1) {
        long i2 = i - 2;
        while (i2 >= 0 && t[i2] != '\n') i2--;                     // start of the line before the '@' line
        if (i2 >= 0 && t[i2 + 1] == '+') {                         // ... it is a '+' line: the '@' line is a quality string
            i2--;
            while (i2 >= 0 && t[i2] != '\n') i2--;
            if (i2 >= 0 && t[i2 + 1] != '+') {
                long k = i2 - 1;
                while (k > 0 && !(t[k] == '@' && t[k - 1] == '\n')) k--;
                i = k > 0 ? k : 0;
            }
        }
    }
    return (size_t)i;
}
size_t fasta_cut(const char* t, size_t get) {                    // prlHashReads.c:856-860
    long i = (long)get - 1;
    while (i > 0 && t[i] != '>') i--;
    return (size_t)i;
}

// One file read the way AIORead hands it out: buffers that end on record boundaries.
struct ChunkStream {
    Source src;
    bool fastq;
    std::vector<char> chunk;
    std::string cache, buf;
    bool last = false;
    ChunkStream(const std::string& path, bool fq) : fastq(fq), chunk(CHUNK) { src.open(path); }
    ~ChunkStream() { src.close(); }
    // false = nothing more to parse; otherwise `buf` holds the next buffer
    bool next() {
        if (last) return false;
        const size_t get = fread(chunk.data(), 1, CHUNK, src.fp);
        if (get > 0) {
            if (get % CHUNK != 0) {
                buf = cache; buf.append(chunk.data(), get); cache.clear();
                last = true;
                return true;
            }
            const size_t cut = fastq ? fastq_cut(chunk.data(), get) : fasta_cut(chunk.data(), get);
            buf = cache; buf.append(chunk.data(), cut);
            cache.assign(chunk.data() + cut, get - cut);
            return true;
        }
        // The file size is a multiple of 32768: the reference re-parses its previous buffer and loses the
        // cached tail record (prlHashReads.c:873-877).
        fprintf(stderr, "Warning : aio_return zero, the size of input file must be N * 32768.\n");
        cache.clear();
        last = true;
        return true;
    }
};

inline int base_code(char c) { return (c & 0x06) >> 1; }            // base2int, inc/def.h:39

// letters -> codes, '.' -> A, everything else skipped (readseq1by1.c:322-339); one table lookup a character
struct BaseTable {
    uint8_t code[256];
    BaseTable() {
        for (int c = 0; c < 256; c++) code[c] = 0xFF;
        for (int c = 'A'; c <= 'Z'; c++) { code[c] = (uint8_t)base_code((char)c); code[c - 'A' + 'a'] = code[c]; }
        code[(unsigned char)'.'] = 0;
    }
};
static const BaseTable g_base_table;
inline int convert_line(const char* s, int n, int max_len, uint8_t* out) {
    if (n > max_len) n = max_len;
    int k = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t v = g_base_table.code[(unsigned char)s[i]];
        out[k] = v;
        k += v != 0xFF;
    }
    return k;
}

// readseqfq (readseq1by1.c:279-360): next record of a FASTQ buffer; returns the read length (0 = none)
static int parse_fastq_scan(const char* buf, size_t end, size_t& start, int max_len, uint8_t* out) {
    size_t p = 0;
    for (size_t m = start; m < end; m++) {
        const char c = buf[m];
        if (c == '@') p = m;
        else if (c == '\n' && buf[p] == '@') p = m;
        else if (c == '\n' && buf[p] == '\n' && m > p) {
            const size_t line_len = m - p - 1;
            // strlen() of the copied line: stops at an embedded NUL
            const size_t sl = strnlen(buf + p + 1, line_len);
            const int n = convert_line(buf + p + 1, (int)sl, max_len, out);
            size_t q = m + 1;
            while (q < end && buf[q] != '\n') q++;                   // the '+' line
            start = q + 2 + sl;                                      // skip the quality line: same length as the sequence line
            return n;
        }
    }
    return 0;
}
#if defined(__x86_64__)
#include <immintrin.h>
// The usual record once more, 32 characters at a time: the sequence line is converted while its end is looked for, and any
// character that is not a letter in front of that end (a digit, '@', NUL, '.', CR ...) sends the record to the scalar forms
// below (-1), which remain the definition.  Reads nothing behind `end`.
__attribute__((target("avx2"))) static int parse_fastq_avx2(const char* buf, size_t end, size_t& start, int max_len, uint8_t* out) {
    if (!(start < end && buf[start] == '@')) return -1;
    const char* e1 = (const char*)memchr(buf + start, '\n', end - start);           // end of the name line
    if (!e1) return -1;
    const char* s = e1 + 1;
    const char* lim = buf + end;
    const __m256i nl = _mm256_set1_epi8('\n'), c20 = _mm256_set1_epi8(0x20), ca = _mm256_set1_epi8((char)('a' - 128)),
                  c26 = _mm256_set1_epi8((char)(26 - 128)), c3 = _mm256_set1_epi8(3);
    size_t sl = 0;
    for (;;) {
        if (s + sl + 32 > lim) return -1;                                             // (the last record of a buffer: scalar)
        const __m256i v = _mm256_loadu_si256((const __m256i*)(s + sl));
        const uint32_t m_nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        // letter: ((v | 0x20) - 'a') < 26, compared as signed bytes after a shift by 128
        const __m256i x = _mm256_sub_epi8(_mm256_or_si256(v, c20), ca);
        const uint32_t m_ok = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi8(c26, x));
        const uint32_t before = m_nl ? ((m_nl & (0u - m_nl)) - 1u) : 0xFFFFFFFFu;     // the characters in front of the line's end
        if ((~m_ok) & before) return -1;
        if ((long)sl < (long)max_len) {
            // base2int: (c & 6) >> 1; 16-bit shift, the bit that crosses a byte is masked off
            const __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), c3);
            uint8_t tmp[32];
            _mm256_storeu_si256((__m256i*)tmp, code);
            const size_t room = (size_t)max_len - sl;
            memcpy(out + sl, tmp, room < 32 ? room : 32);                             // (`out` holds max_len + 8: whole blocks would overrun it)
        }
        if (m_nl) { sl += (size_t)__builtin_ctz(m_nl); break; }
        sl += 32;
    }
    const char* e2 = s + sl;                                                          // the sequence line's '\n'
    const int n = (int)(sl < (size_t)max_len ? sl : (size_t)max_len);
    // the '+' line ends at the next '\n' (usually "+\n"); the quality line is skipped by the sequence's length
    size_t qi;
    if (e2 + 1 < lim && e2[1] == '\n') qi = (size_t)(e2 + 1 - buf);
    else if (e2 + 2 < lim && e2[2] == '\n') qi = (size_t)(e2 + 2 - buf);
    else {
        const char* q = e2 + 1 < lim ? (const char*)memchr(e2 + 1, '\n', (size_t)(lim - (e2 + 1))) : nullptr;
        qi = q ? (size_t)(q - buf) : end;
    }
    start = qi + 2 + sl;
    return n;
}
#endif
#if defined(__x86_64__)
// The usual FASTA record (a '>' line, then the sequence on one line) the same way; -1 = not the usual record, the scalar
// form below decides.
__attribute__((target("avx2"))) static int parse_fasta_avx2(const char* buf, size_t end, size_t& start, int max_len, uint8_t* out) {
    if (!(start < end && buf[start] == '>')) return -1;
    const char* e1 = (const char*)memchr(buf + start, '\n', end - start);           // end of the name line
    if (!e1) return -1;
    const char* s = e1 + 1;
    const char* lim = buf + end;
    const __m256i nl = _mm256_set1_epi8('\n'), c20 = _mm256_set1_epi8(0x20), ca = _mm256_set1_epi8((char)('a' - 128)),
                  c26 = _mm256_set1_epi8((char)(26 - 128)), c3 = _mm256_set1_epi8(3);
    size_t sl = 0;
    for (;;) {
        if (s + sl + 32 > lim) return -1;
        const __m256i v = _mm256_loadu_si256((const __m256i*)(s + sl));
        const uint32_t m_nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        const __m256i x = _mm256_sub_epi8(_mm256_or_si256(v, c20), ca);
        const uint32_t m_ok = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi8(c26, x));
        const uint32_t before = m_nl ? ((m_nl & (0u - m_nl)) - 1u) : 0xFFFFFFFFu;
        if ((~m_ok) & before) return -1;                                              // ('>' in the sequence line starts a record over: scalar)
        if ((long)sl < (long)max_len) {
            const __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), c3);
            uint8_t tmp[32];
            _mm256_storeu_si256((__m256i*)tmp, code);
            const size_t room = (size_t)max_len - sl;
            memcpy(out + sl, tmp, room < 32 ? room : 32);
        }
        if (m_nl) { sl += (size_t)__builtin_ctz(m_nl); break; }
        sl += 32;
    }
    if (sl == 0) return -1;                                                           // (an empty sequence line: the scalar form reports it)
    start = (size_t)(s + sl - buf) + 1;
    return (int)(sl < (size_t)max_len ? sl : (size_t)max_len);
}
#endif
// SOAPDENOVO2_AMD_PARSE_SIMD=0 keeps the scalar forms (looked at on every call: the tests switch it inside one process)
static bool simd_parse_on() {
#if defined(__x86_64__)
    static const bool have = [] { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") != 0; }();
    if (!have) return false;
    const char* e = pg::env_test("SOAPDENOVO2_AMD_PARSE_SIMD");
    return !(e && e[0] == '0');
#else
    return false;
#endif
}

// The same scan for the usual record -- it starts on its '@' and its sequence line holds no '@' -- with the line ends
// found by memchr; anything else goes through the character-by-character scan above.
int parse_fastq(const char* buf, size_t end, size_t& start, int max_len, uint8_t* out, bool simd) {
#if defined(__x86_64__)
    if (simd) { const int r = parse_fastq_avx2(buf, end, start, max_len, out); if (r >= 0) return r; }
#endif
    if (start < end && buf[start] == '@') {
        const char* e1 = (const char*)memchr(buf + start, '\n', end - start);           // end of the name line
        if (!e1) return 0;
        const char* e2 = (const char*)memchr(e1 + 1, '\n', (size_t)(buf + end - (e1 + 1)));   // end of the sequence line
        if (e2 && !memchr(e1 + 1, '@', (size_t)(e2 - (e1 + 1)))) {
            const size_t line_len = (size_t)(e2 - e1 - 1);
            const size_t sl = strnlen(e1 + 1, line_len);
            const int n = convert_line(e1 + 1, (int)sl, max_len, out);
            const char* q = (const char*)memchr(e2 + 1, '\n', (size_t)(buf + end - (e2 + 1)));   // the '+' line
            const size_t qi = q ? (size_t)(q - buf) : end;
            start = qi + 2 + sl;
            return n;
        }
    }
    return parse_fastq_scan(buf, end, start, max_len, out);
}

// readseqInBuf (readseq1by1.c:138-209): next record of a FASTA buffer (single-line sequences)
int parse_fasta(const char* buf, size_t end, size_t& start, int max_len, uint8_t* out, bool simd) {
#if defined(__x86_64__)
    if (simd) { const int r = parse_fasta_avx2(buf, end, start, max_len, out); if (r >= 0) return r; }
#endif
    long p = -1;
    for (size_t m = start; m < end; m++) {
        const char c = buf[m];
        if (c == '>') p = (long)m;
        else if (c == '\n' && p >= 0 && buf[p] == '>') p = (long)m;
        else if (c == '\n' && p >= 0 && buf[p] == '\n') {
            const size_t line_len = m - p - 1;
            const size_t sl = strnlen(buf + p + 1, line_len);
            start = m + 1;
            return convert_line(buf + p + 1, (int)sl, max_len, out);
        }
    }
    return 0;
}

void reverse_complement(uint8_t* s, int n) {                       // reverse2k, readseq1by1.c:788-802
    for (int i = 0, j = n - 1; i < j; i++, j--) { uint8_t a = s[i] ^ 2, b = s[j] ^ 2; s[i] = b; s[j] = a; }
    if (n & 1) s[n / 2] ^= 2;
}

[[noreturn]] void bad_record(const char* buf, size_t size, size_t start) {   // prlHashReads.c:624-630
    fprintf(stderr, "readseqInLib return error! please make sure input file is correct fastq/fasta file \n");
    const std::string rest(buf + std::min(start, size), size - std::min(start, size));
    fprintf(stderr, "invalid data left in buffer:\n%s\n", rest.c_str());
    exit(-1);
}
[[noreturn]] void bad_record(const std::string& buf, size_t start) { bad_record(buf.data(), buf.size(), start); }


// ---- the same stream, parsed by all host threads ------------------------------------------------------------------
// Where a 32 KiB chunk is cut depends on that chunk alone, and every buffer the reference parses is the stretch of file
// between two consecutive cuts, parsed on its own.  So a window of chunks is read at once, the cuts and then the
// buffers are handled in parallel (each thread packs its reads 2 bits a base), and the runs go to the sink in file
// order.  Single-file inputs only; the mate-file interleave stays sequential.
// codes (one byte a base) -> 2 bits a base, 32 bases a word, first base in the top bits (pg_pack_read's layout).  `codes`
// is readable for 8 bytes past n.  With BMI2 eight bases go through one PEXT.
static void pack_plain(const uint8_t* codes, int n, uint64_t* out) {
    const int nw = (n + 31) / 32;
    for (int w = 0; w < nw; w++) {
        uint64_t v = 0;
        const int lo = w * 32, hi = std::min(n, lo + 32);
        for (int i = lo; i < hi; i++) v |= (uint64_t)(codes[i] & 3) << (62 - 2 * (i - lo));
        out[w] = v;
    }
}
#if defined(__x86_64__)
__attribute__((target("bmi2"))) static void pack_bmi2(const uint8_t* codes, int n, uint64_t* out) {
    const int nw = (n + 31) / 32;
    for (int w = 0; w < nw; w++) {
        uint64_t v = 0;
        const int lo = w * 32;
        for (int g = 0; g < 4; g++) {
            const int at = lo + 8 * g;
            if (at >= n) break;
            uint64_t x;
            memcpy(&x, codes + at, 8);
            const int valid = std::min(8, n - at);
            if (valid < 8) x &= (~0ULL) >> (64 - 8 * valid);               // bytes past the read do not count
            const uint64_t bits = __builtin_ia32_pext_di(__builtin_bswap64(x), 0x0303030303030303ULL);   // first base on top
            v |= bits << (48 - 16 * g);
        }
        out[w] = v;
    }
}
#endif
typedef void (*PackFn)(const uint8_t*, int, uint64_t*);
static PackFn pick_pack() {
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("bmi2")) return pack_bmi2;
#endif
    return pack_plain;
}
static const PackFn g_pack = pick_pack();

// (one cache line pair to itself: the parsing threads' runs lie next to each other in a vector, and every record moves the
//  vectors' end pointers and the counters -- unaligned, two threads shared a line and the reader stopped scaling at two threads)
struct alignas(128) PackedRun {
    std::vector<uint64_t> words;
    std::vector<int32_t> lens;
    long long records = 0;
    int min_len = 0x7fffffff, max_len = 0;
    void clear() { words.clear(); lens.clear(); records = 0; min_len = 0x7fffffff; max_len = 0; }
};

// the one-byte-a-base buffer of a parsing thread: written for every base of every read, so it keeps 128 bytes of distance from its
// neighbours' (the threads' buffers used to be 160-byte heap blocks side by side: two threads shared a cache line, and eight threads
// parsed no faster than one -- scripts/parse_bench.cpp)
struct CodeBuf {
    std::vector<uint8_t> raw;
    explicit CodeBuf(size_t n = 0) : raw(n + 256) {}
    uint8_t* data() { return (uint8_t*)(((uintptr_t)raw.data() + 127) & ~(uintptr_t)127); }
};
void parse_range(const InputFile& in, bool fastq, const char* buf, size_t size, CodeBuf& codes, PackedRun& out) {
    size_t start = 0;
    const bool simd = simd_parse_on();
    while (start < size) {
        const int n = fastq ? parse_fastq(buf, size, start, in.max_read_len, codes.data(), simd) : parse_fasta(buf, size, start, in.max_read_len, codes.data(), simd);
        if (n < 1) bad_record(buf, size, start);
        if (in.reverse) reverse_complement(codes.data(), n);
        out.records++;
        const size_t nw = ((size_t)n + 31) / 32, at = out.words.size();
        out.words.resize(at + nw);
        g_pack(codes.data(), n, out.words.data() + at);
        out.lens.push_back(n);
        out.min_len = std::min(out.min_len, n);
        out.max_len = std::max(out.max_len, n);
    }
}

// Threads that live as long as the file is streamed: body(t) for t = 0 .. n - 1, t = 0 on the calling thread.  (A window of the
// file is parsed in some 15 ms; threads created per window start on their creator's processor and are spread by the scheduler's
// balancer only ticks later -- sixteen of them parsed a window hardly faster than three: scripts/parse_bench.cpp shows the same
// with short jobs.  Threads that sleep between windows stay where they were.)
class WorkerPool {
    int n;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    const std::function<void(int)>* job = nullptr;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_start.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                j = job;
            }
            (*j)(t);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
public:
    explicit WorkerPool(int n_) : n(std::max(1, n_)) { for (int t = 1; t < n; t++) th.emplace_back([this, t] { worker(t); }); }
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_start.notify_all();
        for (auto& t : th) t.join();
    }
    void run(const std::function<void(int)>& body) {        // one caller at a time
        if (n > 1) {
            { std::lock_guard<std::mutex> lk(m); job = &body; pending = n - 1; gen++; }
            cv_start.notify_all();
        }
        body(0);
        if (n > 1) { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return pending == 0; }); }
    }
};

// `on_run` gets the packed reads of a stretch of the file, in file order; returning false stops the stream early
typedef std::function<bool(PackedRun&)> RunFn;
long long stream_file_parallel(const InputFile& in, const std::string& path, bool fastq, int nt, size_t window_chunks, const RunFn& on_run) {
    Source src;
    src.open(path);
    std::vector<char> win;
    const char* wdata = nullptr;                             // where the current window (cached tail, then fresh bytes) starts
    std::vector<size_t> cuts;
    std::vector<std::pair<size_t, size_t>> bufs;
    // SOAPDENOVO2_AMD_READER=map: a regular file is mapped and parsed where the page cache holds it instead of being read into
    // buffers (which a pipe always is).  Windows, cuts and buffers are the same.  Not the default: it halves the reader's time on
    // an 8-core box, but on the GPU box (256 hardware threads, 16 granted) the page faults cost more than the four copying
    // threads they replace (19 GB: 2.6 - 2.7 s against 2.0 - 2.3 s).
    const char* map = nullptr;
    size_t map_len = 0;
    if (!src.sequential() && pg::env_user("SOAPDENOVO2_AMD_READER") && !strcmp(pg::env_user("SOAPDENOVO2_AMD_READER"), "map")) {
        struct stat st;
        if (fstat(fileno(src.fp), &st) == 0 && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(src.fp), 0);
            if (m != MAP_FAILED) { map = (const char*)m; map_len = (size_t)st.st_size; (void)madvise(m, map_len, MADV_SEQUENTIAL); }
        }
    }
    // two sets of runs: while the caller's thread hands one window's runs over (copies into the batch buffers: serial, and
    // about as long as the parsing itself once a dozen threads parse), the threads already cut and parse the next window
    std::vector<PackedRun> run_sets[2] = {std::vector<PackedRun>(nt), std::vector<PackedRun>(nt)};
    std::vector<CodeBuf> codes(nt, CodeBuf((size_t)std::max(in.max_read_len, 1) + 8));
    constexpr int READERS = 8;                              // (19 GB: four copy out of the page cache in 0.4 s, the sixteen parsers need 0.3 s)
    WorkerPool workers(nt), readers(READERS);
    std::string last_buf;                                   // the buffer parsed last, for the N x 32768 rerun
    size_t carry = 0;
    long long n_records = 0;
    bool stopped = false;
    auto deliver = [&](std::vector<PackedRun>& runs, int used) {
        for (int t = 0; t < used && !stopped; t++) {
            PackedRun& r = runs[t];
            n_records += r.records;
            if (!r.lens.empty() && !on_run(r)) stopped = true;
        }
    };
    auto parse_all = [&](std::vector<PackedRun>& runs) {
        // contiguous groups of buffers of about equal size, one per thread
        size_t total = 0;
        for (auto& b : bufs) total += b.second - b.first;
        std::vector<size_t> first(nt + 1, bufs.size());
        first[0] = 0;
        size_t acc = 0, bi = 0;
        for (int t = 1; t < nt; t++) {
            const size_t want = total * (size_t)t / (size_t)nt;
            while (bi < bufs.size() && acc < want) { acc += bufs[bi].second - bufs[bi].first; bi++; }
            first[t] = bi;
        }
        workers.run([&](int t) {
            runs[t].clear();
            for (size_t i = first[t]; i < first[t + 1]; i++) parse_range(in, fastq, wdata + bufs[i].first, bufs[i].second - bufs[i].first, codes[t], runs[t]);
        });
    };
    // a regular file is read with a few preads side by side (one thread copying out of the page cache is slower than the
    // parsers); a pipe (.gz) is read as it comes
    double t_parse = 0, t_hand = 0, t_cut = 0, t_read = 0, t_pa = 0;
    auto nowd = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    uint64_t file_off = 0;
    auto read_window = [&](char* dst, size_t want) -> size_t {
        if (src.sequential()) return fread(dst, 1, want, src.fp);
        const int fd = fileno(src.fp);
        const int parts = READERS;
        size_t got_part[READERS] = {};
        readers.run([&](int t) {
            const size_t lo = want * t / parts, hi = want * (t + 1) / parts;
            size_t done = 0;
            while (lo + done < hi) {
                const ssize_t r = pread(fd, dst + lo + done, hi - lo - done, (off_t)(file_off + lo + done));
                if (r <= 0) break;
                done += (size_t)r;
            }
            got_part[t] = done;
        });
        size_t total = 0;
        for (int t = 0; t < parts; t++) {                       // a short part means the file ended inside it
            total += got_part[t];
            if (got_part[t] < want * (t + 1) / parts - want * t / parts) break;
        }
        file_off += total;
        return total;
    };
    // two windows: while the threads parse one, the next stretch of the file is read into the other
    std::vector<char> next_win;
    size_t got = 0;
    auto map_window = [&](uint64_t off) -> size_t {            // the mapped file's next window: nothing moves, the kernel is told to read ahead
        const size_t n = off < map_len ? std::min(window_chunks * CHUNK, map_len - (size_t)off) : 0;
        if (n) (void)posix_fadvise(fileno(src.fp), (off_t)off, (off_t)std::min(2 * window_chunks * CHUNK, map_len - (size_t)off), POSIX_FADV_WILLNEED);
        return n;
    };
    if (map) got = map_window(0);
    else {
        win.resize(window_chunks * CHUNK);
        got = read_window(win.data(), window_chunks * CHUNK);
    }
    // One window (`win`, `carry` cached bytes + `got` fresh ones, got > 0): cut, parse into `runs`, and meanwhile read the
    // next window; leaves win / carry / got describing that next window.  Returns true when this window ended the file
    // inside a chunk (nothing follows).
    auto cut_and_parse = [&](std::vector<PackedRun>& runs) -> bool {
        const auto tr0 = std::chrono::steady_clock::now();
        const size_t n_full = got / CHUNK, rem = got % CHUNK;
        wdata = map ? map + file_off - carry : win.data();     // (mapped: file_off = where the window's fresh bytes start)
        cuts.assign(n_full, 0);
        {
            workers.run([&](int t) {
                for (size_t i = n_full * t / nt; i < n_full * (t + 1) / nt; i++)
                    cuts[i] = fastq ? fastq_cut(wdata + carry + i * CHUNK, CHUNK) : fasta_cut(wdata + carry + i * CHUNK, CHUNK);
            });
        }
        t_cut += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        bufs.clear();
        size_t begin = 0;
        for (size_t i = 0; i < n_full; i++) {
            const size_t end = carry + i * CHUNK + cuts[i];
            bufs.emplace_back(begin, end);
            begin = end;
        }
        if (rem) {                                             // the short last chunk goes out whole, behind the cached tail
            bufs.emplace_back(begin, carry + got);
            parse_all(runs);
            t_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
            return true;
        }
        last_buf.assign(wdata + bufs.back().first, bufs.back().second - bufs.back().first);
        const size_t tail = carry + got - begin;
        if (map) {
            const double a = nowd();
            parse_all(runs);
            t_pa += nowd() - a;
            (void)madvise((void*)(map + ((file_off - carry) & ~(size_t)4095)), ((carry + got - tail) & ~(size_t)4095), MADV_DONTNEED);   // parsed: the mapping lets go of it
            file_off += got;
            carry = tail;
            got = map_window(file_off);
            t_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
            return false;
        }
        next_win.resize(tail + window_chunks * CHUNK);
        memcpy(next_win.data(), win.data() + begin, tail);
        size_t next_got = 0;
        std::thread reader([&]() { const double a = nowd(); next_got = read_window(next_win.data() + tail, window_chunks * CHUNK); t_read += nowd() - a; });
        { const double a = nowd(); parse_all(runs); t_pa += nowd() - a; }
        reader.join();
        win.swap(next_win);
        carry = tail;
        got = next_got;
        t_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        return false;
    };
    if (got == 0) {
        // (an empty file: the reference warns as well, and has no previous buffer to parse again)
        fprintf(stderr, "Warning : aio_return zero, the size of input file must be N * 32768.\n");
    } else {
        int cur = 0;
        bool last = cut_and_parse(run_sets[0]);
        for (;;) {
            // run_sets[cur] holds a parsed window; the next one, if there is one, is cut and parsed by the threads while
            // this thread hands the runs over
            const bool more = !last && got != 0;
            bool next_last = false;
            std::thread ahead;
            if (more) ahead = std::thread([&, cur]() { next_last = cut_and_parse(run_sets[cur ^ 1]); });
            const auto th0 = std::chrono::steady_clock::now();
            deliver(run_sets[cur], nt);
            t_hand += std::chrono::duration<double>(std::chrono::steady_clock::now() - th0).count();
            if (ahead.joinable()) ahead.join();
            if (last || stopped) break;
            if (!more) {
                // the file ended on a chunk boundary: the reference parses its previous buffer again and loses the tail it
                // had cached (prlHashReads.c:873-877)
                fprintf(stderr, "Warning : aio_return zero, the size of input file must be N * 32768.\n");
                run_sets[0][0].clear();
                parse_range(in, fastq, last_buf.data(), last_buf.size(), codes[0], run_sets[0][0]);
                deliver(run_sets[0], 1);
                break;
            }
            cur ^= 1;
            last = next_last;
        }
    }
    if (map) (void)munmap((void*)map, map_len);
    src.close();
    if (pg::env_user("PG_HOST_VERBOSE")) fprintf(stderr, "reader: %.2fs cutting + parsing (with the next window's read), %.2fs handing over beside it (%d threads); of the first: cuts %.2fs, parsing %.2fs, the read beside it %.2fs\n", t_parse, t_hand, nt, t_cut, t_pa, t_read);
    return n_records;
}

long long stream_reads_parallel(const InputFile& in, ReadSink& sink, bool fastq, int nt, size_t window_chunks) {
    return stream_file_parallel(in, in.path1, fastq, nt, window_chunks, [&](PackedRun& r) {
        sink.on_packed(r.words.data(), r.lens.data(), r.lens.size(), r.min_len, r.max_len);
        return true;
    });
}

// Mate files (f1/f2, q1/q2): the reference takes one read from file 1, one from file 2, and so on, stops when file 2
// is used up and fails when file 1 runs dry first (prlHashReads.c:464-609).  Each file is parsed by its own group of
// threads into a short queue of runs; the caller's thread interleaves them read by read.
struct RunQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<PackedRun> q;
    bool done = false, cancel = false;
    bool push(PackedRun& r) {                            // producer; false = the consumer went away
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.size() < 4 || cancel; });
        if (cancel) return false;
        q.emplace_back();
        std::swap(q.back(), r);
        cv.notify_all();
        return true;
    }
    bool pop(PackedRun& r) {                             // consumer; false = the stream is over
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !q.empty() || done; });
        if (q.empty()) return false;
        std::swap(r, q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void finish() { std::lock_guard<std::mutex> lk(m); done = true; cv.notify_all(); }
    void stop() { std::lock_guard<std::mutex> lk(m); cancel = true; cv.notify_all(); }
};

long long stream_mates_parallel(const InputFile& in, ReadSink& sink, bool fastq, int nt, size_t window_chunks) {
    RunQueue qa, qb;
    const int nta = std::max(1, nt / 2), ntb = std::max(1, nt - nta);
    std::thread ta([&]() { stream_file_parallel(in, in.path1, fastq, nta, window_chunks, [&](PackedRun& r) { return qa.push(r); }); qa.finish(); });
    std::thread tb([&]() { stream_file_parallel(in, in.path2, fastq, ntb, window_chunks, [&](PackedRun& r) { return qb.push(r); }); qb.finish(); });
    PackedRun a, b, out;
    size_t ia = 0, ib = 0, wa = 0, wb = 0;               // next read and its first word in the current runs
    bool have_a = false, have_b = false;
    long long n_records = 0;
    auto next_a = [&]() { while (!have_a || ia >= a.lens.size()) { if (!qa.pop(a)) { have_a = false; return false; } have_a = true; ia = 0; wa = 0; } return true; };
    auto next_b = [&]() { while (!have_b || ib >= b.lens.size()) { if (!qb.pop(b)) { have_b = false; return false; } have_b = true; ib = 0; wb = 0; } return true; };
    auto take = [&](PackedRun& r, size_t& i, size_t& w) {
        const int len = r.lens[i];
        const size_t nw = ((size_t)len + 31) / 32;
        out.words.insert(out.words.end(), r.words.begin() + (long)w, r.words.begin() + (long)(w + nw));
        out.lens.push_back(len);
        out.min_len = std::min(out.min_len, len); out.max_len = std::max(out.max_len, len);
        i++; w += nw; n_records++;
    };
    auto flush = [&]() {
        if (!out.lens.empty()) sink.on_packed(out.words.data(), out.lens.data(), out.lens.size(), out.min_len, out.max_len);
        out.clear();
    };
    bool fail = false;
    for (;;) {
        const bool more_a = next_a(), more_b = next_b();
        if (!more_a && !more_b) break;
        if (!more_a) { fail = true; break; }             // file 1 ran dry while file 2 still has reads
        take(a, ia, wa);
        if (!more_b) { fail = true; break; }             // ... and the other way round, after file 1's read went out
        take(b, ib, wb);
        if (out.lens.size() >= (1u << 18)) flush();
        if (!next_b()) break;                            // file 2 used up: the reference stops here
    }
    flush();
    qa.stop(); qb.stop();
    ta.join(); tb.join();
    if (fail) {
        fprintf(stderr, "readseqInLib return error! please make sure input file is correct fastq/fasta file \n");
        fprintf(stderr, "invalid data left in buffer:\n\n");
        exit(-1);
    }
    return n_records;
}

}  // namespace

void ReadSink::on_packed(const uint64_t* words, const int32_t* lens, size_t n, int, int max_len) {
    std::vector<uint8_t> codes((size_t)std::max(max_len, 1) + 32);
    size_t at = 0;
    for (size_t r = 0; r < n; r++) {
        const int len = lens[r];
        for (int i = 0; i < len; i++) codes[i] = (uint8_t)((words[at + (i >> 5)] >> (62 - 2 * (i & 31))) & 3);
        on_read(codes.data(), len);
        at += ((size_t)len + 31) / 32;
    }
}

// ---- b=: BAM (read1seqbam / read1seqInLibBam, readseq1by1.c:449-592,1248-1280; the loops of prlHashReads.c:408-455 and
// prlRead2path.c:902-960).  An own reader over zlib: a BAM file is a series of gzip members (BGZF), which gzread() inflates
// one after the other; inside, the header and then records of block_size bytes (SAM/BAM specification 4.2).  What the
// reference makes of a record: the SEQ column of its SAM text ("=ACMGRSVTWYHKDBN" per 4-bit code; letters go through
// base2int, '=' and '*' are no letters and vanish), cut to the lib's read length, reverse_seq applied.
//   asm_flags = 1: records with the QC-fail flag (0x200) are skipped.
//   otherwise:     records pair up two by two (no names are looked at); a pair with a QC-fail mate is "taken back": the second
//                  mate is not delivered and the LAST KEPT READ of the buffer -- the first mate if it was long enough to be
//                  kept, else whatever was kept before it -- is removed again (prlHashReads.c:414-426).  Hence the delay
//                  line below: reads long enough to be kept wait in a short queue (BAM_HOLD deep) before they reach the
//                  sink, so that several pairs taken back in a row, each with a first mate too short to be kept, can each
//                  remove one earlier kept read as the reference's read_c-- does.
//                  (The reference indexes lenBuffer[read_c - 1] even when its buffer is empty -- a pair taken back right
//                  after a full buffer of reads was flushed -- which reads in front of the array; here nothing is removed
//                  then.)
// `state` (readseq1by1.c:44) is a static, but the reference puts it back to -3 whenever samread() reports the end of a file
// (`if (readstate < 0) state = -3;`, readseq1by1.c:584-587): a file with an odd number of records leaves its last record as a
// first mate without a second, and the next file -- and pass 2, which reads the files again -- starts pairing afresh.  So both
// passes see the same reads and call_pregraph may replay the reads it kept.  pg_host_bam_pair_state stays for tools that want
// to look at or force the state.
static int g_bam_pair_state = -3;
int bam_pair_state(bool set, int value) { if (set) g_bam_pair_state = value; return g_bam_pair_state; }

static long long stream_bam(const InputFile& in, ReadSink& sink) {
    gzFile fp = gzopen(in.path1.c_str(), "rb");
    if (!fp) { fprintf(stderr, "Cannot open %s. Now exit to system...\n", in.path1.c_str()); exit(-1); }
    gzbuffer(fp, 1 << 20);
    auto need = [&](void* dst, size_t n) { return gzread(fp, dst, (unsigned)n) == (int)n; };
    auto skip = [&](size_t n) { char tmp[4096]; while (n) { const size_t k = std::min(n, sizeof tmp); if (!need(tmp, k)) return false; n -= k; } return true; };
    char magic[4];
    int32_t l_text = 0, n_ref = 0;
    bool ok = need(magic, 4) && !memcmp(magic, "BAM\1", 4) && need(&l_text, 4) && l_text >= 0 && skip((size_t)l_text) && need(&n_ref, 4) && n_ref >= 0;
    for (int32_t r = 0; ok && r < n_ref; r++) {
        int32_t l_name = 0, l_ref = 0;
        ok = need(&l_name, 4) && l_name >= 0 && skip((size_t)l_name) && need(&l_ref, 4);
    }
    if (!ok) { fprintf(stderr, "Cannot read the header.\n"); exit(-1); }
    const int max_len = std::max(in.max_read_len, 1);
    constexpr int BAM_HOLD = 256;
    const size_t row = (size_t)max_len + 8;
    std::vector<uint8_t> codes(row), held(row * BAM_HOLD), rec;
    int held_len[BAM_HOLD];
    int held_head = 0, held_n = 0;                                   // kept-length reads waiting: ring of rows, oldest at held_head
    long long n_records = 0;
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    for (;;) {
        int32_t block = 0;
        if (!need(&block, 4) || block < 32) break;                   // end of file (or a truncated one: samread < 0 ends the file too)
        rec.resize((size_t)block);
        if (!need(rec.data(), (size_t)block)) break;
        const uint32_t l_read_name = rec[8];
        uint16_t n_cigar, flag;
        int32_t l_seq;
        memcpy(&n_cigar, rec.data() + 12, 2); memcpy(&flag, rec.data() + 14, 2); memcpy(&l_seq, rec.data() + 16, 4);
        const size_t seq_at = 32 + (size_t)l_read_name + 4 * (size_t)n_cigar;
        if (l_seq < 0 || seq_at + ((size_t)l_seq + 1) / 2 > (size_t)block) break;
        int type = 0;
        if (flag & 0x0200) {
            if (in.asm_flag == 1) continue;                          // not a good read: on to the next record
            switch (g_bam_pair_state) { case -3: g_bam_pair_state = -2; break; case -2: g_bam_pair_state = 0; break; case -1: g_bam_pair_state = 2; break; default: g_bam_pair_state = -3; }
        } else {
            switch (g_bam_pair_state) { case -3: g_bam_pair_state = -1; break; case -2: g_bam_pair_state = 1; break; case -1: g_bam_pair_state = 3; break; default: g_bam_pair_state = -3; }
        }
        // the SEQ column: l_seq characters ("*" when there are none), the first max_read_len of them looked at
        int n = 0;
        const int look = std::min(l_seq == 0 ? 1 : (int)l_seq, max_len);
        for (int j = 0; j < look && l_seq > 0; j++) {
            const char ch = nt16[(rec[seq_at + (size_t)(j >> 1)] >> ((~j & 1) << 2)) & 0xf];
            if (ch >= 'A' && ch <= 'Z') codes[n++] = (uint8_t)base_code(ch);
        }
        if (g_bam_pair_state == 3) g_bam_pair_state = -3;
        else if (g_bam_pair_state == 0 || g_bam_pair_state == 1 || g_bam_pair_state == 2) { g_bam_pair_state = -3; type = -1; }
        if (in.reverse) reverse_complement(codes.data(), n);
        if (type == -1) {                                            // the pair is taken back (prlHashReads.c:412-426)
            n_records--;
            if (held_n > 0) held_n--;                                // the last kept read goes again (read_c--)
            continue;
        }
        n_records++;
        if (n < std::max(in.keep_len, 1)) { sink.on_read(codes.data(), n); continue; }      // nobody keeps it: order does not matter
        if (held_n == BAM_HOLD) {
            sink.on_read(held.data() + row * (size_t)held_head, held_len[held_head]);
            held_head = (held_head + 1) % BAM_HOLD;
            held_n--;
        }
        const int at = (held_head + held_n) % BAM_HOLD;
        memcpy(held.data() + row * (size_t)at, codes.data(), (size_t)n);
        held_len[at] = n;
        held_n++;
    }
    g_bam_pair_state = -3;                                           // end of file, or a truncated one: readseq1by1.c:584-587
    for (; held_n > 0; held_n--, held_head = (held_head + 1) % BAM_HOLD) sink.on_read(held.data() + row * (size_t)held_head, held_len[held_head]);
    gzclose(fp);
    return n_records;
}

long long stream_reads(const InputFile& in, ReadSink& sink) {
    if (in.type == 4) return stream_bam(in, sink);
    const bool fastq = (in.type == 2 || in.type == 6);
    std::vector<uint8_t> codes((size_t)std::max(in.max_read_len, 1) + 8);
    long long n_records = 0;
    auto parse = [&](const std::string& buf, size_t& start) {
        const int n = fastq ? parse_fastq(buf.data(), buf.size(), start, in.max_read_len, codes.data(), /*simd=*/false)    // (the chunk emulation stays on the scalar definition)
                            : parse_fasta(buf.data(), buf.size(), start, in.max_read_len, codes.data(), /*simd=*/false);
        if (n < 1) bad_record(buf, start);
        if (in.reverse) reverse_complement(codes.data(), n);
        n_records++;
        sink.on_read(codes.data(), n);
    };
    int par_threads = host_threads(0);
    if (const char* e = pg::env_user("SOAPDENOVO2_AMD_PARSE_THREADS")) { const int v = atoi(e); if (v > 0) par_threads = v; }
    size_t par_window = 4096;                                    // 128 MiB of text at a time
    if (const char* e = pg::env_test("SOAPDENOVO2_AMD_PARSE_WINDOW")) { const long v = atol(e); if (v > 0) par_window = (size_t)v; }
    size_t par_min_bytes = (size_t)8 << 20;
    if (const char* e = pg::env_test("SOAPDENOVO2_AMD_PARSE_PARALLEL_MIN")) par_min_bytes = (size_t)atol(e);
    auto is_big = [&](const std::string& path) {
        struct stat st;
        return stat(path.c_str(), &st) != 0 || (size_t)st.st_size >= par_min_bytes || !S_ISREG(st.st_mode);
    };
    if (in.type == 1 || in.type == 2) {
        if (par_threads > 1 && is_big(in.path1)) return stream_mates_parallel(in, sink, fastq, par_threads, par_window);
        // mate files: reads alternate file 1, file 2, ... (prlHashReads.c:464-609)
        ChunkStream s1(in.path1, fastq), s2(in.path2, fastq);
        bool ok1 = s1.next(), ok2 = s2.next();
        size_t st1 = 0, st2 = 0;
        int turn = 1;
        while ((ok1 && st1 < s1.buf.size()) || (ok2 && st2 < s2.buf.size())) {
            if (turn == 1) {
                turn = 2;
                if (!(ok1 && st1 < s1.buf.size())) bad_record(s1.buf, st1);
                parse(s1.buf, st1);
                if (st1 >= s1.buf.size()) { st1 = 0; ok1 = s1.next(); if (!ok1) s1.buf.clear(); }
            } else {
                turn = 1;
                if (!(ok2 && st2 < s2.buf.size())) bad_record(s2.buf, st2);
                parse(s2.buf, st2);
                if (st2 >= s2.buf.size()) {
                    if (s2.last) break;
                    st2 = 0; ok2 = s2.next(); if (!ok2) s2.buf.clear();
                }
            }
        }
        return n_records;
    }
    // single file: all host threads, unless the file is small (or SOAPDENOVO2_AMD_PARSE_THREADS=1)
    if (par_threads > 1 && is_big(in.path1)) return stream_reads_parallel(in, sink, fastq, par_threads, par_window);
    ChunkStream s(in.path1, fastq);
    while (s.next()) {
        size_t start = 0;
        while (start < s.buf.size()) parse(s.buf, start);
    }
    return n_records;
}

}  // namespace pg
