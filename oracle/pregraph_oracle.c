/*
 * oracle/pregraph_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the SOAPdenovo2 `pregraph` hot path (pass 1 k-mer counting,
 * the k-mer set layout, low-coverage filter, linear marking, tip clipping, edge construction and the
 * .kmerFreq / .vertex / .edge / .preGraphBasic writers).  It exists only so the tests, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() have something to check the HIP path against.  Nothing in
 * soapdenovo2_amd/ may include, link or call it.
 *
 * It is written from the behaviour of the reference (file:line cited at each function; paths relative to
 * /root/reference/standardPregraph), not copied from it: one generic NW-word k-mer instead of the
 * reference's two compile-time variants, index-free sequential loops instead of the thread pool, and a
 * direct simulation of the "thrd_num" k-mer sets in one thread.
 *
 * Parity pin: validated byte-for-byte against the reference built by oracle/Makefile.ref (oracle/_ref)
 * and against the committed golden files under tests/golden/ (tests/test_oracle_golden.py).
 *
 * Conventions: base codes A0 C1 T2 G3, complement = c ^ 2 (inc/def.h:39-42).  A k-mer is NW 64-bit words,
 * w[0] most significant; NW = 2 restates the 63-mer binary (Kmer{high,low}), NW = 4 the 127-mer binary
 * (Kmer{high1,low1,high2,low2}) (inc/def.h:46-56).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t w[4]; } okmer_t;

typedef struct {
    okmer_t seq;
    uint32_t A;   /* l_links:24 (4 x 6 bit, index = base code) | covs:8        (inc/newhash.h:77-102) */
    uint32_t B;   /* r_links:24 | linear:1 deleted:1 checked:1 single:1 twin:2 inEdge:2               */
    uint64_t first_ord;  /* oracle-only bookkeeping: global ordinal of the first occurrence            */
} onode_t;

#define B_LINEAR  (1u << 24)
#define B_DELETED (1u << 25)
#define B_SINGLE  (1u << 27)
#define B_TWIN_SHIFT 28
#define B_INEDGE_SHIFT 30

typedef struct {
    onode_t *array;
    uint8_t *occ;          /* 1 = slot holds a node (reference: !is_kmer_entity_null)  */
    uint64_t size, count, max;
    float load_factor;
} oset_t;

typedef struct {
    int K, NW, P, D, a_gb;
    oset_t *sets;
    okmer_t filter;
    uint64_t n_kmers;        /* k-mer occurrences seen                                  */
    uint64_t last_put[256];  /* per set: ordinal+1 of the last put (0 = none)           */
    int num_ed, num_vt;
    int max_read_len;
} octx_t;

/* ------------------------------------------------------------------ k-mer arithmetic (kmer.c) */
static int nw_g;  /* words in use; set once per context */

static inline okmer_t k_zero(void) { okmer_t z; memset(&z, 0, sizeof z); return z; }

/* KmerLeftBitMoveBy2 (kmer.c:672-679 / 172-191) */
static inline okmer_t k_shl2(okmer_t a) {
    for (int i = 0; i < nw_g; i++) {
        uint64_t carry = (i + 1 < nw_g) ? (a.w[i + 1] >> 62) : 0;
        a.w[i] = (a.w[i] << 2) | carry;
    }
    return a;
}
static inline okmer_t k_and(okmer_t a, okmer_t b) { for (int i = 0; i < nw_g; i++) a.w[i] &= b.w[i]; return a; }
/* KmerSmaller / KmerLarger / KmerEqual (kmer.c:608-663) */
static inline int k_cmp(okmer_t a, okmer_t b) {
    for (int i = 0; i < nw_g; i++) { if (a.w[i] < b.w[i]) return -1; if (a.w[i] > b.w[i]) return 1; }
    return 0;
}
/* createFilter (kmer.c:738-758 / 345-375): low 2K bits set */
static okmer_t k_filter(int K) {
    okmer_t f = k_zero();
    int bits = 2 * K;
    for (int i = nw_g - 1; i >= 0 && bits > 0; i--) {
        f.w[i] = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
        bits -= 64;
    }
    return f;
}
/* generic right shift by d bits (KmerRightBitMove, kmer.c:760-776) */
static okmer_t k_shr(okmer_t a, int d) {
    okmer_t r = k_zero();
    int ws = d / 64, bs = d % 64;
    for (int i = nw_g - 1; i >= 0; i--) {
        int src = i - ws;
        if (src < 0) continue;
        uint64_t v = a.w[src] >> bs;
        if (bs && src - 1 >= 0) v |= a.w[src - 1] << (64 - bs);
        r.w[i] = v;
    }
    return r;
}
static inline uint64_t rev2(uint64_t x) {   /* reverse the 32 two-bit groups of x */
    x = ((x & 0x3333333333333333ULL) << 2) | ((x >> 2) & 0x3333333333333333ULL);
    x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL);
    return __builtin_bswap64(x);
}
/* reverseComplement (kmer.c:819-855 / 532-591): complement (^2 per base), reverse base order, right-align */
static okmer_t k_rc(okmer_t a, int len) {
    okmer_t r = k_zero();
    for (int i = 0; i < nw_g; i++) r.w[nw_g - 1 - i] = rev2(a.w[i] ^ 0xAAAAAAAAAAAAAAAAULL);
    return k_shr(r, 64 * nw_g - 2 * len);
}
/* nextKmer (kmer.c:696-702) */
static inline okmer_t k_next(okmer_t a, int ch, okmer_t filter) {
    a = k_and(k_shl2(a), filter);
    a.w[nw_g - 1] |= (uint64_t)ch;
    return a;
}
static inline int k_last(okmer_t a) { return (int)(a.w[nw_g - 1] & 3); }                 /* kmer.c:720 */
static inline int k_first(okmer_t a, int K) {                                              /* kmer.c:724 */
    int bit = 2 * (K - 1);
    return (int)((a.w[nw_g - 1 - bit / 64] >> (bit % 64)) & 3);
}

/* ------------------------------------------------------------------ hash_kmer (hashFunction.c:123-158) */
static uint32_t crc_tab[256];
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_tab[i] = c;
    }
}
/* CRC-32 (init 0, final xor) over the raw struct bytes: words in struct order, each little-endian; the
 * reference returns it as `int`, so it is sign-extended to 64 bit before `% thrd_num`. */
static uint64_t hash_kmer(okmer_t a) {
    uint32_t crc = 0;
    for (int i = 0; i < nw_g; i++)
        for (int b = 0; b < 8; b++)
            crc = crc_tab[(crc ^ (uint32_t)(a.w[i] >> (8 * b))) & 0xff] ^ (crc >> 8);
    crc ^= 0xffffffffu;
    return (uint64_t)(int64_t)(int32_t)crc;
}

/* ------------------------------------------------------------------ k-mer set (newhash.c) */
/* is_prime_kh (newhash.c:142-167): trial division by odd i with 3 <= i < (u64)sqrt((float)n), strict '<' */
static int is_prime_kh(uint64_t num) {
    if (num < 4) return 1;
    if (num % 2 == 0) return 0;
    uint64_t max = (uint64_t)sqrt((float)num);
    for (uint64_t i = 3; i < max; i += 2) if (num % i == 0) return 0;
    return 1;
}
static uint64_t next_prime_kh(uint64_t num) {              /* newhash.c:169-185 */
    if (num % 2 == 0) num++;
    while (!is_prime_kh(num)) num += 2;
    return num;
}
/* modular (newhash.c:36-57): exact 128-bit modulus for the 63-mer build, chained 32-bit-chunk reduction for
 * the 127-mer build */
static uint64_t home_slot(const oset_t *s, okmer_t k) {
    if (nw_g == 2) {
        unsigned __int128 t = ((unsigned __int128)k.w[0] << 64) | k.w[1];
        return (uint64_t)(t % s->size);
    }
    uint64_t t;
    t = (k.w[0] % s->size) << 32 | (k.w[1] >> 32 & 0xffffffffULL);
    t = (t % s->size) << 32 | (k.w[1] & 0xffffffffULL);
    t = (t % s->size) << 32 | (k.w[2] >> 32 & 0xffffffffULL);
    t = (t % s->size) << 32 | (k.w[2] & 0xffffffffULL);
    t = (t % s->size) << 32 | (k.w[3] >> 32 & 0xffffffffULL);
    t = (t % s->size) << 32 | (k.w[3] & 0xffffffffULL);
    return t % s->size;
}
static void set_init(oset_t *s, uint64_t init_size, float lf) {   /* init_kmerset, newhash.c:200-233 */
    init_size = init_size < 3 ? 3 : next_prime_kh(init_size);
    s->size = init_size; s->count = 0;
    s->max = (uint64_t)(s->size * lf);
    s->load_factor = lf;
    s->array = calloc(s->size, sizeof(onode_t));
    s->occ = calloc(s->size, 1);
}
/* encap_kmerset (newhash.c:340-455) */
static void set_grow(oset_t *s, int static_pool) {
    if (s->count + 1 <= s->max) return;
    if (static_pool) {
        /* `load_factor < 0.88` compares float with double 0.88: 0.88f < 0.88 is true as well, so a static
         * pool never aborts and never grows */
        if ((double)s->load_factor < 0.88) { s->load_factor = 0.88; s->max = (uint64_t)(s->size * s->load_factor); return; }
        fprintf(stderr, "oracle: static memory pool exploded\n"); abort();
    }
    uint64_t n = s->size;
    do {
        n = (n < 0xFFFFFFFULL) ? (n << 1) : (n + 0xFFFFFFULL);
        n = next_prime_kh(n);
    } while (n * s->load_factor < s->count + 1);       /* float arithmetic, as in the reference */
    uint64_t old = s->size;
    s->array = realloc(s->array, n * sizeof(onode_t));
    uint8_t *newocc = calloc(n, 1);
    uint8_t *pending = s->occ;    /* old flags: 1 = still to be moved */
    s->size = n; s->max = (uint64_t)(n * s->load_factor);
    for (uint64_t i = 0; i < old; i++) {
        if (!pending[i]) continue;
        onode_t key = s->array[i];
        pending[i] = 0;
        for (;;) {
            uint64_t hc = home_slot(s, key.seq);
            while (newocc[hc]) { hc++; if (hc == s->size) hc = 0; }
            newocc[hc] = 1;
            if (hc < old && pending[hc]) {
                onode_t tmp = key; key = s->array[hc]; s->array[hc] = tmp; pending[hc] = 0;
            } else { s->array[hc] = key; break; }
        }
    }
    free(pending);
    s->occ = newocc;
}
static inline uint32_t sat6_inc(uint32_t word, int idx) {
    uint32_t c = (word >> (6 * idx)) & 63u;
    if (c < 63u) word += 1u << (6 * idx);
    return word;
}
/* put_kmerset + set_new_kmer + update_kmer (newhash.c:473-528, 123-140, 74-106) */
static onode_t *set_put(oset_t *s, okmer_t k, int left, int right, int static_pool, uint64_t ord) {
    if (s->count + 1 > s->max) set_grow(s, static_pool);
    uint64_t hc = home_slot(s, k);
    for (;;) {
        if (!s->occ[hc]) {
            onode_t *n = &s->array[hc];
            memset(n, 0, sizeof *n);
            n->seq = k; n->B = B_SINGLE;
            if (left < 4) n->A |= 1u << (6 * left);
            if (right < 4) n->B |= 1u << (6 * right);
            n->A |= 1u << 24;
            n->first_ord = ord;
            s->occ[hc] = 1; s->count++;
            return n;
        }
        if (k_cmp(s->array[hc].seq, k) == 0) {
            onode_t *n = &s->array[hc];
            if (left < 4) n->A = sat6_inc(n->A, left);
            if (right < 4) n->B = sat6_inc(n->B, right);
            if (left < 4 || right < 4) { if ((n->A >> 24) < 255u) n->A += 1u << 24; }
            n->B &= ~B_SINGLE;
            return n;
        }
        if (++hc == s->size) hc = 0;
    }
}
static onode_t *set_search(oset_t *s, okmer_t k) {          /* search_kmerset, newhash.c:277-318 */
    uint64_t hc = home_slot(s, k);
    for (;;) {
        if (!s->occ[hc]) return NULL;
        if (k_cmp(s->array[hc].seq, k) == 0) return &s->array[hc];
        if (++hc == s->size) hc = 0;
    }
}
static inline int Lc(const onode_t *n, int i) { return (n->A >> (6 * i)) & 63; }
static inline int Rc(const onode_t *n, int i) { return (n->B >> (6 * i)) & 63; }
static inline void setL(onode_t *n, int i, uint32_t v) { n->A = (n->A & ~(63u << (6 * i))) | (v << (6 * i)); }
static inline void setR(onode_t *n, int i, uint32_t v) { n->B = (n->B & ~(63u << (6 * i))) | (v << (6 * i)); }
static int n_in(const onode_t *n) { int c = 0; for (int i = 0; i < 4; i++) c += Lc(n, i) > 0; return c; }
static int n_out(const onode_t *n) { int c = 0; for (int i = 0; i < 4; i++) c += Rc(n, i) > 0; return c; }

/* ------------------------------------------------------------------ context */
octx_t *oracle_create(int K, int P, int D, int a_gb, int mer127, int max_read_len) {
    crc_init();
    octx_t *c = calloc(1, sizeof *c);
    /* K clamp (pregraph.c:71-97) */
    if (K % 2 == 0) K++;
    if (K < 13) K = 13; else if (K > (mer127 ? 127 : 63)) K = mer127 ? 127 : 63;
    c->K = K; c->NW = mer127 ? 4 : 2; c->P = P; c->D = D; c->a_gb = a_gb; c->max_read_len = max_read_len;
    nw_g = c->NW;
    c->filter = k_filter(K);
    c->sets = calloc(P, sizeof(oset_t));
    /* initial size (prlHashReads.c:369-390) */
    uint64_t init_size = 1024, k = 0;
    if (a_gb) {
        init_size = (uint64_t)((double)a_gb * 1024.0f * 1024.0f * 1024.0f / (double)P / (mer127 ? 40 : 24));
        do { ++k; } while (k * 0xFFFFFFULL < init_size);
    }
    for (int i = 0; i < P; i++) set_init(&c->sets[i], a_gb ? k * 0xFFFFFFULL : init_size, 0.77f);
    return c;
}
int oracle_K(const octx_t *c) { return c->K; }

/* chopKmer4read + the per-set insert loop (prlHashReads.c:163-259, 79-90).  `seq` = base codes 0..3.
 * Reads shorter than K+1 are skipped by the caller of the reference (prlHashReads.c:642); done here. */
void oracle_add_read(octx_t *c, const uint8_t *seq, int len) {
    int K = c->K;
    nw_g = c->NW;
    if (len < K + 1) return;
    okmer_t word = k_zero();
    for (int i = 0; i < K; i++) { word = k_shl2(word); word.w[nw_g - 1] |= seq[i]; }
    for (int j = 0; j <= len - K; j++) {
        if (j > 0) word = k_next(word, seq[j - 1 + K], c->filter);
        okmer_t bal = k_rc(word, K);
        okmer_t key; int left, right;
        if (k_cmp(word, bal) < 0) {          /* KmerSmaller(word, bal_word) */
            key = word;
            left = j > 0 ? seq[j - 1] : 4;
            right = j < len - K ? seq[j + K] : 4;
        } else {
            key = bal;
            left = j < len - K ? (seq[j + K] ^ 2) : 4;
            right = j > 0 ? (seq[j - 1] ^ 2) : 4;
        }
        int s = (int)(hash_kmer(key) % (uint64_t)c->P);
        uint64_t ord = c->n_kmers++;
        set_put(&c->sets[s], key, left, right, c->a_gb != 0, ord);
        c->last_put[s] = ord + 1;
    }
}

/* thread_delow (prlHashReads.c:953-996) */
static void delow(octx_t *c) {
    for (int p = 0; p < c->P; p++) {
        oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            onode_t *n = &s->array[i];
            for (int b = 0; b < 4; b++) {
                int v = Lc(n, b); if (v > 0 && v <= c->D) setL(n, b, 0);
                v = Rc(n, b);     if (v > 0 && v <= c->D) setR(n, b, 0);
            }
            if ((n->A & 0xFFFFFFu) == 0 && (n->B & 0xFFFFFFu) == 0) n->B |= B_DELETED;
        }
    }
}
/* thread_mark + freqStat (prlHashReads.c:1020-1132) */
static void mark_and_freq(octx_t *c, const char *prefix) {
    long long hist[257]; memset(hist, 0, sizeof hist);
    for (int p = 0; p < c->P; p++) {
        oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            onode_t *n = &s->array[i];
            hist[n->A >> 24]++;
            if (n_in(n) == 1 && n_out(n) == 1) n->B |= B_LINEAR;
        }
    }
    char name[1024]; snprintf(name, sizeof name, "%s.kmerFreq", prefix);
    FILE *fo = fopen(name, "w");
    for (int i = 1; i < 256; i++) fprintf(fo, "%lld\n", hist[i]);
    fclose(fo);
}
/* Mark1in1outNode of cutTipPreGraph.c:532-564 (skips deleted and already-linear nodes) */
static void remark_linear(octx_t *c) {
    for (int p = 0; p < c->P; p++) {
        oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            onode_t *n = &s->array[i];
            if (n->B & (B_DELETED | B_LINEAR)) continue;
            if (n_in(n) == 1 && n_out(n) == 1) n->B |= B_LINEAR;
        }
    }
}

/* canonicalise a walk-oriented word and find its node */
static onode_t *lookup(octx_t *c, okmer_t word, okmer_t *canon, okmer_t *bal_out, int *smaller) {
    okmer_t bal = k_rc(word, c->K);
    if (k_cmp(word, bal) > 0) { okmer_t t = bal; bal = word; word = t; *smaller = 0; } else *smaller = 1;
    *canon = word; *bal_out = bal;
    return set_search(&c->sets[hash_kmer(word) % (uint64_t)c->P], word);
}
/* dislink2prevUncertain / dislink2nextUncertain (newhash.c:681-717) */
static void dislink_prev(onode_t *n, int ch, int smaller) { if (smaller) setL(n, ch, 0); else setR(n, ch ^ 2, 0); }
static void dislink_next(onode_t *n, int ch, int smaller) { if (smaller) setR(n, ch, 0); else setL(n, ch ^ 2, 0); }

/* clipTipFromNode (cutTipPreGraph.c:43-346) */
static int clip_tip(octx_t *c, onode_t *node1, int cut_len, int THIN, int *tip_c) {
    int in_num = n_in(node1), out_num = n_out(node1), ch1, ch = 0, smaller;
    okmer_t pre_word, word, canon, bal;
    if (in_num == 0 && out_num == 1) {
        pre_word = node1->seq;
        for (ch1 = 0; ch1 < 4; ch1++) if (Rc(node1, ch1)) break;
        word = k_next(pre_word, ch1, c->filter);
    } else if (in_num == 1 && out_num == 0) {
        pre_word = k_rc(node1->seq, c->K);
        for (ch1 = 0; ch1 < 4; ch1++) if (Lc(node1, ch1)) break;
        word = k_next(pre_word, ch1 ^ 2, c->filter);
    } else return 0;
    int count = 1;
    onode_t *out = lookup(c, word, &canon, &bal, &smaller);
    if (!out) { fprintf(stderr, "oracle: tip walk lost a k-mer\n"); exit(1); }
    while (out->B & B_LINEAR) {
        count++;
        if (THIN && !(out->B & B_SINGLE)) break;
        if (count > cut_len) return 0;
        if (smaller) {
            pre_word = canon;
            for (ch = 0; ch < 4; ch++) if (Rc(out, ch)) break;
            word = k_next(pre_word, ch, c->filter);
        } else {
            pre_word = bal;
            for (ch = 0; ch < 4; ch++) if (Lc(out, ch)) break;
            word = k_next(pre_word, ch ^ 2, c->filter);
        }
        out = lookup(c, word, &canon, &bal, &smaller);
        if (!out) { fprintf(stderr, "oracle: tip walk lost a k-mer\n"); exit(1); }
    }
    if (n_in(out) + n_out(out) == 1) {
        (*tip_c)++; node1->B |= B_DELETED; out->B |= B_DELETED; return 1;
    }
    ch = k_first(pre_word, c->K);
    if (THIN) {
        (*tip_c)++; node1->B |= B_DELETED;
        dislink_prev(out, ch, smaller);
        out->B &= ~B_LINEAR;
        return 1;
    }
    uint32_t max_links = 0;
    for (ch1 = 0; ch1 < 4; ch1++) {
        uint32_t v = smaller ? Lc(out, ch1) : Rc(out, ch1);
        if (v > max_links) max_links = v;
    }
    uint32_t mine = smaller ? Lc(out, ch) : Rc(out, ch ^ 2);
    if (mine < max_links) {
        (*tip_c)++; node1->B |= B_DELETED;
        dislink_prev(out, ch, smaller);
        if (n_in(out) == 1 && n_out(out) == 1) out->B |= B_LINEAR;
        return 1;
    }
    return 0;
}
/* removeSingleTips / removeMinorTips (cutTipPreGraph.c:363-488) */
static void remove_tips(octx_t *c, int thin_pass) {
    int cut = 2 * c->K, tip_c = 0;
    if (thin_pass) {
        for (int p = 0; p < c->P; p++) {
            oset_t *s = &c->sets[p];
            for (uint64_t i = 0; i < s->size; i++) {
                if (!s->occ[i]) continue;
                onode_t *n = &s->array[i];
                if (!(n->B & B_LINEAR) && !(n->B & B_DELETED) && (n->B & B_SINGLE)) clip_tip(c, n, cut, 1, &tip_c);
            }
        }
        remark_linear(c);
        return;
    }
    int flag = 1;
    while (flag) {
        flag = 0;
        for (int p = 0; p < c->P; p++) {
            oset_t *s = &c->sets[p];
            for (uint64_t i = 0; i < s->size; i++) {
                if (!s->occ[i]) continue;
                onode_t *n = &s->array[i];
                if (!(n->B & B_LINEAR) && !(n->B & B_DELETED)) flag += clip_tip(c, n, cut, 0, &tip_c);
            }
        }
    }
    remark_linear(c);
}

/* ------------------------------------------------------------------ edges (node2edge.c) */
typedef struct { onode_t *node; okmer_t kmer; int smaller; } bead_t;
typedef struct { bead_t *v; int n, cap; } beads_t;
static void beads_push(beads_t *b, onode_t *node, okmer_t kmer, int smaller) {
    if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 1024; b->v = realloc(b->v, b->cap * sizeof(bead_t)); }
    b->v[b->n].node = node; b->v[b->n].kmer = kmer; b->v[b->n].smaller = smaller; b->n++;
}
/* stringBeads (node2edge.c:86-218): follow linear nodes up to and including the first non-linear one */
static void string_beads(octx_t *c, beads_t *b, int nextch) {
    okmer_t word = k_next(b->v[0].kmer, nextch, c->filter), canon, bal;
    int smaller, ch;
    onode_t *out = lookup(c, word, &canon, &bal, &smaller);
    while (out && (out->B & B_LINEAR)) {
        okmer_t oriented = smaller ? canon : bal;
        beads_push(b, out, oriented, smaller);
        if (smaller) { for (ch = 0; ch < 4; ch++) if (Rc(out, ch)) break; word = k_next(oriented, ch, c->filter); }
        else         { for (ch = 0; ch < 4; ch++) if (Lc(out, ch)) break; word = k_next(oriented, ch ^ 2, c->filter); }
        out = lookup(c, word, &canon, &bal, &smaller);
    }
    if (!out) { fprintf(stderr, "oracle: edge walk lost a k-mer\n"); exit(1); }
    beads_push(b, out, smaller ? canon : bal, smaller);
}
/* print_kmer_gz (kmer.c:813-817 / 501-505) */
static void print_kmer(FILE *fp, okmer_t k, char c) {
    for (int i = 0; i < nw_g; i++) fprintf(fp, i ? " %llx" : "%llx", (unsigned long long)k.w[i]);
    fputc(c, fp);
}
/* merge_linearV2 + output_1edge (node2edge.c:430-609, output_pregraph.c:88-110); patch k-mers are pass-2 state
 * and not modelled here */
static void emit_edge(octx_t *c, beads_t *b, int bal_edge, int *edge_c, FILE *fe) {
    int count = b->n, length = count - 1;
    bead_t *first = &b->v[0], *second = &b->v[1], *last = &b->v[count - 1], *second_last = &b->v[count - 2];
    dislink_prev(last->node, k_first(second_last->kmer, c->K), last->smaller);
    dislink_next(first->node, k_last(second->kmer), first->smaller);
    (*edge_c)++;
    long long symbol = 0;
    for (int i = count - 2; i >= 1; i--) {
        onode_t *n = b->v[i].node;
        symbol += Lc(n, 0) + Lc(n, 1) + Lc(n, 2) + Lc(n, 3);
    }
    for (int i = count - 2; i >= 1; i--) {
        onode_t *n = b->v[i].node;
        uint32_t flags = n->B & 0x0FFFFFFFu;   /* keep r_links + linear/deleted/checked/single */
        uint32_t twin = b->v[i].smaller ? (uint32_t)(bal_edge + 1) : (uint32_t)(1 - bal_edge);
        n->B = flags | (twin << B_TWIN_SHIFT) | (1u << B_INEDGE_SHIFT);
        n->A = b->v[i].smaller ? (uint32_t)*edge_c : (uint32_t)(*edge_c + bal_edge);   /* edge id overwrites word A */
    }
    int cvg = 0;
    if (length > 1) { long long v = symbol / (length - 1) * 10; cvg = v > 16000 ? 16000 : (int)v; }
    fprintf(fe, ">length %d,", length);
    print_kmer(fe, first->kmer, ',');
    print_kmer(fe, last->kmer, ',');
    fprintf(fe, "cvg %d, %d\n", cvg, bal_edge);
    for (int i = 0; i < length; i++) {
        fputc("ACTG"[k_last(b->v[i + 1].kmer)], fe);
        if ((i + 1) % 100 == 0) fputc('\n', fe);
    }
    if (length % 100 != 0) fputc('\n', fe);
    *edge_c += bal_edge;
}
/* check_iden_kmerList (node2edge.c:624-649): list equals the reversed list of reverse complements */
static int is_palindrome(octx_t *c, beads_t *b) {
    for (int i = 0; i < b->n; i++) {
        okmer_t rc = k_rc(b->v[b->n - 1 - i].kmer, c->K);
        if (k_cmp(b->v[i].kmer, rc) != 0) return 0;
    }
    return 1;
}
/* make_edge + startEdgeFromNode (node2edge.c:237-411) */
static void make_edges(octx_t *c, const char *prefix) {
    char name[1024]; snprintf(name, sizeof name, "%s.edge", prefix);   /* the oracle writes plain text */
    FILE *fe = fopen(name, "w");
    beads_t b = {0, 0, 0};
    int edge_c = 0;
    for (int p = 0; p < c->P; p++) {
        oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            onode_t *n = &s->array[i];
            if ((n->B & B_LINEAR) || (n->B & B_DELETED)) continue;
            okmer_t word1 = n->seq, bal1 = k_rc(word1, c->K);
            for (int ch = 0; ch < 4; ch++) {
                if (!Rc(n, ch)) continue;
                b.n = 0; beads_push(&b, n, word1, 1);
                string_beads(c, &b, ch);
                emit_edge(c, &b, is_palindrome(c, &b) ? 0 : 1, &edge_c, fe);
            }
            for (int ch = 0; ch < 4; ch++) {
                if (!Lc(n, ch)) continue;
                b.n = 0; beads_push(&b, n, bal1, 0);
                string_beads(c, &b, ch ^ 2);
                emit_edge(c, &b, is_palindrome(c, &b) ? 0 : 1, &edge_c, fe);
            }
        }
    }
    fclose(fe); free(b.v);
    c->num_ed = edge_c;
}
/* output_vertex (output_pregraph.c:50-86) */
static void write_vertex(octx_t *c, const char *prefix) {
    char name[1024]; snprintf(name, sizeof name, "%s.vertex", prefix);
    FILE *fp = fopen(name, "w");
    int cnt = 0;
    for (int p = 0; p < c->P; p++) {
        oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            onode_t *n = &s->array[i];
            if ((n->B & B_LINEAR) || (n->B & B_DELETED)) continue;
            cnt++;
            print_kmer(fp, n->seq, ' ');
            if (cnt % 8 == 0) fputc('\n', fp);
        }
    }
    fputc('\n', fp); fclose(fp);
    c->num_vt = cnt;
    snprintf(name, sizeof name, "%s.preGraphBasic", prefix);
    fp = fopen(name, "w");
    fprintf(fp, "VERTEX %d K %d\n", cnt, c->K);
    fprintf(fp, "\nEDGEs %d\n", c->num_ed);
    fprintf(fp, "\nMaxReadLen %d MinReadLen %d MaxNameLen %d\n", c->max_read_len, 0, 256);
    fclose(fp);
}

/* ------------------------------------------------------------------ node dump after pass 1 (for the per-k-mer parity tests) */
uint64_t oracle_node_count(const octx_t *c) { uint64_t t = 0; for (int p = 0; p < c->P; p++) t += c->sets[p].count; return t; }
uint64_t oracle_kmer_count(const octx_t *c) { return c->n_kmers; }
uint64_t oracle_set_size(const octx_t *c, int p) { return c->sets[p].size; }
uint64_t oracle_set_count(const octx_t *c, int p) { return c->sets[p].count; }
uint64_t oracle_set_last_put(const octx_t *c, int p) { return c->last_put[p]; }
/* Dump every stored node in (set, slot) order: keys[NW*i..], A, B, first_ord, set id, slot. */
uint64_t oracle_dump_nodes(const octx_t *c, uint64_t *keys, uint32_t *A, uint32_t *B, uint64_t *ord, int32_t *setid, uint64_t *slot) {
    uint64_t o = 0;
    for (int p = 0; p < c->P; p++) {
        const oset_t *s = &c->sets[p];
        for (uint64_t i = 0; i < s->size; i++) {
            if (!s->occ[i]) continue;
            const onode_t *n = &s->array[i];
            for (int w = 0; w < c->NW; w++) keys[o * c->NW + w] = n->seq.w[w];
            A[o] = n->A; B[o] = n->B; ord[o] = n->first_ord; setid[o] = p; slot[o] = i; o++;
        }
    }
    return o;
}

/* Everything after the reads are in: [-d] filter, marking + .kmerFreq, tips, edges, vertex (pregraph.c:98-131) */
void oracle_finish(octx_t *c, const char *prefix) {
    nw_g = c->NW;
    if (c->D) delow(c);
    mark_and_freq(c, prefix);
    if (!c->D) remove_tips(c, 1);
    remove_tips(c, 0);
    make_edges(c, prefix);
    write_vertex(c, prefix);
}
/* pass-1-only variant for the counting parity tests */
void oracle_finish_count(octx_t *c, const char *prefix) {
    nw_g = c->NW;
    if (c->D) delow(c);
    mark_and_freq(c, prefix);
}
void oracle_destroy(octx_t *c) {
    for (int p = 0; p < c->P; p++) { free(c->sets[p].array); free(c->sets[p].occ); }
    free(c->sets); free(c);
}

/* One-call convenience: reads as a dense (n_reads x stride) matrix of base codes with per-read lengths. */
void oracle_pregraph(const uint8_t *codes, const int32_t *lens, int64_t n_reads, int64_t stride,
                     int K, int P, int D, int a_gb, int mer127, int max_read_len, const char *prefix) {
    octx_t *c = oracle_create(K, P, D, a_gb, mer127, max_read_len);
    for (int64_t r = 0; r < n_reads; r++) oracle_add_read(c, codes + r * stride, lens ? lens[r] : (int)stride);
    oracle_finish(c, prefix);
    oracle_destroy(c);
}
