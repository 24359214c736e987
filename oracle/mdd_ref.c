/* CPU ORACLE (test infrastructure, NOT product code) -- plain-C restatement of MAP-NEO
 * Matrix MinHash-LSH near-dedup.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file (oracle/libmdd_ref.so).
 *
 * Reference (paths relative to /root/reference):
 *   EN   = Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py
 *   ZH-1 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_minhash.py
 *   ZH-2 = .../minhash_lsh/generate_dup_pairs.py
 * Third-party arithmetic restated (not vendored by the reference, no pinned version):
 *   datasketch MinHash (seed=1, sha1_hash32, p=2^61-1, max_hash=2^32-1), nltk.ngrams,
 *   numpy legacy RandomState (MT19937 init_genrand + masked-rejection bounded uint64).
 * PARITY PIN STATUS: "parity unpinned" w.r.t. a running datasketch; pinned against
 * oracle/ref_py.py (numpy + hashlib), the SHA-1 FIPS vectors and SURVEY.md 8(c) KATs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MDD_P61 ((1ULL << 61) - 1)

/* ------------------------------------------------------------------ SHA-1 (FIPS 180-4) */
static inline uint32_t rol32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

static void sha1_block(uint32_t st[5], const uint8_t blk[64]) {
    uint32_t w[80];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 80; i++) w[i] = rol32(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4];
    for (int i = 0; i < 80; i++) {
        uint32_t f, k;
        if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
        else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
        else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
        else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
        uint32_t t = rol32(a, 5) + f + e + k + w[i];
        e = d; d = c; c = rol32(b, 30); b = a; a = t;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e;
}

/* full 20-byte digest (for the FIPS self-test) */
void ref_sha1(const uint8_t* msg, uint64_t len, uint8_t out[20]) {
    uint32_t st[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    uint64_t i = 0;
    for (; i + 64 <= len; i += 64) sha1_block(st, msg + i);
    uint8_t blk[128];
    uint64_t rem = len - i;
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg + i, rem);
    blk[rem] = 0x80;
    int nb = (rem + 9 > 64) ? 2 : 1;
    uint64_t bits = len * 8;
    for (int k = 0; k < 8; k++) blk[64 * nb - 1 - k] = (uint8_t)(bits >> (8 * k));
    for (int k = 0; k < nb; k++) sha1_block(st, blk + 64 * k);
    for (int k = 0; k < 5; k++) {
        out[4 * k] = st[k] >> 24; out[4 * k + 1] = st[k] >> 16;
        out[4 * k + 2] = st[k] >> 8; out[4 * k + 3] = st[k];
    }
}

/* datasketch.hashfunc.sha1_hash32: struct.unpack('<I', sha1(data).digest()[:4])[0] */
uint32_t ref_sha1_hash32(const uint8_t* msg, uint64_t len) {
    uint8_t d[20];
    ref_sha1(msg, len, d);
    return (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24);
}

/* ------------------------------------------------ numpy legacy RandomState(seed).randint */
typedef struct { uint32_t mt[624]; int idx; } mt_t;

static void mt_seed(mt_t* s, uint32_t seed) { /* numpy _legacy_seeding -> init_genrand */
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static uint32_t mt_next32(mt_t* s) {
    if (s->idx >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (s->mt[k] & 0x80000000u) | (s->mt[(k + 1) % 624] & 0x7fffffffu);
            s->mt[k] = s->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static uint64_t mt_next64(mt_t* s) { uint64_t hi = mt_next32(s); uint64_t lo = mt_next32(s); return (hi << 32) | lo; }
/* randint(low, high, dtype=uint64) for a range wider than 32 bits: masked rejection */
static uint64_t mt_randint64(mt_t* s, uint64_t low, uint64_t high_excl) {
    uint64_t rng = high_excl - low - 1, mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint64_t v;
    do { v = mt_next64(s) & mask; } while (v > rng);
    return low + v;
}

/* datasketch MinHash._init_permutations (a4): interleaved a_i in [1,p), b_i in [0,p) */
void ref_make_perms(uint32_t seed, int K, uint64_t* a, uint64_t* b) {
    mt_t s; mt_seed(&s, seed);
    for (int i = 0; i < K; i++) { a[i] = mt_randint64(&s, 1, MDD_P61); b[i] = mt_randint64(&s, 0, MDD_P61); }
}

/* ------------------------------------------------------------------ shingling (a2) */
/* unit 0 = char (EN:61 / ZH-1:27: code-point windows), unit 1 = word (extension: runs of
 * non-0x20 bytes, windows of `width` tokens spanning first-token-start .. last-token-end).
 * Calls cb(ctx, ptr, len) per shingle. */
typedef void (*shingle_cb)(void*, const uint8_t*, uint64_t);

static void for_each_shingle(const uint8_t* t, uint64_t n, int unit, int width, shingle_cb cb, void* ctx) {
    uint64_t* st = (uint64_t*)malloc((n + 2) * sizeof(uint64_t));
    uint64_t* en = NULL;
    uint64_t m = 0;
    if (unit == 0) {
        for (uint64_t i = 0; i < n; i++) if ((t[i] & 0xC0) != 0x80) st[m++] = i;
        st[m] = n;
        for (uint64_t j = 0; j + width <= m; j++) cb(ctx, t + st[j], st[j + width] - st[j]);
    } else {
        en = (uint64_t*)malloc((n + 2) * sizeof(uint64_t));
        uint64_t i = 0;
        while (i < n) {
            while (i < n && t[i] == 0x20) i++;
            if (i >= n) break;
            st[m] = i;
            while (i < n && t[i] != 0x20) i++;
            en[m++] = i;
        }
        for (uint64_t j = 0; j + width <= m; j++) cb(ctx, t + st[j], en[j + width - 1] - st[j]);
        free(en);
    }
    free(st);
}

/* ------------------------------------------------------------------ signature (a3+a5) */
typedef struct { int K; const uint64_t* a; const uint64_t* b; uint64_t* hv; uint64_t n; } sig_ctx;

static void sig_update(void* vctx, const uint8_t* p, uint64_t len) {
    sig_ctx* c = (sig_ctx*)vctx;
    uint64_t h = ref_sha1_hash32(p, len);
    for (int i = 0; i < c->K; i++) {
        /* numpy uint64: (a*hv + b) wraps mod 2^64 BEFORE % p */
        uint64_t phv = ((c->a[i] * h + c->b[i]) % MDD_P61) & 0xFFFFFFFFULL;
        if (phv < c->hv[i]) c->hv[i] = phv;
    }
    c->n++;
}

/* one document -> K u32 hashvalues; returns the number of shingles */
uint64_t ref_signature(const uint8_t* text, uint64_t len, int unit, int width, int K,
                       const uint64_t* a, const uint64_t* b, uint32_t* sig_out) {
    uint64_t hv[1024];
    if (K > 1024) return (uint64_t)-1;
    for (int i = 0; i < K; i++) hv[i] = 0xFFFFFFFFULL;
    sig_ctx c = {K, a, b, hv, 0};
    for_each_shingle(text, len, unit, width, sig_update, &c);
    for (int i = 0; i < K; i++) sig_out[i] = (uint32_t)hv[i];
    return c.n;
}

/* batch over packed text + offsets[n+1]; returns total shingles */
uint64_t ref_signatures(const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int unit,
                        int width, int K, const uint64_t* a, const uint64_t* b, uint32_t* sig_out) {
    uint64_t tot = 0;
    for (uint64_t d = 0; d < n_docs; d++)
        tot += ref_signature(text + offsets[d], offsets[d + 1] - offsets[d], unit, width, K, a, b,
                             sig_out + d * (uint64_t)K);
    return tot;
}

/* same, documents spread over n_threads host threads (OpenMP): the "optimised CPU" line of bench.py's
 * cpu_baseline; ref_signature keeps no global state */
uint64_t ref_signatures_mt(const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int unit, int width, int K,
                           const uint64_t* a, const uint64_t* b, uint32_t* sig_out, int n_threads) {
    uint64_t tot = 0;
    int64_t d;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : tot)
    for (d = 0; d < (int64_t)n_docs; d++)
        tot += ref_signature(text + offsets[d], offsets[d + 1] - offsets[d], unit, width, K, a, b,
                             sig_out + (uint64_t)d * (uint64_t)K);
    return tot;
}

/* shingle hashes only (for kernel-stage parity tests): writes up to cap hashes, returns count */
typedef struct { uint32_t* out; uint64_t cap; uint64_t n; } hash_ctx;
static void hash_collect(void* vctx, const uint8_t* p, uint64_t len) {
    hash_ctx* c = (hash_ctx*)vctx;
    if (c->n < c->cap) c->out[c->n] = ref_sha1_hash32(p, len);
    c->n++;
}
uint64_t ref_shingle_hashes(const uint8_t* text, uint64_t len, int unit, int width, uint32_t* out, uint64_t cap) {
    hash_ctx c = {out, cap, 0};
    for_each_shingle(text, len, unit, width, hash_collect, &c);
    return c.n;
}

/* ------------------------------------------------------------------ buckets + CC (a9-a13) */
typedef struct { const uint32_t* sig; int K; int rows; uint64_t n; } cmp_env;
static cmp_env g_env;
static int rec_cmp(const void* x, const void* y) {
    uint64_t rx = *(const uint64_t*)x, ry = *(const uint64_t*)y; /* rec = band * n + doc */
    uint64_t bx = rx / g_env.n, by = ry / g_env.n;
    if (bx != by) return bx < by ? -1 : 1;
    const uint32_t* tx = g_env.sig + (rx % g_env.n) * g_env.K + bx * g_env.rows;
    const uint32_t* ty = g_env.sig + (ry % g_env.n) * g_env.K + by * g_env.rows;
    for (int i = 0; i < g_env.rows; i++) if (tx[i] != ty[i]) return tx[i] < ty[i] ? -1 : 1;
    return rx < ry ? -1 : (rx > ry);
}
static uint32_t uf_find(uint32_t* p, uint32_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}
/* Exact-equality bucketing on (band, r-tuple) (EN:136 / ZH-2:30-37), chain edges inside each
 * bucket in ascending doc order (EN:73-80), union-find with label = min doc index of the
 * component (what EN:138-145 converges to).  Writes labels[n]; returns the number of edges
 * (bucket members beyond the first, summed over buckets), and optionally the bucket id of
 * each record (bucket_of[band*n+doc] = smallest record index in the same bucket). */
uint64_t ref_cluster(const uint32_t* sig, uint64_t n, int K, int bands, int rows,
                     uint32_t* labels, uint64_t* bucket_of) {
    uint64_t nrec = n * (uint64_t)bands, n_edges = 0;
    uint64_t* rec = (uint64_t*)malloc(nrec * sizeof(uint64_t));
    for (uint64_t i = 0; i < nrec; i++) rec[i] = i;
    g_env.sig = sig; g_env.K = K; g_env.rows = rows; g_env.n = n;
    qsort(rec, nrec, sizeof(uint64_t), rec_cmp);
    for (uint64_t d = 0; d < n; d++) labels[d] = (uint32_t)d;
    uint64_t head = 0;
    for (uint64_t i = 0; i < nrec; i++) {
        if (i > 0) {
            uint64_t prev = rec[i - 1], cur = rec[i];
            int same = (prev / n == cur / n);
            if (same) {
                const uint32_t* tp = sig + (prev % n) * K + (prev / n) * rows;
                const uint32_t* tc = sig + (cur % n) * K + (cur / n) * rows;
                same = memcmp(tp, tc, rows * sizeof(uint32_t)) == 0;
            }
            if (same) {
                uint32_t ru = uf_find(labels, (uint32_t)(prev % n)), rv = uf_find(labels, (uint32_t)(cur % n));
                if (ru != rv) { if (ru < rv) labels[rv] = ru; else labels[ru] = rv; }
                n_edges++;
            } else head = i;
        }
        if (bucket_of) bucket_of[rec[i]] = rec[head];
    }
    for (uint64_t d = 0; d < n; d++) labels[d] = uf_find(labels, (uint32_t)d);
    free(rec);
    return n_edges;
}
