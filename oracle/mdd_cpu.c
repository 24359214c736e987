/* libmdd_cpu.so -- the C ABI of include/mdd.h ("same symbols built twice ... libmdd_cpu.so (g++, oracle
 * twin)") over the CPU restatement of oracle/mdd_ref.c.
 *
 * TEST INFRASTRUCTURE.  This is the oracle wearing the product's interface: the same ctypes call sequence a caller
 * makes against libmdd_hip.so can be made against it, with HOST pointers where the HIP library takes device pointers,
 * and must give the same signatures, record keys, components and removed mask (tests/test_cpu_twin.py on the CPU,
 * tests/test_gpu_twin.py against the GPU).  Nothing in map_neo_amd/ loads it; the product has no CPU path.
 *
 * Exported: the hot-path subset of the interface -- mdd_version, mdd_last_error, mdd_ctx_create / _destroy / _sync,
 * mdd_make_perms, mdd_signatures, mdd_band_keys, mdd_bucket_edges, mdd_components_init / _union / _finish,
 * mdd_signature_matches, mdd_removed_mask.  Each follows the reference lines its declaration in include/mdd.h cites.
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mdd.h"

void ref_make_perms(uint32_t seed, int K, uint64_t* a, uint64_t* b);
uint64_t ref_signatures_mt(const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int unit, int width, int K,
                           const uint64_t* a, const uint64_t* b, uint32_t* sig_out, int n_threads);

struct mdd_ctx {
    int device;
    int threads;
};

static _Thread_local char g_err[512];
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return fail(MDD_ERR_INVALID, __VA_ARGS__); \
    } while (0)

int mdd_version(void) { return 100; }
const char* mdd_last_error(void) { return g_err; }

int mdd_ctx_create(int device, void* hip_stream, mdd_ctx** out) {
    (void)hip_stream;
    REQUIRE(out, "mdd_ctx_create: out is NULL");
    mdd_ctx* c = (mdd_ctx*)calloc(1, sizeof *c);
    if (!c) return fail(MDD_ERR_INVALID, "mdd_ctx_create: out of memory");
    c->device = device;
    const char* t = getenv("MDD_CPU_THREADS");
    c->threads = t && atoi(t) > 0 ? atoi(t) : 1;
    *out = c;
    return MDD_OK;
}
int mdd_ctx_destroy(mdd_ctx* ctx) {
    free(ctx);
    return MDD_OK;
}
int mdd_ctx_sync(mdd_ctx* ctx) {
    REQUIRE(ctx, "mdd_ctx_sync: ctx is NULL");
    return MDD_OK;
}

int mdd_make_perms(uint32_t seed, int K, uint64_t* a_host, uint64_t* b_host) {
    REQUIRE(a_host && b_host, "mdd_make_perms: NULL argument");
    REQUIRE(K >= 1 && K <= MDD_MAX_PERM, "mdd_make_perms: K=%d outside [1,%d]", K, MDD_MAX_PERM);
    ref_make_perms(seed, K, a_host, b_host);
    return MDD_OK;
}

int mdd_signatures(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs, int unit,
                   int width, int K, const uint64_t* a, const uint64_t* b, uint32_t* sig_out) {
    REQUIRE(ctx, "mdd_signatures: ctx is NULL");
    REQUIRE(K >= 1 && K <= MDD_MAX_PERM, "mdd_signatures: K=%d outside [1,%d]", K, MDD_MAX_PERM);
    REQUIRE(unit == MDD_UNIT_CHAR || unit == MDD_UNIT_WORD, "mdd_signatures: unknown unit %d", unit);
    REQUIRE(width >= 1 && width <= 128, "mdd_signatures: width=%d outside [1,128]", width);
    if (n_docs == 0) return MDD_OK;
    REQUIRE(text && offsets && a && b && sig_out, "mdd_signatures: NULL argument");
    for (uint64_t d = 0; d < n_docs; d++)
        REQUIRE(offsets[d] <= offsets[d + 1] && offsets[d + 1] <= text_bytes, "mdd_signatures: offsets[%llu] out of order or past text_bytes",
                (unsigned long long)d);
    ref_signatures_mt(text, offsets, n_docs, unit, width, K, a, b, sig_out, ctx->threads);
    return MDD_OK;
}

/* the surrogate of map_neo_amd/csrc/mdd_hip.hip::tuple_key (implementation-defined: it only places records; buckets
 * are decided on the exact tuple).  Restated so that the two libraries hand out identical (key, meta) records. */
static uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
static uint64_t tuple_key(const uint32_t* t, int rows, uint32_t band, int key_bits) {
    uint64_t h = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)band * 0xD6E8FEB86659FD93ULL);
    for (int i = 0; i < rows; i++) h = mix64(h ^ t[i]) + 0x9E3779B97F4A7C15ULL;
    h = mix64(h);
    if (key_bits < 64) h &= (1ULL << key_bits) - 1;
    return h ? h : 1ULL;
}

int mdd_band_keys(mdd_ctx* ctx, const uint32_t* sig, uint64_t n_docs, int K, int bands, int rows, uint64_t doc_base, int key_bits,
                  uint64_t* key, uint64_t* meta) {
    REQUIRE(ctx, "mdd_band_keys: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(sig && key && meta, "mdd_band_keys: NULL argument");
    REQUIRE(bands >= 1 && bands <= MDD_MAX_BANDS && rows >= 1 && (int64_t)bands * rows <= K, "mdd_band_keys: bands=%d rows=%d do not fit K=%d",
            bands, rows, K);
    REQUIRE(key_bits >= 1 && key_bits <= 64, "mdd_band_keys: key_bits=%d outside [1,64]", key_bits);
    REQUIRE(doc_base + n_docs <= 0xFFFFFFFFull, "mdd_band_keys: global document index exceeds 2^32-1");
    for (uint64_t d = 0; d < n_docs; d++)
        for (int j = 0; j < bands; j++) { /* EN:66-71: band j = hashvalues[j*r : (j+1)*r] */
            key[d * bands + j] = tuple_key(sig + d * (uint64_t)K + (uint64_t)j * rows, rows, (uint32_t)j, key_bits);
            meta[d * bands + j] = ((uint64_t)j << 56) | (doc_base + d);
        }
    return MDD_OK;
}

/* EN:136 groupBy((band, key)) / ZH-2:30-37: exact equality of (band, r-tuple).  Open addressing on the surrogate, the
 * entry holds the lowest record index of the bucket so far; records are visited in index order, so "the lowest-indexed
 * record of its bucket" is the first one met and every later one emits (its document, that record's document). */
int mdd_bucket_edges(mdd_ctx* ctx, const uint64_t* key, const uint64_t* meta, uint64_t n_rec, const uint32_t* tuple_base,
                     uint64_t rec_per_doc, uint64_t doc_stride, uint64_t band_stride, int rows, uint64_t* table, uint64_t tab_size,
                     uint32_t* edge_u, uint32_t* edge_v, uint64_t edge_cap, uint64_t* n_edges, uint64_t* n_coll, uint64_t* rec_bucket) {
    REQUIRE(ctx && n_edges && n_coll, "mdd_bucket_edges: NULL argument");
    *n_edges = 0;
    *n_coll = 0;
    if (n_rec == 0) return MDD_OK;
    REQUIRE(key && meta && tuple_base && table && (edge_cap == 0 || (edge_u && edge_v)), "mdd_bucket_edges: NULL buffer");
    REQUIRE(rows >= 1 && rec_per_doc >= 1, "mdd_bucket_edges: rows / rec_per_doc must be positive");
    REQUIRE(tab_size >= 2 && (tab_size & (tab_size - 1)) == 0 && (double)tab_size >= 1.25 * (double)n_rec,
            "mdd_bucket_edges: tab_size=%llu must be a power of two >= 1.25 * n_rec", (unsigned long long)tab_size);
    for (uint64_t s = 0; s < tab_size; s++) table[s] = ~0ULL;
    for (uint64_t i = 0; i < n_rec; i++) {
        const uint32_t* ti = tuple_base + (i / rec_per_doc) * doc_stride + (i % rec_per_doc) * band_stride;
        uint64_t slot = mix64(key[i]) & (tab_size - 1);
        for (;;) {
            uint64_t head = table[slot];
            if (head == ~0ULL) {
                table[slot] = i;
                if (rec_bucket) rec_bucket[i] = i;
                break;
            }
            const uint32_t* th = tuple_base + (head / rec_per_doc) * doc_stride + (head % rec_per_doc) * band_stride;
            if ((meta[head] >> 56) == (meta[i] >> 56) && memcmp(th, ti, (size_t)rows * sizeof(uint32_t)) == 0) {
                uint32_t du = (uint32_t)meta[i], dv = (uint32_t)meta[head];
                if (du != dv) {
                    if (*n_edges < edge_cap) {
                        edge_u[*n_edges] = du;
                        edge_v[*n_edges] = dv;
                    }
                    (*n_edges)++;
                }
                if (rec_bucket) rec_bucket[i] = head;
                break;
            }
            if (key[head] == key[i]) (*n_coll)++;
            slot = (slot + 1) & (tab_size - 1);
        }
    }
    return MDD_OK;
}

/* EN:82-115,138-145 / ZH-3:11-27: label = smallest document index of the component */
static uint32_t uf_find(uint32_t* p, uint32_t x) {
    while (p[x] != x) {
        p[x] = p[p[x]];
        x = p[x];
    }
    return x;
}
int mdd_components_init(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs) {
    REQUIRE(ctx && (labels || n_docs == 0), "mdd_components_init: NULL argument");
    REQUIRE(n_docs <= 0xFFFFFFFFull, "mdd_components_init: n_docs exceeds 2^32-1");
    for (uint64_t d = 0; d < n_docs; d++) labels[d] = (uint32_t)d;
    return MDD_OK;
}
int mdd_components_union(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs, const uint32_t* edge_u, const uint32_t* edge_v, uint64_t n_edges) {
    REQUIRE(ctx, "mdd_components_union: ctx is NULL");
    if (n_edges == 0) return MDD_OK;
    REQUIRE(labels && edge_u && edge_v, "mdd_components_union: NULL buffer");
    for (uint64_t e = 0; e < n_edges; e++) {
        if (edge_u[e] >= n_docs || edge_v[e] >= n_docs) continue; /* as the device kernel: edges outside [0, n_docs) are ignored */
        uint32_t ru = uf_find(labels, edge_u[e]), rv = uf_find(labels, edge_v[e]);
        if (ru < rv) labels[rv] = ru;
        else if (rv < ru) labels[ru] = rv;
    }
    return MDD_OK;
}
int mdd_components_finish(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs) {
    REQUIRE(ctx && (labels || n_docs == 0), "mdd_components_finish: NULL argument");
    for (uint64_t d = 0; d < n_docs; d++) labels[d] = uf_find(labels, (uint32_t)d);
    return MDD_OK;
}

int mdd_signature_matches(mdd_ctx* ctx, const uint32_t* sig, uint64_t n_docs, int K, const uint32_t* edge_u, const uint32_t* edge_v,
                          uint64_t n_edges, uint32_t* matches) {
    REQUIRE(ctx, "mdd_signature_matches: ctx is NULL");
    REQUIRE(K >= 1 && K <= 4096, "mdd_signature_matches: K=%d outside [1,4096]", K);
    if (n_edges == 0) return MDD_OK;
    REQUIRE(sig && edge_u && edge_v && matches, "mdd_signature_matches: NULL buffer");
    for (uint64_t e = 0; e < n_edges; e++) {
        uint32_t c = 0;
        if (edge_u[e] < n_docs && edge_v[e] < n_docs)
            for (int i = 0; i < K; i++) c += sig[(uint64_t)edge_u[e] * K + i] == sig[(uint64_t)edge_v[e] * K + i];
        matches[e] = c;
    }
    return MDD_OK;
}

/* EN:145-149 / ZH-4:17-22 */
int mdd_removed_mask(mdd_ctx* ctx, const uint32_t* labels, uint64_t first, uint64_t n, uint8_t* removed, uint64_t* n_removed) {
    REQUIRE(ctx && n_removed, "mdd_removed_mask: NULL argument");
    *n_removed = 0;
    if (n == 0) return MDD_OK;
    REQUIRE(labels && removed, "mdd_removed_mask: NULL buffer");
    for (uint64_t d = 0; d < n; d++) {
        removed[d] = labels[first + d] != (uint32_t)(first + d);
        *n_removed += removed[d];
    }
    return MDD_OK;
}
