/* mdd.h -- C ABI of libmdd_hip.so: MI355X (gfx950) MinHash-LSH near-dedup hot path.
 *
 * Drop-in boundary for the document near-deduplication stage of MAP-NEO's Matrix pipeline.
 * The reference has no FFI of its own (the stage is Python scripts whose arithmetic lives in
 * datasketch / hashlib / numpy); each entry point below names the reference
 * function(s) it replaces.  Reference paths are relative to /root/reference:
 *   EN   = Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py
 *   ZH-1 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_minhash.py
 *   ZH-2 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_dup_pairs.py
 *   ZH-3 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_connected_components.py
 *   ZH-4 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_dup_line_id_for_each_file.py
 *
 * Conventions
 *  - every function returns 0 (MDD_OK) or a negative error code; the message of the last
 *    error on the calling thread is returned by mdd_last_error().  No C++ exception and no
 *    torch type crosses this boundary: plain pointers and sizes only.
 *  - unless a parameter says "host", every pointer is a DEVICE pointer on the context's GPU.
 *    The caller owns every buffer (inputs, outputs AND workspaces); the compute entry points
 *    allocate nothing, keep no pointer after they return, and enqueue their kernels on the
 *    context's stream (asynchronous; call mdd_ctx_sync or synchronise the stream yourself).
 *  - a context is bound to one (device, stream) pair and is not thread-safe: one per rank.
 *  - there is no CPU fallback: on a machine without a usable GPU mdd_ctx_create fails.
 */
#ifndef MDD_H
#define MDD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDD_OK 0
#define MDD_ERR_INVALID (-1)     /* bad argument (null pointer, size out of range, misalignment) */
#define MDD_ERR_HIP (-2)         /* a HIP runtime call failed */
#define MDD_ERR_UNSUPPORTED (-3) /* parameter combination this build does not implement */

#define MDD_UNIT_CHAR 0 /* shingle unit = Unicode code point (EN:61, ZH-1:27: nltk.ngrams over a str) */
#define MDD_UNIT_WORD 1 /* shingle unit = space-separated token (BASELINE.json config 1; an extension: the reference has
                         * no word unit, so its parity is unpinned).  DEFINITION: tokens are maximal runs of bytes other than
                         * 0x20, and a shingle is the TEXT SPAN from the first byte of its first token to the last byte of its
                         * last token (no copy of the tokens is made).  On single-space-normalised text (what EN:31-40
                         * produces) that is exactly " ".join(tokens[i:i+width]), i.e. nltk.ngrams over text.split(" ");
                         * on text with runs of spaces the run stays inside the shingle.  Normalise first (--normalize en)
                         * when the nltk-equivalent is wanted. */

#define MDD_TEXT_PAD 128 /* bytes the text allocation must extend past text_bytes (kernels load unclamped) */
#define MDD_MAX_PERM 256
#define MDD_MAX_BANDS 64
#define MDD_NO_EDGE 0xFFFFFFFFu

typedef struct mdd_ctx mdd_ctx;

/* Library identification / error reporting. */
int mdd_version(void);
const char* mdd_last_error(void);

/* Context = (device, stream).  `hip_stream` is the hipStream_t every kernel of this context
 * is enqueued on, e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's default (null)
 * stream.  The library never creates a stream of its own, so its kernels are ordered with
 * the caller's copies and kernels on that stream. */
int mdd_ctx_create(int device, void* hip_stream, mdd_ctx** out);
int mdd_ctx_destroy(mdd_ctx* ctx);
int mdd_ctx_sync(mdd_ctx* ctx);
/* Long documents: mdd_signatures cuts a document longer than `seg_bytes` (default 16384, a
 * multiple of 256; 0 restores the default) into segments hashed by different waves whose
 * minima are combined with atomicMin -- results are identical, only load balance changes.
 * The context owns the small segment queue (8 MiB, allocated by mdd_ctx_create). */
int mdd_ctx_set_segment_bytes(mdd_ctx* ctx, uint32_t seg_bytes);
/* Permute-phase filter of mdd_signatures (no counterpart in the reference; results never
 * change, only speed).  Once a work item has hashed `n_shingles - 1` shingles, a shingle hash
 * is first tested with the low product only (u = lo32(a*h + b + 1) + 7 > sat(min + 6) proves it
 * cannot lower that permutation's minimum) and the full a*h+b fold runs only for hashes some
 * permutation can still use.  0 = default (256 shingles per 64 permutations of K);
 * 1 = filter from the first shingle; 0xFFFFFFFF = never filter.  A tuning / testing knob. */
int mdd_ctx_set_filter_from(mdd_ctx* ctx, uint32_t n_shingles);

/* Permutation table.  Replaces datasketch MinHash.__init__/_init_permutations, which the
 * reference re-runs for every document through `MinHash(num_perm=128)` (EN:62, ZH-1:28):
 * numpy legacy RandomState(seed); for i<K: a_i = randint(1, 2^61-1), b_i = randint(0, 2^61-1)
 * (interleaved draws, dtype uint64).  HOST pointers, K <= MDD_MAX_PERM.  Needs no GPU. */
int mdd_make_perms(uint32_t seed, int K, uint64_t* a_host, uint64_t* b_host);

/* Signatures.  Replaces generate_minhash_value (EN:56-64) and the first half of
 * generate_hash_values (ZH-1:23-29) for a whole batch of already-normalised documents:
 * shingle (code-point or token windows of `width`, no padding) -> utf-8 bytes -> SHA-1 ->
 * hv = little-endian u32 of digest[0:4] -> for i<K:
 *   sig[i] = min(sig[i], ((a_i*hv + b_i) mod 2^64) mod (2^61-1) & 0xFFFFFFFF), init 0xFFFFFFFF.
 *  text       packed utf-8 of all documents, 4-byte aligned; the allocation must extend
 *             MDD_TEXT_PAD bytes past text_bytes (padding content is ignored)
 *  offsets    n_docs+1 ascending byte offsets into text (document d = [offsets[d], offsets[d+1]))
 *             each document must be shorter than 4 GiB
 *  a, b       device copies of the permutation table (K entries each)
 *  sig_out    n_docs*K u32, row-major (the reference keeps uint64 hashvalues < 2^32)
 * 1 <= K <= 256; char: 1 <= width <= 128; word: 1 <= width <= 128. */
int mdd_signatures(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets,
                   uint64_t n_docs, int unit, int width, int K, const uint64_t* a, const uint64_t* b,
                   uint32_t* sig_out);

/* Allocation-free variant of mdd_signatures.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory around the call -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same arguments, same results (bit for bit) as mdd_signatures; the segment queue and the two counters
 * its kernels need live in a file-scope `static __device__` array of the library (2 MiB + 256 B per GPU, held until
 * the process ends; its device address is resolved once per device by mdd_ctx_create with hipGetSymbolAddress)
 * instead of the context's hipMalloc'd queue.  The call itself performs no hipMalloc / hipFree and calls no
 * allocating routine: it only enqueues one 16-byte memset and two kernels on the context's stream.
 * Limit: at most 2^18 queued segments, i.e. text_bytes / segment_bytes <= 262144 (4 GiB of text at the default
 * 16 KiB segments; raise mdd_ctx_set_segment_bytes for more) -- beyond that the call fails with MDD_ERR_INVALID and
 * enqueues nothing (mdd_signatures would lengthen the segments instead).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two contexts / streams) race; calls
 * on one stream are ordered and each call clears the counters it needs. */
int mdd_signatures_noalloc(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets,
                           uint64_t n_docs, int unit, int width, int K, const uint64_t* a, const uint64_t* b,
                           uint32_t* sig_out);

/* mdd_signatures for a corpus in PINNED HOST memory (the measured region starts at packed text in
 * pinned host memory).  No counterpart in the reference (its text lives in Python strings).  ONE kernel launch on
 * the context's stream hashes the documents while the text is still arriving: the call enqueues the chunked
 * host-to-device copies of text_host[0, text_bytes) into text_dev on `copy_stream` (a hipStream_t other than the
 * context's stream; created with hipStreamNonBlocking when the context runs on the default NULL stream, which would
 * otherwise synchronise with it implicitly -- the call checks), back to back, with an event behind each chunk; behind
 * that event, on a stream of the context's own, a host function (hipLaunchHostFunc) publishes the number of bytes that
 * have landed in a pinned host word the kernel polls (round 3: on the copy stream itself the host functions kept the
 * copy engine waiting for a host thread between any two chunks); `copy_stream` ends up behind the last of them.  A wave
 * starts a document only when every byte it can touch has landed.  Results are those of
 * mdd_signatures on the same text, bit for bit.
 *   text_host   HOST pointer, pinned (hipHostMalloc / hipHostRegister / torch pin_memory)
 *   text_dev    device buffer, 256-byte aligned, text_bytes + MDD_TEXT_PAD capacity; holds the text afterwards
 *   offsets     DEVICE, n_docs+1 (copy them first, on the context's stream)
 *   chunk_bytes copy granularity (0 = 16 MiB; a multiple of 256, >= 1 MiB)
 * Asynchronous.  The kernel cannot wait forever: a wave that has waited 4 s for its text sets a status word and
 * every wave of that launch leaves; the verdict is sticky -- the next mdd_ctx_sync returns MDD_ERR_HIP (and clears it)
 * however many streamed calls were enqueued in between, and the signatures of all of them are then suspect.  The first
 * call on a context allocates a small pinned table, a stream and three events (set-up, all or nothing), and a call
 * creates one event per chunk the first time it needs it; after that calls allocate nothing. */
int mdd_signatures_from_host(mdd_ctx* ctx, void* copy_stream, const uint8_t* text_host, uint64_t text_bytes,
                             uint8_t* text_dev, const uint64_t* offsets, uint64_t n_docs, int unit, int width, int K,
                             const uint64_t* a, const uint64_t* b, uint32_t* sig_out, uint64_t chunk_bytes);

/* Statistics of the most recent mdd_signatures_from_host on this context: the 100 MHz wall-clock ticks its waves
 * spent waiting for text, summed over waves, in *ticks_host (HOST).  Synchronises the context's stream. */
int mdd_stream_wait_ticks(mdd_ctx* ctx, uint64_t* ticks_host);

/* Stage probe for parity tests: the shingle hashes (a2+a3: datasketch sha1_hash32 of every
 * shingle, in shingle order) of ONE document [doc_begin, doc_end) of `text`.  Writes at most
 * `cap` hashes; *n_out (device u64) receives the number of shingles of the document. */
int mdd_shingle_hashes(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, uint64_t doc_begin,
                       uint64_t doc_end, int unit, int width, uint32_t* out, uint64_t cap, uint64_t* n_out);

/* Band records.  Replaces generate_bands (EN:66-71) / tail of generate_hash_values
 * (ZH-1:30-33): record i = doc*bands + band carries the r-tuple sig[doc][band*rows ..+rows).
 * The reference serialises the tuple as 8*rows big-endian bytes; here the tuple stays in
 * `sig` and the record gets
 *   key[i]  = 64-bit surrogate hash of (band, tuple), never 0 (used for hashing only --
 *             bucket identity is decided on the exact tuple in mdd_bucket_edges)
 *   meta[i] = (band << 56) | (doc_base + doc)      (global document index, < 2^32)
 * bands*rows <= K, bands <= MDD_MAX_BANDS.  `key_bits` (1..64) truncates the surrogate
 * (tests use a few bits to force surrogate collisions); pass 64. */
int mdd_band_keys(mdd_ctx* ctx, const uint32_t* sig, uint64_t n_docs, int K, int bands, int rows,
                  uint64_t doc_base, int key_bits, uint64_t* key, uint64_t* meta);

/* Same surrogate for records in PACKED layout (after the multi-GPU exchange): tuple of record
 * i = tuples[i*rows ..+rows), band = meta[i] >> 56. */
int mdd_packed_keys(mdd_ctx* ctx, const uint32_t* tuples, const uint64_t* meta, uint64_t n_rec, int rows,
                    int key_bits, uint64_t* key);

/* Multi-GPU routing (new design; no counterpart in the reference, which
 * leaves the shuffle to Spark EN:136 / the filesystem ZH-1:58-61): destination rank of a
 * record = (key >> 32) mod world.  mdd_route_count histograms records per rank into
 * counts[world] (device u64, zeroed by the call); mdd_route_pack writes the records in
 * PACKED layout grouped by rank at send_offsets[rank] (device u64[world], exclusive prefix of
 * counts, advanced by the call -- pass a scratch copy). */
int mdd_route_count(mdd_ctx* ctx, const uint64_t* key, uint64_t n_rec, int world, uint64_t* counts);
int mdd_route_pack(mdd_ctx* ctx, const uint32_t* sig, int K, int bands, int rows, const uint64_t* key,
                   const uint64_t* meta, uint64_t n_rec, int world, uint64_t* send_cursor,
                   uint32_t* tuples_out, uint64_t* meta_out);

/* Multi-GPU exchange (no counterpart in the reference, which leaves the shuffle to Spark EN:136,
 * 140-142 / the filesystem ZH-1:58-61).  One process per GPU; the library issues the collectives itself, on the
 * context's stream, against the RCCL the process has loaded (bound at run time with dlopen; torch's copy when torch is
 * loaded).  mdd_comm_unique_id (rank 0; HOST buffer of MDD_COMM_ID_BYTES, to be handed to every rank by any means, e.g.
 * torch.distributed's store) + mdd_comm_init (collective: every rank calls it) create the communicator of a context;
 * mdd_ctx_destroy / mdd_comm_destroy release it.  world <= 64.
 *
 *   exchange_counts  send_counts: DEVICE u64[world] (records this rank holds for every owner: mdd_route_count).  Writes
 *                    them to send_counts_host and what every peer holds for this rank to recv_counts_host (HOST
 *                    u64[world] each).  SYNCHRONOUS -- the one host synchronisation of the exchange: send / recv sizes
 *                    are host arguments and these numbers are computed on the device.
 *   exchange_bands   band records in PACKED layout grouped by owner (mdd_route_pack) -> the records this rank owns,
 *                    grouped by sender: tuples (rows u32 per record) and meta travel in the same send/recv group, in
 *                    rounds of at most 512 MiB per pair and array.  recv buffers hold sum(recv_counts_host) records.
 *   allgather_counts *n_local (DEVICE u64) of every rank -> counts_host (HOST u64[world]).  SYNCHRONOUS.
 *   allgather_edges  edges of all ranks, rank by rank, into all_u / all_v (sum(counts_host) entries). */
#define MDD_COMM_ID_BYTES 128
int mdd_comm_unique_id(uint8_t* id_host);
int mdd_comm_init(mdd_ctx* ctx, const uint8_t* id_host, int rank, int world);
int mdd_comm_destroy(mdd_ctx* ctx);
int mdd_exchange_counts(mdd_ctx* ctx, const uint64_t* send_counts, uint64_t* send_counts_host, uint64_t* recv_counts_host);
int mdd_exchange_bands(mdd_ctx* ctx, const uint32_t* tuples_send, const uint64_t* meta_send, const uint64_t* send_counts_host,
                       const uint64_t* recv_counts_host, int rows, uint32_t* tuples_recv, uint64_t* meta_recv);
int mdd_allgather_counts(mdd_ctx* ctx, const uint64_t* n_local, uint64_t* counts_host);
int mdd_allgather_edges(mdd_ctx* ctx, const uint32_t* edge_u, const uint32_t* edge_v, const uint64_t* counts_host,
                        uint32_t* all_u, uint32_t* all_v);

/* Bucketing + edges.  Replaces groupBy((band,key)) + generate_edges (EN:136, EN:73-80) and
 * the per-band dict of ZH-2:19-40.  Bucket identity = exact equality of (band, r-tuple).
 * Records are described by (key, meta) plus tuple addressing:
 *   tuple(i) = tuple_base + (i / rec_per_doc)*doc_stride + (i % rec_per_doc)*band_stride
 *   SIG layout:    tuple_base = sig, rec_per_doc = bands, doc_stride = K, band_stride = rows
 *   PACKED layout: tuple_base = tuples, rec_per_doc = 1, doc_stride = rows, band_stride = 0
 * Workspace (caller-allocated): table u64[tab_size], tab_size a power of two >= 1.25*n_rec (1.5 - 2 x n_rec is the
 * sensible range: open addressing with linear probing, one atomic per probe).  The call clears it.
 * Output: every record except the lowest-indexed one of its bucket emits one edge
 * (its document, the document of a lower-indexed record of the same bucket), appended to
 * edge_u/edge_v (u32 global doc indices; which lower record, and the order of the edges, depend
 * on scheduling -- the connected components do not).  Edges between records of one document
 * are dropped.  *n_edges (device u64) counts them; at most edge_cap are written (n_rec always
 * suffices).
 * The 64-bit surrogate key only picks the slot and a 32-bit tag; a tag match is confirmed on the
 * exact (band, tuple), so buckets are exact and nothing is left to the caller.  *n_coll (device
 * u64) counts the tag matches that turned out to be different tuples (statistics).
 * rec_bucket (optional, may be NULL): u64[n_rec], lowest record index of the record's bucket
 * (costs a second pass; used by tests). */
int mdd_bucket_edges(mdd_ctx* ctx, const uint64_t* key, const uint64_t* meta, uint64_t n_rec,
                     const uint32_t* tuple_base, uint64_t rec_per_doc, uint64_t doc_stride,
                     uint64_t band_stride, int rows, uint64_t* table, uint64_t tab_size,
                     uint32_t* edge_u, uint32_t* edge_v, uint64_t edge_cap, uint64_t* n_edges,
                     uint64_t* n_coll, uint64_t* rec_bucket);

/* Connected components.  Replaces the large-star/small-star loop (EN:82-115, EN:138-144)
 * and networkit ConnectedComponents (ZH-3:11-27): union-find over n_docs global indices;
 * labels[d] = smallest document index of d's component (what EN:145 keeps).
 * mdd_components_init sets labels[d] = d; mdd_components_union may be called repeatedly
 * (e.g. once per gathered edge shard); mdd_components_finish flattens. */
int mdd_components_init(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs);
int mdd_components_union(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs, const uint32_t* edge_u,
                         const uint32_t* edge_v, uint64_t n_edges);
int mdd_components_finish(mdd_ctx* ctx, uint32_t* labels, uint64_t n_docs);

/* Spanning forest of flattened labels (multi-GPU merge; no counterpart in the
 * reference): one edge (d, labels[d]) per document with labels[d] != d, appended to edge_u /
 * edge_v (at most edge_cap are written; *n_edges, device u64, counts them).  A rank runs its
 * own bucket edges through init/union/finish and ships this forest -- at most one edge per
 * touched document -- instead of the raw edges. */
int mdd_forest_edges(mdd_ctx* ctx, const uint32_t* labels, uint64_t n_docs, uint32_t* edge_u, uint32_t* edge_v,
                     uint64_t edge_cap, uint64_t* n_edges);

/* Re-verification of candidate pairs from their signatures (an EXTENSION: the reference unions
 * every bucket pair unverified, EN:136-145).  matches[e] = number of i < K with sig[edge_u[e]][i] == sig[edge_v[e]][i],
 * i.e. K x datasketch's MinHash.jaccard() of the two documents; an edge naming a document >= n_docs gets 0.  The caller
 * keeps the edges whose count reaches its threshold before mdd_components_union (map_neo_amd.dedup: min_jaccard). */
int mdd_signature_matches(mdd_ctx* ctx, const uint32_t* sig, uint64_t n_docs, int K, const uint32_t* edge_u,
                          const uint32_t* edge_v, uint64_t n_edges, uint32_t* matches);

/* Removal rule.  Replaces EN:145-149 (every non-minimum member of a component is removed)
 * and ZH-4:17-22 (members [1:] of each component): removed[d] = labels[first+d] != first+d for
 * the n documents starting at global index `first`; *n_removed (device u64) = their count. */
int mdd_removed_mask(mdd_ctx* ctx, const uint32_t* labels, uint64_t first, uint64_t n, uint8_t* removed,
                     uint64_t* n_removed);

/* Exact grouping of byte ranges (the exact-match stages around
 * MinHash in the EN pipeline).  Replaces the "hash the unit with SHA-256, group by hash, keep the
 * first" pattern of Exact_Document_Deduplication/exact_document_deduplication.py:16-18
 * (dropDuplicates on the content), Paragraph_Deduplication/paragraph.py:18-23 (sha2 +
 * row_number over (id, pos)) and Exact_Substring_Deduplication/exact_substr_dedup.py:23-24,
 * 44-47,83 (_get_hash + groupByKey + dup_list[1:]) -- all under
 * Matrix/english_data_clean_pipeline/dedup/.  Unit u = bytes [ubeg[u], uend[u]) of `text`
 * (8-byte aligned, text_bytes long + MDD_TEXT_PAD spare bytes; ranges are clamped to it); units may overlap.
 *   first_of[u] = smallest unit index whose bytes are identical to unit u's (u itself if none).
 * Equal SHA-256 <=> equal bytes, so these are the reference's groups with "first" = lowest index.
 * A 64-bit content hash (truncated to key_bits; pass 64) only picks the table slot and a 32-bit
 * tag; a tag match is confirmed by comparing the bytes inside the kernels, so the result is exact.
 * *n_coll (device u64) counts the tag matches that were different bytes (statistics).
 * Workspaces: key_scratch u64[n_units]; table u64[tab_size], tab_size a power of two
 * >= 2*n_units.  avg_unit_bytes picks the hash kernel: a thread per unit (< 64), eight lanes per unit (< 2048) or a wave
 * per unit; all three compute the same key. */
int mdd_exact_group(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* ubeg, const uint64_t* uend, uint64_t n_units,
                    uint64_t avg_unit_bytes, int key_bits, uint64_t* key_scratch, uint64_t* table, uint64_t tab_size,
                    uint32_t* first_of, uint64_t* n_coll);
/* the same, and unit_cps[u] (u32[n_units]) = code points of unit u, counted by the hash pass that reads the unit's bytes
 * anyway (paragraph dedup needs them for PAR:29's length filter: mdd_para_doc_sums) */
int mdd_exact_group_cps(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* ubeg, const uint64_t* uend,
                        uint64_t n_units, uint64_t avg_unit_bytes, int key_bits, uint64_t* key_scratch, uint64_t* table,
                        uint64_t tab_size, uint32_t* first_of, uint64_t* n_coll, uint32_t* unit_cps);

/* Paragraph dedup.  Replaces paragraph_dedup of
 * Matrix/english_data_clean_pipeline/dedup/Paragraph_Deduplication/paragraph.py:15-30 around
 * mdd_exact_group; documents in (doc_id, then input) order, a paragraph = a "\n"-separated piece.
 *   count:   units_per_doc[d] = paragraphs of document d (PAR:16-17 split + posexplode)
 *   units:   byte ranges ubeg/uend of all paragraphs; unit_base[d] = exclusive prefix sum of
 *            units_per_doc (u64[n_docs]), n_units their total.  mdd_exact_group on these ranges is
 *            PAR:18-23 (first_of[u] == u <=> the paragraph survives).
 *   compact_count: kept_bytes[d] / kept_cps[d] = bytes / code points of "\n".join(surviving
 *            paragraphs of d) (PAR:26-27), has_unit[d] = 1 when at least one survives;
 *   doc_sums (round 3): the same three per-document results as compact_count, computed from the UNITS instead of the text
 *            (lengths from ubeg / uend, code points from mdd_exact_group_cps: ~28 bytes per paragraph instead of the
 *            document), plus whole_doc[d] = 1 when every paragraph of d survives (the output is then the document itself);
 *   compact_write: that text for every document with keep_doc[d] != 0 (the caller applies PAR:29) at
 *            out_text[out_offsets[d] ..); nothing is written at or beyond out_capacity.  whole_doc (may be NULL): documents
 *            flagged there are copied instead of being re-assembled paragraph by paragraph. */
int mdd_para_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                   uint64_t* units_per_doc);
int mdd_para_units(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                   const uint64_t* unit_base, uint64_t n_units, uint64_t* ubeg, uint64_t* uend);
int mdd_para_compact_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                           const uint64_t* unit_base, uint64_t n_units, const uint32_t* first_of, uint64_t* kept_bytes,
                           uint64_t* kept_cps, uint8_t* has_unit);
int mdd_para_compact_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                           const uint64_t* unit_base, uint64_t n_units, const uint32_t* first_of, const uint8_t* keep_doc,
                           const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity, const uint8_t* whole_doc);
int mdd_para_doc_sums(mdd_ctx* ctx, const uint64_t* offsets, uint64_t n_docs, uint64_t text_bytes, const uint64_t* unit_base,
                      uint64_t n_units, const uint64_t* ubeg, const uint64_t* uend, const uint32_t* unit_cps, const uint32_t* first_of,
                      uint64_t* kept_bytes, uint64_t* kept_cps, uint8_t* has_unit, uint8_t* whole_doc);

/* Exact substring dedup.  Replaces text2hash + groupByKey +
 * get_remove_list + remove_dup_content + the min_length filter of
 * Matrix/english_data_clean_pipeline/dedup/Exact_Substring_Deduplication/exact_substr_dedup.py:34-47,
 * 55-70,81-90 without materialising the windows.  Documents in input order (zipWithIndex, SUB:75).
 *   mark:  dup[p] = 1 for every byte position p where a window of `window` code points starts
 *          (offsets range(len - window), SUB:38) whose bytes already occur at a smaller position.
 *          table: workspace u64[tab_size], tab_size >= 2 * number of windows (any size, not only
 *          powers of two; sum over documents of max(0, bytes - window) is a safe window count);
 *          dup: u8[text_bytes].  When the table for all windows does not fit, call n_parts times
 *          with part = 0 .. n_parts-1: each call handles the windows whose content key falls into
 *          that partition (tab_size >= 2 * windows / n_parts, plus slack) and clears the table;
 *          part 0 clears dup.
 *   count: kept_bytes[d] / kept_cps[d] = bytes / code points of document d that are covered by no
 *          marked window (SUB:58-67).
 *   write: the surviving bytes of every document with keep_doc[d] != 0 (the caller applies
 *          SUB:90: kept_cps >= min_length) to out_text[out_offsets[d] ..); nothing is written at
 *          or beyond out_capacity.
 * 1 <= window <= 128; text 4-byte aligned with MDD_TEXT_PAD spare bytes, text_bytes < 2^40. */
int mdd_substr_mark(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                    int window, int part, int n_parts, uint64_t* table, uint64_t tab_size, uint8_t* dup);
/*   mark_sorted: the same marks as `mark`, computed without random atomics on a big table (mdd_substr_part.hip): every
 *          window becomes an 8-byte entry, two 2048-way partition passes (tiles sorted by bin in LDS, so the stores are
 *          runs, not single entries) sort the entries into 2^22 leaves by content key, and one wave per leaf groups its
 *          ~500 windows in an LDS table (tag matches confirmed on the bytes, so the result is exact).  Leaves above
 *          16384 windows (boilerplate, separator lines) are sliced over all waves and joined through a small global
 *          table.  About twice as fast as `mark` on corpora of millions of documents (C2: 75 ms against 154 ms).
 *          text_bytes < 2^36.  Workspaces (caller-allocated, contents irrelevant):
 *            ent_a, ent_b  u64[ent_cap] each, ent_cap >= 1.05 x windows of the pass + 2048 x 28 x CUs + 12288
 *            d1_ws         u16[ent_cap]
 *            ws            u64[ws_words], ws_words >= MDD_SUBSTR_SORTED_WS_WORDS (89 MiB)
 *          part / n_parts as for `mark` (windows are split by content key; part 0 clears dup).  An entry array that is
 *          too small, or more than 768 distinct windows in one leaf (or one 4096-window slice of a big leaf), is
 *          counted and reported by mdd_substr_check: redo with more parts, or use `mark`.
 *          cand (round 3; NULL = every window takes part): the bitmap mdd_substr_candidates made (bit p & 63 of u64 word
 *          p >> 6 = byte position p; MDD_SUBSTR_CAND_BYTES(text_bytes) bytes, 8-byte aligned) -- windows whose bit is 0
 *          are skipped.  The marks are the same with and without it; ent_cap then only has to hold the
 *          candidate windows of the pass; dup is NOT cleared by part 0 then (mdd_substr_candidates has cleared it and,
 *          for window = 50, already holds marks).
 *   candidates: the filter in front of mark_sorted (same reference lines: it decides nothing, it only proves windows
 *          unique).  A window that occurs twice holds the same 32-byte k-grams at the same offsets in both copies, so
 *          it can only take part in a group when the k-gram its bytes select -- the one with the smallest hash among
 *          those starting in its first 19 bytes, leftmost on ties -- occurs twice in the corpus.  The selected k-grams
 *          ("anchors", one per ~10 windows) are sorted and grouped like windows (same workspaces; ent_cap >=
 *          max(windows / 6, (text_bytes + 71) / 8) + the slack above is ample: anchors beyond it switch the filter
 *          off, they do not fail the call), and bit p of cand is set for every byte position p with a repeated anchor in
 *          [p, p + 18] (cand: MDD_SUBSTR_CAND_BYTES(text_bytes) bytes; the call also uses it as scratch).  *n_cand (DEVICE) = number of flagged positions, an upper bound of the windows mark_sorted
 *          will sort.  Asynchronous.  50 <= window <= 128 (a window of w code points has >= w bytes; below 50 bytes
 *          the selection does not fit: call mark_sorted with cand = NULL); cand 8-byte aligned.
 *          dup: u8[text_bytes], cleared here.  For window = 50 the repeated anchors are settled on the spot: the 68 bytes
 *          around an anchor are compared with those around the earlier anchors of its key, which decides all 19 windows
 *          that can select it; their marks go straight into dup and only what that comparison cannot vouch for (bytes
 *          >= 0x80 in the region, keys with more than 32 anchors, leaves of more than 768 anchors) is flagged in cand for
 *          mark_sorted.  *n_cand == 0 then means dup is final.  The anchors' leaves follow the size of the text (2^20
 *          leaves up to ~2.2 GB, 2^21 / 2^22 beyond) and a leaf above 384 anchors -- every leaf once a call holds more
 *          than ~9 GB -- is settled in rounds over shares of the tag space, so one call takes any text below 2^36 bytes
 *          (27 GB / 12.5 M documents: 0.33 s for the stage, one pass).  On C2 (1 M documents, 15 % near-copies) 2 % of the
 *          windows remain for mark_sorted and the stage takes 24.5 instead of 79 ms (filter alone, window != 50: 28 %
 *          remain, 61 ms). */
#define MDD_SUBSTR_SORTED_WS_WORDS 11672587ull
#define MDD_SUBSTR_CAND_BYTES(text_bytes) ((((uint64_t)(text_bytes) >> 6) + 2) * 8) /* candidate bitmap: a bit per text byte + a spare word */
int mdd_substr_mark_sorted(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                           int window, int part, int n_parts, uint64_t* ent_a, uint64_t* ent_b, uint64_t ent_cap,
                           uint16_t* d1_ws, uint64_t* ws, uint64_t ws_words, uint8_t* dup, const uint8_t* cand);
int mdd_substr_candidates(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                          int window, uint64_t* ent_a, uint64_t* ent_b, uint64_t ent_cap, uint16_t* d1_ws, uint64_t* ws,
                          uint64_t ws_words, uint8_t* cand, uint64_t* n_cand, uint8_t* dup);
/*   check: mark is asynchronous and cannot validate tab_size against the (device-side) number of windows.  A window
 *          that finds neither its content nor a free slot is counted on the device; mdd_substr_check synchronises
 *          the stream, stores that count in *n_overflow_host (HOST, may be NULL), clears it and returns
 *          MDD_ERR_INVALID when it is non-zero (the marks are then incomplete: redo the passes with a larger
 *          table).  Call it after the last mark pass, before trusting dup. */
int mdd_substr_check(mdd_ctx* ctx, uint64_t* n_overflow_host);
int mdd_substr_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                     int window, const uint8_t* dup, uint64_t* kept_bytes, uint64_t* kept_cps);
int mdd_substr_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                     int window, const uint8_t* dup, const uint8_t* keep_doc, const uint64_t* out_offsets,
                     uint8_t* out_text, uint64_t out_capacity);

/* EN normalisation of pure-ASCII documents on the device.  Replaces
 * normalize_text EN:31-40 (fix_text/NFC: identity on clean ASCII; lower; delete
 * string.punctuation; strip; whitespace runs -> one space) for the documents where that needs no
 * Unicode tables.  Two calls with the caller's exclusive scan of out_len in between:
 *   count: out_len[d] = normalised length, non_ascii[d] = 1 when document d holds a byte >= 0x80
 *          (then out_len[d] = 0 and the caller normalises that document with the Python steps);
 *   write: normalised bytes of every document with non_ascii[d] == 0 to
 *          out_text[out_offsets[d] .. out_offsets[d] + out_len[d]).
 * All pointers are device memory; out_text holds out_capacity bytes (nothing is written beyond). */
int mdd_normalize_en_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                           uint64_t* out_len, uint8_t* non_ascii);
int mdd_normalize_en_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                           const uint8_t* non_ascii, const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity);

/* Packed corpus of selected documents: output document k = input document sel[k] (any order, repeats allowed).
 * This is the row movement between the EN steps -- the survivors of dropDuplicates (EXD:17) and of the left_anti
 * join (EN:149), the doc_id order paragraph dedup wants (PAR:21-26 orderBy) -- done on the packed text in HBM.
 * out_offsets[n_sel+1] is the caller's exclusive scan of the selected lengths; document k is copied to
 * out_text[out_offsets[k] ..) and nothing is written at or beyond out_capacity.  text and out_text must be 4-byte
 * aligned (dword copies with a byte-aligned source); sel entries >= n_docs select nothing. */
int mdd_gather_docs(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                    const uint64_t* sel, uint64_t n_sel, const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity);

/* Device-side field extraction for UNCOMPRESSED JSON-lines shards (the ingest on the input side of
 * the path; csrc/mdd_jsonl.hip).  Replaces, for the lines it can vouch for, the reference's per-line host work --
 * ZH-1:40-46 `for idx, line in enumerate(f): json.loads(line)[field]`, EN:120-123 spark.read.json + row[field], row["doc_id"]
 * -- under the contract of the host scanner (include/mdd_io.h): a line is extracted exactly as json.loads would, or handed
 * back untouched (MDD_JSONL_FALLBACK) for the reference's own Python steps.  `raw` = the shard's bytes in HBM, 16-byte
 * aligned, with MDD_TEXT_PAD spare bytes after raw_bytes.
 *   count_lines  tile_counts[t] = "\n" bytes of 16 KiB tile t (u32[ceil(raw_bytes / MDD_JSONL_TILE)]); totals (device
 *                u64[3], cleared by the call): [0] = "\n" bytes, [1] = lone "\r" bytes (a text-mode reader ends a line there:
 *                send such a shard to the host reader), [2] = all "\r" bytes (0: the shard's lines can be copied as byte
 *                ranges, ZH-5's text-mode rewrite is the identity).  n_lines = totals[0] + (last byte is not "\n").
 *   line_begins  line_begin[0 .. n_lines] (u64; line i = [line_begin[i], line_begin[i+1]), its "\n" included); tile_base =
 *                the caller's exclusive prefix sum of tile_counts (u64 per tile).
 *   scan         per line: status (MDD_JSONL_*), content_pos = position of the opening quote of the content string,
 *                content_len = its decoded utf-8 length (both 0 unless status is OK), and -- when id_key is not NULL --
 *                id_kind (MDD_IO_ID_* of include/mdd_io.h: 0 absent / null, 1 plain string, 2 JSON token) with the byte
 *                range [id_beg, id_end) of the id inside raw.  Keys: HOST strings of 1..64 bytes that need no JSON
 *                escaping (otherwise MDD_ERR_UNSUPPORTED: use the host reader).  The last duplicate key wins.  One wave
 *                per line, 64 bytes per step (string bodies, digit runs and whitespace are crossed with ballots).  TEST
 *                HOOK: the environment variable MDD_JSONL_SCAN=thread selects the first implementation (one thread
 *                per line), which the tests hold the wave scanner against.
 *   decode       the decoded content of every OK line at out_text[out_offsets[i] ..) (out_offsets = the caller's
 *                exclusive prefix sum of content_len; nothing is written at or beyond out_capacity): escapes,
 *                \uXXXX and surrogate pairs resolved -- the packed utf-8 mdd_signatures consumes. */
#define MDD_JSONL_TILE 16384
#define MDD_JSONL_OK 0        /* content extracted */
#define MDD_JSONL_NOCONTENT 1 /* valid JSON object without a usable string in the content field: never hashed, always kept */
#define MDD_JSONL_FALLBACK 2  /* not vouched for: the caller runs json.loads on the line */
#define MDD_JSONL_BLANK 3     /* only whitespace: no row */
int mdd_jsonl_count_lines(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, uint32_t* tile_counts, uint64_t* totals);
int mdd_jsonl_line_begins(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, const uint64_t* tile_base, uint64_t* line_begin,
                          uint64_t n_lines);
int mdd_jsonl_scan(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, const uint64_t* line_begin, uint64_t n_lines,
                   const char* content_key, const char* id_key, uint8_t* status, uint64_t* content_pos, uint64_t* content_len,
                   uint8_t* id_kind, uint64_t* id_beg, uint64_t* id_end);
int mdd_jsonl_decode(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, uint64_t n_lines, const uint8_t* status,
                     const uint64_t* content_pos, const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity);

/* Measurement helper: runs `iters` iterations of a dependency-free
 * v_xor/v_add/v_lshr mix on every CU and reports the measured int32 VALU lane-op rate in
 * *tops_host (HOST double, 1e12 lane-ops/s).  Synchronous. */
int mdd_valu_calibrate(mdd_ctx* ctx, int iters, double* tops_host);

#ifdef __cplusplus
}
#endif
#endif /* MDD_H */
