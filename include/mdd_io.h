/* mdd_io.h -- C ABI of libmdd_io.so: native JSON-lines ingest for the dedup stage
 * (the data format on the input side of the hot path).
 *
 * Replaces, per shard, the reference's Python line loop
 *   ZH-1:40-46  open(file) / for idx, line in enumerate(f) / json.loads(line)[CONTENT_FIELD_NAME]
 *   EN:120-125  spark.read.json + row[CONTENT_FIELD_NAME] / row["doc_id"] + normalize_text (EN:31-40)
 * (paths as in include/mdd.h) and produces what mdd_signatures consumes: packed utf-8 text +
 * offsets.  Host code only (g++ / pthread-free / zlib); no GPU involved.
 *
 * Exactness strategy: the scanner implements strict RFC 8259 JSON plus Python's rules
 * (no raw control characters in strings, nothing but whitespace after the value).  Every line
 * it is not 100 % sure about is marked MDD_IO_FALLBACK and handed back untouched, and the
 * caller runs the original Python code (json.loads + normalize_text) on exactly that line.
 * So the native path can only be faster, never different.
 */
#ifndef MDD_IO_H
#define MDD_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-line status */
#define MDD_IO_OK 0        /* content extracted (and normalised if asked): the line is a document to hash */
#define MDD_IO_NOCONTENT 1 /* valid object, content field absent / null / not a string / not utf-8 encodable
                              (lone surrogate): never hashed, always kept (EN:124, ZH-1:52-54) */
#define MDD_IO_FALLBACK 2  /* caller must run the Python path on the raw line */

/* id kinds */
#define MDD_IO_ID_ABSENT 0
#define MDD_IO_ID_STRING 1 /* plain string without escapes: bytes [id_begin, id_end) of the raw buffer, utf-8 */
#define MDD_IO_ID_TOKEN 2  /* any other JSON value: raw token [id_begin, id_end); decode with json.loads */

#define MDD_IO_NORMALIZE_NONE 0
#define MDD_IO_NORMALIZE_EN_ASCII 1 /* EN:31-40 on pure-ASCII content (lower, drop string.punctuation, strip,
                                       collapse \s+ runs); non-ASCII content is marked MDD_IO_FALLBACK */

typedef struct mdd_io_shard mdd_io_shard;

/* Reads `path` (gzip is detected by its magic bytes), splits it on '\n' and scans every
 * non-blank line.  content_field / id_field are top-level keys (id_field may be NULL).
 * Returns 0 or a negative code; mdd_io_last_error() has the message. */
int mdd_io_read_jsonl(const char* path, const char* content_field, const char* id_field, int normalize,
                      mdd_io_shard** out);
void mdd_io_free(mdd_io_shard* s);
const char* mdd_io_last_error(void);

/* Non-blank lines, in file order. */
uint64_t mdd_io_n_lines(const mdd_io_shard* s);
const uint8_t* mdd_io_status(const mdd_io_shard* s);      /* [n_lines] MDD_IO_OK / NOCONTENT / FALLBACK */
const uint64_t* mdd_io_line_no(const mdd_io_shard* s);    /* [n_lines] physical 0-based line number in the file */
const uint64_t* mdd_io_line_begin(const mdd_io_shard* s); /* [n_lines+1]... raw byte range of line i = [begin[i], end[i]) */
const uint64_t* mdd_io_line_end(const mdd_io_shard* s);
const uint8_t* mdd_io_id_kind(const mdd_io_shard* s);     /* [n_lines] */
const uint64_t* mdd_io_id_begin(const mdd_io_shard* s);   /* [n_lines] into raw */
const uint64_t* mdd_io_id_end(const mdd_io_shard* s);
const uint8_t* mdd_io_raw(const mdd_io_shard* s);         /* decompressed file */
uint64_t mdd_io_raw_bytes(const mdd_io_shard* s);

/* Packed content of the MDD_IO_OK lines, in line order: text[offsets[k] .. offsets[k+1]). */
uint64_t mdd_io_n_docs(const mdd_io_shard* s);
const uint8_t* mdd_io_text(const mdd_io_shard* s);
const uint64_t* mdd_io_offsets(const mdd_io_shard* s);    /* [n_docs+1] */

/* ---- egress: the output side of the exact-match stages ------------------------------------
 * Replaces the reference's  df.select(doc_id, raw_content).write.json(path, compression="gzip")
 * (paragraph.py:27-28,33-36; exact_substr_dedup.py:92-95) for one output part.  Writes n_docs
 * lines
 *     {<key_id_json>: <id k>, <key_content_json>: "<content k, JSON-escaped>"}\n
 * byte for byte what Python's  json.dumps(row, ensure_ascii=False) + "\n"  produces: ", " and
 * ": " separators, only  "  \  and the control characters below 0x20 escaped (\b \f \n \r \t,
 * \u00XX for the rest), everything else copied as utf-8.
 *   key_id_json / key_content_json   the two keys already JSON-encoded (quotes included)
 *   ids_json[ids_bytes]              the n_docs id values already JSON-encoded, separated by one
 *                                    '\n' (a JSON value never holds a raw newline)
 *   text / offsets[n_docs+1]         packed utf-8 contents: document k = text[offsets[k], offsets[k+1])
 *   gzip_level                       1-9: gzip member at that zlib level; 0: plain file
 * Contents that are not valid utf-8 (Python's .decode("utf8") would raise) fail the call with
 * a negative code and leave no file behind.  *bytes_out = uncompressed bytes written. */
int mdd_io_write_jsonl(const char* path, const char* key_id_json, const char* key_content_json, const uint8_t* ids_json,
                       uint64_t ids_bytes, const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int gzip_level,
                       uint64_t* bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* MDD_IO_H */
