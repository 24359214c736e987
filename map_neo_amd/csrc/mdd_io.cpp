// libmdd_io.so -- native JSON-lines ingest (C ABI: include/mdd_io.h).  Host code only.
// Strict scanner + "when in doubt, hand the line back" (MDD_IO_FALLBACK): the caller then runs the
// reference's own Python steps on that line, so this path can only be faster, never different.
#include <zlib.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mdd_io.h"

struct mdd_io_shard {
    std::vector<uint8_t> raw;
    std::vector<uint8_t> status, id_kind;
    std::vector<uint64_t> line_no, line_begin, line_end, id_begin, id_end;
    std::vector<uint8_t> text;
    std::vector<uint64_t> offsets;
};

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint8_t magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, f);
    // a single-member gzip file ends with its uncompressed size mod 2^32: used only to reserve the buffer once
    size_t isize_hint = 0;
    uint8_t tail[4];
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b && fseek(f, -4, SEEK_END) == 0 && fread(tail, 1, 4, f) == 4)
        isize_hint = (size_t)tail[0] | (size_t)tail[1] << 8 | (size_t)tail[2] << 16 | (size_t)tail[3] << 24;
    fclose(f);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        gzFile g = gzopen(path, "rb");
        if (!g) return false;
        gzbuffer(g, 1 << 20);
        out.reserve(isize_hint);
        std::vector<uint8_t> buf(4 << 20);
        for (;;) {
            int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) {
                gzclose(g);
                return false;
            }
            if (n == 0) break;
            out.insert(out.end(), buf.begin(), buf.begin() + n);
        }
        gzclose(g);
        return true;
    }
    f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    size_t rd = sz > 0 ? fread(out.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return rd == out.size();
}

// what Python's str.strip() / str.isspace() treat as whitespace, restricted to ASCII
inline bool py_ascii_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
inline bool json_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

struct Scanner {
    const uint8_t* p;
    const uint8_t* end;
    bool bad = false;  // anything we are not sure about -> fallback

    void ws() {
        while (p < end && json_ws(*p)) p++;
    }
    // validates utf-8 continuation structure of one multi-byte sequence starting at *p (lead byte >= 0x80)
    bool utf8_seq() {
        uint8_t c = *p;
        int n = (c >= 0xC2 && c <= 0xDF) ? 1 : (c >= 0xE0 && c <= 0xEF) ? 2 : (c >= 0xF0 && c <= 0xF4) ? 3 : -1;
        if (n < 0 || end - p <= n) return false;
        for (int i = 1; i <= n; i++)
            if ((p[i] & 0xC0) != 0x80) return false;
        if (c == 0xE0 && p[1] < 0xA0) return false;                 // overlong
        if (c == 0xED && p[1] >= 0xA0) return false;                // utf-8 encoded surrogate
        if (c == 0xF0 && p[1] < 0x90) return false;                 // overlong
        if (c == 0xF4 && p[1] >= 0x90) return false;                // > U+10FFFF
        p += n + 1;
        return true;
    }
    static int hexv(uint8_t c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    // Scans a string starting at the opening quote.  If `out` is non-null the decoded utf-8 is appended.
    // *plain = no escape was seen; *lone = a lone surrogate escape was seen (value not utf-8 encodable).
    bool string(std::vector<uint8_t>* out, bool* plain, bool* lone) {
        if (p >= end || *p != '"') return false;
        p++;
        bool pl = true, ln = false;
        for (;;) {
            if (p >= end) return false;
            uint8_t c = *p;
            if (c == '"') {
                p++;
                break;
            }
            if (c < 0x20) return false;  // Python json (strict) rejects raw control characters
            if (c == '\\') {
                pl = false;
                if (end - p < 2) return false;
                uint8_t e = p[1];
                p += 2;
                uint32_t cp;
                switch (e) {
                    case '"': cp = '"'; break;
                    case '\\': cp = '\\'; break;
                    case '/': cp = '/'; break;
                    case 'b': cp = 8; break;
                    case 'f': cp = 12; break;
                    case 'n': cp = 10; break;
                    case 'r': cp = 13; break;
                    case 't': cp = 9; break;
                    case 'u': {
                        if (end - p < 4) return false;
                        int h0 = hexv(p[0]), h1 = hexv(p[1]), h2 = hexv(p[2]), h3 = hexv(p[3]);
                        if ((h0 | h1 | h2 | h3) < 0) return false;
                        cp = (uint32_t)(h0 << 12 | h1 << 8 | h2 << 4 | h3);
                        p += 4;
                        if (cp >= 0xD800 && cp <= 0xDBFF) {
                            // high surrogate: pairs with a following \uDC00-\uDFFF (as Python's decoder does)
                            if (end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                                int g0 = hexv(p[2]), g1 = hexv(p[3]), g2 = hexv(p[4]), g3 = hexv(p[5]);
                                if ((g0 | g1 | g2 | g3) < 0) return false;
                                uint32_t lo = (uint32_t)(g0 << 12 | g1 << 8 | g2 << 4 | g3);
                                if (lo >= 0xDC00 && lo <= 0xDFFF) {
                                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                                    p += 6;
                                } else {
                                    ln = true;
                                }
                            } else {
                                ln = true;
                            }
                        } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                            ln = true;
                        }
                        break;
                    }
                    default: return false;
                }
                if (out && !ln) {
                    if (cp < 0x80) out->push_back((uint8_t)cp);
                    else if (cp < 0x800) {
                        out->push_back(0xC0 | (cp >> 6));
                        out->push_back(0x80 | (cp & 0x3F));
                    } else if (cp < 0x10000) {
                        out->push_back(0xE0 | (cp >> 12));
                        out->push_back(0x80 | ((cp >> 6) & 0x3F));
                        out->push_back(0x80 | (cp & 0x3F));
                    } else {
                        out->push_back(0xF0 | (cp >> 18));
                        out->push_back(0x80 | ((cp >> 12) & 0x3F));
                        out->push_back(0x80 | ((cp >> 6) & 0x3F));
                        out->push_back(0x80 | (cp & 0x3F));
                    }
                }
                continue;
            }
            if (c < 0x80) {
                // longest run of bytes that stand for themselves: one bulk append instead of a push_back per byte
                const uint8_t* run = p;
                do p++;
                while (p < end && *p >= 0x20 && *p < 0x80 && *p != '"' && *p != '\\');
                if (out) out->insert(out->end(), run, p);
            } else {
                const uint8_t* s = p;
                if (!utf8_seq()) return false;
                if (out) out->insert(out->end(), s, p);
            }
        }
        if (plain) *plain = pl;
        if (lone) *lone = ln;
        return true;
    }
    bool number() {
        if (p < end && *p == '-') p++;
        if (p >= end) return false;
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') {
            while (p < end && *p >= '0' && *p <= '9') p++;
        } else return false;  // includes NaN / Infinity, which Python accepts: fallback decides
        if (p < end && *p == '.') {
            p++;
            if (p >= end || *p < '0' || *p > '9') return false;
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            p++;
            if (p < end && (*p == '+' || *p == '-')) p++;
            if (p >= end || *p < '0' || *p > '9') return false;
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        return true;
    }
    bool literal(const char* w) {
        size_t n = strlen(w);
        if ((size_t)(end - p) < n || memcmp(p, w, n) != 0) return false;
        p += n;
        return true;
    }
    bool value(int depth) {  // skips any value
        if (depth > 200) return false;
        ws();
        if (p >= end) return false;
        uint8_t c = *p;
        if (c == '"') return string(nullptr, nullptr, nullptr);
        if (c == '{') {
            p++;
            ws();
            if (p < end && *p == '}') {
                p++;
                return true;
            }
            for (;;) {
                ws();
                if (!string(nullptr, nullptr, nullptr)) return false;
                ws();
                if (p >= end || *p != ':') return false;
                p++;
                if (!value(depth + 1)) return false;
                ws();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == '}') {
                    p++;
                    return true;
                }
                return false;
            }
        }
        if (c == '[') {
            p++;
            ws();
            if (p < end && *p == ']') {
                p++;
                return true;
            }
            for (;;) {
                if (!value(depth + 1)) return false;
                ws();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == ']') {
                    p++;
                    return true;
                }
                return false;
            }
        }
        if (c == 't') return literal("true");
        if (c == 'f') return literal("false");
        if (c == 'n') return literal("null");
        return number();
    }
};

const uint8_t PUNCT[] = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~";  // string.punctuation

// EN:31-40 on pure-ASCII text (NFC is the identity on ASCII): lower, drop punctuation, strip, \s+ -> ' '
struct PunctTable {
    bool is[256];
    PunctTable() {
        memset(is, 0, sizeof is);
        for (const uint8_t* q = PUNCT; *q; q++) is[*q] = true;
    }
};
const PunctTable g_punct;  // built at load time: safe to read from any thread

void normalize_en_ascii(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
    const bool* is_punct = g_punct.is;
    bool pending_space = false, any = false;
    for (uint8_t c : in) {
        if (is_punct[c]) continue;
        if (py_ascii_space(c)) {
            pending_space = true;
            continue;
        }
        if (pending_space && any) out.push_back(' ');
        pending_space = false;
        any = true;
        out.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c);
    }
}

}  // namespace

extern "C" {

const char* mdd_io_last_error(void) { return g_err; }

int mdd_io_read_jsonl(const char* path, const char* content_field, const char* id_field, int normalize, mdd_io_shard** out) {
    if (!path || !content_field || !out) return fail(-1, "mdd_io_read_jsonl: NULL argument");
    if (normalize != MDD_IO_NORMALIZE_NONE && normalize != MDD_IO_NORMALIZE_EN_ASCII)
        return fail(-1, "mdd_io_read_jsonl: unknown normalize mode %d", normalize);
    mdd_io_shard* s = new mdd_io_shard();
    if (!read_file(path, s->raw)) {
        delete s;
        return fail(-2, "mdd_io_read_jsonl: cannot read %s", path);
    }
    // a key that needs escaping in JSON cannot be matched byte-wise: leave such files to Python
    const size_t clen = strlen(content_field), ilen = id_field ? strlen(id_field) : 0;
    bool key_simple = true;
    for (size_t i = 0; i < clen; i++) key_simple &= (uint8_t)content_field[i] >= 0x20 && content_field[i] != '"' && content_field[i] != '\\';
    for (size_t i = 0; i < ilen; i++) key_simple &= (uint8_t)id_field[i] >= 0x20 && id_field[i] != '"' && id_field[i] != '\\';

    const uint8_t* base = s->raw.data();
    const size_t n = s->raw.size();
    s->offsets.push_back(0);
    s->text.reserve(n);  // decoded contents are never longer than the lines they come from
    std::vector<uint8_t> val, norm;
    uint64_t lineno = 0;
    size_t pos = 0;
    while (pos < n) {
        const void* nl = memchr(base + pos, '\n', n - pos);
        size_t e = nl ? (size_t)((const uint8_t*)nl - base) : n;
        size_t next = e < n ? e + 1 : n;
        bool blank = true;
        for (size_t i = pos; i < e; i++)
            if (!py_ascii_space(base[i])) {
                blank = false;
                break;
            }
        if (!blank) {
            uint8_t status = MDD_IO_FALLBACK, idk = MDD_IO_ID_ABSENT;
            uint64_t idb = 0, ide = 0;
            bool have_content = false, content_ok = false;
            val.clear();
            Scanner sc{base + pos, base + e};
            bool ok = key_simple;
            if (ok) {
                sc.ws();
                ok = sc.p < sc.end && *sc.p == '{';
            }
            if (ok) {
                sc.p++;
                sc.ws();
                if (sc.p < sc.end && *sc.p == '}') {
                    sc.p++;
                } else {
                    for (;;) {
                        sc.ws();
                        const uint8_t* k0 = sc.p + 1;
                        bool kplain = true;
                        if (!sc.string(nullptr, &kplain, nullptr)) { ok = false; break; }
                        const uint8_t* k1 = sc.p - 1;
                        if (!kplain) { ok = false; break; }  // escaped key: let Python decide what it equals
                        sc.ws();
                        if (sc.p >= sc.end || *sc.p != ':') { ok = false; break; }
                        sc.p++;
                        sc.ws();
                        size_t klen = (size_t)(k1 - k0);
                        bool is_c = klen == clen && memcmp(k0, content_field, clen) == 0;
                        bool is_i = id_field && klen == ilen && memcmp(k0, id_field, ilen) == 0;
                        if (is_c) {  // duplicate keys: the last one wins, as in Python's dict
                            have_content = true;
                            content_ok = false;
                            val.clear();
                            if (sc.p < sc.end && *sc.p == '"') {
                                bool lone = false;
                                if (!sc.string(&val, nullptr, &lone)) { ok = false; break; }
                                content_ok = !lone;
                            } else if (!sc.value(1)) { ok = false; break; }
                        } else if (is_i) {
                            const uint8_t* v0 = sc.p;
                            if (sc.p < sc.end && *sc.p == '"') {
                                bool plain = true, lone = false;
                                if (!sc.string(nullptr, &plain, &lone)) { ok = false; break; }
                                if (plain) {
                                    idk = MDD_IO_ID_STRING;
                                    idb = (uint64_t)(v0 + 1 - base);
                                    ide = (uint64_t)(sc.p - 1 - base);
                                } else {
                                    idk = MDD_IO_ID_TOKEN;
                                    idb = (uint64_t)(v0 - base);
                                    ide = (uint64_t)(sc.p - base);
                                }
                            } else {
                                if (!sc.value(1)) { ok = false; break; }
                                idk = MDD_IO_ID_TOKEN;
                                idb = (uint64_t)(v0 - base);
                                ide = (uint64_t)(sc.p - base);
                                if (ide - idb == 4 && memcmp(v0, "null", 4) == 0) idk = MDD_IO_ID_ABSENT;  // row.get() is None
                            }
                        } else if (!sc.value(1)) { ok = false; break; }
                        sc.ws();
                        if (sc.p < sc.end && *sc.p == ',') {
                            sc.p++;
                            continue;
                        }
                        if (sc.p < sc.end && *sc.p == '}') {
                            sc.p++;
                            break;
                        }
                        ok = false;
                        break;
                    }
                }
            }
            if (ok) {
                sc.ws();
                ok = sc.p == sc.end;  // json.loads raises on trailing data
            }
            if (ok) {
                if (!have_content || !content_ok) {
                    status = MDD_IO_NOCONTENT;
                } else if (normalize == MDD_IO_NORMALIZE_EN_ASCII) {
                    bool ascii = true;
                    for (uint8_t c : val)
                        if (c >= 0x80) {
                            ascii = false;
                            break;
                        }
                    if (ascii) {
                        norm.clear();
                        normalize_en_ascii(val, norm);
                        s->text.insert(s->text.end(), norm.begin(), norm.end());
                        s->offsets.push_back(s->text.size());
                        status = MDD_IO_OK;
                    }  // else: Unicode lower / NFC / \s are Python's business
                } else {
                    s->text.insert(s->text.end(), val.begin(), val.end());
                    s->offsets.push_back(s->text.size());
                    status = MDD_IO_OK;
                }
            }
            if (status == MDD_IO_FALLBACK) {
                idk = MDD_IO_ID_ABSENT;
                idb = ide = 0;
            }
            s->status.push_back(status);
            s->id_kind.push_back(idk);
            s->id_begin.push_back(idb);
            s->id_end.push_back(ide);
            s->line_no.push_back(lineno);
            s->line_begin.push_back(pos);
            s->line_end.push_back(e);
        }
        lineno++;
        pos = next;
    }
    *out = s;
    return 0;
}

int mdd_io_write_jsonl(const char* path, const char* key_id_json, const char* key_content_json, const uint8_t* ids_json,
                       uint64_t ids_bytes, const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int gzip_level,
                       uint64_t* bytes_out) {
    if (!path || !key_id_json || !key_content_json || !bytes_out) return fail(-1, "mdd_io_write_jsonl: NULL argument");
    if (n_docs && (!ids_json || !offsets)) return fail(-1, "mdd_io_write_jsonl: NULL ids / offsets");
    if (gzip_level < 0 || gzip_level > 9) return fail(-1, "mdd_io_write_jsonl: gzip_level=%d outside [0,9]", gzip_level);
    *bytes_out = 0;
    gzFile g = nullptr;
    FILE* f = nullptr;
    if (gzip_level > 0) {
        char mode[8];
        snprintf(mode, sizeof mode, "wb%d", gzip_level);
        g = gzopen(path, mode);
        if (g) gzbuffer(g, 1 << 18);
    } else {
        f = fopen(path, "wb");
    }
    if (!g && !f) return fail(-2, "mdd_io_write_jsonl: cannot create %s", path);

    static const char HEX[] = "0123456789abcdef";
    const size_t FLUSH = 1 << 20;
    std::vector<uint8_t> buf;
    buf.reserve(FLUSH + (1 << 16));
    const std::string head = std::string("{") + key_id_json + ": ", mid = std::string(", ") + key_content_json + ": \"";
    uint64_t total = 0;
    int rc = 0;
    auto flush = [&]() -> bool {
        if (buf.empty()) return true;
        bool ok = g ? gzwrite(g, buf.data(), (unsigned)buf.size()) == (int)buf.size()
                    : fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        total += buf.size();
        buf.clear();
        return ok;
    };
    const uint8_t* idp = ids_json;
    const uint8_t* const id_end = ids_json + ids_bytes;
    for (uint64_t k = 0; k < n_docs && rc == 0; k++) {
        if (k > 0) {
            if (idp >= id_end) { rc = fail(-1, "mdd_io_write_jsonl: %llu ids for %llu documents", (unsigned long long)k, (unsigned long long)n_docs); break; }
            idp++;  // the separator
        }
        const uint8_t* idq = (const uint8_t*)memchr(idp, '\n', (size_t)(id_end - idp));
        if (!idq) idq = id_end;
        if (idq == idp) { rc = fail(-1, "mdd_io_write_jsonl: id %llu is empty", (unsigned long long)k); break; }
        buf.insert(buf.end(), head.begin(), head.end());
        buf.insert(buf.end(), idp, idq);
        idp = idq;
        buf.insert(buf.end(), mid.begin(), mid.end());
        if (offsets[k + 1] < offsets[k]) { rc = fail(-1, "mdd_io_write_jsonl: offsets decrease at document %llu", (unsigned long long)k); break; }
        Scanner sc{text + offsets[k], text + offsets[k + 1]};
        while (sc.p < sc.end) {
            // longest run that is copied verbatim
            const uint8_t* run = sc.p;
            while (sc.p < sc.end && *sc.p >= 0x20 && *sc.p < 0x80 && *sc.p != '"' && *sc.p != '\\') sc.p++;
            buf.insert(buf.end(), run, sc.p);
            if (sc.p >= sc.end) break;
            uint8_t c = *sc.p;
            if (c >= 0x80) {
                const uint8_t* s0 = sc.p;
                if (!sc.utf8_seq()) { rc = fail(-3, "mdd_io_write_jsonl: document %llu is not valid utf-8 at byte %llu", (unsigned long long)k, (unsigned long long)(s0 - (text + offsets[k]))); break; }
                buf.insert(buf.end(), s0, sc.p);
                continue;
            }
            sc.p++;
            buf.push_back('\\');
            switch (c) {
                case '"': buf.push_back('"'); break;
                case '\\': buf.push_back('\\'); break;
                case '\b': buf.push_back('b'); break;
                case '\f': buf.push_back('f'); break;
                case '\n': buf.push_back('n'); break;
                case '\r': buf.push_back('r'); break;
                case '\t': buf.push_back('t'); break;
                default:
                    buf.push_back('u');
                    buf.push_back('0');
                    buf.push_back('0');
                    buf.push_back(HEX[c >> 4]);
                    buf.push_back(HEX[c & 15]);
            }
        }
        if (rc) break;
        buf.push_back('"');
        buf.push_back('}');
        buf.push_back('\n');
        if (buf.size() >= FLUSH && !flush()) rc = fail(-2, "mdd_io_write_jsonl: write to %s failed", path);
    }
    if (rc == 0 && n_docs && idp != id_end) rc = fail(-1, "mdd_io_write_jsonl: more ids than the %llu documents", (unsigned long long)n_docs);
    if (rc == 0 && !flush()) rc = fail(-2, "mdd_io_write_jsonl: write to %s failed", path);
    bool closed = g ? gzclose(g) == Z_OK : fclose(f) == 0;
    if (rc == 0 && !closed) rc = fail(-2, "mdd_io_write_jsonl: closing %s failed", path);
    if (rc) {
        remove(path);
        return rc;
    }
    *bytes_out = total;
    return 0;
}

void mdd_io_free(mdd_io_shard* s) { delete s; }
uint64_t mdd_io_n_lines(const mdd_io_shard* s) { return s->status.size(); }
const uint8_t* mdd_io_status(const mdd_io_shard* s) { return s->status.data(); }
const uint64_t* mdd_io_line_no(const mdd_io_shard* s) { return s->line_no.data(); }
const uint64_t* mdd_io_line_begin(const mdd_io_shard* s) { return s->line_begin.data(); }
const uint64_t* mdd_io_line_end(const mdd_io_shard* s) { return s->line_end.data(); }
const uint8_t* mdd_io_id_kind(const mdd_io_shard* s) { return s->id_kind.data(); }
const uint64_t* mdd_io_id_begin(const mdd_io_shard* s) { return s->id_begin.data(); }
const uint64_t* mdd_io_id_end(const mdd_io_shard* s) { return s->id_end.data(); }
const uint8_t* mdd_io_raw(const mdd_io_shard* s) { return s->raw.data(); }
uint64_t mdd_io_raw_bytes(const mdd_io_shard* s) { return s->raw.size(); }
uint64_t mdd_io_n_docs(const mdd_io_shard* s) { return s->offsets.size() - 1; }
const uint8_t* mdd_io_text(const mdd_io_shard* s) { return s->text.data(); }
const uint64_t* mdd_io_offsets(const mdd_io_shard* s) { return s->offsets.data(); }

}  // extern "C"
