// Device-side field extraction for UNCOMPRESSED JSON-lines shards (the ingest on the input side of
// the path).  Replaces, for the lines it can vouch for, what the reference does per line on a host core:
//   ZH-1:40-46  for idx, line in enumerate(f): json.loads(line)[content_field]        (generate_minhash.py)
//   EN:120-123  spark.read.json(...) -> row[content_field], row["doc_id"]              (minhashLsh.py)
// and what libmdd_io.so's host scanner (csrc/mdd_io.cpp, include/mdd_io.h) does on 16 threads at ~0.5 GB/s each.
// Same contract as that scanner: a line is either extracted EXACTLY as json.loads would (status OK / NOCONTENT) or handed
// back untouched (status FALLBACK) for the reference's own Python steps -- strict RFC 8259 grammar plus Python's extra
// rules (no raw control characters in strings, only whitespace after the value; NaN / Infinity, escaped keys, nesting
// deeper than 64, invalid utf-8 all go back), so this path can only be faster, never different.
//
// Three steps, all on the raw shard bytes in HBM:
//   line index   newline positions (SWAR test per dword, ballots for ranks): count per 16 KiB tile, caller's prefix sum,
//                then line_begin[]; lone "\r" bytes are counted (text-mode readers end a line there: the caller sends such
//                a shard to the host reader)
//   scan         ONE WAVE PER LINE runs the host scanner's grammar as an automaton over 64-byte chunks (string bodies, digit
//                runs and whitespace are crossed with ballots; jsonl_scan_wave_kernel): status, position of the content
//                string, its decoded length, the id token.  The first implementation, one THREAD per line walking the
//                bytes (jsonl_scan_kernel), is kept as the independent one the tests compare with (MDD_JSONL_SCAN=thread).
//   decode       ONE WAVE PER LINE copies / unescapes the content string into the packed text mdd_signatures consumes:
//                a lane per byte, escape starts from the parity of backslash runs (ballot + count-leading-ones, carried
//                across 64-byte chunks), \uXXXX and surrogate pairs decoded by the starting lane, output positions from
//                ballots over the lanes' byte counts.  Chunks without a backslash are a straight coalesced copy.
#include <cstdlib>
#include <cstring>

#include "mdd_common.h"

namespace {
using namespace mddi;

constexpr uint32_t TILE_BYTES = 16384;  // one workgroup of 256 threads: 4 waves x 16 steps x 256 bytes
constexpr int KEY_MAX = 64;

// bit 7 of every byte of w that equals the byte b (exact SWAR zero-byte test on w ^ bbbb)
__device__ __forceinline__ uint32_t swar_eq(uint32_t w, uint32_t b) {
    const uint32_t y = w ^ (b * 0x01010101u);
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// WRITE = false: tile_counts[tile] = newlines of the tile, totals[0] += them, totals[1] += lone "\r", totals[2] += all "\r".
// WRITE = true:  line_begin[k + 1] = position after the k-th newline (k counted over the whole shard; tile_base[tile] =
//                newlines before the tile).
template <bool WRITE>
__global__ void __launch_bounds__(256) jsonl_index_kernel(const uint8_t* __restrict__ raw, uint64_t n, uint32_t* tile_counts,
                                                          unsigned long long* totals, const uint64_t* __restrict__ tile_base,
                                                          uint64_t* line_begin, uint64_t n_lines) {
    __shared__ uint32_t wave_cnt[4];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(raw);
    const uint64_t seg = (uint64_t)blockIdx.x * TILE_BYTES + (uint64_t)wave * (TILE_BYTES / 4);  // this wave's 4096 bytes
    uint32_t z[16];
    uint32_t cnt = 0, lone = 0, crs = 0;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const uint64_t p = seg + (uint64_t)s * 256 + 4 * lane;
        uint32_t w = 0;
        if (p < n) w = tw[p >> 2];  // the dword may reach up to 3 bytes past n: inside the buffer's padding
        uint32_t in = 0x80808080u;
        if (p + 4 > n) {
            in = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (p + k < n) in |= 0x80u << (8 * k);
        }
        z[s] = swar_eq(w, 10u) & in;
        cnt += (uint32_t)__builtin_popcount(z[s]);
        if (!WRITE) {
            const uint32_t cr = swar_eq(w, 13u) & in;
            if (cr) {
                crs += (uint32_t)__builtin_popcount(cr);
                // followed by "\n"?  bytes 0..2: the next byte of the same dword; byte 3: the next dword's first byte
                uint32_t follow = (z[s] >> 8);
                if ((cr >> 31) && p + 4 < n && raw[p + 4] == 10u) follow |= 0x80000000u;
                lone += (uint32_t)__builtin_popcount(cr & ~follow);
            }
        }
    }
    uint32_t wsum = cnt;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, o, 64);
    if (!WRITE) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lone += (uint32_t)__shfl_xor((int)lone, o, 64);
            crs += (uint32_t)__shfl_xor((int)crs, o, 64);
        }
        if (lane == 0) {
            wave_cnt[wave] = wsum;
            if (lone) atomicAdd(&totals[1], (unsigned long long)lone);
            if (crs) atomicAdd(&totals[2], (unsigned long long)crs);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            tile_counts[blockIdx.x] = t;  // (totals[0] is their sum: tile_sum_kernel -- one atomic per tile on one address
        }                                 // is ~11 ns each, 1.5 ms for the 134 k tiles of a 2.2 GB shard)
        return;
    }
    if (lane == 0) wave_cnt[wave] = wsum;
    if (blockIdx.x == 0 && threadIdx.x == 0) line_begin[n_lines] = n;
    __syncthreads();
    uint64_t k = tile_base[blockIdx.x];
    for (uint32_t w2 = 0; w2 < wave; w2++) k += wave_cnt[w2];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const uint32_t c = (uint32_t)__builtin_popcount(z[s]);
        const uint64_t B0 = __ballot(c & 1u), B1 = __ballot(c & 2u), B2 = __ballot(c & 4u);
        if (c) {
            uint64_t mine = k + __builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) + 4 * __builtin_popcountll(B2 & lt);
            const uint64_t p = seg + (uint64_t)s * 256 + 4 * lane;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (z[s] & (0x80u << (8 * b))) {
                    if (mine + 1 <= n_lines) line_begin[mine + 1] = p + b + 1;  // tile_base is the caller's: stay inside line_begin
                    mine++;
                }
        }
        k += __builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2);
    }
}

// totals[0] = sum of the tile counts (one workgroup)
__global__ void __launch_bounds__(1024) tile_sum_kernel(const uint32_t* __restrict__ tile_counts, uint64_t n_tiles, unsigned long long* totals) {
    __shared__ unsigned long long part[16];
    unsigned long long acc = 0;
    for (uint64_t i = threadIdx.x; i < n_tiles; i += 1024) acc += tile_counts[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; i++) t += part[i];
        totals[0] = t;
    }
}

// ---- scan: the host scanner's state machine (csrc/mdd_io.cpp: Scanner), one thread per line ------------------------------
// The cursor reads the shard through a 64-byte window kept in LDS (one slot per thread, refilled with four 16-byte loads):
// a byte-wise walk straight from global memory is a chain of dependent loads (~0.3-1 us each; a 2 KB line then takes
// milliseconds, which a shard of a few thousand lines cannot hide behind occupancy), an LDS byte costs ~30 ns.  Slots are
// 17 dwords apart, so the lanes of a wave hit different banks.
constexpr uint32_t WIN = 64, WIN_STRIDE = 68;
struct Sc {
    const uint8_t* __restrict__ b;  // shard (16-byte aligned, MDD_TEXT_PAD spare bytes after its end)
    uint64_t p, end;                // cursor, end of the line (its "\n" excluded)
    uint8_t* win;                   // this thread's LDS slot
    uint64_t wbase;                 // shard position of win[0]; 2^63 = nothing loaded (i - 2^63 >= WIN for every position i)
    __device__ __forceinline__ uint32_t at(uint64_t i) {
        if (i - wbase >= WIN) {
            wbase = i & ~15ull;
            const uint4* src = reinterpret_cast<const uint4*>(b + wbase);
            uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
            uint32_t* w = reinterpret_cast<uint32_t*>(win);
            w[0] = v0.x, w[1] = v0.y, w[2] = v0.z, w[3] = v0.w;
            w[4] = v1.x, w[5] = v1.y, w[6] = v1.z, w[7] = v1.w;
            w[8] = v2.x, w[9] = v2.y, w[10] = v2.z, w[11] = v2.w;
            w[12] = v3.x, w[13] = v3.y, w[14] = v3.z, w[15] = v3.w;
        }
        return win[i - wbase];
    }
    __device__ __forceinline__ uint32_t back(uint64_t i) { return at(i); }  // a byte behind the cursor (keys, the id's "null")
    __device__ __forceinline__ bool line_ok() const { return true; }         // everything is checked where it is read
    __device__ __forceinline__ void ws() {
        while (p < end) {
            const uint32_t c = at(p);
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') p++;
            else break;
        }
    }
};
__device__ __forceinline__ int hexv(uint32_t c) {
    if (c - '0' <= 9u) return (int)(c - '0');
    if (c - 'a' <= 5u) return (int)(c - 'a' + 10);
    if (c - 'A' <= 5u) return (int)(c - 'A' + 10);
    return -1;
}
// a string starting at its opening quote; *dec += decoded utf-8 length, *plain &= no escape seen, *lone |= a lone surrogate
// escape seen (the value cannot be utf-8 encoded: never hashed).  false = not something this scanner vouches for.
__device__ bool sc_string(Sc& s, uint64_t* dec, bool* plain, bool* lone) {
    if (s.p >= s.end || s.at(s.p) != '"') return false;
    s.p++;
    uint64_t n = 0;
    bool pl = true, ln = false;
    for (;;) {
        if (s.p >= s.end) return false;
        const uint32_t c = s.at(s.p);
        if (c == '"') {
            s.p++;
            break;
        }
        if (c < 0x20u) return false;  // Python's json (strict) rejects raw control characters
        if (c == '\\') {
            pl = false;
            if (s.end - s.p < 2) return false;
            const uint32_t e = s.at(s.p + 1);
            s.p += 2;
            if (e == 'u') {
                if (s.end - s.p < 4) return false;
                const int h0 = hexv(s.at(s.p)), h1 = hexv(s.at(s.p + 1)), h2 = hexv(s.at(s.p + 2)), h3 = hexv(s.at(s.p + 3));
                if ((h0 | h1 | h2 | h3) < 0) return false;
                uint32_t cp = (uint32_t)(h0 << 12 | h1 << 8 | h2 << 4 | h3);
                s.p += 4;
                if (cp - 0xD800u <= 0x3FFu) {  // high surrogate: pairs with a following \uDC00-\uDFFF (as Python's decoder does)
                    bool paired = false;
                    if (s.end - s.p >= 6 && s.at(s.p) == '\\' && s.at(s.p + 1) == 'u') {
                        const int g0 = hexv(s.at(s.p + 2)), g1 = hexv(s.at(s.p + 3)), g2 = hexv(s.at(s.p + 4)), g3 = hexv(s.at(s.p + 5));
                        if ((g0 | g1 | g2 | g3) < 0) return false;
                        const uint32_t lo = (uint32_t)(g0 << 12 | g1 << 8 | g2 << 4 | g3);
                        if (lo - 0xDC00u <= 0x3FFu) {
                            s.p += 6;
                            paired = true;
                        }
                    }
                    if (paired) n += 4;
                    else ln = true;
                } else if (cp - 0xDC00u <= 0x3FFu) {
                    ln = true;
                } else {
                    n += cp < 0x80u ? 1u : (cp < 0x800u ? 2u : 3u);
                }
            } else if (e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't') {
                n += 1;
            } else {
                return false;
            }
            continue;
        }
        if (c < 0x80u) {
            s.p++;
            n++;
            continue;
        }
        // one multi-byte utf-8 sequence, validated like Python's decoder would
        const int k = (c >= 0xC2u && c <= 0xDFu) ? 1 : (c >= 0xE0u && c <= 0xEFu) ? 2 : (c >= 0xF0u && c <= 0xF4u) ? 3 : -1;
        if (k < 0 || s.end - s.p <= (uint64_t)k) return false;
        const uint32_t c1 = s.at(s.p + 1);
        for (int i = 1; i <= k; i++)
            if ((s.at(s.p + i) & 0xC0u) != 0x80u) return false;
        if ((c == 0xE0u && c1 < 0xA0u) || (c == 0xEDu && c1 >= 0xA0u) || (c == 0xF0u && c1 < 0x90u) || (c == 0xF4u && c1 >= 0x90u)) return false;
        s.p += (uint64_t)k + 1;
        n += (uint64_t)k + 1;
    }
    if (dec) *dec = n;
    if (plain) *plain = pl;
    if (lone) *lone = ln;
    return true;
}
template <class S>
__device__ bool sc_number(S& s) {
    auto dig = [&](uint64_t i) { return i < s.end && s.at(i) - '0' <= 9u; };
    if (s.p < s.end && s.at(s.p) == '-') s.p++;
    if (s.p >= s.end) return false;
    if (s.at(s.p) == '0') s.p++;
    else if (s.at(s.p) - '1' <= 8u) {
        while (dig(s.p)) s.p++;
    } else return false;  // includes NaN / Infinity, which Python accepts: the host decides
    if (s.p < s.end && s.at(s.p) == '.') {
        s.p++;
        if (!dig(s.p)) return false;
        while (dig(s.p)) s.p++;
    }
    if (s.p < s.end && (s.at(s.p) == 'e' || s.at(s.p) == 'E')) {
        s.p++;
        if (s.p < s.end && (s.at(s.p) == '+' || s.at(s.p) == '-')) s.p++;
        if (!dig(s.p)) return false;
        while (dig(s.p)) s.p++;
    }
    return true;
}
template <class S>
__device__ bool sc_literal(S& s, const char* w, uint32_t n) {
    if (s.end - s.p < n) return false;
    for (uint32_t i = 0; i < n; i++)
        if (s.at(s.p + i) != (uint32_t)(uint8_t)w[i]) return false;
    s.p += n;
    return true;
}
// skips any value (the host scanner recurses; here: a bit stack, 1 = inside an object, 0 = inside an array, 64 levels)
template <class S>
__device__ bool sc_value(S& s) {
    uint64_t stack = 0;
    int depth = 0;
    for (;;) {
        // ---- one value starts here
        s.ws();
        if (s.p >= s.end) return false;
        const uint32_t c = s.at(s.p);
        bool opened = false;
        if (c == '"') {
            if (!sc_string(s, nullptr, nullptr, nullptr)) return false;
        } else if (c == '{' || c == '[') {
            s.p++;
            s.ws();
            if (s.p < s.end && s.at(s.p) == (c == '{' ? '}' : ']')) {
                s.p++;  // empty container: a complete value
            } else {
                if (depth >= 64) return false;
                stack = (stack << 1) | (c == '{' ? 1u : 0u);
                depth++;
                opened = true;
            }
        } else if (c == 't') {
            if (!sc_literal(s, "true", 4)) return false;
        } else if (c == 'f') {
            if (!sc_literal(s, "false", 5)) return false;
        } else if (c == 'n') {
            if (!sc_literal(s, "null", 4)) return false;
        } else if (!sc_number(s)) {
            return false;
        }
        if (!opened) {
            // ---- a value has ended: close containers / move to the next member
            for (;;) {
                if (depth == 0) return true;
                s.ws();
                if (s.p >= s.end) return false;
                const uint32_t d = s.at(s.p);
                const bool obj = stack & 1u;
                if (d == ',') {
                    s.p++;
                    break;  // next member of the innermost container
                }
                if (d == (obj ? '}' : ']')) {
                    s.p++;
                    stack >>= 1;
                    depth--;
                    continue;
                }
                return false;
            }
        }
        if (stack & 1u) {  // inside an object: a key and its colon come before the value
            s.ws();
            if (!sc_string(s, nullptr, nullptr, nullptr)) return false;
            s.ws();
            if (s.p >= s.end || s.at(s.p) != ':') return false;
            s.p++;
        }
    }
}

struct ScanArgs {
    const uint8_t* raw;
    uint64_t raw_bytes;
    const uint64_t* line_begin;
    uint64_t n_lines;
    uint8_t ckey[KEY_MAX], ikey[KEY_MAX];
    uint32_t clen, ilen;  // ilen = 0: no id field
    uint8_t* status;
    uint64_t* content_pos;  // position of the content string's opening quote
    uint64_t* content_len;  // its decoded length in bytes (0 unless status OK)
    uint8_t* id_kind;
    uint64_t* id_beg;
    uint64_t* id_end;
};
constexpr uint8_t ST_OK = 0, ST_NOCONTENT = 1, ST_FALLBACK = 2, ST_BLANK = 3;
constexpr uint8_t ID_ABSENT = 0, ID_STRING = 1, ID_TOKEN = 2;  // as MDD_IO_ID_* (include/mdd_io.h)

struct LineResult {
    uint8_t status, idk;
    uint64_t idb, ide, cpos, cdec;
};

// one line that is not blank, cursor at its first byte: the object's members, the content string, the id token
template <class S>
__device__ void scan_line(S& s, const ScanArgs& A, LineResult& R) {
    uint8_t status = ST_FALLBACK, idk = ID_ABSENT;
    uint64_t idb = 0, ide = 0, cpos = 0, cdec = 0;
    bool have_content = false, content_ok = false, ok = true;
    s.ws();
    ok = s.p < s.end && s.at(s.p) == '{';
    if (ok) {
        s.p++;
        s.ws();
        if (s.p < s.end && s.at(s.p) == '}') {
            s.p++;
        } else {
            for (;;) {
                s.ws();
                const uint64_t k0 = s.p + 1;
                bool kplain = true;
                if (!sc_string(s, nullptr, &kplain, nullptr) || !kplain) {  // escaped key: Python decides what it equals
                    ok = false;
                    break;
                }
                const uint64_t klen = s.p - 1 - k0;
                s.ws();
                if (s.p >= s.end || s.at(s.p) != ':') {
                    ok = false;
                    break;
                }
                s.p++;
                s.ws();
                bool is_c = klen == A.clen, is_i = A.ilen && klen == A.ilen;
                for (uint32_t t = 0; is_c && t < A.clen; t++) is_c = s.back(k0 + t) == A.ckey[t];
                for (uint32_t t = 0; is_i && t < A.ilen; t++) is_i = s.back(k0 + t) == A.ikey[t];
                if (is_c) {  // duplicate keys: the last one wins, as in Python's dict
                    have_content = true;
                    content_ok = false;
                    if (s.p < s.end && s.at(s.p) == '"') {
                        bool lone = false;
                        cpos = s.p;
                        if (!sc_string(s, &cdec, nullptr, &lone)) {
                            ok = false;
                            break;
                        }
                        content_ok = !lone;
                    } else if (!sc_value(s)) {
                        ok = false;
                        break;
                    }
                } else if (is_i) {
                    const uint64_t v0 = s.p;
                    if (s.p < s.end && s.at(s.p) == '"') {
                        bool plain = true;
                        if (!sc_string(s, nullptr, &plain, nullptr)) {
                            ok = false;
                            break;
                        }
                        if (plain) idk = ID_STRING, idb = v0 + 1, ide = s.p - 1;
                        else idk = ID_TOKEN, idb = v0, ide = s.p;
                    } else {
                        if (!sc_value(s)) {
                            ok = false;
                            break;
                        }
                        idk = ID_TOKEN, idb = v0, ide = s.p;
                        if (ide - idb == 4 && s.back(v0) == 'n' && s.back(v0 + 1) == 'u' && s.back(v0 + 2) == 'l' && s.back(v0 + 3) == 'l')
                            idk = ID_ABSENT;  // row.get() is None
                    }
                } else if (!sc_value(s)) {
                    ok = false;
                    break;
                }
                s.ws();
                if (s.p < s.end && s.at(s.p) == ',') {
                    s.p++;
                    continue;
                }
                if (s.p < s.end && s.at(s.p) == '}') {
                    s.p++;
                    break;
                }
                ok = false;
                break;
            }
        }
    }
    if (ok) {
        s.ws();
        ok = s.p == s.end;  // json.loads raises on trailing data
    }
    if (ok) ok = s.line_ok();  // (wave scanner: utf-8 and escapes are checked chunk by chunk, over the whole line)
    if (ok) status = (have_content && content_ok) ? ST_OK : ST_NOCONTENT;
    if (status == ST_FALLBACK) idk = ID_ABSENT, idb = ide = 0;
    R.status = status, R.idk = idk, R.idb = idb, R.ide = ide;
    R.cpos = status == ST_OK ? cpos : 0;
    R.cdec = status == ST_OK ? cdec : 0;
}

__device__ __forceinline__ bool py_blank(uint32_t c) { return c == ' ' || (c - 9u <= 4u) || (c - 0x1Cu <= 3u); }  // str.strip's ASCII whitespace

__global__ void __launch_bounds__(256) jsonl_scan_kernel(ScanArgs A) {
    __shared__ __attribute__((aligned(4))) uint8_t lds_win[256 * WIN_STRIDE];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_lines) return;
    uint64_t beg = A.line_begin[i], end = A.line_begin[i + 1];
    end = end < A.raw_bytes ? end : A.raw_bytes;
    beg = beg < end ? beg : end;
    if (end > beg && A.raw[end - 1] == 10u) end--;  // the line without its "\n"
    Sc s{A.raw, beg, end, lds_win + threadIdx.x * WIN_STRIDE, 1ull << 63};
    bool blank = true;
    for (uint64_t q = beg; q < end; q++)
        if (!py_blank(s.at(q))) {
            blank = false;
            break;
        }
    LineResult R{ST_BLANK, ID_ABSENT, 0, 0, 0, 0};
    if (!blank) scan_line(s, A, R);
    A.status[i] = R.status;
    A.content_pos[i] = R.cpos;
    A.content_len[i] = R.cdec;
    if (A.id_kind) {
        A.id_kind[i] = R.idk;
        A.id_beg[i] = R.idb;
        A.id_end[i] = R.ide;
    }
}

// ---- scan, one WAVE per line ------------------------------------------------------------------------------------------------
// A thread per line walks its ~2 KB byte by byte -- a dependent chain of LDS loads and branches -- while 63 other lines of the
// wave wait for the longest of them: 80 % of the extraction's time.  Here the line's bytes sit in the lanes, 64 at a time, and
// what is per-byte work inside a string body happens once per chunk with ballots:
//   * escape starts from the parity of backslash runs (carried across chunks), the bytes each escape consumes, the closing
//     quote = the first quote that is neither;
//   * per lane: the escape's validity (\" \\ \/ \b \f \n \r \t \uXXXX), its decoded utf-8 length, surrogate pairs (a high half
//     directly followed by a low one is ONE code point of 4 bytes; any other half is "lone": the value cannot be encoded);
//   * utf-8: every lead byte checks its own continuation bytes (its dword holds them) and the line's continuation bytes are
//     balanced against what its leads ask for -- so no byte >= 0x80 passes that Python's decoder would reject;
//   * control characters, whitespace and digit runs.
// The grammar is the thread scanner's (scan_line / sc_value / sc_number / sc_literal), written as an explicit automaton so
// that the chunk is loaded at ONE place in the code (a cursor that fetches the next chunk wherever it is advanced compiled
// to 200 KB of code and 248 VGPRs); every value the wave branches on is the same in all lanes.  utf-8 and escape checks
// apply to the WHOLE line here -- outside strings such bytes are no JSON anyway -- which can only turn a line into a fallback.
enum : uint32_t {
    W_START,     // whitespace, then the line's '{'
    W_OPEN_OBJ,  // after '{': whitespace, then '}' (an empty container is a complete value) or the first key
    W_OPEN_ARR,  // after '[': whitespace, then ']' or the first element
    W_KEY,       // whitespace, then the '"' of a key
    W_COLON,     // whitespace, then ':'
    W_VALUE,     // whitespace, then the first byte of a value
    W_AFTER,     // a value has ended: whitespace, then ',' or the closer of the innermost container
    W_END,       // the object is closed: only whitespace may follow
    W_STRING,    // inside a string body (the states from here on take no whitespace)
    W_NUM_MINUS,
    W_NUM_ZERO,
    W_NUM_INT,
    W_NUM_FRAC0,
    W_NUM_FRAC,
    W_NUM_EXP0,
    W_NUM_EXP1,
    W_NUM_EXP,
    W_LITERAL,   // the rest of true / false / null
    W_FAIL
};
enum : uint32_t { P_KEY_NESTED, P_KEY_TOP, P_CONTENT, P_ID, P_OTHER };  // what the string being crossed is
enum : uint32_t { K_NONE, K_CONTENT, K_ID };                            // which member of the line's object the next value belongs to

__device__ __forceinline__ bool hex4v(const uint8_t* __restrict__ r, uint64_t q, uint32_t& cp) {
    const int h0 = hexv(r[q]), h1 = hexv(r[q + 1]), h2 = hexv(r[q + 2]), h3 = hexv(r[q + 3]);
    cp = (uint32_t)((h0 & 15) << 12 | (h1 & 15) << 8 | (h2 & 15) << 4 | (h3 & 15));
    return (h0 | h1 | h2 | h3) >= 0;
}

// one wave per line, persistent: wave w takes lines w, w + #waves, ...
__global__ void __launch_bounds__(256) jsonl_scan_wave_kernel(ScanArgs A) {
    const uint32_t lane = lane_id();
    const uint8_t* __restrict__ b = A.raw;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t w0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    for (uint64_t i = w0; i < A.n_lines; i += nw) {
        uint64_t beg = A.line_begin[i], end = A.line_begin[i + 1];
        end = end < A.raw_bytes ? end : A.raw_bytes;
        beg = beg < end ? beg : end;
        if (end > beg && b[end - 1] == 10u) end--;  // the line without its "\n"
        bool blank = true;
        for (uint64_t q0 = beg; q0 < end; q0 += 64) {
            const uint64_t q = q0 + lane;
            const uint32_t x = q < end ? b[q] : (uint32_t)' ';
            if (__ballot(!py_blank(x))) {
                blank = false;
                break;
            }
        }
        uint8_t status = ST_BLANK, idk = ID_ABSENT;
        uint64_t idb = 0, ide = 0, cpos = 0, cdec = 0;
        if (!blank) {
            // ---- the chunk in the lanes and its masks
            uint64_t p = beg, base = beg;
            bool loaded = false;
            uint32_t c = ' ';
            // the NEXT chunk's bytes are requested while this one is worked on: a wave crosses its line chunk by chunk, and a
            // load issued only when the chunk is needed left it waiting a memory round trip per 64 bytes (17 ms per 2.2 GB)
            // (unconditional loads at an address clamped to the buffer's padding, masked when consumed: a load under a branch
            // makes the compiler wait for it where the branch joins)
            auto fetch = [&](uint64_t q) -> uint32_t { return *reinterpret_cast<const mdd_u32_a1*>(b + (q < A.raw_bytes ? q : A.raw_bytes)); };
            uint32_t w_next = fetch(beg + lane);
            uint64_t Q = 0, S = 0, CTRL = 0, NWS = 0, LONE = 0, C0 = 0, C1 = 0, C2 = 0, DIG = 0;
            uint64_t carry_s = 0, prev_S = 0, prev_SU = 0;  // previous chunk: its last byte starts an escape; its escape starts; its \u starts
            uint64_t need = 0, conts = 0;                   // continuation bytes the line's leads ask for / continuation bytes seen
            bool bad = false;                               // an invalid escape or utf-8 sequence somewhere in the line
            // ---- the automaton
            uint32_t state = W_START, purpose = P_OTHER, pend = K_NONE, depth = 0, lit_left = 0;
            uint64_t stack = 0, lit = 0;  // stack: the containers INSIDE the line's object, 1 = object (64 levels); lit: bytes still to match
            uint64_t str_start = 0, str_n = 0, id_v0 = 0;
            bool str_plain = true, str_lone = false, id_open = false, have_content = false, content_ok = false;
#define VALUE_DONE(pos)                                                                          \
    do {                                                                                         \
        if (depth == 1u) {                                                                       \
            if (id_open) {                                                                       \
                idk = ID_TOKEN, idb = id_v0, ide = (pos), id_open = false;                       \
                if (ide - idb == 4 && b[idb] == 'n' && b[idb + 1] == 'u' && b[idb + 2] == 'l' && b[idb + 3] == 'l') \
                    idk = ID_ABSENT; /* row.get() is None */                                     \
            }                                                                                    \
            pend = K_NONE;                                                                       \
        }                                                                                        \
        state = W_AFTER;                                                                         \
    } while (0)
            for (;;) {
                if (state == W_FAIL || p >= end) break;
                if (!loaded || p - base >= 64) {
                    base = loaded ? base + 64 : beg;
                    loaded = true;
                    const uint64_t q = base + lane;
                    const bool in = q < end;
                    // the lane's byte and the three behind it (bytes past `end` are the next line's or the buffer's padding:
                    // every use below checks positions against `end`)
                    const uint32_t w = in ? w_next : 0x20202020u;
                    w_next = fetch(q + 64);
                    c = w & 0xFFu;
                    const uint32_t c1 = (w >> 8) & 0xFFu, c2 = (w >> 16) & 0xFFu, c3 = w >> 24;
                    bool badlane = false;
                    uint32_t cnt = 1;
                    if (__ballot(c >= 0x80u)) {
                        uint32_t k = 0;
                        if (c >= 0xC0u) {
                            k = (c >= 0xC2u && c <= 0xDFu) ? 1u : (c >= 0xE0u && c <= 0xEFu) ? 2u : (c >= 0xF0u && c <= 0xF4u) ? 3u : 0u;
                            bool ok = k != 0u && q + k < end && (c1 & 0xC0u) == 0x80u && (k < 2u || (c2 & 0xC0u) == 0x80u) &&
                                      (k < 3u || (c3 & 0xC0u) == 0x80u);
                            ok = ok && !((c == 0xE0u && c1 < 0xA0u) || (c == 0xEDu && c1 >= 0xA0u) || (c == 0xF0u && c1 < 0x90u) ||
                                         (c == 0xF4u && c1 >= 0x90u));
                            badlane = !ok;
                        }
                        conts += __builtin_popcountll(__ballot((c & 0xC0u) == 0x80u));
                        need += __builtin_popcountll(__ballot(k & 1u)) + 2 * __builtin_popcountll(__ballot(k & 2u));
                    }
                    if (__ballot(badlane)) bad = true;
                    badlane = false;
                    const uint64_t B = __ballot(c == '\\');
                    // the common chunk: 64 bytes of a string body without a quote, a backslash or a control character, no
                    // escape reaching in from the chunk before.  Nothing else has to be known about it (the automaton's
                    // masks and its dispatch are ~200 scalar instructions per chunk, and the scalar unit is what this
                    // kernel waits for): 64 more decoded bytes, on to the next chunk
                    if (state == W_STRING && p == base && base + 64 <= end && !(B | carry_s | prev_SU) &&
                        !__ballot(c == '"' || c < 0x20u)) {
                        str_n += 64;
                        p = base + 64;
                        prev_S = 0;
                        // printable ASCII goes on 256 bytes at a time, a dword per lane (any byte >= 0x80 or < 0x20, a quote or a
                        // backslash ends this); the chunks then continue from wherever it stopped
                        bool moved = false;
                        while (p + 256 <= end) {
                            const uint32_t x = *reinterpret_cast<const mdd_u32_a1*>(b + p + 4 * lane);
                            const uint32_t special = (x | (x - 0x20202020u)) & 0x80808080u;  // high bit set, or below 0x20 (no borrow
                            if (__ballot((special | swar_eq(x, '"') | swar_eq(x, '\\')) != 0u)) break;  // crosses a byte >= 0x20)
                            str_n += 256, p += 256, moved = true;
                        }
                        if (moved) {
                            base = p - 64;  // (the next load takes the chunk at p)
                            w_next = fetch(p + lane);
                        }
                        continue;
                    }
                    uint64_t body = 0, SU = 0;
                    S = 0, LONE = 0;
                    if (B || carry_s || prev_SU) {
                        // escape starts: a backslash with an even number of backslashes directly before it in its run
                        bool st = false;
                        if (c == '\\') {
                            uint32_t r = 0;
                            if (lane) {
                                const uint64_t x = B << (64 - lane);  // the lanes below, aligned to the top
                                r = (uint32_t)__builtin_clzll(~x);     // consecutive backslashes directly below (64 when x is all ones)
                                if (r > lane) r = lane;
                            }
                            st = ((r + ((r == lane && carry_s) ? 1u : 0u)) & 1u) == 0u;
                        }
                        S = __ballot(st);
                        const uint32_t e = (st && q + 1 < end) ? c1 : 0u;
                        SU = __ballot(st && e == 'u');
                        body = (S << 1) | carry_s | (SU << 2) | (SU << 3) | (SU << 4) | (SU << 5) | (prev_SU >> 62) | (prev_SU >> 61) |
                               (prev_SU >> 60) | (prev_SU >> 59);
                        bool lonelane = false;
                        if (st) {
                            if (e == 'u') {
                                uint32_t cp = 0;
                                if (q + 5 < end && hex4v(b, q + 2, cp)) {
                                    if (cp - 0xD800u <= 0x3FFu) {
                                        // a high half pairs with a low half that follows directly (Python's decoder); a following
                                        // \u with bad digits is flagged by its own lane
                                        uint32_t lo = 0;
                                        const bool paired = q + 11 < end && b[q + 6] == '\\' && b[q + 7] == 'u' && hex4v(b, q + 8, lo) &&
                                                            lo - 0xDC00u <= 0x3FFu;
                                        if (paired) cnt = 4;
                                        else lonelane = true;
                                    } else if (cp - 0xDC00u <= 0x3FFu) {
                                        // a low half belongs to the escape 6 bytes before it when that one is a high half
                                        bool paired = false;
                                        if (q >= beg + 6 && b[q - 6] == '\\' && b[q - 5] == 'u') {
                                            uint32_t hi = 0;
                                            const bool st6 = lane >= 6 ? (((S >> (lane - 6)) & 1u) != 0) : (((prev_S >> (lane + 58)) & 1u) != 0);
                                            paired = st6 && hex4v(b, q - 4, hi) && hi - 0xD800u <= 0x3FFu;
                                        }
                                        if (paired) cnt = 0;
                                        else lonelane = true;
                                    } else {
                                        cnt = cp < 0x80u ? 1u : (cp < 0x800u ? 2u : 3u);
                                    }
                                } else {
                                    badlane = true;
                                }
                            } else if (!(e == '"' || e == '\\' || e == '/' || e == 'b' || e == 'f' || e == 'n' || e == 'r' || e == 't')) {
                                badlane = true;  // includes a backslash that is the line's last byte (e = 0)
                            }
                        } else if ((body >> lane) & 1u) {
                            cnt = 0;
                        }
                        LONE = __ballot(lonelane);
                    }
                    if (__ballot(badlane)) bad = true;
                    C0 = __ballot(cnt & 1u), C1 = __ballot(cnt & 2u), C2 = __ballot(cnt & 4u);
                    Q = __ballot(c == '"') & ~body;
                    CTRL = __ballot(in && c < 0x20u);
                    NWS = __ballot(!(c == ' ' || c == '\t' || c == '\n' || c == '\r'));
                    DIG = __ballot(c - '0' <= 9u);
                    carry_s = S >> 63;
                    prev_S = S;
                    prev_SU = SU;
                    continue;
                }
                uint64_t ge = ~0ull << (uint32_t)(p - base);
                if (state <= W_END) {  // the states that begin with whitespace
                    const uint64_t m = NWS & ge;
                    if (!m) {
                        p = base + 64;
                        continue;
                    }
                    p = base + (uint64_t)__builtin_ctzll(m);  // (< end: the lanes at and beyond `end` hold spaces)
                    ge = ~0ull << (uint32_t)(p - base);
                }
                const uint32_t ch = (uint32_t)__builtin_amdgcn_readlane((int)c, __builtin_amdgcn_readfirstlane((int)(uint32_t)(p - base)));
                const bool in_obj = depth == 1u || (stack & 1u);
                switch (state) {
                    case W_START:
                        if (ch == '{') p++, state = W_OPEN_OBJ;
                        else state = W_FAIL;
                        break;
                    case W_OPEN_OBJ:
                        if (ch == '}') {
                            p++;
                            if (depth == 0u) state = W_END;  // the line is "{}"
                            else VALUE_DONE(p);
                        } else if (depth >= 65u) {
                            state = W_FAIL;  // deeper than the host scanner follows
                        } else {
                            if (depth) stack = (stack << 1) | 1u;
                            depth++;
                            state = W_KEY;
                        }
                        break;
                    case W_OPEN_ARR:
                        if (ch == ']') {
                            p++;
                            VALUE_DONE(p);
                        } else if (depth >= 65u) {
                            state = W_FAIL;
                        } else {
                            stack = stack << 1;
                            depth++;
                            state = W_VALUE;
                        }
                        break;
                    case W_KEY:
                        if (ch == '"') {
                            str_start = p, str_n = 0, str_plain = true, str_lone = false;
                            purpose = depth == 1u ? P_KEY_TOP : P_KEY_NESTED;
                            p++, state = W_STRING;
                        } else {
                            state = W_FAIL;
                        }
                        break;
                    case W_COLON:
                        if (ch == ':') p++, state = W_VALUE;
                        else state = W_FAIL;
                        break;
                    case W_VALUE: {
                        const bool top = depth == 1u;
                        if (top && pend == K_CONTENT) have_content = true, content_ok = false;  // duplicate keys: the last one wins
                        if (ch == '"') {
                            str_start = p, str_n = 0, str_plain = true, str_lone = false;
                            purpose = !top ? P_OTHER : pend == K_CONTENT ? P_CONTENT : pend == K_ID ? P_ID : P_OTHER;
                            p++, state = W_STRING;
                            break;
                        }
                        if (top && pend == K_ID) id_v0 = p, id_open = true;
                        if (ch == '{') p++, state = W_OPEN_OBJ;
                        else if (ch == '[') p++, state = W_OPEN_ARR;
                        else if (ch == 't') p++, state = W_LITERAL, lit = 0x657572ull, lit_left = 3;      // "rue"
                        else if (ch == 'f') p++, state = W_LITERAL, lit = 0x65736C61ull, lit_left = 4;    // "alse"
                        else if (ch == 'n') p++, state = W_LITERAL, lit = 0x6C6C75ull, lit_left = 3;      // "ull"
                        else if (ch == '-') p++, state = W_NUM_MINUS;
                        else if (ch == '0') p++, state = W_NUM_ZERO;
                        else if (ch - '1' <= 8u) p++, state = W_NUM_INT;
                        else state = W_FAIL;  // includes NaN / Infinity, which Python accepts: the host decides
                        break;
                    }
                    case W_AFTER:
                        if (ch == ',') {
                            p++, state = in_obj ? W_KEY : W_VALUE;
                        } else if (ch == (in_obj ? (uint32_t)'}' : (uint32_t)']')) {
                            p++;
                            if (depth == 1u) {
                                depth = 0, state = W_END;
                            } else {
                                stack >>= 1, depth--;
                                VALUE_DONE(p);
                            }
                        } else {
                            state = W_FAIL;
                        }
                        break;
                    case W_END:
                        state = W_FAIL;  // json.loads raises on trailing data
                        break;
                    case W_STRING: {
                        const uint64_t m = Q & ge;
                        const uint32_t stop = m ? (uint32_t)__builtin_ctzll(m) : 64u;
                        const uint64_t region = stop < 64u ? (ge & ((1ull << stop) - 1ull)) : ge;  // the body's bytes in this chunk
                        if (CTRL & region) {  // Python's json (strict) rejects raw control characters
                            state = W_FAIL;
                            break;
                        }
                        str_n += __builtin_popcountll(C0 & region) + 2 * __builtin_popcountll(C1 & region) + 4 * __builtin_popcountll(C2 & region);
                        if (S & region) str_plain = false;
                        if (LONE & region) str_lone = true;
                        if (!m) {
                            p = base + 64;
                            break;
                        }
                        const uint64_t close = base + stop;
                        p = close + 1;
                        if (purpose == P_KEY_NESTED) {
                            state = W_COLON;
                        } else if (purpose == P_KEY_TOP) {
                            if (!str_plain) {  // escaped key: Python decides what it equals
                                state = W_FAIL;
                                break;
                            }
                            const uint64_t k0 = str_start + 1, klen = close - k0;
                            bool is_c = klen == A.clen, is_i = A.ilen && klen == A.ilen;
                            for (uint32_t t = 0; is_c && t < A.clen; t++) is_c = b[k0 + t] == A.ckey[t];
                            for (uint32_t t = 0; is_i && t < A.ilen; t++) is_i = b[k0 + t] == A.ikey[t];
                            pend = is_c ? K_CONTENT : is_i ? K_ID : K_NONE;
                            state = W_COLON;
                        } else {
                            if (purpose == P_CONTENT) {
                                cpos = str_start, cdec = str_n, content_ok = !str_lone;
                            } else if (purpose == P_ID) {
                                if (str_plain) idk = ID_STRING, idb = str_start + 1, ide = close;
                                else idk = ID_TOKEN, idb = str_start, ide = p;
                            }
                            VALUE_DONE(p);
                        }
                        break;
                    }
                    case W_NUM_MINUS:
                        if (ch == '0') p++, state = W_NUM_ZERO;
                        else if (ch - '1' <= 8u) p++, state = W_NUM_INT;
                        else state = W_FAIL;
                        break;
                    case W_NUM_ZERO:
                        if (ch == '.') p++, state = W_NUM_FRAC0;
                        else if (ch == 'e' || ch == 'E') p++, state = W_NUM_EXP0;
                        else VALUE_DONE(p);
                        break;
                    case W_NUM_INT:
                    case W_NUM_FRAC:
                    case W_NUM_EXP:
                        if (ch - '0' <= 9u) {  // the rest of the digit run in this chunk
                            const uint64_t m = ~DIG & ge;
                            p = m ? base + (uint64_t)__builtin_ctzll(m) : base + 64;
                        } else if (ch == '.' && state == W_NUM_INT) {
                            p++, state = W_NUM_FRAC0;
                        } else if ((ch == 'e' || ch == 'E') && state != W_NUM_EXP) {
                            p++, state = W_NUM_EXP0;
                        } else {
                            VALUE_DONE(p);
                        }
                        break;
                    case W_NUM_FRAC0:
                        if (ch - '0' <= 9u) p++, state = W_NUM_FRAC;
                        else state = W_FAIL;
                        break;
                    case W_NUM_EXP0:
                        if (ch == '+' || ch == '-') p++;
                        state = W_NUM_EXP1;
                        break;
                    case W_NUM_EXP1:
                        if (ch - '0' <= 9u) p++, state = W_NUM_EXP;
                        else state = W_FAIL;
                        break;
                    case W_LITERAL:
                        if (ch == (uint32_t)(lit & 0xFFu)) {
                            p++, lit >>= 8;
                            if (--lit_left == 0u) VALUE_DONE(p);
                        } else {
                            state = W_FAIL;
                        }
                        break;
                    default:
                        state = W_FAIL;
                        break;
                }
            }
#undef VALUE_DONE
            // the loop ends at the end of the line (every chunk has been in the lanes by then) or on a failure
            const bool ok = state == W_END && !bad && need == conts;
            status = !ok ? ST_FALLBACK : (have_content && content_ok) ? ST_OK : ST_NOCONTENT;
            if (status == ST_FALLBACK) idk = ID_ABSENT, idb = ide = 0;
            if (status != ST_OK) cpos = cdec = 0;
        }
        if (lane == 0) {
            A.status[i] = status;
            A.content_pos[i] = cpos;
            A.content_len[i] = cdec;
            if (A.id_kind) {
                A.id_kind[i] = idk;
                A.id_beg[i] = idb;
                A.id_end[i] = ide;
            }
        }
    }
}

// ---- decode: one wave per line ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t enc_utf8(uint32_t cp, uint32_t& len) {
    if (cp < 0x80u) {
        len = 1;
        return cp;
    }
    if (cp < 0x800u) {
        len = 2;
        return (0xC0u | (cp >> 6)) | ((0x80u | (cp & 0x3Fu)) << 8);
    }
    if (cp < 0x10000u) {
        len = 3;
        return (0xE0u | (cp >> 12)) | ((0x80u | ((cp >> 6) & 0x3Fu)) << 8) | ((0x80u | (cp & 0x3Fu)) << 16);
    }
    len = 4;
    return (0xF0u | (cp >> 18)) | ((0x80u | ((cp >> 12) & 0x3Fu)) << 8) | ((0x80u | ((cp >> 6) & 0x3Fu)) << 16) | ((0x80u | (cp & 0x3Fu)) << 24);
}
__device__ __forceinline__ uint32_t hex4(const uint8_t* __restrict__ r, uint64_t q) {  // validated by the scan: four hex digits
    return (uint32_t)(hexv(r[q]) << 12 | hexv(r[q + 1]) << 8 | hexv(r[q + 2]) << 4 | hexv(r[q + 3]));
}

__global__ void __launch_bounds__(256) jsonl_decode_kernel(const uint8_t* __restrict__ raw, uint64_t raw_bytes, uint64_t n_lines,
                                                           const uint8_t* __restrict__ status, const uint64_t* __restrict__ content_pos,
                                                           const uint64_t* __restrict__ out_offs, uint8_t* __restrict__ out_text,
                                                           uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n_lines; i += nw) {
        if (status[i] != ST_OK) continue;
        const uint64_t start = content_pos[i] + 1;  // first byte after the opening quote
        uint64_t done = out_offs[i];
        uint64_t carry_s = 0, prev_su = 0;  // previous chunk: its last byte starts an escape; its \u starters
        // (requested a chunk ahead, as in the scan: unconditional, the address clamped to the padding, masked when consumed)
        uint32_t c_next = raw[start + lane < raw_bytes ? start + lane : raw_bytes];
        for (uint64_t q0 = start; q0 < raw_bytes; q0 += 64) {  // (ends at the closing quote: the scan found one)
            if (!(carry_s | prev_su)) {
                // 256 bytes at a time while there is neither a quote nor a backslash among them: a copy, a dword per lane
                bool moved = false;
                while (q0 + 256 <= raw_bytes && done + 256 <= out_cap) {
                    const uint32_t w = *reinterpret_cast<const mdd_u32_a1*>(raw + q0 + 4 * lane);
                    if (__ballot((swar_eq(w, '"') | swar_eq(w, '\\')) != 0u)) break;
                    *reinterpret_cast<mdd_u32_a1*>(out_text + done + 4 * lane) = w;
                    done += 256, q0 += 256, moved = true;
                }
                if (moved) c_next = raw[q0 + lane < raw_bytes ? q0 + lane : raw_bytes];
            }
            const uint64_t q = q0 + lane;
            const uint32_t c = q < raw_bytes ? c_next : (uint32_t)'"';
            c_next = raw[q + 64 < raw_bytes ? q + 64 : raw_bytes];
            const uint64_t B = __ballot(c == '\\');
            uint32_t cnt = 1, enc = c;
            uint64_t S = 0, SU = 0, body = 0;
            if (B || carry_s || prev_su) {
                // escape starts: a backslash with an even number of backslashes directly before it in its run
                bool s = false;
                if (c == '\\') {
                    uint32_t r = 0;
                    if (lane) {
                        const uint64_t x = B << (64 - lane);  // the lanes below, aligned to the top
                        r = (uint32_t)__builtin_clzll(~x);     // consecutive backslashes directly below (64 when x is all ones)
                        if (r > lane) r = lane;
                    }
                    s = ((r + ((r == lane && carry_s) ? 1u : 0u)) & 1u) == 0u;
                }
                S = __ballot(s);
                uint32_t e = 0;
                if (s) e = q + 1 < raw_bytes ? raw[q + 1] : 0u;
                SU = __ballot(s && e == 'u');
                body = (S << 1) | carry_s | (SU << 2) | (SU << 3) | (SU << 4) | (SU << 5) | (prev_su >> 62) | (prev_su >> 61) |
                       (prev_su >> 60) | (prev_su >> 59);
                if (s) {
                    uint32_t cp;
                    switch (e) {
                        case 'b': cp = 8; break;
                        case 'f': cp = 12; break;
                        case 'n': cp = 10; break;
                        case 'r': cp = 13; break;
                        case 't': cp = 9; break;
                        case 'u': cp = hex4(raw, q + 2); break;
                        default: cp = e; break;  // " \ /
                    }
                    if (e == 'u' && cp - 0xD800u <= 0x3FFu) cp = 0x10000u + ((cp - 0xD800u) << 10) + (hex4(raw, q + 8) - 0xDC00u);  // its partner follows (scan)
                    if (e == 'u' && cp - 0xDC00u <= 0x3FFu) cnt = 0, enc = 0;  // the low half: emitted by its partner
                    else enc = enc_utf8(cp, cnt);
                } else if ((body >> lane) & 1u) {
                    cnt = 0;
                }
            }
            // the closing quote: the first quote that is neither an escape start nor part of an escape
            const uint64_t E = __ballot(c == '"') & ~body;
            const uint32_t stop = E ? (uint32_t)__builtin_ctzll(E) : 64u;
            if (lane >= stop) cnt = 0;
            const uint64_t c0 = __ballot(cnt & 1u), c1 = __ballot(cnt & 2u), c2 = __ballot(cnt & 4u);
            if (cnt) {
                const uint64_t pos = done + __builtin_popcountll(c0 & lt) + 2 * __builtin_popcountll(c1 & lt) + 4 * __builtin_popcountll(c2 & lt);
                if (cnt == 1u) {
                    if (pos < out_cap) out_text[pos] = (uint8_t)enc;  // out_offsets are the caller's: stay inside out_text
                } else {
                    store_low_bytes(out_text, pos, out_cap, enc, cnt);
                }
            }
            done += __builtin_popcountll(c0) + 2 * __builtin_popcountll(c1) + 4 * __builtin_popcountll(c2);
            if (E) break;
            carry_s = S >> 63;
            prev_su = SU;
        }
    }
}

}  // namespace

extern "C" {

int mdd_jsonl_count_lines(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, uint32_t* tile_counts, uint64_t* totals) {
    REQUIRE(ctx, "mdd_jsonl_count_lines: ctx is NULL");
    REQUIRE(totals, "mdd_jsonl_count_lines: NULL totals");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(totals, 0, 3 * sizeof(uint64_t), ctx->stream));
    if (raw_bytes == 0) return MDD_OK;
    REQUIRE(raw && tile_counts, "mdd_jsonl_count_lines: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(raw) & 3u) == 0, "mdd_jsonl_count_lines: raw must be 4-byte aligned");
    const uint64_t tiles = (raw_bytes + TILE_BYTES - 1) / TILE_BYTES;
    REQUIRE(tiles <= 0x7FFFFFFFull, "mdd_jsonl_count_lines: shard too large for one launch (32 TiB)");
    hipLaunchKernelGGL(jsonl_index_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, raw, raw_bytes, tile_counts,
                       reinterpret_cast<unsigned long long*>(totals), (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0);
    hipLaunchKernelGGL(tile_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile_counts, tiles, reinterpret_cast<unsigned long long*>(totals));
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

int mdd_jsonl_line_begins(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, const uint64_t* tile_base, uint64_t* line_begin,
                          uint64_t n_lines) {
    REQUIRE(ctx, "mdd_jsonl_line_begins: ctx is NULL");
    REQUIRE(line_begin, "mdd_jsonl_line_begins: NULL line_begin");
    HIP_TRY(hipSetDevice(ctx->device));
    // line 0 starts at 0; the last entry (the end of the shard, whether or not its last line has a "\n") is written by the kernel
    HIP_TRY(hipMemsetAsync(line_begin, 0, sizeof(uint64_t), ctx->stream));
    if (raw_bytes == 0 || n_lines == 0) return MDD_OK;
    REQUIRE(raw && tile_base, "mdd_jsonl_line_begins: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(raw) & 3u) == 0, "mdd_jsonl_line_begins: raw must be 4-byte aligned");
    const uint64_t tiles = (raw_bytes + TILE_BYTES - 1) / TILE_BYTES;
    REQUIRE(tiles <= 0x7FFFFFFFull, "mdd_jsonl_line_begins: shard too large for one launch");
    hipLaunchKernelGGL(jsonl_index_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, raw, raw_bytes, (uint32_t*)nullptr,
                       (unsigned long long*)nullptr, tile_base, line_begin, n_lines);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

int mdd_jsonl_scan(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, const uint64_t* line_begin, uint64_t n_lines,
                   const char* content_key, const char* id_key, uint8_t* status, uint64_t* content_pos, uint64_t* content_len,
                   uint8_t* id_kind, uint64_t* id_beg, uint64_t* id_end) {
    REQUIRE(ctx, "mdd_jsonl_scan: ctx is NULL");
    REQUIRE(content_key, "mdd_jsonl_scan: NULL content_key");
    const size_t cl = strlen(content_key), il = id_key ? strlen(id_key) : 0;
    // a key that needs escaping in JSON cannot be matched byte-wise, and the kernel holds 64 bytes per key: such shards
    // belong to the host reader
    bool simple = cl >= 1 && cl <= (size_t)KEY_MAX && il <= (size_t)KEY_MAX;
    for (size_t i = 0; i < cl; i++) simple = simple && (uint8_t)content_key[i] >= 0x20 && content_key[i] != '"' && content_key[i] != '\\';
    for (size_t i = 0; i < il; i++) simple = simple && (uint8_t)id_key[i] >= 0x20 && id_key[i] != '"' && id_key[i] != '\\';
    if (!simple) return fail(MDD_ERR_UNSUPPORTED, "mdd_jsonl_scan: field names must be 1..%d bytes without '\"', '\\\\' or control characters", KEY_MAX);
    if (n_lines == 0) return MDD_OK;
    REQUIRE(raw && line_begin && status && content_pos && content_len, "mdd_jsonl_scan: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15u) == 0, "mdd_jsonl_scan: raw must be 16-byte aligned");
    REQUIRE(il == 0 || (id_kind && id_beg && id_end), "mdd_jsonl_scan: id outputs are NULL");
    REQUIRE(n_lines <= MAX_ELEMS_PER_LAUNCH, "mdd_jsonl_scan: too many lines for one launch");
    HIP_TRY(hipSetDevice(ctx->device));
    ScanArgs A{};
    A.raw = raw;
    A.raw_bytes = raw_bytes;
    A.line_begin = line_begin;
    A.n_lines = n_lines;
    memcpy(A.ckey, content_key, cl);
    A.clen = (uint32_t)cl;
    if (il) memcpy(A.ikey, id_key, il);
    A.ilen = (uint32_t)il;
    A.status = status;
    A.content_pos = content_pos;
    A.content_len = content_len;
    A.id_kind = il ? id_kind : nullptr;
    A.id_beg = id_beg;
    A.id_end = id_end;
    // TEST HOOK: MDD_JSONL_SCAN=thread runs the thread-per-line scanner (the first implementation, kept as the independent
    // one the tests hold the wave scanner against)
    const char* how = getenv("MDD_JSONL_SCAN");
    if (how && strcmp(how, "thread") == 0) {
        hipLaunchKernelGGL(jsonl_scan_kernel, dim3(grid_for(n_lines, 256)), dim3(256), 0, ctx->stream, A);
    } else {
        const uint64_t blocks = (n_lines + 3) / 4, cap = (uint64_t)ctx->cus * 6;  // 24 waves per CU: all resident (106 SGPRs)
        hipLaunchKernelGGL(jsonl_scan_wave_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, ctx->stream, A);
    }
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

int mdd_jsonl_decode(mdd_ctx* ctx, const uint8_t* raw, uint64_t raw_bytes, uint64_t n_lines, const uint8_t* status,
                     const uint64_t* content_pos, const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity) {
    REQUIRE(ctx, "mdd_jsonl_decode: ctx is NULL");
    if (n_lines == 0 || raw_bytes == 0) return MDD_OK;
    REQUIRE(raw && status && content_pos && out_offsets && (out_text || out_capacity == 0), "mdd_jsonl_decode: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    uint64_t blocks = (n_lines + 3) / 4, cap = (uint64_t)ctx->cus * 64;
    hipLaunchKernelGGL(jsonl_decode_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, ctx->stream, raw, raw_bytes,
                       n_lines, status, content_pos, out_offsets, out_text, out_capacity);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

}  // extern "C"
