// EN text normalisation on the device.
// Restates normalize_text, Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/
// minhashLsh.py:31-40.  Pure-ASCII stretches need no Unicode tables:
//   ftfy.fix_text(normalization="NFC")  identity on clean ASCII (NFC never changes ASCII; with ftfy
//                                       installed the host keeps doing this step, see hostio.py)
//   .lower()                            A-Z -> a-z
//   translate(PUNCTUATION -> None)      the 32 characters of string.punctuation are deleted
//   .strip()                            leading / trailing whitespace removed
//   re.sub(r"\s+", " ")                 every remaining whitespace run becomes one space
// ASCII whitespace as Python's str.isspace / re \s see it: 0x09-0x0D, 0x1C-0x1F, 0x20.
// Round 3: 256-byte steps that hold a byte >= 0x80 take a table-driven path (mdd_unicode_tab.h, generated from the
// interpreter's own unicodedata / str.lower / str.isspace by tools/gen_unicode_tables.py): utf-8 is validated and decoded,
// every code point is looked up -- 1:1 lower case, Unicode whitespace (what str.strip / re's \s see), and a "cannot vouch"
// flag for everything where the Python steps are not per-character (not NFC-inert: combining marks, composable or
// decomposable characters; U+0130 and U+03A3, whose lower() is not 1:1 / context-free; unassigned code points; invalid
// utf-8).  Only documents that hold a flagged code point are handed back (flag = 1, no output): the caller runs the
// Python steps on exactly those, as it did for every non-ASCII document before.  (The byte-per-lane kernel for unaligned
// text keeps flagging every non-ASCII document.)  One wave per document, 64 bytes per step (4 steps per load batch); the output position of
// every byte comes from ballots + popcounts, so there is no scan and no LDS.  HBM-bound: the text
// is read twice (count, write) and the normalised text written once.
#include "mdd_common.h"
#include "mdd_unicode_tab.h"

namespace {
using namespace mddi;

__device__ __forceinline__ bool is_punct(uint32_t c) {
    return (c - 33u <= 14u) | (c - 58u <= 6u) | (c - 91u <= 5u) | (c - 123u <= 3u);
}
__device__ __forceinline__ bool is_space(uint32_t c) { return (c - 9u <= 4u) | (c - 28u <= 4u); }

template <bool WRITE>
__global__ void __launch_bounds__(256) norm_en_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                      const uint64_t* __restrict__ offs, uint64_t n, uint64_t* out_len,
                                                      uint8_t* non_ascii, const uint64_t* __restrict__ out_offs,
                                                      uint8_t* __restrict__ out_text, uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;  // lanes below this one
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        if (WRITE && non_ascii[d]) continue;
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t o = WRITE ? out_offs[d] : 0;
        uint64_t cnt = 0;
        bool prev_ws = false, seen = false, bad = false;  // wave-uniform carries between 64-byte steps
        for (uint64_t p = b; p < e && !bad; p += 256) {
            uint32_t cc[4];  // four independent loads in flight: the loop is latency-bound otherwise
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = p + 64 * j + lane;
                cc[j] = i < e ? text[i] : 0u;
            }
            if (!WRITE && __ballot((cc[0] | cc[1] | cc[2] | cc[3]) >= 128u)) {
                bad = true;
                break;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c = cc[j];
                const bool in = p + 64 * j + lane < e;
                const bool keep = in && !is_punct(c);
                const bool ws = is_space(c);
                const uint64_t K = __ballot(keep), W = __ballot(keep && ws), N = K & ~W;
                const bool is_n = (N >> lane) & 1;
                const uint64_t pk = K & lt;
                const bool pws = pk ? ((W >> (63 - __builtin_clzll(pk))) & 1) : prev_ws;  // previous kept byte is whitespace
                const bool sp = is_n && pws && (seen || (N & lt));                          // a space precedes this byte
                const uint64_t S = __ballot(sp);
                if (WRITE && is_n) {
                    uint64_t pos = o + cnt + __builtin_popcountll(N & lt) + __builtin_popcountll(S & lt);
                    if (sp && pos < out_cap) out_text[pos] = ' ';
                    pos += sp;
                    if (pos < out_cap) out_text[pos] = (uint8_t)((c - 65u <= 25u) ? c + 32u : c);  // out_offsets are the caller's: stay inside out_text
                }
                cnt += __builtin_popcountll(N) + __builtin_popcountll(S);
                if (K) prev_ws = (W >> (63 - __builtin_clzll(K))) & 1;
                seen = seen || N;
            }
        }
        if (!WRITE && lane == 0) {
            out_len[d] = bad ? 0 : cnt;
            non_ascii[d] = bad;
        }
    }
}

// table entry of a code point >= 0x80 (mdd_unicode_tab.h): MDD_UNI_FALLBACK | MDD_UNI_SPACE | 21-bit (lower(c) - c)
__device__ __forceinline__ uint32_t uni_entry(uint32_t cp) {
    if (cp < MDD_UNI_LIMIT) return mdd_uni_blk[((uint32_t)mdd_uni_idx[cp >> 8] << 8) | (cp & 255u)];
    for (int r = 0; r < MDD_UNI_NRANGES; r++)
        if (cp - mdd_uni_hi[r][0] <= mdd_uni_hi[r][1] - mdd_uni_hi[r][0]) return 0u;  // uncased, not whitespace
    return MDD_UNI_FALLBACK;
}
// utf-8 of cp as little-endian bytes of the result, length in len (1..4)
__device__ __forceinline__ uint32_t utf8_encode(uint32_t cp, uint32_t& len) {
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

// One lane's 4 bytes of a step that holds non-ASCII bytes.  v8 = the lane's dword and the one after it (a code point
// that starts in this lane is emitted by this lane, its continuation bytes may lie in the next one); inb = bit k set when
// byte k belongs to the document; p = text position of byte 0; e = end of the document.  Results: Wm / Nm = bit k set
// when a kept whitespace / non-whitespace code point STARTS at byte k; enc[k] / len[k] its lower-cased utf-8 (Nm only);
// bal += continuation bytes the lane's lead bytes own - continuation bytes the lane holds (0 over a valid document).
// Returns false when the document must go to the host (invalid utf-8 or a flagged code point).
__device__ __forceinline__ bool classify_unicode(uint64_t v8, uint32_t inb, uint64_t p, uint64_t e, uint32_t& Wm, uint32_t& Nm,
                                                 uint32_t (&enc)[4], uint32_t (&len)[4], int32_t& bal) {
    bool ok_all = true;
    Wm = Nm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        len[k] = 0;
        enc[k] = 0;
        if (!((inb >> k) & 1u)) continue;
        const uint32_t v = (uint32_t)(v8 >> (8 * k));
        const uint32_t b0 = v & 0xFFu;
        if (b0 < 0x80u) {
            if (is_punct(b0)) continue;
            if (is_space(b0)) {
                Wm |= 1u << k;
            } else {
                Nm |= 1u << k;
                enc[k] = (b0 - 65u <= 25u) ? b0 + 32u : b0;
                len[k] = 1;
            }
        } else if ((b0 & 0xC0u) == 0x80u) {
            bal -= 1;
        } else {
            const uint32_t L = b0 >= 0xF0u ? 4u : (b0 >= 0xE0u ? 3u : 2u);
            const uint32_t b1 = (v >> 8) & 0xFFu, b2 = (v >> 16) & 0xFFu, b3 = v >> 24;
            bool ok = (p + k + L <= e) && b0 >= 0xC2u && b0 <= 0xF4u && (b1 & 0xC0u) == 0x80u && (L < 3u || (b2 & 0xC0u) == 0x80u) &&
                      (L < 4u || (b3 & 0xC0u) == 0x80u);
            const uint32_t cp = L == 2u   ? ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu)
                                : L == 3u ? ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu)
                                          : ((b0 & 7u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
            ok = ok && (L == 2u || (L == 3u && cp >= 0x800u && cp - 0xD800u > 0x7FFu) || (L == 4u && cp - 0x10000u <= 0xFFFFFu));
            if (!ok) {
                ok_all = false;
                continue;
            }
            bal += (int32_t)L - 1;
            const uint32_t ent = uni_entry(cp);
            if (ent & MDD_UNI_FALLBACK) {
                ok_all = false;
            } else if (ent & MDD_UNI_SPACE) {
                Wm |= 1u << k;
            } else {
                Nm |= 1u << k;
                enc[k] = utf8_encode((cp + ent) & 0x1FFFFFu, len[k]);
            }
        }
    }
    return ok_all;
}

// Count and write passes, dword per lane (text 4-byte aligned): a lane owns 4 consecutive bytes, a wave step covers 256.
// The byte classes come from SWAR range tests on the whole dword (bytes are < 0x80 in every document that
// survives, so x + (128 - lo) and x + (127 - hi) never carry into the neighbour byte; bit 7 of a byte then
// says x >= lo resp. x > hi).  Inside the lane the "previous kept byte is whitespace" / "a word was seen"
// state runs over the 4 bytes; between lanes it comes from 3 ballots (lane has a kept byte, its last kept
// byte is whitespace, lane has a non-whitespace byte) instead of 12 for the same 256 bytes.  Per-lane counts
// are summed once per document.
__device__ __forceinline__ uint32_t swar_in_range(uint32_t x7, uint32_t lo, uint32_t hi) {  // bit 7 of each byte: lo <= byte <= hi
    return (x7 + (128u - lo) * 0x01010101u) & ~(x7 + (127u - hi) * 0x01010101u) & 0x80808080u;
}

// WRITE: the same walk, second time: the bytes of non-flagged documents go out.  A lane emits 0..8 bytes per step (each of
// its 4 bytes, and the single space before a word); its first output position = the document's offset + what earlier
// steps wrote + the counts of the lanes below it, which four ballots over the bits of the count deliver.
template <bool WRITE>
__global__ void __launch_bounds__(256) norm_en4_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                       const uint64_t* __restrict__ offs, uint64_t n, uint64_t* out_len,
                                                       uint8_t* non_ascii, const uint64_t* __restrict__ out_offs,
                                                       uint8_t* __restrict__ out_text, uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;  // lanes below this one
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        if (WRITE && non_ascii[d]) continue;
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t w0 = b >> 2, w1 = (e + 3) >> 2;  // dwords that hold the document (the last one lies inside the text's padding)
        uint64_t done = WRITE ? out_offs[d] : 0;      // WRITE: next output position of the wave (wave-uniform)
        uint32_t cnt = 0;                             // count pass: this lane's bytes, summed once per document
        bool carry_ws = false, carry_seen = false, bad = false;  // wave-uniform carries between steps
        int32_t bal = 0;  // continuation bytes owned by lead bytes - continuation bytes seen: 0 over valid utf-8
        for (uint64_t wi = w0; wi < w1 && !bad; wi += 128) {
            uint32_t ww[2];  // two independent loads in flight
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                ww[j] = (b < e && i < w1) ? tw[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                const uint64_t p = i << 2;  // position of the dword's byte 0
                // bytes of the dword that belong to the document, as bit 7 of each byte
                uint32_t in = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i < w1 && p + k >= b && p + k < e) in |= 0x80u << (8 * k);
                const uint32_t w = ww[j];
                if (__ballot((w & in) != 0u)) {  // a byte >= 0x80 in this step: the table-driven path (same carries, same counters)
                    const uint32_t nx = i < w1 ? tw[i + 1] : 0u;  // the dword after the lane's (inside the text's padding at the end)
                    const uint32_t inb = ((in >> 7) & 1u) | ((in >> 14) & 2u) | ((in >> 21) & 4u) | ((in >> 28) & 8u);
                    uint32_t Wm, Nm, enc[4], len[4];
                    const bool okl = classify_unicode((uint64_t)w | ((uint64_t)nx << 32), inb, p, e, Wm, Nm, enc, len, bal);
                    if (__ballot(!okl)) {  // (cannot happen in the write pass: flagged documents are skipped)
                        bad = true;
                        break;
                    }
                    const uint32_t Km = Wm | Nm;
                    const bool has_k = Km != 0u, has_n = Nm != 0u;
                    const bool last_ws = has_k && ((Wm >> (31u - (uint32_t)__builtin_clz(Km))) & 1u);
                    const uint64_t BK = __ballot(has_k), BW = __ballot(last_ws), BN = __ballot(has_n);
                    const uint64_t pk = BK & lt;
                    bool pw = pk ? ((BW >> (63 - __builtin_clzll(pk))) & 1) : carry_ws;
                    bool sn = carry_seen || (BN & lt);
                    uint32_t c = 0, spm = 0;  // bytes this lane emits; bit k: a space goes before the code point at byte k
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((Nm >> k) & 1u) {
                            if (pw && sn) {
                                spm |= 1u << k;
                                c++;
                            }
                            c += len[k];
                            sn = true;
                            pw = false;
                        } else if ((Wm >> k) & 1u) {
                            pw = true;
                        }
                    }
                    if (!WRITE) {
                        cnt += c;
                    } else {  // c <= 20: five count planes
                        const uint64_t c0 = __ballot(c & 1u), c1 = __ballot(c & 2u), c2 = __ballot(c & 4u), c3 = __ballot(c & 8u), c4 = __ballot(c & 16u);
                        uint64_t pos = done + __builtin_popcountll(c0 & lt) + 2 * __builtin_popcountll(c1 & lt) + 4 * __builtin_popcountll(c2 & lt) +
                                       8 * __builtin_popcountll(c3 & lt) + 16 * __builtin_popcountll(c4 & lt);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (!((Nm >> k) & 1u)) continue;
                            if ((spm >> k) & 1u) {
                                if (pos < out_cap) out_text[pos] = ' ';
                                pos++;
                            }
                            for (uint32_t q = 0; q < len[k]; q++, pos++)
                                if (pos < out_cap) out_text[pos] = (uint8_t)(enc[k] >> (8 * q));  // out_offsets are the caller's: stay inside out_text
                        }
                        done += __builtin_popcountll(c0) + 2 * __builtin_popcountll(c1) + 4 * __builtin_popcountll(c2) + 8 * __builtin_popcountll(c3) +
                                16 * __builtin_popcountll(c4);
                    }
                    if (BK) carry_ws = (BW >> (63 - __builtin_clzll(BK))) & 1;
                    carry_seen = carry_seen || BN;
                    continue;
                }
                const uint32_t x = w & 0x7F7F7F7Fu;
                const uint32_t punct = swar_in_range(x, 33, 47) | swar_in_range(x, 58, 64) | swar_in_range(x, 91, 96) |
                                       swar_in_range(x, 123, 126);
                const uint32_t space = swar_in_range(x, 9, 13) | swar_in_range(x, 28, 32);
                const uint32_t upper = swar_in_range(x, 65, 90);
                const uint32_t K = in & ~punct, W = K & space, N = K & ~W;
                // lane summary for the cross-lane state
                const bool has_k = K != 0u, has_n = N != 0u;
                const uint32_t last_k = K ? 31u - (uint32_t)__builtin_clz(K) : 0u;  // bit index of the last kept byte's flag
                const bool last_ws = has_k && ((W >> last_k) & 1u);
                const uint64_t BK = __ballot(has_k), BW = __ballot(last_ws), BN = __ballot(has_n);
                const uint64_t pk = BK & lt;
                bool pw = pk ? ((BW >> (63 - __builtin_clzll(pk))) & 1) : carry_ws;  // previous kept byte is whitespace
                bool sn = carry_seen || (BN & lt);                                      // a non-whitespace byte was seen
                if (!WRITE) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t f = 0x80u << (8 * k);
                        if (N & f) {
                            cnt += 1u + ((pw && sn) ? 1u : 0u);  // the byte, and the single space before it
                            sn = true;
                            pw = false;
                        } else if (W & f) {
                            pw = true;
                        }
                    }
                } else {
                    // what this lane emits, lower-cased (A-Z | 0x20), as up to 8 bytes in two dwords
                    const uint32_t lowered = x | (upper >> 2);
                    uint32_t lo = 0, hi = 0, c = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t f = 0x80u << (8 * k);
                        if (N & f) {
                            if (pw && sn) {
                                if (c < 4) lo |= 0x20u << (8 * c);
                                else hi |= 0x20u << (8 * (c - 4));
                                c++;
                            }
                            const uint32_t ch = (lowered >> (8 * k)) & 0xFFu;
                            if (c < 4) lo |= ch << (8 * c);
                            else hi |= ch << (8 * (c - 4));
                            c++;
                            sn = true;
                            pw = false;
                        } else if (W & f) {
                            pw = true;
                        }
                    }
                    const uint64_t c0 = __ballot(c & 1u), c1 = __ballot(c & 2u), c2 = __ballot(c & 4u), c3 = __ballot(c & 8u);
                    uint64_t pos = done + __builtin_popcountll(c0 & lt) + 2 * __builtin_popcountll(c1 & lt) + 4 * __builtin_popcountll(c2 & lt) +
                                   8 * __builtin_popcountll(c3 & lt);
                    store_low_bytes(out_text, pos, out_cap, lo, c < 4u ? c : 4u);  // out_offsets are the caller's: stay inside out_text
                    if (c > 4u) store_low_bytes(out_text, pos + 4, out_cap, hi, c - 4u);
                    done += __builtin_popcountll(c0) + 2 * __builtin_popcountll(c1) + 4 * __builtin_popcountll(c2) + 8 * __builtin_popcountll(c3);
                }
                if (BK) carry_ws = (BW >> (63 - __builtin_clzll(BK))) & 1;
                carry_seen = carry_seen || BN;
            }
        }
        if (!WRITE) {
            uint32_t tot = cnt;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                tot += (uint32_t)__shfl_xor((int)tot, o, 64);
                bal += __shfl_xor(bal, o, 64);
            }
            bad = bad || bal != 0;  // a continuation byte no lead byte owns
            if (lane == 0) {
                out_len[d] = bad ? 0 : tot;
                non_ascii[d] = bad;
            }
        }
    }
}

// Output document k = input document sel[k].  One wave per output document: up to 3 head bytes bring the
// destination to a dword boundary, the body moves aligned destination dwords assembled from two aligned source
// dwords (v_alignbyte; four independent loads in flight), up to 3 tail bytes finish.  The source dword after a
// document's last byte lies inside the MDD_TEXT_PAD bytes every text buffer carries.
__global__ void __launch_bounds__(256) gather_docs_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                          const uint64_t* __restrict__ offs, uint64_t n_src,
                                                          const uint64_t* __restrict__ sel, uint64_t n_sel,
                                                          const uint64_t* __restrict__ out_offs, uint8_t* __restrict__ out_text,
                                                          uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; k < n_sel; k += nw) {
        const uint64_t s = sel[k];
        if (s >= n_src) continue;
        uint64_t b, e;
        doc_range(offs, s, text_bytes, b, e);
        const uint64_t o = out_offs[k];
        if (o >= out_cap) continue;  // out_offsets are the caller's: stay inside out_text
        uint64_t len = e - b;
        len = len < out_cap - o ? len : out_cap - o;
        const uint64_t head = ((4 - (o & 3)) & 3) < len ? ((4 - (o & 3)) & 3) : len;
        if (lane < head) out_text[o + lane] = text[b + lane];
        const uint64_t nd = (len - head) >> 2, src = b + head;
        const uint32_t m = (uint32_t)(src & 3);
        const uint32_t* __restrict__ sw = reinterpret_cast<const uint32_t*>(text + (src - m));
        uint32_t* __restrict__ dw = reinterpret_cast<uint32_t*>(out_text + o + head);
        for (uint64_t j0 = 0; j0 < nd; j0 += 256) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t j = j0 + 64 * u + lane;
                lo[u] = j < nd ? sw[j] : 0u;
                hi[u] = (j < nd && m) ? sw[j + 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t j = j0 + 64 * u + lane;
                if (j < nd) dw[j] = __builtin_amdgcn_alignbyte(hi[u], lo[u], m);
            }
        }
        const uint64_t done = head + 4 * nd;
        if (lane < len - done) out_text[o + done + lane] = text[b + done + lane];
    }
}

unsigned norm_grid(const mdd_ctx* ctx, uint64_t n) {
    uint64_t blocks = (n + 3) / 4, cap = (uint64_t)ctx->cus * 64;
    return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace

extern "C" int mdd_normalize_en_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                      uint64_t* out_len, uint8_t* non_ascii) {
    REQUIRE(ctx, "mdd_normalize_en_count: NULL context");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(text && offsets && out_len && non_ascii, "mdd_normalize_en_count: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)  // the usual case (device allocations are aligned): dword per lane
        hipLaunchKernelGGL(norm_en4_kernel<false>, dim3(norm_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets, n_docs,
                           out_len, non_ascii, (const uint64_t*)nullptr, (uint8_t*)nullptr, (uint64_t)0);
    else
        hipLaunchKernelGGL(norm_en_kernel<false>, dim3(norm_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, out_len, non_ascii, (const uint64_t*)nullptr, (uint8_t*)nullptr, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_normalize_en_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                      const uint8_t* non_ascii, const uint64_t* out_offsets, uint8_t* out_text, uint64_t out_capacity) {
    REQUIRE(ctx, "mdd_normalize_en_write: NULL context");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(text && offsets && non_ascii && out_offsets && out_text, "mdd_normalize_en_write: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)
        hipLaunchKernelGGL(norm_en4_kernel<true>, dim3(norm_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets, n_docs,
                           (uint64_t*)nullptr, const_cast<uint8_t*>(non_ascii), out_offsets, out_text, out_capacity);
    else
        hipLaunchKernelGGL(norm_en_kernel<true>, dim3(norm_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets, n_docs,
                           (uint64_t*)nullptr, const_cast<uint8_t*>(non_ascii), out_offsets, out_text, out_capacity);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_gather_docs(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                               const uint64_t* sel, uint64_t n_sel, const uint64_t* out_offsets, uint8_t* out_text,
                               uint64_t out_capacity) {
    REQUIRE(ctx, "mdd_gather_docs: NULL context");
    if (n_sel == 0) return MDD_OK;
    REQUIRE(offsets && sel && out_offsets && (text || text_bytes == 0) && (out_text || out_capacity == 0),
            "mdd_gather_docs: NULL buffer");
    REQUIRE(((reinterpret_cast<uintptr_t>(text) | reinterpret_cast<uintptr_t>(out_text)) & 3u) == 0,
            "mdd_gather_docs: text and out_text must be 4-byte aligned");
    if (n_docs == 0 || out_capacity == 0) return MDD_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gather_docs_kernel, dim3(norm_grid(ctx, n_sel)), dim3(256), 0, ctx->stream, text, text_bytes, offsets, n_docs,
                       sel, n_sel, out_offsets, out_text, out_capacity);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}
