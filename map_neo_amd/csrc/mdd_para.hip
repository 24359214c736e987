// Paragraph dedup on the device.  Restates
// Matrix/english_data_clean_pipeline/dedup/Paragraph_Deduplication/paragraph.py
//   :16-17  split(content, "\n") + posexplode            -> para_count / para_units: unit ranges per document
//   :18-23  sha2(paragraph) + row_number over (id, pos)   -> mdd_exact_group on those ranges (first_of)
//   :26-27  collect_list ordered by pos + concat_ws("\n") -> para_compact<count/write>
//   :29     length >= min_len                             -> caller, from kept_cps
// One wave per document, 64 bytes per step; a byte's paragraph index is unit_base[d] + the
// newlines before it, so ballots + popcounts place every unit boundary and every surviving byte
// without scans or LDS.  HBM-bound.  Round 3: the text is read three times plus once by the grouping (count, units, hash,
// write) instead of five: the per-document sums come from the UNITS (mdd_para_doc_sums: lengths from the unit ranges, code
// points counted by the grouping's hash pass, ~230 MB instead of the text), and a document all of whose paragraphs survive
// -- nearly every document -- leaves the write pass as a straight copy.
#include "mdd_common.h"

namespace {
using namespace mddi;

unsigned wave_grid(const mdd_ctx* ctx, uint64_t n) {
    uint64_t blocks = (n + 3) / 4, cap = (uint64_t)ctx->cus * 64;
    return (unsigned)(blocks < cap ? blocks : cap);
}

// UNITS = false: units_per_doc[d] = newlines + 1.  UNITS = true: ubeg / uend of every paragraph.
template <bool UNITS>
__global__ void __launch_bounds__(256) para_scan_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                        const uint64_t* __restrict__ offs, uint64_t n, uint64_t* units_per_doc,
                                                        const uint64_t* __restrict__ unit_base, uint64_t n_units, uint64_t* ubeg,
                                                        uint64_t* uend) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        uint64_t u = UNITS ? unit_base[d] : 0;  // paragraph that holds the first byte of the step
        uint64_t nl = 0;
        if (UNITS && lane == 0 && u < n_units) ubeg[u] = b;
        for (uint64_t p = b; p < e; p += 64) {
            const uint64_t i = p + lane;
            const bool is_nl = i < e && text[i] == 10u;
            const uint64_t N = __ballot(is_nl);
            if (UNITS && is_nl) {
                uint64_t mine = u + __builtin_popcountll(N & lt);  // the paragraph this newline ends
                if (mine + 1 < n_units) {                          // unit_base is the caller's: stay inside ubeg / uend
                    uend[mine] = i;
                    ubeg[mine + 1] = i + 1;
                }
            }
            uint64_t c = __builtin_popcountll(N);
            u += c;
            nl += c;
        }
        if (UNITS) {
            if (lane == 0 && u < n_units) uend[u] = e;
        } else if (lane == 0) {
            units_per_doc[d] = nl + 1;
        }
    }
}

// The same scan with a dword per lane (text 4-byte aligned): a wave step covers 256 bytes.  Newline bytes
// are found with an exact SWAR zero-byte test on w ^ 0x0A0A0A0A; a lane holds 0..4 of them, so the rank of
// a lane's first newline is a prefix sum over 3-bit counts = 3 ballots (UNITS) or nothing at all (count:
// per-lane sums, added up once per document).
template <bool UNITS>
__global__ void __launch_bounds__(256) para_scan4_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                         const uint64_t* __restrict__ offs, uint64_t n, uint64_t* units_per_doc,
                                                         const uint64_t* __restrict__ unit_base, uint64_t n_units, uint64_t* ubeg,
                                                         uint64_t* uend) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t w0 = b >> 2, w1 = b < e ? (e + 3) >> 2 : w0;  // dwords that hold the document
        uint64_t u = UNITS ? unit_base[d] : 0;  // paragraph that holds the first byte of the step
        uint32_t nl = 0;
        if (UNITS && lane == 0 && u < n_units) ubeg[u] = b;
        for (uint64_t wi = w0; wi < w1; wi += 256) {
            uint32_t ww[4];  // four independent loads in flight
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = wi + 64 * j + lane;
                ww[j] = i < w1 ? tw[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = wi + 64 * j + lane;
                const uint64_t p = i << 2;
                const uint32_t y = ww[j] ^ 0x0A0A0A0Au;
                uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);  // bit 7 of every byte that is '\n'
                if (i >= w1) z = 0;
                if (p < b || p + 4 > e) {  // first / last dword: drop bytes outside [b, e)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (p + k < b || p + k >= e) z &= ~(0x80u << (8 * k));
                }
                const uint32_t c = (uint32_t)__builtin_popcount(z);
                if (UNITS) {
                    const uint64_t B0 = __ballot(c & 1u), B1 = __ballot(c & 2u), B2 = __ballot(c & 4u);
                    uint64_t mine = u + __builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) +
                                    4 * __builtin_popcountll(B2 & lt);  // the paragraph this lane's first newline ends
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (z & (0x80u << (8 * k))) {
                            if (mine + 1 < n_units) {  // unit_base is the caller's: stay inside ubeg / uend
                                uend[mine] = p + k;
                                ubeg[mine + 1] = p + k + 1;
                            }
                            mine++;
                        }
                    }
                    u += __builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2);
                } else {
                    nl += c;
                }
            }
        }
        if (UNITS) {
            if (lane == 0 && u < n_units) uend[u] = e;
        } else {
            uint32_t tot = nl;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o, 64);
            if (lane == 0) units_per_doc[d] = (uint64_t)tot + 1;
        }
    }
}

template <bool WRITE>
__global__ void __launch_bounds__(256) para_compact_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                           const uint64_t* __restrict__ offs, uint64_t n,
                                                           const uint64_t* __restrict__ unit_base, uint64_t n_units,
                                                           const uint32_t* __restrict__ first_of, uint64_t* kept_bytes,
                                                           uint64_t* kept_cps, uint8_t* has_unit, const uint8_t* __restrict__ keep_doc,
                                                           const uint64_t* __restrict__ out_offs, uint8_t* __restrict__ out_text,
                                                           uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t incl = lt | (1ULL << lane);
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    auto kept_unit = [&](uint64_t u) -> bool { return u < n_units && first_of[u] == (uint32_t)u; };
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        if (WRITE && !keep_doc[d]) continue;
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t o = WRITE ? out_offs[d] : 0;
        uint64_t u = unit_base[d];
        uint64_t cnt_b = 0, cnt_c = 0;
        bool any = false;  // a paragraph that ended before this step survives
        for (uint64_t p = b; p < e; p += 64) {
            const uint64_t i = p + lane;
            const bool in = i < e;
            const uint32_t c = in ? text[i] : 0u;
            const bool is_nl = in && c == 10u;
            const uint64_t N = __ballot(is_nl);
            const uint64_t mine = u + __builtin_popcountll(N & lt);  // paragraph of this byte (a newline belongs to the one it ends)
            const bool ku = in && kept_unit(mine);
            const uint64_t Kn = __ballot(is_nl && ku);               // newlines that end a surviving paragraph
            bool keep;
            if (is_nl)  // the "\n" in front of a surviving paragraph that is not its document's first survivor
                keep = kept_unit(mine + 1) && (any || (Kn & incl) != 0);
            else
                keep = ku;
            const uint64_t Kb = __ballot(keep);
            if (WRITE && keep) {
                uint64_t pos = o + cnt_b + __builtin_popcountll(Kb & lt);
                if (pos < out_cap) out_text[pos] = (uint8_t)c;
            }
            cnt_b += __builtin_popcountll(Kb);
            cnt_c += __builtin_popcountll(__ballot(keep && (c & 0xC0u) != 0x80u));
            any = any || Kn != 0;
            u += __builtin_popcountll(N);
        }
        if (!WRITE && lane == 0) {
            kept_bytes[d] = cnt_b;
            kept_cps[d] = cnt_c;
            has_unit[d] = any || kept_unit(u);  // u = the document's last paragraph
        }
    }
}

// Count pass of the compaction with a dword per lane (text 4-byte aligned).  The per-byte rules of
// para_compact_kernel add up to: kept bytes = bytes of the surviving paragraphs + one "\n" for every survivor
// but the document's first ("\n".join), code points likewise, has_unit = a paragraph survives.  So the pass
// only needs, per lane, the paragraph its first byte lies in (3 ballots over the lanes' newline counts) and
// that paragraph's verdict; lanes that hold a newline walk their 4 bytes.  Sums are reduced once per document.
__global__ void __launch_bounds__(256) para_compact_count4_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                                  const uint64_t* __restrict__ offs, uint64_t n,
                                                                  const uint64_t* __restrict__ unit_base, uint64_t n_units,
                                                                  const uint32_t* __restrict__ first_of, uint64_t* kept_bytes,
                                                                  uint64_t* kept_cps, uint8_t* has_unit) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    auto kept_unit = [&](uint64_t u) -> bool { return u < n_units && first_of[u] == (uint32_t)u; };
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t w0 = b >> 2, w1 = b < e ? (e + 3) >> 2 : w0;
        uint64_t u = unit_base[d];
        uint32_t cnt_b = 0, cnt_c = 0, ns = (lane == 0 && kept_unit(u)) ? 1u : 0u;  // per-lane sums; ns = surviving paragraphs
        for (uint64_t wi = w0; wi < w1; wi += 256) {
            uint32_t ww[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = wi + 64 * j + lane;
                ww[j] = i < w1 ? tw[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = wi + 64 * j + lane;
                const uint64_t p = i << 2;
                const uint32_t w = ww[j];
                uint32_t in = i < w1 ? 0x80808080u : 0u;  // bytes of the dword inside [b, e), as bit 7 of each byte
                if (p < b || p + 4 > e) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (p + k < b || p + k >= e) in &= ~(0x80u << (8 * k));
                }
                const uint32_t y = w ^ 0x0A0A0A0Au;
                const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu) & in;  // '\n' bytes
                const uint32_t cb = w & ~(w << 1) & 0x80808080u;                                // utf-8 continuation bytes (10xxxxxx)
                const uint32_t c = (uint32_t)__builtin_popcount(z);
                const uint64_t B0 = __ballot(c & 1u), B1 = __ballot(c & 2u), B2 = __ballot(c & 4u);
                uint64_t cur = u + __builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) +
                               4 * __builtin_popcountll(B2 & lt);  // paragraph of the lane's first byte
                bool kc = in != 0u && kept_unit(cur);
                if (c == 0u) {
                    if (kc) {
                        cnt_b += (uint32_t)__builtin_popcount(in);
                        cnt_c += (uint32_t)__builtin_popcount(in & ~cb);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t f = 0x80u << (8 * k);
                        if (z & f) {  // a newline ends `cur`; the next paragraph starts here
                            cur++;
                            kc = kept_unit(cur);
                            ns += kc ? 1u : 0u;
                        } else if ((in & f) && kc) {
                            cnt_b++;
                            cnt_c += (cb & f) ? 0u : 1u;
                        }
                    }
                }
                u += __builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            cnt_b += (uint32_t)__shfl_xor((int)cnt_b, o, 64);
            cnt_c += (uint32_t)__shfl_xor((int)cnt_c, o, 64);
            ns += (uint32_t)__shfl_xor((int)ns, o, 64);
        }
        if (lane == 0) {
            const uint64_t joins = ns ? ns - 1u : 0u;
            kept_bytes[d] = (uint64_t)cnt_b + joins;
            kept_cps[d] = (uint64_t)cnt_c + joins;
            has_unit[d] = ns != 0u;
        }
    }
}

// Write pass with a dword per lane.  Per byte the rules are those of para_compact_kernel: a byte that is not a newline goes out
// with its paragraph; a newline goes out when the paragraph AFTER it survives and a surviving paragraph has ended at or
// before it ("\n".join of the survivors).  A lane knows the paragraph of its first byte from 3 ballots over the lanes' newline
// counts, looks up the verdicts of the (at most 5) paragraphs its 4 bytes touch, and learns from one more ballot whether a
// survivor ended in a lane below it; its kept bytes (0..4) are placed by 3 ballots over the counts and leave as one
// unaligned dword store when all four are kept.
__global__ void __launch_bounds__(256) para_compact_write4_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                                  const uint64_t* __restrict__ offs, uint64_t n,
                                                                  const uint64_t* __restrict__ unit_base, uint64_t n_units,
                                                                  const uint32_t* __restrict__ first_of, const uint8_t* __restrict__ keep_doc,
                                                                  const uint64_t* __restrict__ out_offs, uint8_t* __restrict__ out_text,
                                                                  uint64_t out_cap, const uint8_t* __restrict__ whole_doc) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    auto kept_unit = [&](uint64_t u) -> bool { return u < n_units && first_of[u] == (uint32_t)u; };
    auto rl64 = [](uint64_t x, uint32_t j) -> uint64_t {  // lane j's value (j wave-uniform)
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)j) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)j);
    };
    // 64 documents at a time: every lane fetches one document's range, verdicts, output offset and first paragraph, so the
    // five dependent loads a document needs cost one round trip per 64 documents instead of one per document (a document
    // that is copied whole is two steps of work: the kernel was waiting on those loads)
    for (uint64_t D = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; D < n; D += nw * 64) {
        const uint64_t dl = D + lane;
        uint64_t b_l = 0, e_l = 0, o_l = 0, u_l = 0;
        bool keep_l = false, whole_l = false;
        if (dl < n) {
            keep_l = keep_doc[dl] != 0;
            whole_l = whole_doc && whole_doc[dl] != 0;
            doc_range(offs, dl, text_bytes, b_l, e_l);
            o_l = out_offs[dl];
            u_l = unit_base[dl];
        }
        const uint64_t wmask = __ballot(whole_l);
        for (uint64_t km = __ballot(keep_l); km; km &= km - 1) {
        const uint32_t j = (uint32_t)__builtin_ctzll(km);
        const uint64_t b = rl64(b_l, j), e = rl64(e_l, j);
        if ((wmask >> j) & 1ull) {
            // every paragraph survives ("\n".join of all of them = the document): a copy, four unaligned dwords in flight per lane
            const uint64_t o = rl64(o_l, j), len = e - b;
            for (uint64_t i = 4ull * lane; i < len; i += 1024) {
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint64_t q = i + 256ull * j;
                    v[j] = q < len ? *reinterpret_cast<const mdd_u32_a1*>(text + b + q) : 0u;  // (over-read < 4 bytes: inside MDD_TEXT_PAD)
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint64_t q = i + 256ull * j;
                    if (q < len) store_low_bytes(out_text, o + q, out_cap, v[j], len - q < 4 ? (uint32_t)(len - q) : 4u);
                }
            }
            continue;
        }
        const uint64_t w0 = b >> 2, w1 = b < e ? (e + 3) >> 2 : w0;
        uint64_t u = rl64(u_l, j);
        uint64_t done = rl64(o_l, j);  // next output position of the wave
        bool any = false;              // a surviving paragraph ended in an earlier step
        for (uint64_t wi = w0; wi < w1; wi += 128) {
            uint32_t ww[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                ww[j] = i < w1 ? tw[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                const uint64_t p = i << 2;
                const uint32_t w = ww[j];
                uint32_t in = i < w1 ? 0x80808080u : 0u;  // bytes of the dword inside [b, e), as bit 7 of each byte
                if (p < b || p + 4 > e) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (p + k < b || p + k >= e) in &= ~(0x80u << (8 * k));
                }
                const uint32_t y = w ^ 0x0A0A0A0Au;
                const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu) & in;  // '\n' bytes
                const uint32_t c = (uint32_t)__builtin_popcount(z);
                const uint64_t B0 = __ballot(c & 1u), B1 = __ballot(c & 2u), B2 = __ballot(c & 4u);
                const uint64_t cur = u + __builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) +
                                     4 * __builtin_popcountll(B2 & lt);  // paragraph of the lane's first byte
                // verdicts of the paragraphs cur .. cur + c (bit q = paragraph cur + q survives)
                uint32_t ku = (in != 0u && kept_unit(cur)) ? 1u : 0u;
                if (c) {
#pragma unroll
                    for (uint32_t q = 1; q <= 4; q++)
                        if (q <= c && kept_unit(cur + q)) ku |= 1u << q;
                }
                const bool ends_surv = (ku & ((1u << c) - 1u)) != 0u;  // one of the lane's newlines ends a survivor
                const uint64_t BA = __ballot(ends_surv);
                bool a = any || (BA & lt) != 0;
                uint32_t keep = 0;  // bit 7 of each kept byte
                if (c == 0u) {
                    if (ku & 1u) keep = in;
                } else {
                    uint32_t q = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t f = 0x80u << (8 * k);
                        if (z & f) {
                            a = a || ((ku >> q) & 1u);
                            q++;
                            if (((ku >> q) & 1u) && a) keep |= f;
                        } else if ((in & f) && ((ku >> q) & 1u)) {
                            keep |= f;
                        }
                    }
                }
                const uint32_t kc = (uint32_t)__builtin_popcount(keep);
                const uint64_t K0 = __ballot(kc & 1u), K1 = __ballot(kc & 2u), K2 = __ballot(kc & 4u);
                if (kc) {
                    const uint64_t pos = done + __builtin_popcountll(K0 & lt) + 2 * __builtin_popcountll(K1 & lt) + 4 * __builtin_popcountll(K2 & lt);
                    uint32_t v = w;
                    if (kc != 4u) {  // squeeze the kept bytes together
                        v = 0;
                        uint32_t at = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (keep & (0x80u << (8 * k))) {
                                v |= ((w >> (8 * k)) & 0xFFu) << (8 * at);
                                at++;
                            }
                    }
                    store_low_bytes(out_text, pos, out_cap, v, kc);  // out_offsets are the caller's: stay inside out_text
                }
                done += __builtin_popcountll(K0) + 2 * __builtin_popcountll(K1) + 4 * __builtin_popcountll(K2);
                any = any || BA != 0;
                u += __builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2);
            }
        }
        }
    }
}

// per-document sums of the compaction from the units alone (no text): what para_compact_count4_kernel computes, for
// callers that hold the units' ranges and code-point counts (mdd_exact_group_cps).  One thread per document.
__global__ void __launch_bounds__(256) para_doc_sums_kernel(const uint64_t* __restrict__ offs, uint64_t n, uint64_t text_bytes,
                                                            const uint64_t* __restrict__ unit_base, uint64_t n_units,
                                                            const uint64_t* __restrict__ ubeg, const uint64_t* __restrict__ uend,
                                                            const uint32_t* __restrict__ ucps, const uint32_t* __restrict__ first_of,
                                                            uint64_t* kept_bytes, uint64_t* kept_cps, uint8_t* has_unit, uint8_t* whole_doc) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    uint64_t b, e;
    doc_range(offs, d, text_bytes, b, e);
    uint64_t u0 = unit_base[d], u1 = d + 1 < n ? unit_base[d + 1] : n_units;
    u1 = u1 < n_units ? u1 : n_units;
    u0 = u0 < u1 ? u0 : u1;
    uint64_t ns = 0, kb = 0, kc = 0;
    for (uint64_t u = u0; u < u1; u++)
        if (first_of[u] == (uint32_t)u) {
            uint64_t ue = uend[u], ub = ubeg[u];
            ue = ue < text_bytes ? ue : text_bytes;
            ub = ub < ue ? ub : ue;
            ns++;
            kb += ue - ub;
            kc += ucps[u];
        }
    const uint64_t joins = ns ? ns - 1 : 0;
    kept_bytes[d] = kb + joins;
    kept_cps[d] = kc + joins;
    has_unit[d] = ns != 0;
    whole_doc[d] = ns != 0 && ns == u1 - u0 && kb + joins == e - b;  // nothing dropped: the output is the document itself
}

}  // namespace

extern "C" int mdd_para_doc_sums(mdd_ctx* ctx, const uint64_t* offsets, uint64_t n_docs, uint64_t text_bytes, const uint64_t* unit_base,
                                 uint64_t n_units, const uint64_t* ubeg, const uint64_t* uend, const uint32_t* unit_cps,
                                 const uint32_t* first_of, uint64_t* kept_bytes, uint64_t* kept_cps, uint8_t* has_unit,
                                 uint8_t* whole_doc) {
    REQUIRE(ctx, "mdd_para_doc_sums: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && unit_base && kept_bytes && kept_cps && has_unit && whole_doc && ((ubeg && uend && unit_cps && first_of) || n_units == 0),
            "mdd_para_doc_sums: NULL buffer");
    REQUIRE(n_docs <= MAX_ELEMS_PER_LAUNCH, "mdd_para_doc_sums: n_docs=%llu exceeds 2^32-256", (unsigned long long)n_docs);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(para_doc_sums_kernel, dim3(grid_for(n_docs, 256)), dim3(256), 0, ctx->stream, offsets, n_docs, text_bytes, unit_base,
                       n_units, ubeg, uend, unit_cps, first_of, kept_bytes, kept_cps, has_unit, whole_doc);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_para_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                              uint64_t* units_per_doc) {
    REQUIRE(ctx, "mdd_para_count: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && units_per_doc && (text || text_bytes == 0), "mdd_para_count: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)  // the usual case (device allocations are aligned): dword per lane
        hipLaunchKernelGGL(para_scan4_kernel<false>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, units_per_doc, (const uint64_t*)nullptr, (uint64_t)0, (uint64_t*)nullptr, (uint64_t*)nullptr);
    else
        hipLaunchKernelGGL(para_scan_kernel<false>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, units_per_doc, (const uint64_t*)nullptr, (uint64_t)0, (uint64_t*)nullptr, (uint64_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_para_units(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                              const uint64_t* unit_base, uint64_t n_units, uint64_t* ubeg, uint64_t* uend) {
    REQUIRE(ctx, "mdd_para_units: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && unit_base && ubeg && uend && (text || text_bytes == 0), "mdd_para_units: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)
        hipLaunchKernelGGL(para_scan4_kernel<true>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, (uint64_t*)nullptr, unit_base, n_units, ubeg, uend);
    else
        hipLaunchKernelGGL(para_scan_kernel<true>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, (uint64_t*)nullptr, unit_base, n_units, ubeg, uend);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_para_compact_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets,
                                      uint64_t n_docs, const uint64_t* unit_base, uint64_t n_units, const uint32_t* first_of,
                                      uint64_t* kept_bytes, uint64_t* kept_cps, uint8_t* has_unit) {
    REQUIRE(ctx, "mdd_para_compact_count: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && unit_base && first_of && kept_bytes && kept_cps && has_unit && (text || text_bytes == 0),
            "mdd_para_compact_count: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)  // the usual case (device allocations are aligned): dword per lane
        hipLaunchKernelGGL(para_compact_count4_kernel, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, unit_base, n_units, first_of, kept_bytes, kept_cps, has_unit);
    else
        hipLaunchKernelGGL(para_compact_kernel<false>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes,
                           offsets, n_docs, unit_base, n_units, first_of, kept_bytes, kept_cps, has_unit, (const uint8_t*)nullptr,
                           (const uint64_t*)nullptr, (uint8_t*)nullptr, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_para_compact_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets,
                                      uint64_t n_docs, const uint64_t* unit_base, uint64_t n_units, const uint32_t* first_of,
                                      const uint8_t* keep_doc, const uint64_t* out_offsets, uint8_t* out_text,
                                      uint64_t out_capacity, const uint8_t* whole_doc) {
    REQUIRE(ctx, "mdd_para_compact_write: ctx is NULL");
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && unit_base && first_of && keep_doc && out_offsets && ((text && out_text) || text_bytes == 0),
            "mdd_para_compact_write: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if ((reinterpret_cast<uintptr_t>(text) & 3u) == 0)  // the usual case (device allocations are aligned): dword per lane
        hipLaunchKernelGGL(para_compact_write4_kernel, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, unit_base, n_units, first_of, keep_doc, out_offsets, out_text, out_capacity, whole_doc);
    else
        hipLaunchKernelGGL(para_compact_kernel<true>, dim3(wave_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, unit_base, n_units, first_of, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint8_t*)nullptr, keep_doc,
                           out_offsets, out_text, out_capacity);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}
