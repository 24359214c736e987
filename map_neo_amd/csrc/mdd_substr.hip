// Exact substring dedup on the device.  Restates
// Matrix/english_data_clean_pipeline/dedup/Exact_Substring_Deduplication/exact_substr_dedup.py
//   :34-47  every window text[o : o+W] of W = 50 code points for o in range(len(text) - W) is hashed
//           (SHA-256) and grouped; all occurrences but the first go to the remove list
//   :55-70  a document loses every code point covered by one of its listed windows
//   :90     documents left with fewer than min_length code points are dropped
// without materialising the windows: the shingle walker of the MinHash kernel (mdd_shingle.h,
// width = W) hands every wave 64 windows per round.
//   win_kernel         content key of each window -> open-addressing table of 8-byte entries
//                      (24-bit tag | 40-bit byte position of the earliest window with these bytes).
//                      A tag match is confirmed by comparing the bytes, so the table is exact:
//                      no collision list, nothing for the host to resolve.  ONE pass: a window that
//                      finds an earlier position in its slot marks itself (dup[first byte] = 1), a
//                      window that replaces a later position (atomicMin) marks the one it displaced,
//                      so when the kernel ends every occurrence but the earliest is marked.  One
//                      random atomic per window -- the device's random-atomic rate (~17 G/s on
//                      tables >> 256 MB, tools/atomic_ubench.hip) bounds the stage.
//   compact_kernel<false/true>  one wave per document, 64 bytes per step: a code point is gone
//                      when one of the W code points ending at it starts a marked window; ballots
//                      + popcounts give every byte its output position (count pass, caller's
//                      prefix sum over documents, write pass).
// Equal SHA-256 <=> equal bytes, so grouping on the bytes is the reference's grouping; "first" is
// the smallest (document, offset) = smallest byte position.
#include "mdd_shingle.h"

namespace {

constexpr unsigned long long EMPTY = ~0ULL;
constexpr uint64_t POS_MASK = (1ULL << 40) - 1;

struct WinArgs {
    const uint8_t* text;
    uint64_t text_bytes;
    const uint64_t* offsets;
    uint64_t n_docs;
    uint32_t window;
    uint32_t seg_bytes;
    const uint64_t* items;
    const uint64_t* n_items;
    uint64_t max_items;
    unsigned long long* work_counter;
    unsigned long long* table;
    uint64_t tab_size;
    uint8_t* dup;
    unsigned long long* overflow;  // windows that found neither their content nor a free slot (table too small)
    uint32_t part, n_parts;  // this launch handles the windows whose key falls into partition `part` of `n_parts`
    uint32_t batch_shift;    // a claim on the work counter takes (items left) >> batch_shift items, clamped to [1, 8]
};

// 4 bytes at byte address p of the text (any alignment); reads at most 7 bytes past p
__device__ __forceinline__ uint32_t ld4(const uint32_t* tw, uint64_t p) {
    uint64_t i = p >> 2;
    return __builtin_amdgcn_alignbyte(tw[i + 1], tw[i], (uint32_t)p & 3u);
}

// 64-bit content key of the L bytes at byte offset s from tv.docw
__device__ __forceinline__ uint64_t window_key(const TextView& tv, uint32_t s, uint32_t L) {
    uint32_t base = s >> 2, sh = s & 3u;
    uint32_t h1 = 0x9E3779B9u ^ L, h2 = 0x85EBCA6Bu;
    uint32_t lo = tv.docw[base];
    uint32_t nd = (L + 3u) >> 2;
    for (uint32_t k = 0; k < nd; k++) {
        uint32_t hi = tv.docw[base + k + 1];
        uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
        uint32_t rem = L - 4u * k;
        if (rem < 4u) w &= 0xFFFFFFFFu >> (8u * (4u - rem));
        h1 = (h1 ^ w) * 0xCC9E2D51u;
        h1 = __builtin_amdgcn_alignbit(h1, h1, 19);
        h2 = (h2 + w) * 0x1B873593u;
        h2 = __builtin_amdgcn_alignbit(h2, h2, 17) ^ h1;
        lo = hi;
    }
    return mix64(((uint64_t)h1 << 32) | h2);
}

// the L bytes at text positions p and q are identical (q + L must not leave the text: checked by the caller)
__device__ __forceinline__ bool same_bytes(const uint32_t* tw, uint64_t p, uint64_t q, uint32_t L) {
    uint32_t nd = (L + 3u) >> 2;
    for (uint32_t k = 0; k < nd; k++) {
        uint32_t x = ld4(tw, p + 4u * k) ^ ld4(tw, q + 4u * k);
        uint32_t rem = L - 4u * k;
        if (rem < 4u) x &= 0xFFFFFFFFu >> (8u * (4u - rem));
        if (x) return false;
    }
    return true;
}

__global__ void __launch_bounds__(64) win_kernel(WinArgs A) {
    __shared__ WaveLds lds;
    const uint32_t lane = lane_id();
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(A.text);
    const uint64_t extra = *A.n_items;  // final: seg_plan_kernel ran before this kernel on the stream
    const uint64_t total = A.n_docs + (extra < A.max_items ? extra : A.max_items);
    // persistent waves claiming documents (then queued segments of long documents) from one counter, in batches (guided
    // self-scheduling as in sig_kernel, mdd_hip.hip: a same-address atomic costs ~11 ns and the wave waits for it);
    // every wave leaves once the counter passes `total`
    auto fetch = [&](uint32_t n) -> uint64_t {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(A.work_counter, (unsigned long long)n);
        uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    uint64_t item = 0, batch_end = 0;
    for (;; item++) {
        if (item >= batch_end) {  // `item` = end of this wave's previous batch: a lower bound of the counter
            const uint64_t want = (total > item ? total - item : 0) >> A.batch_shift;
            const uint32_t n = want < 1u ? 1u : (want > 8u ? 8u : (uint32_t)want);
            item = fetch(n);
            if (item >= total) break;
            batch_end = item + n < total ? item + n : total;
        }
        uint64_t d = item;
        uint32_t seg = 0;
        if (d >= A.n_docs) {
            uint64_t it = A.items[d - A.n_docs];
            d = it >> 24;
            seg = (uint32_t)it & 0xFFFFFFu;
        }
        uint64_t beg, end;
        doc_range(A.offsets, d, A.text_bytes, beg, end);
        const uint32_t len = (uint32_t)(end - beg);
        const bool split = len > A.seg_bytes;
        const uint32_t seg_beg = seg * A.seg_bytes;
        const uint32_t seg_end = (uint64_t)seg_beg + A.seg_bytes < len ? seg_beg + A.seg_bytes : len;
        TextView tv;
        tv.docw = tw + (beg >> 2);
        const uint32_t boff = (uint32_t)beg & 3u;
        const bool ascii = segment_is_ascii(tv, boff, len, seg_beg, seg_end, A.window, lane);
        for_each_round(lds, A.text, beg, len, tv, boff, MDD_UNIT_CHAR, A.window, ascii, split, seg_beg, seg_end, lane,
                       [&](uint32_t s, uint32_t L, uint32_t nvalid, uint32_t) {
                           const uint32_t sb = s - boff;  // doc-relative first byte
                           // SUB:38 range(len(text) - duplicate_len): the window that ends with the document is never hashed
                           if (lane >= nvalid || sb + L >= len) return;
                           const uint64_t pos = beg + sb;
                           const uint64_t key = window_key(tv, s, L);
                           if (A.n_parts > 1 && (uint32_t)((key >> 24) % A.n_parts) != A.part) return;
                           const unsigned long long tag = (key & 0xFFFFFFull) << 40;
                           const unsigned long long val = tag | pos;
                           uint64_t slot = __umul64hi(key, A.tab_size);
                           // the table is at most half full (caller's contract), so chains are short; the cap only keeps a
                           // mis-sized table from spinning: a window that runs out of probes is COUNTED and
                           // mdd_substr_check turns the count into MDD_ERR_INVALID (never a silent "unique")
                           const uint64_t max_probe = A.tab_size < 65536 ? A.tab_size : 65536;
                           for (uint64_t probe = 0; probe < max_probe; probe++) {
                               unsigned long long cur = atomicCAS(&A.table[slot], EMPTY, val);
                               if (cur == EMPTY) return;  // first occurrence so far; a later, earlier one will mark this window
                               uint64_t q = cur & POS_MASK;
                               if ((cur & ~POS_MASK) == tag && q + L <= A.text_bytes && same_bytes(tw, pos, q, L)) {
                                   // the slot's content never changes, only its position shrinks.  Whoever loses the
                                   // comparison is a repeat: this window if the entry is earlier, else the displaced one
                                   if (cur > val) cur = atomicMin(&A.table[slot], val);
                                   A.dup[cur > val ? (cur & POS_MASK) : pos] = 1;
                                   return;
                               }
                               slot = slot + 1 == A.tab_size ? 0 : slot + 1;
                           }
                           atomicAdd(A.overflow, 1ull);
                       });
    }
}

// One wave per document.  since = code points after the most recent marked window start (saturating;
// "none yet" = huge); a code point is gone while since < W.  S / D / G: ballots of code-point starts,
// marked starts and gone starts of the 64-byte step.
template <bool WRITE>
__global__ void __launch_bounds__(256) compact_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                      const uint64_t* __restrict__ offs, uint64_t n, const uint8_t* __restrict__ dup,
                                                      uint32_t W, uint64_t* kept_bytes, uint64_t* kept_cps,
                                                      const uint8_t* __restrict__ keep_doc, const uint64_t* __restrict__ out_offs,
                                                      uint8_t* __restrict__ out_text, uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t incl = lt | (1ULL << lane);
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        if (WRITE && !keep_doc[d]) continue;
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t o = WRITE ? out_offs[d] : 0;
        uint64_t cnt_b = 0, cnt_c = 0;
        uint32_t since = 0x7FFFFFFFu;
        bool carry_gone = false;
        for (uint64_t p = b; p < e; p += 64) {
            const uint64_t i = p + lane;
            const bool in = i < e;
            const uint32_t c = in ? text[i] : 0x80u;
            const bool start = in && (c & 0xC0u) != 0x80u;
            const bool mark = start && dup[i] != 0;
            const uint64_t S = __ballot(start), D = __ballot(mark);
            const uint64_t Dq = D & incl;
            uint32_t since_l;
            if (Dq) {
                uint32_t q = 63u - (uint32_t)__builtin_clzll(Dq);
                since_l = (uint32_t)__builtin_popcountll(S & incl & ~(~0ULL >> (63u - q)));  // starts in (q, lane]
            } else {
                uint32_t add = (uint32_t)__builtin_popcountll(S & incl);
                since_l = since > 0x7FFFFFFFu - add ? 0x7FFFFFFFu : since + add;
            }
            const uint64_t G = __ballot(start && since_l < W);
            const uint64_t Sq = S & incl;
            const bool gone = Sq ? ((G >> (63u - (uint32_t)__builtin_clzll(Sq))) & 1) : carry_gone;
            const bool keep = in && !gone;
            const uint64_t Kb = __ballot(keep);
            if (WRITE && keep) {
                uint64_t pos = o + cnt_b + __builtin_popcountll(Kb & lt);
                if (pos < out_cap) out_text[pos] = (uint8_t)c;  // out_offsets are the caller's: stay inside out_text
            }
            cnt_b += __builtin_popcountll(Kb);
            cnt_c += __builtin_popcountll(S & ~G);
            if (D) {
                uint32_t q = 63u - (uint32_t)__builtin_clzll(D);
                since = (uint32_t)__builtin_popcountll(S & ~(~0ULL >> (63u - q)));
            } else {
                uint32_t add = (uint32_t)__builtin_popcountll(S);
                since = since > 0x7FFFFFFFu - add ? 0x7FFFFFFFu : since + add;
            }
            if (S) carry_gone = (G >> (63u - (uint32_t)__builtin_clzll(S))) & 1;
        }
        if (!WRITE && lane == 0) {
            kept_bytes[d] = cnt_b;
            kept_cps[d] = cnt_c;
        }
    }
}

// The same pass with a dword per lane (text and dup 4-byte aligned): a wave step covers 256 bytes.  Per lane: flags of its
// code-point starts S and marked starts M (SWAR on the text / dup dwords).  Across lanes: the number of starts below a lane
// (3 ballots over the lanes' counts), the start index of the most recent mark below it (one ballot + one shuffle), whether
// the last start below it is gone (two ballots) -- 6 ballots per 256 bytes where the byte-per-lane pass needs 16 -- and,
// for the write pass, the kept bytes below it (3 more).  Kept bytes leave as one unaligned dword store when all four stay.
template <bool WRITE>
__global__ void __launch_bounds__(256) compact4_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes,
                                                       const uint64_t* __restrict__ offs, uint64_t n, const uint8_t* __restrict__ dup,
                                                       uint32_t W, uint64_t* kept_bytes, uint64_t* kept_cps,
                                                       const uint8_t* __restrict__ keep_doc, const uint64_t* __restrict__ out_offs,
                                                       uint8_t* __restrict__ out_text, uint64_t out_cap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    const uint32_t* __restrict__ dw = reinterpret_cast<const uint32_t*>(dup);
    auto sat_add = [](uint32_t x, uint32_t add) -> uint32_t { return x > 0x7FFFFFFFu - add ? 0x7FFFFFFFu : x + add; };
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        if (WRITE && !keep_doc[d]) continue;
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        const uint64_t w0 = b >> 2, w1 = b < e ? (e + 3) >> 2 : w0;
        uint64_t done = WRITE ? out_offs[d] : 0;  // write pass: next output position of the wave
        uint32_t cnt_b = 0, cnt_c = 0;            // count pass: per-lane sums
        uint32_t since = 0x7FFFFFFFu;             // starts since the most recent marked start, at the end of the previous step
        bool carry_gone = false;
        for (uint64_t wi = w0; wi < w1; wi += 128) {
            uint32_t ww[2], dd[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                ww[j] = i < w1 ? tw[i] : 0u;  // (the text's last dword lies inside MDD_TEXT_PAD)
                dd[j] = 0u;
                if (i < w1) {
                    if ((i + 1) * 4 <= text_bytes) {
                        dd[j] = dw[i];
                    } else {  // dup is exactly text_bytes long: its last, partial dword byte by byte
                        for (uint64_t q = i * 4; q < text_bytes; q++) dd[j] |= (uint32_t)dup[q] << (8 * (q - i * 4));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint64_t i = wi + 64 * j + lane;
                const uint64_t p = i << 2;
                const uint32_t w = ww[j], du = dd[j];
                uint32_t in = i < w1 ? 0x80808080u : 0u;  // bytes of the dword inside [b, e), as bit 7 of each byte
                if (p < b || p + 4 > e) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (p + k < b || p + k >= e) in &= ~(0x80u << (8 * k));
                }
                const uint32_t cb = w & ~(w << 1) & 0x80808080u;                                    // continuation bytes (10xxxxxx)
                const uint32_t S = in & ~cb;                                                        // code-point starts
                const uint32_t nz = (((du & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | du) & 0x80808080u;        // dup bytes != 0
                // The common step: 256 bytes inside the document without a mark, the most recent marked start at least W starts
                // back and no removed code point reaching in -- every byte stays.  (Most documents hold no mark at all: the six
                // ballots and the per-byte fates below were 2.9 + 3.3 ms of the stage for nothing.)
                if (since >= W && !carry_gone && !__ballot(nz != 0u || in != 0x80808080u)) {
                    if (WRITE) {
                        store_low_bytes(out_text, done + 4 * lane, out_cap, w, 4);
                        done += 256;
                    } else {
                        cnt_b += 4;
                        cnt_c += (uint32_t)__builtin_popcount(S);
                    }
                    continue;
                }
                const uint32_t M = S & nz;                                                          // marked window starts
                const uint32_t s = (uint32_t)__builtin_popcount(S);
                const uint64_t B0 = __ballot(s & 1u), B1 = __ballot(s & 2u), B2 = __ballot(s & 4u);
                const uint32_t P = (uint32_t)(__builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) + 4 * __builtin_popcountll(B2 & lt));
                const uint32_t T = (uint32_t)(__builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2));
                // index (1-based among the step's starts) of this lane's last marked start, and of the last one below this lane
                const bool has_m = M != 0u;
                const uint32_t last_m = has_m ? 31u - (uint32_t)__builtin_clz(M) : 0u;
                const uint32_t Vm = P + (uint32_t)__builtin_popcount(S & (0xFFFFFFFFu >> (31u - last_m)));
                const uint64_t BM = __ballot(has_m);
                const uint64_t pm = BM & lt;
                const bool have_prev = pm != 0;
                const uint32_t Vprev = (uint32_t)__shfl((int)Vm, have_prev ? 63 - __builtin_clzll(pm) : 0, 64);
                // which of the lane's starts are gone: fewer than W starts since the most recent marked one
                uint32_t gs = 0, rank = 0, mr = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t f = 0x80u << (8 * k);
                    if (S & f) {
                        rank++;
                        if (M & f) mr = rank;
                        const uint32_t sc = mr ? rank - mr : (have_prev ? P + rank - Vprev : sat_add(since, P + rank));
                        if (sc < W) gs |= f;
                    }
                }
                // a continuation byte shares the fate of its code point's start, which may lie in a lane below or in the previous step
                const bool has_s = S != 0u;
                const bool g_last = has_s && ((gs >> (31u - (uint32_t)__builtin_clz(S | 1u))) & 1u);
                const uint64_t BS = __ballot(has_s), BG = __ballot(g_last);
                const uint64_t ps = BS & lt;
                bool g = ps ? ((BG >> (63 - __builtin_clzll(ps))) & 1) : carry_gone;
                uint32_t keep = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t f = 0x80u << (8 * k);
                    if (S & f) g = (gs & f) != 0u;
                    if ((in & f) && !g) keep |= f;
                }
                const uint32_t kc = (uint32_t)__builtin_popcount(keep);
                if (WRITE) {
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
                } else {
                    cnt_b += kc;
                    cnt_c += (uint32_t)__builtin_popcount(keep & S);
                }
                // carries into the next step
                if (BM) since = T - (uint32_t)__shfl((int)Vm, 63 - __builtin_clzll(BM), 64);
                else since = sat_add(since, T);
                if (BS) carry_gone = (BG >> (63 - __builtin_clzll(BS))) & 1;
            }
        }
        if (!WRITE) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                cnt_b += (uint32_t)__shfl_xor((int)cnt_b, o, 64);
                cnt_c += (uint32_t)__shfl_xor((int)cnt_c, o, 64);
            }
            if (lane == 0) {
                kept_bytes[d] = cnt_b;
                kept_cps[d] = cnt_c;
            }
        }
    }
}

unsigned compact_grid(const mdd_ctx* ctx, uint64_t n) {
    uint64_t blocks = (n + 3) / 4, cap = (uint64_t)ctx->cus * 64;
    return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace

extern "C" int mdd_substr_mark(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                               int window, int part, int n_parts, uint64_t* table, uint64_t tab_size, uint8_t* dup) {
    REQUIRE(ctx, "mdd_substr_mark: ctx is NULL");
    REQUIRE(window >= 1 && window <= 128, "mdd_substr_mark: window=%d outside [1,128]", window);
    REQUIRE(text_bytes < (1ULL << 40), "mdd_substr_mark: text_bytes exceeds 2^40");
    REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "mdd_substr_mark: part=%d outside [0,%d)", part, n_parts);
    if (n_docs == 0 || text_bytes == 0) return MDD_OK;
    REQUIRE(text && offsets && table && dup, "mdd_substr_mark: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(text) & 3u) == 0, "mdd_substr_mark: text must be 4-byte aligned");
    REQUIRE(tab_size >= 2, "mdd_substr_mark: tab_size=%llu too small", (unsigned long long)tab_size);
    REQUIRE(n_docs < (1ull << 40), "mdd_substr_mark: n_docs too large");
    HIP_TRY(hipSetDevice(ctx->device));
    WinArgs A{};
    A.text = text;
    A.text_bytes = text_bytes;
    A.offsets = offsets;
    A.n_docs = n_docs;
    A.window = (uint32_t)window;
    uint64_t seg = ctx->seg_bytes;
    uint64_t need = text_bytes / SEG_ITEM_CAP + 1;
    if (seg < need) seg = need;
    seg = (seg + 255) / 256 * 256;
    REQUIRE(seg <= 0xFFFFFF00ull, "mdd_substr_mark: text too large for the segment plan");
    uint64_t max_items = text_bytes / seg;
    if (max_items > SEG_ITEM_CAP) max_items = SEG_ITEM_CAP;
    A.seg_bytes = (uint32_t)seg;
    A.items = ctx->seg_items;
    A.n_items = ctx->seg_n_items;
    A.max_items = max_items;
    A.work_counter = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 1);
    A.table = reinterpret_cast<unsigned long long*>(table);
    A.tab_size = tab_size;
    A.dup = dup;
    A.overflow = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 2);
    A.part = (uint32_t)part;
    A.n_parts = (uint32_t)n_parts;
    HIP_TRY(hipMemsetAsync(table, 0xFF, sizeof(uint64_t) * tab_size, ctx->stream));
    if (part == 0) HIP_TRY(hipMemsetAsync(dup, 0, text_bytes, ctx->stream));
    uint64_t waves = (uint64_t)ctx->cus * 4 * 8;
    if (waves > n_docs + max_items) waves = n_docs + max_items;
    HIP_TRY(hipMemsetAsync(ctx->seg_n_items, 0, 2 * sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(seg_plan_kernel, dim3(grid_for(n_docs, 256)), dim3(256), 0, ctx->stream, offsets, n_docs, text_bytes,
                       A.seg_bytes, ctx->seg_items, reinterpret_cast<unsigned long long*>(ctx->seg_n_items), max_items,
                       (uint32_t*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    while ((1ull << A.batch_shift) < waves * 4) A.batch_shift++;
    hipLaunchKernelGGL(win_kernel, dim3((unsigned)waves), dim3(64), 0, ctx->stream, A);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_substr_check(mdd_ctx* ctx, uint64_t* n_overflow_host) {
    REQUIRE(ctx, "mdd_substr_check: ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    uint64_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, ctx->seg_n_items + 2, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->seg_n_items + 2, 0, sizeof n, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n_overflow_host) *n_overflow_host = n;
    if (n)
        return mddi::fail(MDD_ERR_INVALID,
                          "mdd_substr_mark: %llu windows found no room (table method: tab_size below 2 x the windows of a pass; "
                          "sorted method: ent_cap below the windows of a pass, or more than 768 distinct windows in one leaf): "
                          "the marks are incomplete -- use more passes",
                          (unsigned long long)n);
    return MDD_OK;
}

extern "C" int mdd_substr_count(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                int window, const uint8_t* dup, uint64_t* kept_bytes, uint64_t* kept_cps) {
    REQUIRE(ctx, "mdd_substr_count: ctx is NULL");
    REQUIRE(window >= 1 && window <= 128, "mdd_substr_count: window=%d outside [1,128]", window);
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && kept_bytes && kept_cps && ((text && dup) || text_bytes == 0), "mdd_substr_count: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if (((reinterpret_cast<uintptr_t>(text) | reinterpret_cast<uintptr_t>(dup)) & 3u) == 0)  // the usual case: dword per lane
        hipLaunchKernelGGL(compact4_kernel<false>, dim3(compact_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, dup, (uint32_t)window, kept_bytes, kept_cps, (const uint8_t*)nullptr, (const uint64_t*)nullptr,
                           (uint8_t*)nullptr, (uint64_t)0);
    else
        hipLaunchKernelGGL(compact_kernel<false>, dim3(compact_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, dup, (uint32_t)window, kept_bytes, kept_cps, (const uint8_t*)nullptr, (const uint64_t*)nullptr,
                           (uint8_t*)nullptr, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_substr_write(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                int window, const uint8_t* dup, const uint8_t* keep_doc, const uint64_t* out_offsets,
                                uint8_t* out_text, uint64_t out_capacity) {
    REQUIRE(ctx, "mdd_substr_write: ctx is NULL");
    REQUIRE(window >= 1 && window <= 128, "mdd_substr_write: window=%d outside [1,128]", window);
    if (n_docs == 0) return MDD_OK;
    REQUIRE(offsets && keep_doc && out_offsets && ((text && dup && out_text) || text_bytes == 0), "mdd_substr_write: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    if (((reinterpret_cast<uintptr_t>(text) | reinterpret_cast<uintptr_t>(dup)) & 3u) == 0)
        hipLaunchKernelGGL(compact4_kernel<true>, dim3(compact_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, dup, (uint32_t)window, (uint64_t*)nullptr, (uint64_t*)nullptr, keep_doc, out_offsets, out_text,
                           out_capacity);
    else
        hipLaunchKernelGGL(compact_kernel<true>, dim3(compact_grid(ctx, n_docs)), dim3(256), 0, ctx->stream, text, text_bytes, offsets,
                           n_docs, dup, (uint32_t)window, (uint64_t*)nullptr, (uint64_t*)nullptr, keep_doc, out_offsets, out_text,
                           out_capacity);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}
