// Exact substring dedup, bandwidth-bound formulation (same reference lines as mdd_substr.hip:
// exact_substr_dedup.py:34-47 -- hash every window, group, all occurrences but the first are repeats).
//
// mdd_substr.hip's win_kernel puts one random 8-byte atomic per window on a table far larger than any cache; such
// atomics execute at the memory side (~17.5 G/s on this chip, 86 bytes written per atomic: 184 GB for C2's 2.13 G
// windows).  Here the windows are SORTED INTO LEAVES first and grouped in LDS:
//   emit      every window -> one 8-byte entry [d2:11 | tag:17 | position:36] appended to a flat array (each wave
//             reserves chunks of it); d1 = key bits 63..53 goes to a parallel u16 array and is counted in LDS
//   scatter<1>  entries -> their d1 partition (2048 partitions), tiles sorted by bin in LDS before they are written
//   count2 / scatter<2>   the same inside every partition on d2 = key bits 52..42  ->  2^22 leaves of ~500 windows
//   leaf      one wave per leaf: an LDS table of 8-byte entries (tag | lowest position); a tag match is confirmed by
//             COMPARING THE BYTES of the two windows (so the result is exact), the later window of a matching pair is
//             marked.  Leaves of any size work: above 640 windows a leaf is grouped in rounds over shares of the tag space,
//             above 16384 (boilerplate puts millions of identical windows into one leaf) it is sliced over all waves and
//             the slices meet in a small global table (bigleaf_kernel); more than LEAF_SLOTS * 3/4 DISTINCT windows in one
//             round (hash adversary) raise the overflow count that mdd_substr_check reports.
// Equal 64-bit keys land in the same leaf, so every pair of identical windows meets in one LDS table.
//
// CANDIDATE FILTER (mdd_substr_candidates, round 3).  Most windows of a corpus occur once; sorting them all only to
// find that out is what the method above spends its time on.  A window that occurs twice contains, at the same place
// in both copies, every k-gram of its bytes -- so a window can only be a repeat (or the first of a repeated content)
// when the k-gram its content SELECTS occurs twice in the corpus.  The selection is a minimizer: of the 19 32-byte
// k-grams that start in the window's first 19 bytes (a window of >= 50 code points has >= 50 bytes, so all of them lie
// inside it) the one with the smallest hash, leftmost on ties -- a function of the window's bytes alone.  Consecutive
// windows mostly select the same k-gram: one ANCHOR per ~10 windows.  The anchors (10 % of the entries) go through the
// same two-level sort; a leaf pass flags every anchor whose key occurs twice (key equality only: a false match merely
// lets windows through); a streaming pass turns anchor flags into window flags (bit p of cand = any flagged anchor in
// [p, p + 18]: a superset of "p's own anchor is flagged"; both are bitmaps).  emit_kernel then skips every window whose bit is clear.
// Everything the filter does errs on the side of keeping a window, so the marks are those of the unfiltered method.
#include "mdd_shingle.h"

namespace {

constexpr int D1_BITS = 11, BINS1 = 1 << D1_BITS;  // level-1 partitions
constexpr int D2_BITS = 11, BINS2 = 1 << D2_BITS;  // leaves per partition
constexpr int TAG_BITS = 17;
constexpr int POS_BITS = 36;  // text_bytes < 64 GiB per call
static_assert(D2_BITS + TAG_BITS + POS_BITS == 64, "entry layout");
constexpr uint64_t POS_M = (1ULL << POS_BITS) - 1;
constexpr uint64_t ENT_INVALID = ~0ULL;
constexpr uint16_t D1_INVALID = 0xFFFFu;
constexpr uint32_t EMIT_CHUNK = 2048;  // entries a wave reserves at a time
constexpr int XCDS = 8;
constexpr int LEAF_SLOTS = 1024;  // LDS table entries per leaf (one wave per leaf): up to 768 DISTINCT windows

// 64-bit content key of the L bytes at byte offset s from tv.docw.  Equal bytes give equal keys; that is all exactness
// needs (the leaves compare bytes), the rest is balance.  The per-dword step is add / rotate / xor only -- v_mul_lo_u32
// issues at a quarter of the rate, so the multiplies (mix64) come once per window, not once per dword as in win_kernel's
// window_key -- and the dwords are loaded eight at a time (a loop of single dependent loads left the kernel waiting on
// the L1 round trip thirteen times per window).
// (A one-multiply step, h' = (lo32(h) ^ w) * C + h as a single v_mad_u64_u32, is two instructions instead of five, but it is
// not a bijection of the state: on repetitive text distinct windows pile into single leaves -- test_substr_sorted_method_big_leaves
// then reports windows without room -- so the step stays add / rotate / xor.)
__device__ __forceinline__ uint64_t wkey(const TextView& tv, uint32_t s, uint32_t L) {
    const uint32_t* src = tv.docw + (s >> 2);
    const uint32_t sh = s & 3u;
    uint32_t h1 = 0x9E3779B9u ^ L, h2 = 0x85EBCA6Bu;
    uint32_t lo = src[0];
    const uint32_t nd = (L + 3u) >> 2;
    for (uint32_t k0 = 0; k0 < nd; k0 += 8) {
        uint32_t r[8];  // (at most 32 bytes past the window: inside MDD_TEXT_PAD)
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = src[k0 + 1 + j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t k = k0 + j;
            if (k < nd) {
                uint32_t w = __builtin_amdgcn_alignbyte(r[j], lo, sh);
                const uint32_t rem = L - 4u * k;
                if (rem < 4u) w &= 0xFFFFFFFFu >> (8u * (4u - rem));
                h1 = __builtin_amdgcn_alignbit(h1, h1, 27) + w;         // rotl 5
                h2 = (__builtin_amdgcn_alignbit(h2, h2, 19) ^ w) + h1;  // rotl 13
                lo = r[j];
            }
        }
    }
    return mix64(((uint64_t)h1 << 32) | h2);
}
// wkey for a length known at compile time (the same value as wkey(tv, s, L)): no loop control, no per-dword tests, the
// tail mask a constant.  emit_kernel takes it for the windows of pure-ASCII segments at the default window of 50.
template <uint32_t L>
__device__ __forceinline__ uint64_t wkey_fixed(const TextView& tv, uint32_t s) {
    constexpr uint32_t ND = (L + 3u) / 4u;
    const uint32_t* src = tv.docw + (s >> 2);
    const uint32_t sh = s & 3u;
    uint32_t r[ND + 1];
#pragma unroll
    for (uint32_t j = 0; j <= ND; j++) r[j] = src[j];
    uint32_t h1 = 0x9E3779B9u ^ L, h2 = 0x85EBCA6Bu;
#pragma unroll
    for (uint32_t k = 0; k < ND; k++) {
        uint32_t w = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh);
        if (L - 4u * k < 4u) w &= 0xFFFFFFFFu >> (8u * (4u - (L - 4u * k)));
        h1 = __builtin_amdgcn_alignbit(h1, h1, 27) + w;
        h2 = (__builtin_amdgcn_alignbit(h2, h2, 19) ^ w) + h1;
    }
    return mix64(((uint64_t)h1 << 32) | h2);
}
__device__ __forceinline__ uint32_t d1_of(uint64_t key) { return (uint32_t)(key >> (64 - D1_BITS)); }
__device__ __forceinline__ uint64_t entry_of(uint64_t key, uint64_t pos) {  // [d2:11 | tag:17 | pos:36] = key bits 52..25, position
    return (((key >> (64 - D1_BITS - D2_BITS - TAG_BITS)) & ((1ULL << (D2_BITS + TAG_BITS)) - 1)) << POS_BITS) | pos;
}
__device__ __forceinline__ uint32_t d2_of(uint64_t e) { return (uint32_t)(e >> (POS_BITS + TAG_BITS)); }

// wave_sync() without its fences, which make the compiler wait for the global loads in flight (the next leaf's entries)
// as well.  Between the lanes of ONE wave the LDS needs no fence -- DS instructions of a wave execute in issue order --
// only the compiler must keep its hands off the order: a compiler-level memory barrier on both sides of the wave barrier.
__device__ __forceinline__ void lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct EmitArgs {
    const uint8_t* text;
    uint64_t text_bytes;
    const uint64_t* offsets;
    uint64_t n_docs;
    uint32_t window, seg_bytes;
    const uint64_t* items;
    const uint64_t* n_items;
    uint64_t max_items;
    unsigned long long* work_counter;
    uint64_t* ent;
    uint16_t* d1;               // level-1 digit of every entry (the entry itself has no room for it)
    uint64_t ent_cap;
    unsigned long long* n_ent;  // entries handed out so far (whole chunks)
    unsigned int* hist1;        // [BINS1]
    unsigned long long* overflow;
    uint32_t part, n_parts;
    uint32_t batch_shift;  // a claim on the work counter takes (items left) >> batch_shift items, clamped to [1, 16]
    const uint64_t* cand;  // candidate bitmap, a bit per byte position (mdd_substr_candidates), or NULL: every window takes part
    uint32_t max_batch;    // items per claim at most (16 without the filter; with it an item is cheap and batches of 64 keep the claims rare)
};

// persistent waves pulling documents (then queued segments of long documents) from one counter, four waves per workgroup
// sharing the LDS histogram of d1; every wave leaves once the counter passes `total`
constexpr int EMIT_WAVES = 4;
constexpr int CPLANE_WORDS = 264;  // candidate bits per wave in LDS: 264 x 64 positions = a 16 KiB segment + alignment
__global__ void __launch_bounds__(64 * EMIT_WAVES) emit_kernel(EmitArgs A) {
    __shared__ WaveLds lds_all[EMIT_WAVES];
    __shared__ unsigned int h1[BINS1];
    __shared__ uint64_t cplane_all[EMIT_WAVES][CPLANE_WORDS];
    WaveLds& lds = lds_all[threadIdx.x >> 6];
    uint64_t* cplane = cplane_all[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    for (int i = threadIdx.x; i < BINS1; i += 64 * EMIT_WAVES) h1[i] = 0;
    __syncthreads();
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(A.text);
    const uint64_t extra = *A.n_items;
    const uint64_t total = A.n_docs + (extra < A.max_items ? extra : A.max_items);
    // work items are claimed in batches (guided self-scheduling as in sig_kernel, mdd_hip.hip: same-address atomics retire
    // at ~11 ns each and the wave waits for its claim, so one claim per document would cap this kernel at ~90 M documents/s)
    auto fetch = [&](uint32_t n) -> uint64_t {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(A.work_counter, (unsigned long long)n);
        uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    uint64_t chunk_pos = 0, chunk_end = 0;  // this wave's current chunk of the entry array (wave-uniform)
    uint64_t item = 0, batch_end = 0;
    for (;; item++) {
        if (item >= batch_end) {  // `item` = end of this wave's previous batch: a lower bound of the counter
            const uint64_t want = (total > item ? total - item : 0) >> A.batch_shift;
            const uint32_t n = want < 1u ? 1u : (want > A.max_batch ? A.max_batch : (uint32_t)want);
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
        // With the candidate filter few windows are left and a round of 64 windows would wait a memory round trip for its
        // flags only to find none set.  So the item's words of the candidate bitmap are read ONCE into LDS (all loads in
        // flight): a window then finds its flag with one LDS read, and an item without any candidate is dropped before its
        // text is touched.
        uint64_t plane_u0 = 0;    // index (in the bitmap) of the first word held in LDS
        bool planes = false;
        if (A.cand) {
            const uint64_t p0 = beg + seg_beg, p1 = beg + seg_end;  // window starts of this item lie in [p0, p1)
            if (p0 >= p1) continue;
            const uint64_t u0 = p0 >> 6, u1 = (p1 - 1) >> 6;
            uint64_t acc = 0;
            if (u1 - u0 < (uint64_t)CPLANE_WORDS) {
                planes = true;
                plane_u0 = u0;
                lds_sync();  // (the previous item's readers are done)
                for (uint64_t i = lane; u0 + i <= u1; i += 64) {
                    const uint64_t v = A.cand[u0 + i];
                    cplane[i] = v;
                    acc |= v;
                }
                lds_sync();
            } else {  // a segment beyond the LDS words (segments grow with texts beyond 16 GiB): flags straight from memory
                for (uint64_t i = u0 + lane; i <= u1; i += 64) acc |= A.cand[i];
            }
            if (!__ballot(acc != 0ull)) continue;  // (bits of the neighbours in the first and last word only cost a look)
        }
        TextView tv;
        tv.docw = tw + (beg >> 2);
        const uint32_t boff = (uint32_t)beg & 3u;
        const bool ascii = segment_is_ascii(tv, boff, len, seg_beg, seg_end, A.window, lane);
        // the lanes with `on` append their window (content key, absolute position) to the entry array
        auto put = [&](bool on, uint64_t key, uint64_t pos) {
            const uint64_t bal = __ballot(on);
            const uint32_t cnt = (uint32_t)__popcll(bal);
            if (cnt == 0) return;
            if (chunk_pos + cnt > chunk_end) {  // not enough room left: void the rest (< 64 entries) and take a new chunk
                if (chunk_pos + lane < chunk_end && chunk_pos + lane < A.ent_cap) {
                    A.ent[chunk_pos + lane] = ENT_INVALID;
                    A.d1[chunk_pos + lane] = D1_INVALID;
                }
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(A.n_ent, (unsigned long long)EMIT_CHUNK);
                uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
                chunk_pos = ((uint64_t)hi << 32) | lo;
                chunk_end = chunk_pos + EMIT_CHUNK;
            }
            if (on) {
                uint64_t o = chunk_pos + below(bal);
                if (o < A.ent_cap) {
                    A.ent[o] = entry_of(key, pos);
                    A.d1[o] = (uint16_t)d1_of(key);
                    atomicAdd(&h1[d1_of(key)], 1u);
                } else if (below(bal) == 0) {
                    atomicAdd(A.overflow, 1ull);  // entry array too small (caller's contract): reported by mdd_substr_check
                }
            }
            chunk_pos += cnt;
        };
        if (planes && ascii) {
            // Pure-ASCII item under the filter: every byte offset is a window start, so the candidates are the set bits of
            // the planes.  They are queued (the boundary ring is idle on this path) and hashed 64 at a time with every lane
            // busy, instead of walking all rounds of 64 windows to find most of them empty.
            if (len <= A.window) continue;
            const uint32_t sb_end = seg_end < len - A.window ? seg_end : len - A.window;  // SUB:38: sb + window < len
            uint32_t q_head = 0, q_tail = 0;
            auto hash_queued = [&](uint32_t n) {
                lds_sync();
                const bool on0 = lane < n;
                const uint32_t sb = lds.ringS[(q_head + (on0 ? lane : 0u)) & (RING - 1)];
                bool on = on0;
                uint64_t key = 0;
                if (on) {
                    key = A.window == 50u ? wkey_fixed<50>(tv, boff + sb) : wkey(tv, boff + sb, A.window);
                    if (A.n_parts > 1 && (uint32_t)((key & 0x3FFFu) % A.n_parts) != A.part) on = false;
                }
                put(on, key, beg + sb);
                q_head += n;
                lds_sync();
            };
            const uint64_t u1 = (beg + seg_end - 1) >> 6;
            for (uint64_t wd = 0; plane_u0 + wd <= u1; wd++) {
                const uint64_t B = cplane[wd];
                if (B == 0) continue;
                const uint64_t q = ((plane_u0 + wd) << 6) + lane;  // absolute position of this lane's bit
                const bool sel = ((B >> lane) & 1ull) && q >= beg + seg_beg && q < beg + sb_end;
                const uint64_t sbal = __ballot(sel);
                if (sbal == 0) continue;
                if (sel) lds.ringS[(q_tail + below(sbal)) & (RING - 1)] = (uint32_t)(q - beg);
                q_tail += (uint32_t)__popcll(sbal);
                if (q_tail - q_head >= 64u) hash_queued(64u);
            }
            if (q_tail != q_head) hash_queued(q_tail - q_head);
            continue;
        }
        for_each_round(lds, A.text, beg, len, tv, boff, MDD_UNIT_CHAR, A.window, ascii, split, seg_beg, seg_end, lane,
                       [&](uint32_t s, uint32_t L, uint32_t nvalid, uint32_t) {
                           const uint32_t sb = s - boff;
                           // SUB:38 range(len(text) - duplicate_len): the window that ends with the document is never hashed
                           bool on = lane < nvalid && sb + L < len;
                           // candidate filter: a window whose selected k-gram occurs once in the corpus occurs once itself
                           if (A.cand && on) {
                               const uint64_t q = beg + sb;
                               const uint64_t wq = planes ? cplane[(q >> 6) - plane_u0] : A.cand[q >> 6];
                               on = (wq >> (q & 63u)) & 1ull;
                           }
                           uint64_t key = 0;
                           if (on) {
                               // (selecting the pass from a cheap hash of the window's first bytes BEFORE the key was tried:
                               // no gain -- a wave computes the key as long as ONE of its 64 lanes needs it)
                               // (pure-ASCII segment: every window is A.window bytes long -- a wave-uniform length lets the
                               // compiler do the loop control and the tail mask of the key on the scalar unit)
                               key = ascii ? (A.window == 50u ? wkey_fixed<50>(tv, s) : wkey(tv, s, A.window)) : wkey(tv, s, L);
                               if (A.n_parts > 1 && (uint32_t)((key & 0x3FFFu) % A.n_parts) != A.part) on = false;
                           }
                           put(on, key, beg + sb);
                       });
    }
    for (uint64_t i = chunk_pos + lane; i < chunk_end && i < A.ent_cap; i += 64) {  // void what is left of the last chunk
        A.ent[i] = ENT_INVALID;
        A.d1[i] = D1_INVALID;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS1; i += 64 * EMIT_WAVES)
        if (h1[i]) atomicAdd(&A.hist1[i], h1[i]);
}

// exclusive scan of n u32 counts -> u64 offsets[n+1] in three small launches: per-block sums (coalesced), scan of the
// block sums (one workgroup), per-block exclusive scan
constexpr int SCAN_BLOCK = 4096;  // counts per workgroup of 256 threads
__global__ void __launch_bounds__(256) scan_sums_kernel(const unsigned int* cnt, uint64_t n, uint64_t* sums) {
    __shared__ uint64_t red[256];
    uint64_t lo = (uint64_t)blockIdx.x * SCAN_BLOCK, s = 0;
    for (int k = 0; k < SCAN_BLOCK / 256; k++) {
        uint64_t i = lo + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) s += cnt[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(1024) scan_top_kernel(uint64_t* sums, uint64_t nb, uint64_t* total_out) {  // nb <= 1024
    __shared__ uint64_t v[1024];
    v[threadIdx.x] = threadIdx.x < nb ? sums[threadIdx.x] : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (uint64_t t = 0; t < nb; t++) {
            uint64_t x = v[t];
            v[t] = acc;
            acc += x;
        }
        *total_out = acc;
    }
    __syncthreads();
    if (threadIdx.x < nb) sums[threadIdx.x] = v[threadIdx.x];
}
__global__ void __launch_bounds__(256) scan_apply_kernel(const unsigned int* cnt, uint64_t n, const uint64_t* sums, uint64_t* off) {
    __shared__ uint64_t tsum[256];
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * (SCAN_BLOCK / 256);
    uint32_t c[SCAN_BLOCK / 256];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_BLOCK / 256; k++) {
        c[k] = lo + k < n ? cnt[lo + k] : 0u;
        s += c[k];
    }
    tsum[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = sums[blockIdx.x];
        for (int t = 0; t < 256; t++) {
            uint64_t x = tsum[t];
            tsum[t] = acc;
            acc += x;
        }
    }
    __syncthreads();
    s = tsum[threadIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_BLOCK / 256; k++) {
        if (lo + k < n) off[lo + k] = s;
        s += c[k];
    }
}
// sums: scratch of ceil(n / SCAN_BLOCK) <= 1024 words
static int exclusive_scan(hipStream_t st, const unsigned int* cnt, uint64_t n, uint64_t* off, uint64_t* sums) {
    unsigned nb = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, st, cnt, n, sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, st, sums, (uint64_t)nb, off + n);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, st, cnt, n, sums, off);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

// ---- the two scatter levels -------------------------------------------------------------------------------------
// A scatter that stores each entry straight from the thread that loaded it issues 64 different cache lines per store
// instruction; the texture path takes ~5 cycles per line, which made 2.1 G stores cost 20 ms whatever the tile shape.
// So a tile is SORTED BY BIN IN LDS first (rank from an LDS counter, block scan of the counts, ONE reservation per
// non-empty bin and tile) and written out in sorted order: consecutive lanes hold consecutive entries of a bin's run and
// a store instruction touches few lines.  What then bounds a level is the number of RUNS: a run is a random DRAM access,
// and the chip sustains ~17.5 G of those per second (the same figure as its random atomics).
//   level 1: 2048 bins, tiles of 12288 -> runs of 6 entries scattered over the whole entry array (17 ms on C2)
//   level 2: 2048 bins, tiles of 16384 -> runs of 8 entries, but inside one partition (8 MB), where the XCD's L2 merges
//            them into full lines (9 ms): tiles never straddle a partition (tile_off = running sum of ceil(size_p / 16384)) and
//            workgroup b takes tile (b % 8) * ceil(T / 8) + b / 8, so the workgroups of one XCD (b % 8) walk through one
//            contiguous eighth of the partitions and all stores into a partition pass through the same L2.
// (Tried and dropped: 1024 + 4096 bins -- level 1 16 ms, level 2 18 ms; per-XCD output sub-streams at level 1 -- no gain, 8 x the open pages; reservations per 8 tiles
// from a histogram pre-pass -- fewer atomics, but a bin's lines then fill over 8 tiles' time and leave the L2 half-written.)
constexpr int SC_THREADS = 1024;
template <int LEVEL>
struct ScatterLds {
    static constexpr int BINS = LEVEL == 1 ? BINS1 : BINS2;
    static constexpr int N = LEVEL == 1 ? 12288 : 16384;  // entries per tile
    uint64_t sorted[N];
    uint16_t sbin[LEVEL == 1 ? N : 1];  // level 1: the entry does not carry d1, its bin is staged beside it
    unsigned int start[BINS];           // counts, then exclusive starts inside the tile
    unsigned long long gbase[BINS];     // global index of the bin's run minus its start inside the tile
    unsigned int wsum[SC_THREADS / 64 + 1];
    uint32_t p;
};
constexpr int TILE1 = ScatterLds<1>::N, TILE2 = ScatterLds<2>::N;
static_assert(TILE1 % SC_THREADS == 0 && TILE2 % SC_THREADS == 0, "whole entries per thread");
static_assert(BINS1 % SC_THREADS == 0 && BINS2 % SC_THREADS == 0, "whole bins per thread");

__global__ void __launch_bounds__(1024) tile_map_kernel(const uint64_t* off1, unsigned int* tile_off) {
    __shared__ unsigned int v[BINS1];
    for (int i = threadIdx.x; i < BINS1; i += 1024) v[i] = (unsigned int)((off1[i + 1] - off1[i] + TILE2 - 1) / TILE2);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int i = 0; i < BINS1; i++) {
            unsigned int x = v[i];
            v[i] = acc;
            acc += x;
        }
        tile_off[BINS1] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS1; i += 1024) tile_off[i] = v[i];
}

// the level-2 tile of this workgroup: partition p and entry range [lo, hi); false = nothing to do
__device__ __forceinline__ bool level2_tile(const uint64_t* off1, const unsigned int* tile_off, uint32_t* s_p, uint32_t& p, uint64_t& lo,
                                            uint64_t& hi) {
    const uint32_t n_tiles = tile_off[BINS1], per = (n_tiles + XCDS - 1) / XCDS;
    if (blockIdx.x / XCDS >= per) return false;
    const uint32_t tile = (blockIdx.x % XCDS) * per + blockIdx.x / XCDS;
    if (tile >= n_tiles) return false;
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = BINS1;  // invariant: tile_off[a] <= tile < tile_off[b]
        while (b - a > 1) {
            uint32_t mid = (a + b) >> 1;
            if (tile_off[mid] <= tile) a = mid;
            else b = mid;
        }
        *s_p = a;
    }
    __syncthreads();
    p = *s_p;
    lo = off1[p] + (uint64_t)(tile - tile_off[p]) * TILE2;
    const uint64_t pend = off1[p + 1];
    hi = lo + TILE2 < pend ? lo + TILE2 : pend;
    return true;
}

// leaf sizes: cnt2[partition * 2048 + d2]
__global__ void __launch_bounds__(256) count2_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ off1,
                                                     const unsigned int* __restrict__ tile_off, unsigned int* cnt2) {
    __shared__ unsigned int l_cnt[BINS2];
    __shared__ uint32_t s_p;
    for (int i = threadIdx.x; i < BINS2; i += 256) l_cnt[i] = 0;
    uint32_t p;
    uint64_t lo, hi;
    if (!level2_tile(off1, tile_off, &s_p, p, lo, hi)) return;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&l_cnt[d2_of(src[i])], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < BINS2; i += 256)
        if (l_cnt[i]) atomicAdd(&cnt2[(uint64_t)p * BINS2 + i], l_cnt[i]);
}

template <int LEVEL>
__global__ void __launch_bounds__(SC_THREADS) scatter_kernel(const uint64_t* __restrict__ src, const uint16_t* __restrict__ d1,
                                                             const unsigned long long* n_ent, uint64_t cap, const uint64_t* __restrict__ off1,
                                                             const unsigned int* __restrict__ tile_off, uint64_t* __restrict__ dst,
                                                             unsigned long long* cursor) {
    __shared__ ScatterLds<LEVEL> L;
    constexpr int N = ScatterLds<LEVEL>::N, BINS = ScatterLds<LEVEL>::BINS, EPT = N / SC_THREADS, BPT = BINS / SC_THREADS;
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < BINS; i += SC_THREADS) L.start[i] = 0;
    uint64_t lo, hi;
    if (LEVEL == 1) {
        const uint64_t n = *n_ent < cap ? *n_ent : cap;
        lo = (uint64_t)blockIdx.x * N;
        if (lo >= n) return;
        hi = lo + N < n ? lo + N : n;
        __syncthreads();
    } else {
        uint32_t p;
        if (!level2_tile(off1, tile_off, &L.p, p, lo, hi)) return;
        cursor += (uint64_t)p * BINS;
    }
    // 1. load, rank inside the bin
    uint64_t e[EPT];
    uint16_t bin[EPT], rk[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const uint64_t i = lo + (uint64_t)k * SC_THREADS + tid;
        e[k] = i < hi ? src[i] : ENT_INVALID;
        if (LEVEL == 1) bin[k] = i < hi ? (uint16_t)(d1[i] & (BINS - 1)) : (uint16_t)0;
        else bin[k] = (uint16_t)d2_of(e[k]);
    }
#pragma unroll
    for (int k = 0; k < EPT; k++) rk[k] = e[k] != ENT_INVALID ? (uint16_t)atomicAdd(&L.start[bin[k]], 1u) : (uint16_t)0;
    __syncthreads();
    // 2. exclusive scan of the counts (BPT consecutive bins per thread), one reservation per non-empty bin
    unsigned int c[BPT], mine = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        c[j] = L.start[BPT * tid + j];
        mine += c[j];
    }
    unsigned int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned int y = (unsigned int)__shfl_up((int)incl, o, 64);
        if ((tid & 63) >= (uint32_t)o) incl += y;
    }
    if ((tid & 63) == 63) L.wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned int st = incl - mine;
    for (uint32_t w = 0; w < (tid >> 6); w++) st += L.wsum[w];
#pragma unroll
    for (int j = 0; j < BPT; j++) {  // (a thread's counts are read and replaced by itself only)
        L.start[BPT * tid + j] = st;
        if (c[j]) L.gbase[BPT * tid + j] = atomicAdd(&cursor[BPT * tid + j], (unsigned long long)c[j]) - st;
        st += c[j];
    }
    if (tid == SC_THREADS - 1) L.wsum[SC_THREADS / 64] = st;  // entries of the tile
    __syncthreads();
    // 3. sorted order in LDS
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (e[k] != ENT_INVALID) {
            const unsigned int at = L.start[bin[k]] + rk[k];
            L.sorted[at] = e[k];
            if (LEVEL == 1) L.sbin[at] = bin[k];
        }
    __syncthreads();
    // 4. out, consecutive lanes = consecutive entries
    const unsigned int total = L.wsum[SC_THREADS / 64];
#pragma unroll 4
    for (int k = 0; k < EPT; k++) {
        const unsigned int i = (unsigned int)k * SC_THREADS + tid;
        if (i >= total) break;
        const uint64_t v = L.sorted[i];
        const uint32_t b = LEVEL == 1 ? (uint32_t)L.sbin[i] : d2_of(v);
        dst[L.gbase[b] + i] = v;
    }
}

// Are the windows of W code points that start at text positions p and q identical?  (p is a window start, so its W code
// points exist and are followed by at least one more byte of its document.)  Both windows are loaded 64 bytes at a
// time (eight 16-byte loads in flight: one memory round trip covers a 50-character ASCII window); code-point
// starts of p's bytes are counted on the way and the comparison stops at the byte where p's (W+1)-th code point starts.
// (q is a window start too; while the bytes agree both windows have the same code-point structure, so no load goes
// further than 70 bytes past the end of either window: inside MDD_TEXT_PAD.)
constexpr int CMP_DW = 15;  // dwords compared per batch (60 bytes: a 50-character ASCII window in one batch)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at a 4-byte aligned address
__device__ __forceinline__ bool same_window_cp(const uint8_t* text, uint64_t p, uint64_t q, uint32_t W) {
    const uint32_t* tp = reinterpret_cast<const uint32_t*>(text) + (p >> 2);
    const uint32_t* tq = reinterpret_cast<const uint32_t*>(text) + (q >> 2);
    const uint32_t sp = (uint32_t)p & 3u, sq = (uint32_t)q & 3u;
    uint32_t starts = 0;  // code-point starts of p seen so far
    for (uint32_t off = 0;; off += CMP_DW) {
        // every lane reads its own two places of the text, so what a load costs is the number of lane requests:
        // four 16-byte loads per side instead of sixteen 4-byte ones
        uint32_t ra[CMP_DW + 1], rb[CMP_DW + 1];
#pragma unroll
        for (int k = 0; k <= CMP_DW; k += 4) {
            const u32x4_a4 va = *reinterpret_cast<const u32x4_a4*>(tp + off + k);
            const u32x4_a4 vb = *reinterpret_cast<const u32x4_a4*>(tq + off + k);
            ra[k] = va.x, ra[k + 1] = va.y, ra[k + 2] = va.z, ra[k + 3] = va.w;
            rb[k] = vb.x, rb[k + 1] = vb.y, rb[k + 2] = vb.z, rb[k + 3] = vb.w;
        }
#pragma unroll
        for (int k = 0; k < CMP_DW; k++) {
            const uint32_t a = __builtin_amdgcn_alignbyte(ra[k + 1], ra[k], sp);
            const uint32_t b = __builtin_amdgcn_alignbyte(rb[k + 1], rb[k], sq);
            // bit 0 of every byte of m: that byte is NOT a continuation byte (10xxxxxx)
            const uint32_t m = ~((a >> 7) & ~(a >> 6)) & 0x01010101u;
            const uint32_t n = __builtin_popcount(m);
            const uint32_t x = a ^ b;
            if (starts + n <= W) {  // the whole dword belongs to the window
                if (x) return false;
                starts += n;
                continue;
            }
            // the window ends inside this dword: at the byte holding start number W+1 (counting from 1)
            uint32_t need = W + 1 - starts, seen = 0, keep = 0;
#pragma unroll
            for (int by = 0; by < 4; by++) {
                seen += (m >> (8 * by)) & 1u;
                if (seen < need) keep |= 0xFFu << (8 * by);
            }
            return (x & keep) == 0;
        }
    }
}

struct LeafArgs {
    const uint64_t* ent;
    const uint64_t* off2;  // [2^22 + 1] leaf offsets
    uint64_t n_leaves;
    const uint8_t* text;
    uint64_t text_bytes;
    uint32_t window;
    uint8_t* dup;
    unsigned long long* overflow;
    unsigned long long* prof;  // MDD_LEAF_PROF builds: cycles per phase, summed over waves
    unsigned int* n_big;       // leaves queued for bigleaf_kernel
    unsigned int* big_list;    // [BIG_LIST_CAP]
    unsigned long long* big_tab;  // [2^BIG_TAB_BITS], all-ones = empty
    uint32_t one_round, per_round;  // LEAF_ONE_ROUND / LEAF_PER_ROUND (a testing knob lowers them: MDD_SUBSTR_LEAF_ROUND)
    const unsigned long long* n_ent;  // entries handed out by emit_kernel (whole chunks: an upper bound of the windows sorted)
};
// With the candidate filter a pass may sort a few percent of the windows: 2^22 leaves of ten entries cost what 2^22 leaves of
// 500 cost (ranges, table clear, round trips).  Adjacent leaves are adjacent in memory and the table matches on (d2, tag),
// so 2^shift of them are grouped as ONE leaf until a leaf holds ~256-512 entries.  Every wave derives the same shift.
__device__ __forceinline__ uint32_t leaf_merge_shift(const LeafArgs& A) {
    const unsigned long long ne = *A.n_ent;
    uint32_t shift = 0;
    while (shift < 8u && (ne >> (D1_BITS + D2_BITS - shift)) < 256ull) shift++;
    return shift;
}
constexpr uint32_t LEAF_ONE_ROUND = 640;  // a leaf (slice) up to this many windows is grouped in one go (the table holds 768 distinct)
constexpr uint32_t LEAF_PER_ROUND = 448;  // beyond: rounds of about this many
constexpr uint32_t BIG_LEAF = 16384;      // a leaf above this many windows is sliced over all waves
constexpr uint32_t BIG_SLICE = 4096;
constexpr uint32_t BIG_LIST_CAP = 1u << 18;
constexpr int BIG_TAB_BITS = 20;
#ifdef MDD_LEAF_PROF
#define LEAF_T(i)                                            \
    {                                                        \
        uint64_t t_now = __builtin_amdgcn_s_memtime();       \
        t_acc[i] += t_now - t_last;                          \
        t_last = t_now;                                      \
    }
#else
#define LEAF_T(i)
#endif

// One WAVE per leaf (leaves are ~500 windows: a workgroup per leaf spends its time in barriers), leaves dealt out
// statically (wave w takes leaves w, w + #waves, ...: a work counter would be 4 M same-address atomics = 46 ms).
// The table entry is [d2 | tag : 28 | lowest position : 36] of one distinct window content; all-ones = empty.
// A leaf costs memory round trips, not bandwidth, so the kernel is built to need few of them:
//   * the leaf's entries are loaded LEAF_CH per lane in one go, and the NEXT leaf's entries are requested before
//     this leaf is worked on;
//   * inserting never touches the text: an entry that meets its tag in the table becomes a CANDIDATE (entry, slot) in an
//     LDS list and leaves the slot unchanged;
//   * candidates are resolved 64 at a time, every lane busy (a leaf of 500 windows has ~80): compare the bytes with the
//     slot's window; equal -> the later window of the pair is marked and the slot keeps the lower position (atomicMin);
//     different (a tag collision: the 17 tag bits that vary inside a leaf meet by chance about once per leaf) -> go on
//     probing, verifying on the way.  A slot's CONTENT never changes once claimed, and every window of one content walks
//     the same probe sequence, so deferring cannot split a group of equal windows over two slots.
constexpr int LEAF_CH = 10;     // entries per lane and chunk: 640 = LEAF_ONE_ROUND, above nearly every leaf's size
constexpr int LEAF_ILP = 1;     // entries a lane inserts side by side (2 measured: 29.3 vs 29.5 ms, no gain)
constexpr int CAND_CAP = 128;   // candidate list entries (whole turns of 64 are drained when a step could overflow it); 256 cost
                                // 1.25 KB more LDS per wave: 14 instead of 16 waves per CU
constexpr unsigned LEAF_WAVES_PER_CU = 16;  // single-wave workgroups resident per CU: 16 x sizeof(LeafLds) = 148 KB of the 160 KB LDS
struct LeafLds {
    unsigned long long tab[LEAF_SLOTS];
    unsigned long long cval[CAND_CAP];
    uint16_t cslot[CAND_CAP];
};
__device__ __forceinline__ uint32_t leaf_home(unsigned long long val) {
    return ((uint32_t)(val >> POS_BITS) * 0x9E3779B1u) >> (32 - 10);
}
static_assert(LEAF_SLOTS == 1024, "leaf_home yields 10 bits");
static_assert(LEAF_WAVES_PER_CU * sizeof(LeafLds) <= 160 * 1024, "the leaf kernels' workgroups must all be resident");

// the full protocol for one window, from `slot` on; true = it claimed an empty slot (a new distinct content)
__device__ __forceinline__ bool leaf_resolve(LeafLds& L, const LeafArgs& A, unsigned long long val, uint32_t slot) {
    const uint64_t pos = val & POS_M;
    for (uint32_t probe = 0; probe < LEAF_SLOTS; probe++) {
        unsigned long long cur = atomicCAS(&L.tab[slot], ENT_INVALID, val);
        if (cur == ENT_INVALID) return true;
        if ((cur >> POS_BITS) == (val >> POS_BITS) && same_window_cp(A.text, pos, cur & POS_M, A.window)) {
            if (cur > val) cur = atomicMin(&L.tab[slot], val);
            A.dup[cur > val ? (cur & POS_M) : pos] = 1;
            return false;
        }
        slot = (slot + 1) & (LEAF_SLOTS - 1);
    }
    atomicAdd(A.overflow, 1ull);
    return false;
}

// one wave's state while it works on a leaf (or on a slice of a big leaf)
struct LeafWave {
    LeafLds& L;
    const LeafArgs& A;
    const uint32_t lane;
    uint32_t n_cand = 0, distinct = 0;  // wave-uniform
    __device__ __forceinline__ LeafWave(LeafLds& l, const LeafArgs& a, uint32_t ln) : L(l), A(a), lane(ln) {}

    __device__ __forceinline__ void load_chunk(unsigned long long (&e)[LEAF_CH], uint64_t lo, uint64_t hi) const {
#pragma unroll
        for (int k = 0; k < LEAF_CH; k++) {
            const uint64_t i = lo + (uint64_t)k * 64 + lane;
            e[k] = i < hi ? A.ent[i] : ENT_INVALID;
        }
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < LEAF_SLOTS / 64; i++) L.tab[i * 64 + lane] = ENT_INVALID;
        lds_sync();
        distinct = 0;
    }
    // all = false (the list is filling up in the middle of a leaf): only whole turns of 64 are resolved, what is left over
    // moves to the front of the list and waits for company
    __device__ __forceinline__ void drain(bool all) {
        lds_sync();
        uint32_t base = 0;
        for (; all ? base < n_cand : base + 64 <= n_cand; base += 64) {
            bool claimed = false;
            if (base + lane < n_cand) claimed = leaf_resolve(L, A, L.cval[base + lane], L.cslot[base + lane]);
            distinct += (uint32_t)__popcll(__ballot(claimed));
        }
        const uint32_t rest = base < n_cand ? n_cand - base : 0u;  // < 64, only when !all
        if (rest && base) {
            unsigned long long v = 0;
            uint16_t sl = 0;
            if (lane < rest) v = L.cval[base + lane], sl = L.cslot[base + lane];
            lds_sync();
            if (lane < rest) L.cval[lane] = v, L.cslot[lane] = sl;
        }
        lds_sync();
        n_cand = rest;
    }
    // (a rolled loop over a rotating register array: unrolled, the body -- with the candidate drain inside -- came to
    // 150 KB of code, more than the instruction cache holds)
    // R > 1: only the entries whose tag falls into share r of R take part (the others wait for their round)
    __device__ __forceinline__ void insert_chunk(unsigned long long (&e)[LEAF_CH], uint32_t R, uint32_t r) {
        static_assert(LEAF_CH % LEAF_ILP == 0, "whole groups of entries per step");
#pragma unroll 1
        for (int k = 0; k < LEAF_CH; k += LEAF_ILP) {
            // LEAF_ILP entries per lane and step: what a step costs is the round trip of an LDS atomic with return (hundreds
            // of cycles) times the length of the dependent probe chain; independent chains overlap their round trips
            unsigned long long val[LEAF_ILP];
#pragma unroll
            for (int u = 0; u < LEAF_ILP; u++) val[u] = e[u];
#pragma unroll
            for (int j = 0; j + LEAF_ILP < LEAF_CH; j++) e[j] = e[j + LEAF_ILP];
#pragma unroll
            for (int u = 0; u < LEAF_ILP; u++) e[LEAF_CH - 1 - u] = ENT_INVALID;
            bool open[LEAF_ILP], claimed[LEAF_ILP], cand[LEAF_ILP];
            uint32_t slot[LEAF_ILP];
            bool any = false;
#pragma unroll
            for (int u = 0; u < LEAF_ILP; u++) {
                if (R > 1u && val[u] != ENT_INVALID && ((((uint32_t)(val[u] >> POS_BITS) & ((1u << TAG_BITS) - 1u)) * R) >> TAG_BITS) != r)
                    val[u] = ENT_INVALID;
                open[u] = val[u] != ENT_INVALID;
                claimed[u] = cand[u] = false;
                slot[u] = leaf_home(val[u]);
                any = any || open[u];
            }
            if (__ballot(any) == 0) continue;
            // four probes straight-line (a loop with two exits costs ~20 scalar mask instructions per turn, and nearly
            // every entry is settled within four), then the general loop for the few lanes still probing
#pragma unroll
            for (int probe = 0; probe < 4; probe++) {
                unsigned long long cur[LEAF_ILP];
#pragma unroll
                for (int u = 0; u < LEAF_ILP; u++)
                    if (open[u]) cur[u] = atomicCAS(&L.tab[slot[u]], ENT_INVALID, val[u]);
#pragma unroll
                for (int u = 0; u < LEAF_ILP; u++)
                    if (open[u]) {
                        claimed[u] = cur[u] == ENT_INVALID;
                        cand[u] = !claimed[u] && (cur[u] >> POS_BITS) == (val[u] >> POS_BITS);
                        open[u] = !claimed[u] && !cand[u];
                        if (open[u]) slot[u] = (slot[u] + 1) & (LEAF_SLOTS - 1);
                    }
            }
#pragma unroll
            for (int u = 0; u < LEAF_ILP; u++) {
                if (open[u]) {
                    for (uint32_t probe = 4; probe < LEAF_SLOTS; probe++) {
                        const unsigned long long c = atomicCAS(&L.tab[slot[u]], ENT_INVALID, val[u]);
                        if (c == ENT_INVALID) {
                            claimed[u] = true;
                            break;
                        }
                        if ((c >> POS_BITS) == (val[u] >> POS_BITS)) {
                            cand[u] = true;
                            break;
                        }
                        slot[u] = (slot[u] + 1) & (LEAF_SLOTS - 1);
                    }
                    if (!claimed[u] && !cand[u]) atomicAdd(A.overflow, 1ull);  // table full
                }
                distinct += (uint32_t)__popcll(__ballot(claimed[u]));
                const uint64_t cb = __ballot(cand[u]);
                if (cand[u]) {
                    const uint32_t at = n_cand + below(cb);
                    L.cval[at] = val[u];
                    L.cslot[at] = (uint16_t)slot[u];
                }
                n_cand += (uint32_t)__popcll(cb);
            }
            if (n_cand > CAND_CAP - 64 * LEAF_ILP) drain(false);
        }
    }
    // share r of R of the entries [lo, hi) into the (cleared) table; `first` holds the first chunk already
    __device__ __forceinline__ void run(uint64_t lo, uint64_t hi, unsigned long long (&first)[LEAF_CH], uint32_t R, uint32_t r) {
        insert_chunk(first, R, r);  // kept apart from the loop below so that the wait for `first` does not cover later requests
        for (uint64_t c0 = lo + 64 * LEAF_CH; c0 < hi && distinct <= LEAF_SLOTS * 3 / 4; c0 += 64 * LEAF_CH) {
            unsigned long long more[LEAF_CH];
            load_chunk(more, c0, hi);
            insert_chunk(more, R, r);
        }
        drain(true);
        // the table is filling up: chains get long and then it is full -- refuse (mdd_substr_check)
        if (distinct > LEAF_SLOTS * 3 / 4 && lane == 0) atomicAdd(A.overflow, 1ull);
    }
    // A range of up to BIG_LEAF entries: more than LEAF_ONE_ROUND of them are grouped in R rounds, round r taking the
    // entries whose tag lies in share r of the tag space (equal windows have equal tags, so they meet in one round);
    // the entries are read R times, from the L2.  after_round(): what the caller does with a round's table.
    template <typename F>
    __device__ __forceinline__ void group(uint64_t lo, uint64_t hi, unsigned long long (&first)[LEAF_CH], F after_round) {
        const uint32_t n = (uint32_t)(hi - lo);
        const uint32_t R = n > A.one_round ? (n + A.per_round - 1) / A.per_round : 1u;
        for (uint32_t r = 0; r < R; r++) {
            clear();
            if (r) load_chunk(first, lo, hi);
            run(lo, hi, first, R, r);
            after_round();
        }
    }
};

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) leaf_kernel(LeafArgs A) {
    __shared__ LeafLds L;
    const uint32_t lane = lane_id();
    const uint64_t n_waves = gridDim.x;
    LeafWave W(L, A, lane);
    const uint32_t mshift = leaf_merge_shift(A);
    A.n_leaves >>= mshift;
    auto leaf_range = [&](uint64_t leaf, uint64_t& lo, uint64_t& hi) {
        lo = hi = 0;
        if (leaf < A.n_leaves) {
            lo = A.off2[leaf << mshift];
            hi = A.off2[(leaf + 1) << mshift];
        }
    };
#ifdef MDD_LEAF_PROF
    uint64_t t_acc[6] = {0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#endif
    uint64_t leaf = blockIdx.x;
    uint64_t lo, hi, nlo, nhi;
    leaf_range(leaf, lo, hi);
    leaf_range(leaf + n_waves, nlo, nhi);
    unsigned long long e[LEAF_CH], nx[LEAF_CH];
    W.load_chunk(e, lo, hi);
    while (leaf < A.n_leaves) {
        // requests for the following leaf (entries) and the one after it (range) go out before this leaf's work
        W.load_chunk(nx, nlo, nhi);
        uint64_t nnlo, nnhi;
        leaf_range(leaf + 2 * n_waves, nnlo, nnhi);
        LEAF_T(0)
        if (hi - lo > BIG_LEAF) {
            // boilerplate / runs of one character: one wave would be busy for milliseconds.  Queue it for bigleaf_kernel.
            if (lane == 0) {
                const unsigned int at = atomicAdd(A.n_big, 1u);
                if (at < BIG_LIST_CAP) A.big_list[at] = (unsigned int)leaf;
                else atomicAdd(A.overflow, 1ull);  // (cannot happen below 2^32 entries: 2^32 / BIG_LEAF < BIG_LIST_CAP)
            }
        } else if (hi - lo >= 2) {  // a leaf with one window has no repeats
            W.group(lo, hi, e, [] {});
            LEAF_T(3)
        }
#pragma unroll
        for (int k = 0; k < LEAF_CH; k++) e[k] = nx[k];
        lo = nlo, hi = nhi, nlo = nnlo, nhi = nnhi;
        leaf += n_waves;
        LEAF_T(4)
    }
#ifdef MDD_LEAF_PROF
    if (lane == 0)
        for (int i = 0; i < 5; i++) atomicAdd(&A.prof[i], (unsigned long long)t_acc[i]);
#endif
}

// The same protocol as leaf_resolve on the global table that joins the slices of big leaves.  Different leaves share
// the table; a slot of another leaf (or another content) simply fails the byte comparison and the probe goes on.
__device__ __forceinline__ void big_resolve(const LeafArgs& A, unsigned long long val, uint32_t leaf) {
    const uint64_t pos = val & POS_M;
    uint32_t slot = (((uint32_t)(val >> POS_BITS) ^ (leaf * 0x85EBCA6Bu)) * 0x9E3779B1u) >> (32 - BIG_TAB_BITS);
    for (uint32_t probe = 0; probe < (1u << BIG_TAB_BITS); probe++) {
        unsigned long long cur = atomicCAS(&A.big_tab[slot], ENT_INVALID, val);
        if (cur == ENT_INVALID) return;
        if ((cur >> POS_BITS) == (val >> POS_BITS) && same_window_cp(A.text, pos, cur & POS_M, A.window)) {
            if (cur > val) cur = atomicMin(&A.big_tab[slot], val);
            A.dup[cur > val ? (cur & POS_M) : pos] = 1;
            return;
        }
        slot = (slot + 1) & ((1u << BIG_TAB_BITS) - 1);
    }
    atomicAdd(A.overflow, 1ull);
}

// Big leaves, all waves on each of them: a wave takes slices of BIG_SLICE entries, groups a slice in its LDS table exactly
// as leaf_kernel groups a leaf (every window but the slice's first of each content is marked), then enters the slice's
// survivors -- one per distinct content, what is left in the LDS table -- into a global table, which marks all but
// the first across slices.  Every wave reads the same list and runs the same loops, so all of them reach the end.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) bigleaf_kernel(LeafArgs A) {
    __shared__ LeafLds L;
    const uint32_t lane = lane_id();
    LeafWave W(L, A, lane);
    unsigned int n_big = *A.n_big;
    if (n_big > BIG_LIST_CAP) n_big = BIG_LIST_CAP;
    const uint32_t mshift = leaf_merge_shift(A);  // the numbering leaf_kernel queued the leaves under
    A.n_leaves >>= mshift;
    for (unsigned int b = 0; b < n_big; b++) {
        const uint32_t leaf = A.big_list[b];
        if (leaf >= A.n_leaves) continue;
        const uint64_t lo = A.off2[(uint64_t)leaf << mshift], hi = A.off2[((uint64_t)leaf + 1) << mshift];
        const uint64_t n_slices = (hi - lo + BIG_SLICE - 1) / BIG_SLICE;
        for (uint64_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
            const uint64_t slo = lo + sl * BIG_SLICE, shi = slo + BIG_SLICE < hi ? slo + BIG_SLICE : hi;
            unsigned long long e[LEAF_CH];
            W.load_chunk(e, slo, shi);
            W.group(slo, shi, e, [&] {
                for (int i = 0; i < LEAF_SLOTS / 64; i++) {
                    const unsigned long long v = L.tab[i * 64 + lane];
                    if (v != ENT_INVALID) big_resolve(A, v, leaf);
                }
                lds_sync();
            });
        }
    }
}


// ---- candidate filter: anchors ------------------------------------------------------------------------------------
constexpr uint32_t AN_KG = 32;                        // bytes of an anchor k-gram
constexpr uint32_t AN_SPAN = 19;                      // k-gram positions a window selects from: its first 19 byte offsets
constexpr uint32_t AN_MINWIN = AN_KG + AN_SPAN - 1;   // = 50: the smallest window (in bytes) the selection fits into
constexpr uint32_t AN_LANES = 50;                     // lanes of a step whose 4 positions get a window minimum: their windows reach
                                                      // position 4 * 49 + 3 + 18 = 217 (lane 54), whose k-gram ends in lane 62's dword
static_assert(AN_MINWIN == 50, "mdd_substr_candidates: the filter is defined for windows of at least 50 code points");
constexpr int AN_D1_BITS = 9;                         // level-1 digit of an anchor: 9 bits up to ~2.2 GB of text, 10 / 11 beyond (a window entry has 11)
constexpr uint32_t AN_C0 = 0x9E3779B1u, AN_C1 = 0x85EBCA77u, AN_C2 = 0xC2B2AE3Du, AN_C3 = 0x27D4EB2Fu;

// lane l reads x of lane l + n (the top n lanes read the bottom ones: their positions are never used)
__device__ __forceinline__ uint32_t lane_up(uint32_t x, uint32_t lane, uint32_t n) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + n) & 63u) << 2), (int)x);
}
// 64-bit key of the 32 bytes whose dwords are w[0..7]: two multiply-with-carry chains over dwords 0-3 and 4-7,
// one v_xor + one v_mad_u64_u32 per PAIR of dwords.  Any function of the BYTES would do for exactness; this one is cheap.
__device__ __forceinline__ uint64_t anchor_key(const uint32_t (&w)[8]) {
    // a' = (lo(a) ^ w_even) * C + (w_odd : hi(a)): the odd dword rides in the high half of the addend and meets a multiplier
    // one step later (the last ones in the final fold)
    uint64_t a = (uint64_t)(w[0] ^ 0x243F6A88u) * AN_C0 + (((uint64_t)w[1] << 32) | 0x03707344u);
    uint64_t b = (uint64_t)(w[4] ^ 0xA4093822u) * AN_C1 + (((uint64_t)w[5] << 32) | 0x082EFA98u);
    a = (uint64_t)((uint32_t)a ^ w[2]) * AN_C2 + (((uint64_t)w[3] << 32) | (a >> 32));
    b = (uint64_t)((uint32_t)b ^ w[6]) * AN_C3 + (((uint64_t)w[7] << 32) | (b >> 32));
    const uint64_t u = a ^ ((b << 32) | (b >> 32));
    return (uint64_t)(uint32_t)u * AN_C0 + (uint64_t)(uint32_t)(u >> 32) * 0x9FB21C651E98DF25ull;
}
// what is sorted: [d2:11 | tag:17 | pos:36] like a window entry; d1 has 9 bits up to ~2.2 GB of text (512 partitions x 2048 = 2^20
// leaves of ~200 anchors: small leaves mean small LDS tables and many resident waves for anchor_resolve_kernel, which waits
// on memory) and 10 or 11 bits beyond (2^21 / 2^22 leaves: the leaves stay small; an_d1_bits).  The ORDER among a window's
// k-grams uses bits 31..8 of the key, so d1 / d2 / tag avoid the top of the low half (anchors are minima: those bits are
// mostly zero).
__device__ __forceinline__ uint32_t anchor_d1(uint64_t key, uint32_t bits) { return (uint32_t)(key >> (64 - bits)); }
__device__ __forceinline__ uint64_t anchor_entry(uint64_t key, uint64_t pos, uint32_t bits) {
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    // 11 + 17 bits below the `bits` of d1: the rest of the high half, then bits 8.. of the low half (bits = 9: 23 + 5)
    const uint64_t d2tag = ((uint64_t)(hi & (0xFFFFFFFFu >> bits)) << (bits - 4u)) | ((lo >> 8) & ((1u << (bits - 4u)) - 1u));
    return (d2tag << POS_BITS) | pos;
}
// level-1 digit width for a text of this size: ~one anchor per 10 bytes, ~200 per leaf
inline uint32_t an_d1_bits(uint64_t text_bytes) {
    uint32_t bits = AN_D1_BITS;
    while (bits < (uint32_t)D1_BITS && text_bytes / 10 > (220ull << (bits + D2_BITS))) bits++;
    if (const char* knob = getenv("MDD_SUBSTR_D1_BITS")) {  // testing: more leaves than the size asks for, or so few (4 bits: 2^15)
        const int b = atoi(knob);                           // that a moderate text fills them beyond RS_MAX
        if (b >= 4 && b <= D1_BITS) bits = (uint32_t)b;
    }
    return bits;
}

struct AnchorArgs {
    const uint8_t* text;
    uint64_t text_bytes;
    const uint64_t* offsets;
    uint64_t n_docs;
    uint32_t seg_bytes;
    const uint64_t* items;
    const uint64_t* n_items;
    uint64_t max_items;
    unsigned long long* work_counter;
    uint64_t* ent;
    uint16_t* d1;
    uint64_t ent_cap;
    unsigned long long* n_ent;
    unsigned int* hist1;
    unsigned long long* overflow;  // anchors that found no room: the filter then lets every window through (smear_kernel)
    uint32_t batch_shift;
    uint32_t d1_bits;              // an_d1_bits(text_bytes)
};

// One wave per document (or segment of a long one), persistent, batched claims -- the skeleton of emit_kernel.  A step
// covers 64 aligned dwords of the text: lane l owns the 4 byte positions of dword l.  It gets the 8 following dwords from
// the lanes above (so a step reads its 256 bytes once), hashes the four 32-byte k-grams that start in its dword, and the
// window minimum over 19 positions follows from in-lane prefix / suffix minima and the whole-dword minima of lanes
// l+1 .. l+5.  The low byte of a packed hash is the position inside the step, so the minimum IS the argmin, leftmost on
// ties.  Window p emits its anchor when it differs from window p-1's (argmin positions only move forward), through a
// flag byte in LDS that the anchor's own lane -- which holds its key -- picks up.  Steps advance by 50 lanes.
__global__ void __launch_bounds__(64 * EMIT_WAVES) anchor_kernel(AnchorArgs A) {
    __shared__ uint32_t flag_all[EMIT_WAVES][64];
    __shared__ unsigned int h1[BINS1];
    uint32_t* flags = flag_all[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    for (int i = threadIdx.x; i < BINS1; i += 64 * EMIT_WAVES) h1[i] = 0;
    __syncthreads();
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(A.text);
    const uint64_t extra = *A.n_items;
    const uint64_t total = A.n_docs + (extra < A.max_items ? extra : A.max_items);
    auto fetch = [&](uint32_t n) -> uint64_t {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(A.work_counter, (unsigned long long)n);
        uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    uint64_t chunk_pos = 0, chunk_end = 0;
    uint64_t item = 0, batch_end = 0;
    for (;; item++) {
        if (item >= batch_end) {
            const uint64_t want = (total > item ? total - item : 0) >> A.batch_shift;
            const uint32_t n = want < 1u ? 1u : (want > 32u ? 32u : (uint32_t)want);
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
        const uint64_t len = end - beg;
        if (len < AN_MINWIN) continue;
        // window starts of this item: absolute byte positions [lo_abs, hi_abs); every window of >= 50 code points that the
        // stage hashes starts at a p with p + 50 < end, so "p + 50 <= end" takes a superset (other code-point starts,
        // continuation bytes): more anchors than needed, never fewer
        uint64_t seg_beg = 0, seg_end = len;
        if (len > A.seg_bytes) {
            seg_beg = (uint64_t)seg * A.seg_bytes;
            seg_end = seg_beg + A.seg_bytes < len ? seg_beg + A.seg_bytes : len;
        }
        const uint64_t lo_abs = beg + seg_beg;
        const uint64_t hi_abs = beg + (seg_end < len - (AN_MINWIN - 1) ? seg_end : len - (AN_MINWIN - 1));
        if (lo_abs >= hi_abs) continue;
        const uint64_t last_dw = (end - 1) >> 2;  // the document's last dword: lanes past it re-read it (their positions are not used)
        uint64_t a_prev = ~0ull;                  // anchor of the previous step's last window (absolute position)
        uint32_t d_next;  // the next step's dword is requested before this step is worked on
        {
            const uint64_t i0 = (lo_abs >> 2) + lane;
            d_next = tw[i0 < last_dw ? i0 : last_dw];
        }
        for (uint64_t base_dw = lo_abs >> 2; 4 * base_dw < hi_abs; base_dw += AN_LANES) {
            const uint64_t my_dw = base_dw + lane;
            uint32_t D[9];
            D[0] = d_next;
            {
                const uint64_t i1 = my_dw + AN_LANES;
                d_next = tw[i1 < last_dw ? i1 : last_dw];
            }
#pragma unroll
            for (uint32_t k = 1; k <= 8; k++) D[k] = lane_up(D[0], lane, k);
            uint32_t keylo[4], keyhi[4], pk[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                uint32_t w[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) w[k] = j ? __builtin_amdgcn_alignbyte(D[k + 1], D[k], j) : D[k];
                const uint64_t key = anchor_key(w);
                keylo[j] = (uint32_t)key;
                keyhi[j] = (uint32_t)(key >> 32);
                pk[j] = (keylo[j] & 0xFFFFFF00u) | (4u * lane + j);
            }
            // suffix minima s_j = min(pk[j..3]), prefix minima p_j = min(pk[0..j])
            const uint32_t s3 = pk[3], s2 = pk[2] < s3 ? pk[2] : s3, s1 = pk[1] < s2 ? pk[1] : s2, s0 = pk[0] < s1 ? pk[0] : s1;
            const uint32_t p1 = pk[0] < pk[1] ? pk[0] : pk[1], p2 = p1 < pk[2] ? p1 : pk[2];
            const uint32_t M1 = lane_up(s0, lane, 1), M2 = lane_up(s0, lane, 2), M3 = lane_up(s0, lane, 3), M4 = lane_up(s0, lane, 4);
            const uint32_t P2_4 = lane_up(p2, lane, 4), P0_5 = lane_up(pk[0], lane, 5), P1_5 = lane_up(p1, lane, 5);
            uint32_t Q = M1 < M2 ? M1 : M2;
            Q = Q < M3 ? Q : M3;
            const uint32_t QQ = Q < M4 ? Q : M4;
            // window of position 4l + j: positions 4l + j .. 4l + j + 18
            uint32_t win[4];
            win[0] = s0 < Q ? s0 : Q;
            win[0] = win[0] < P2_4 ? win[0] : P2_4;  // l+4: positions 0..2
            win[1] = s1 < QQ ? s1 : QQ;              // l+4: positions 0..3
            win[2] = s2 < QQ ? s2 : QQ;
            win[2] = win[2] < P0_5 ? win[2] : P0_5;  // l+5: position 0
            win[3] = s3 < QQ ? s3 : QQ;
            win[3] = win[3] < P1_5 ? win[3] : P1_5;  // l+5: positions 0..1
            uint32_t code[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) code[j] = win[j] & 0xFFu;
            // the anchor of the window before this lane's first one
            const uint32_t carry_code = a_prev == ~0ull ? 0xFFFFFFFFu : (uint32_t)(a_prev - 4 * base_dw);  // (a_prev >= 4 * base_dw - 1 ... any mismatch is harmless)
            uint32_t prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 63u) & 63u) << 2), (int)code[3]);
            if (lane == 0) prev = carry_code;
            flags[lane] = 0;
            lds_sync();
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint64_t p_abs = 4 * my_dw + j;
                const bool valid = lane < AN_LANES && p_abs >= lo_abs && p_abs < hi_abs;
                if (valid && (p_abs == lo_abs || code[j] != prev)) reinterpret_cast<uint8_t*>(flags)[code[j]] = 1;
                prev = code[j];
            }
            lds_sync();
            const uint32_t F = flags[lane];  // byte j set: position 4 * lane + j is a new anchor
            lds_sync();
            a_prev = 4 * base_dw + __builtin_amdgcn_readlane(code[3], AN_LANES - 1);
            const uint32_t cnt = (uint32_t)__builtin_popcount(F);
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            const uint32_t tot = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
            if (tot == 0) continue;
            if (chunk_pos + tot > chunk_end) {  // void the rest of the chunk (< 256 entries) and take a new one
                for (uint64_t i = chunk_pos + lane; i < chunk_end && i < A.ent_cap; i += 64) {
                    A.ent[i] = ENT_INVALID;
                    A.d1[i] = D1_INVALID;
                }
                unsigned long long v = 0;
                if (lane == 0) v = atomicAdd(A.n_ent, (unsigned long long)EMIT_CHUNK);
                uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
                chunk_pos = ((uint64_t)hi << 32) | lo;
                chunk_end = chunk_pos + EMIT_CHUNK;
            }
            uint64_t o = chunk_pos + below(b0) + 2u * below(b1) + 4u * below(b2);
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (F & (0xFFu << (8 * j))) {
                    const uint64_t key = ((uint64_t)keyhi[j] << 32) | keylo[j];
                    if (o < A.ent_cap) {
                        A.ent[o] = anchor_entry(key, 4 * my_dw + j, A.d1_bits);
                        A.d1[o] = (uint16_t)anchor_d1(key, A.d1_bits);
                        atomicAdd(&h1[anchor_d1(key, A.d1_bits)], 1u);
                    } else {
                        atomicAdd(A.overflow, 1ull);
                    }
                    o++;
                }
            chunk_pos += tot;
        }
    }
    for (uint64_t i = chunk_pos + lane; i < chunk_end && i < A.ent_cap; i += 64) {
        A.ent[i] = ENT_INVALID;
        A.d1[i] = D1_INVALID;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS1; i += 64 * EMIT_WAVES)
        if (h1[i]) atomicAdd(&A.hist1[i], h1[i]);
}

// the anchor flags are a bitmap (a bit per text position; cleared by the caller): 272 MB for C2 instead of 2.2 GB to clear,
// to scatter into and to stream through smear_kernel
__device__ __forceinline__ void set_flag(uint32_t* flag, uint64_t pos) { atomicOr(&flag[pos >> 5], 1u << (pos & 31u)); }

// One wave per leaf of anchors: flag every anchor whose (d2, tag) occurs twice in the leaf.  Key equality only -- a false
// match lets a few windows through, nothing else -- so a leaf too big for the table is flagged whole.
constexpr uint32_t ALEAF_CH = 12;  // 768 anchors per leaf at most go through the table (1024 slots)
__global__ void __launch_bounds__(64) anchor_leaf_kernel(const uint64_t* __restrict__ ent, const uint64_t* __restrict__ off2, uint32_t n_leaves,
                                                         uint32_t* flag, uint64_t text_bytes) {
    __shared__ unsigned long long tab[LEAF_SLOTS];
    const uint32_t lane = lane_id();
    for (uint32_t leaf = blockIdx.x; leaf < n_leaves; leaf += gridDim.x) {
        const uint64_t lo = off2[leaf], hi = off2[leaf + 1];
        if (hi - lo < 2) continue;
        if (hi - lo > 64 * ALEAF_CH) {
            for (uint64_t i = lo + lane; i < hi; i += 64) {
                const uint64_t pos = ent[i] & POS_M;
                if (pos < text_bytes) set_flag(flag, pos);
            }
            continue;
        }
        unsigned long long e[ALEAF_CH];
#pragma unroll
        for (uint32_t k = 0; k < ALEAF_CH; k++) {
            const uint64_t i = lo + (uint64_t)k * 64 + lane;
            e[k] = i < hi ? ent[i] : ENT_INVALID;
        }
#pragma unroll
        for (int i = 0; i < LEAF_SLOTS / 64; i++) tab[i * 64 + lane] = ENT_INVALID;
        lds_sync();
#pragma unroll 1
        for (uint32_t k = 0; k < ALEAF_CH; k++) {
            const unsigned long long val = e[0];
#pragma unroll
            for (uint32_t j = 0; j + 1 < ALEAF_CH; j++) e[j] = e[j + 1];
            e[ALEAF_CH - 1] = ENT_INVALID;
            if (val == ENT_INVALID) continue;
            uint32_t slot = leaf_home(val);
            for (uint32_t probe = 0; probe < LEAF_SLOTS; probe++) {
                const unsigned long long cur = atomicCAS(&tab[slot], ENT_INVALID, val);
                if (cur == ENT_INVALID) break;
                if ((cur >> POS_BITS) == (val >> POS_BITS)) {
                    const uint64_t p = val & POS_M, q = cur & POS_M;
                    if (p < text_bytes) set_flag(flag, p);
                    if (q < text_bytes) set_flag(flag, q);
                    break;
                }
                slot = (slot + 1) & (LEAF_SLOTS - 1);
            }
        }
        lds_sync();
    }
}

// ---- resolving repeated anchors on the spot (window == 50) ----------------------------------------------------------
// vmap bit p = 1 when a window of `window` code points that the stage hashes starts at byte p: p is a code-point start of its
// document and more than `window` code points start at or after it (SUB:38 range(len(text) - duplicate_len)).  One wave per
// document: count the code points (and see whether there is a byte >= 0x80 at all), then write -- pure-ASCII documents are
// not read a second time.  A BITMAP (bit p & 31 of word p >> 5; cleared by the caller): an eighth of the bytes to write, and
// what anchor_resolve_kernel reads of it -- 19 bits per region, scattered -- stays in the caches (272 MB for C2 instead of
// 2.2 GB).  A lane holds the 4 bits of its dword; the lanes of one word combine theirs (three shuffles) and the first of
// them stores the word, with an atomic OR where the word is shared with another document or another step.
__global__ void __launch_bounds__(256) vmap_kernel(const uint8_t* __restrict__ text, uint64_t text_bytes, const uint64_t* __restrict__ offs,
                                                   uint64_t n, uint32_t window, uint32_t* __restrict__ vmap) {
    const uint32_t lane = lane_id();
    const uint64_t lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t* __restrict__ tw = reinterpret_cast<const uint32_t*>(text);
    for (uint64_t d = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n; d += nw) {
        uint64_t b, e;
        doc_range(offs, d, text_bytes, b, e);
        if (b >= e) continue;
        const uint64_t w0 = b >> 2, w1 = (e + 3) >> 2;
        auto inside = [&](uint64_t i) -> uint32_t {  // bytes of dword i inside [b, e), as bit 7 of each byte
            const uint64_t p = i << 2;
            uint32_t in = 0x80808080u;
            if (p < b || p + 4 > e) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (p + k < b || p + k >= e) in &= ~(0x80u << (8 * k));
            }
            return in;
        };
        // every lane of the wave calls it (i = first dword of the step + lane); v: bit k = byte k of dword i starts a hashed window
        auto put = [&](uint64_t i, bool live, uint32_t in, uint32_t v) {
            uint32_t val = live ? (v & 0xFu) << (4u * (uint32_t)(i & 7u)) : 0u;
            bool whole = live && in == 0x80808080u;  // the lane's dword lies inside the document
            const uint64_t word = i >> 3;
#pragma unroll
            for (uint32_t sft = 1; sft <= 4; sft <<= 1) {  // OR over the lanes of the same word (consecutive lanes)
                const uint32_t ov = (uint32_t)__shfl_down((int)val, sft, 64);
                const int ow = __shfl_down((int)whole, sft, 64);
                if (lane + sft < 64u && ((i + sft) >> 3) == word) val |= ov, whole = whole && ow;
            }
            const bool first = live && (lane == 0u || (i & 7u) == 0u);  // the word's first lane in this step
            if (first) {
                // all 8 dwords of the word in this step and inside the document: nobody else writes the word
                if ((i & 7u) == 0u && lane + 7u < 64u && whole) vmap[word] = val;
                else if (val) atomicOr(&vmap[word], val);
            }
        };
        uint32_t cnt = 0, hi_bits = 0;
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
                if (i < w1) {
                    const uint32_t in = inside(i), w = ww[j];
                    cnt += (uint32_t)__builtin_popcount(in & ~(w & ~(w << 1)));  // not a continuation byte 10xxxxxx
                    hi_bits |= w & in;
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
        const bool ascii = __ballot((hi_bits & 0x80808080u) != 0u) == 0ull;
        const uint64_t ncp = cnt;  // code points of the document; window c (0-based code point index) is hashed iff c + window < ncp
        if (ascii) {
            const uint64_t nv = ncp > window ? ncp - window : 0;  // valid starts: bytes b .. b + nv - 1
            // the valid starts are the bytes [b, b + nv): a lane writes a whole word of the map (32 positions) from the range alone
            const uint64_t v_end = b + nv;
            for (uint64_t wj = (b >> 5) + lane; wj <= ((e - 1) >> 5); wj += 64) {
                const uint64_t lo = wj << 5;
                uint32_t m = 0;
                if (v_end > lo && b < lo + 32) {
                    const uint32_t s0 = b > lo ? (uint32_t)(b - lo) : 0u;
                    const uint32_t t0 = v_end < lo + 32 ? (uint32_t)(v_end - lo) : 32u;  // exclusive
                    if (t0 > s0) m = (t0 - s0 == 32u) ? 0xFFFFFFFFu : (((1u << (t0 - s0)) - 1u) << s0);
                }
                if (m) {
                    if (lo >= b && lo + 32 <= e) vmap[wj] = m;  // the word's 32 positions all belong to this document
                    else atomicOr(&vmap[wj], m);
                }
            }
        } else {
            uint64_t done = 0;  // code points that start before this step
            for (uint64_t wi = w0; wi < w1; wi += 64) {
                const uint64_t i = wi + lane;
                const uint32_t in = i < w1 ? inside(i) : 0u;
                const uint32_t w = i < w1 ? tw[i] : 0u;
                const uint32_t S = in & ~(w & ~(w << 1));  // code-point starts
                const uint32_t c = (uint32_t)__builtin_popcount(S);
                const uint64_t B0 = __ballot(c & 1u), B1 = __ballot(c & 2u), B2 = __ballot(c & 4u);
                uint64_t idx = done + __builtin_popcountll(B0 & lt) + 2 * __builtin_popcountll(B1 & lt) + 4 * __builtin_popcountll(B2 & lt);
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (S & (0x80u << (8 * k))) {
                        if (idx + window < ncp) v |= 1u << k;
                        idx++;
                    }
                put(i, i < w1, in, v);
                done += __builtin_popcountll(B0) + 2 * __builtin_popcountll(B1) + 4 * __builtin_popcountll(B2);
            }
        }
    }
}

// One wave per leaf of anchors, window = 50.  Anchors with equal keys form a group (an LDS table slot keeps the group's
// LOWEST anchor, a chain through the leaf's entries lists its members).  For a member a and an earlier member b the 68 bytes
// [a - 18, a + 50) -- every window that has the k-gram at a among its first 19 positions -- are compared with the bytes
// around b: with l matching bytes before the k-gram and r behind it, the windows a - 18 + u, u in [18 - l, r], equal the
// windows b - 18 + u; those that are hashed windows on both sides (vmap) are repeats: dup = 1.  A member compares with the
// group's lowest anchor first (in a cluster of near-copies that is the original: one comparison explains all 19 windows) and
// walks the chain only while a hashed window of its own is still unexplained.  Every identical earlier window has its own
// selected k-gram in the same group, so after the walk a window is marked iff the stage marks it.  What this pass cannot
// vouch for -- a byte >= 0x80 in the region (windows are then longer than 50 bytes), a group above RS_GROUP members, a leaf
// too big for the table, an anchor in the first 20 bytes of the text -- is flagged, WITH ITS WHOLE GROUP, for the window
// sort (mdd_substr_mark_sorted): both copies of a repeated window are then sorted, as the filter alone would have it.
constexpr uint32_t RS_SLOTS = 512;   // LDS table slots per leaf
constexpr uint32_t RS_MAX = 384;     // anchors per leaf that go through the table (a leaf holds ~200)
constexpr uint32_t RS_CH = RS_MAX / 64;
constexpr uint32_t RS_GROUP = 32;    // a group (anchors of one key) above this many members is handed to the window sort
constexpr uint32_t RS_LIST = 128;    // work list: members with an earlier anchor in their group
constexpr uint32_t RS_NIL = 511u;    // end of a chain (member indices stay below RS_MAX)
static_assert(RS_MAX <= RS_NIL, "member index and NIL share 9 bits");
__device__ __forceinline__ uint32_t rs_home(unsigned long long val) { return ((uint32_t)(val >> POS_BITS) * 0x9E3779B1u) >> (32 - 9); }
static_assert(RS_SLOTS == 512, "rs_home yields 9 bits");
// BIG = the variant for leaves above RS_MAX anchors (a text beyond ~9 GB per call: even 2^22 leaves then hold more than the
// table takes; or heavy repetition): such a leaf is grouped in rounds over shares of the tag space -- equal keys have equal
// tags and meet in one round -- and a member is numbered inside its round, midx[] mapping the number back to the leaf.
template <bool BIG>
struct ResolveLds {
    unsigned long long tab[RS_SLOTS];  // [d2 | tag | position] of the group's lowest anchor; all-ones = empty
    uint32_t head[RS_SLOTS];           // [members : 23 | index (inside the leaf / round) of the most recently chained member : 9]
    uint16_t next[RS_MAX];
    uint32_t fb[RS_SLOTS / 32];        // groups handed to the window sort
    unsigned long long lval[RS_LIST];
    uint16_t lslot[RS_LIST];
    uint16_t lidx[RS_LIST];              // index of the member inside the leaf / round
    uint32_t clone[RS_MAX / 32];         // members that add nothing to their group's lowest anchor (see drain)
    uint16_t midx[BIG ? RS_MAX : 1];     // BIG: offset inside the leaf of the round's member
};
constexpr unsigned RS_WAVES_PER_CU = 18, RS_WAVES_PER_CU_BIG = 16;
constexpr uint32_t RS_ROUND = 320;       // BIG: anchors per round, on average (their number is Poisson: 384 is 3.6 sigma away; a
                                         // share that overflows goes to the window sort -- rare, and fewer rounds are worth more)
constexpr uint32_t RS_BIG_MAX = 0xFFFFu; // BIG: a leaf above this many anchors is flagged whole (midx is 16 bits wide)
static_assert(RS_WAVES_PER_CU * sizeof(ResolveLds<false>) <= 160 * 1024 && RS_WAVES_PER_CU_BIG * sizeof(ResolveLds<true>) <= 160 * 1024,
              "the resolve kernel's workgroups must all be resident");
struct ResolveArgs {
    const uint64_t* ent;
    const uint64_t* off2;
    uint32_t n_leaves;
    const uint8_t* text;
    uint64_t text_bytes;
    const uint32_t* vmap;  // bitmap of hashed window starts (vmap_kernel)
    uint8_t* dup;
    uint32_t* flag;      // a bit per text position: the anchor there goes to the window sort (set_flag)
    uint32_t max_group;  // RS_GROUP (a tuning knob lowers / raises it: MDD_SUBSTR_GROUP)
    uint32_t big_hi;     // a leaf of RS_MAX + 1 .. big_hi anchors is settled in rounds (anchor_resolve_kernel<true>), a larger one
                         // is flagged whole.  RS_MAX when the leaves are expected to hold ~200 (a leaf above that is heavy repetition,
                         // whose groups exceed max_group anyway: rounds over it were 3.5 ms of nothing on C2); a few times the
                         // expected size when the text outgrows 2^22 leaves
};
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// bit u of the result: bit p + u of the bitmap, u < 19
__device__ __forceinline__ uint32_t vmask19(const uint32_t* vmap, uint64_t p) {
    const u32x2_a4 v = *reinterpret_cast<const u32x2_a4*>(vmap + (p >> 5));
    const uint64_t x = ((uint64_t)v.y << 32) | v.x;
    return (uint32_t)(x >> (p & 31u)) & 0x7FFFFu;
}

// region of an anchor at a: the 72 bytes from a - 20 as 18 dwords
__device__ __forceinline__ void load_region(const uint8_t* text, uint64_t a, uint32_t (&R)[18]) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(text) + ((a - 20) >> 2);
    const uint32_t sh = (uint32_t)(a - 20) & 3u;
    uint32_t r[20];
#pragma unroll
    for (int k = 0; k < 20; k += 4) {
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(q + k);
        r[k] = v.x, r[k + 1] = v.y, r[k + 2] = v.z, r[k + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 18; k++) R[k] = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh);
}
// windows (bit u = the window at a - 18 + u) whose 50 bytes equal the window at the same offset from b
__device__ __forceinline__ uint32_t region_match(const uint32_t (&Ra)[18], const uint32_t (&Rb)[18]) {
    uint32_t x[18], kg = 0;
#pragma unroll
    for (int k = 0; k < 18; k++) x[k] = Ra[k] ^ Rb[k];
#pragma unroll
    for (int k = 5; k < 13; k++) kg |= x[k];  // the k-gram itself: region bytes 20 .. 51
    if (kg) return 0u;
    uint32_t l = 0, r = 0;
    bool open = true;
#pragma unroll
    for (int k = 4; k >= 0; k--)  // bytes 19 down to 0
        if (open) {
            if (x[k] == 0u) l += 4;
            else l += (uint32_t)__builtin_clz(x[k]) >> 3, open = false;
        }
    open = true;
#pragma unroll
    for (int k = 13; k < 18; k++)  // bytes 52 up to 71
        if (open) {
            if (x[k] == 0u) r += 4;
            else r += (uint32_t)__builtin_ctz(x[k]) >> 3, open = false;
        }
    l = l < 18u ? l : 18u;
    r = r < 18u ? r : 18u;
    if (18u - l > r) return 0u;
    return (0x7FFFFu >> (18u - r)) & (0x7FFFFu << (18u - l)) & 0x7FFFFu;  // bits 18 - l .. r
}

template <bool BIG>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 5))) anchor_resolve_kernel(ResolveArgs A) {
    __shared__ ResolveLds<BIG> L;
    const uint32_t lane = lane_id();
    auto find_slot = [&](unsigned long long val) -> uint32_t {  // the slot of val's group (val has been inserted)
        uint32_t slot = rs_home(val);
        for (uint32_t probe = 0; probe < RS_SLOTS && (L.tab[slot] >> POS_BITS) != (val >> POS_BITS); probe++) slot = (slot + 1) & (RS_SLOTS - 1);
        return slot;
    };
    for (uint32_t leaf = blockIdx.x; leaf < A.n_leaves; leaf += gridDim.x) {
        const uint64_t lo = A.off2[leaf], hi = A.off2[leaf + 1];
        if (hi - lo < 2) continue;
        if (BIG && (hi - lo <= RS_MAX || hi - lo > A.big_hi)) continue;  // (the other instance of this kernel takes the leaf)
        if (!BIG && hi - lo > RS_MAX) {
            if (hi - lo > A.big_hi)                // heavy repetition: the window sort's business
                for (uint64_t i = lo + lane; i < hi; i += 64) {
                    const uint64_t pos = A.ent[i] & POS_M;
                    if (pos < A.text_bytes) set_flag(A.flag, pos);
                }
            continue;
        }
        // BIG: round r takes the anchors whose tag lies in share r of the tag space
        const uint32_t rounds = BIG ? ((uint32_t)(hi - lo) + RS_ROUND - 1) / RS_ROUND : 1u;
        auto mine = [&](unsigned long long val, uint32_t r) -> bool {
            return !BIG || ((((uint32_t)(val >> POS_BITS) & 0xFFFFu) * rounds) >> 16) == r;
        };
      for (uint32_t round = 0; round < rounds; round++) {
#pragma unroll
        for (int i = 0; i < RS_SLOTS / 64; i++) {
            L.tab[i * 64 + lane] = ENT_INVALID;
            L.head[i * 64 + lane] = RS_NIL;  // no members, empty chain
        }
        if (lane < RS_SLOTS / 32) L.fb[lane] = 0;
        if (lane < RS_MAX / 32) L.clone[lane] = 0;
        lds_sync();
        // ---- 1. groups: the slot keeps the lowest anchor of its key, the chain every member
        auto insert = [&](unsigned long long val, uint32_t idx) {
            uint32_t slot = rs_home(val);
            for (uint32_t probe = 0; probe < RS_SLOTS; probe++) {
                const unsigned long long cur = atomicCAS(&L.tab[slot], ENT_INVALID, val);
                if (cur == ENT_INVALID) break;
                if ((cur >> POS_BITS) == (val >> POS_BITS)) {
                    if (val < cur) atomicMin(&L.tab[slot], val);
                    break;
                }
                slot = (slot + 1) & (RS_SLOTS - 1);
            }
            // chain this member in: head = [members so far : 23 | most recent member : 9], updated as one word
            uint32_t old = L.head[slot];
            for (;;) {
                const uint32_t seen = atomicCAS(&L.head[slot], old, (((old >> 9) + 1u) << 9) | idx);
                if (seen == old) break;
                old = seen;
            }
            L.next[idx] = (uint16_t)(old & RS_NIL);
        };
        uint32_t n_mem = 0;  // BIG: members of the round
        if constexpr (!BIG) {
            unsigned long long e[RS_CH];
#pragma unroll
            for (uint32_t k = 0; k < RS_CH; k++) {
                const uint64_t i = lo + (uint64_t)k * 64 + lane;
                e[k] = i < hi ? A.ent[i] : ENT_INVALID;
            }
#pragma unroll 1
            for (uint32_t k = 0; k < RS_CH; k++) {
                const unsigned long long val = e[0];
#pragma unroll
                for (uint32_t j = 0; j + 1 < RS_CH; j++) e[j] = e[j + 1];
                e[RS_CH - 1] = ENT_INVALID;
                if (val == ENT_INVALID) continue;
                insert(val, k * 64 + lane);
            }
        } else {
            // the round's anchors are a share of the leaf: they are queued (lval / lidx are free until step 2) and inserted 64
            // at a time, so the probe loops run with every lane busy (inserting straight from the leaf's chunks left half of
            // them idle: 276 ms for the 2.66 G anchors of a C4-sized shard)
            uint32_t n_q = 0;
            auto flush = [&](uint32_t cnt) {  // the first cnt <= 64 queued anchors become members n_mem ..
                lds_sync();
                const uint32_t idx = n_mem + lane;
                if (lane < cnt && idx < RS_MAX) {
                    L.midx[idx] = L.lidx[lane];
                    insert(L.lval[lane], idx);
                }
                n_mem += cnt;
            };
            for (uint64_t c0 = lo; c0 < hi; c0 += 64) {
                const uint64_t i = c0 + lane;
                const unsigned long long val = i < hi ? A.ent[i] : ENT_INVALID;
                const bool take = val != ENT_INVALID && mine(val, round);
                const uint64_t tb = __ballot(take);
                if (take) {
                    const uint32_t at = n_q + below(tb);
                    L.lval[at] = val;
                    L.lidx[at] = (uint16_t)(i - lo);
                }
                n_q += (uint32_t)__popcll(tb);
                if (n_q >= 64) {
                    flush(64);
                    const uint32_t rest = n_q - 64;
                    unsigned long long v = 0;
                    uint16_t ix = 0;
                    if (lane < rest) v = L.lval[64 + lane], ix = L.lidx[64 + lane];
                    lds_sync();
                    if (lane < rest) L.lval[lane] = v, L.lidx[lane] = ix;
                    n_q = rest;
                }
            }
            if (n_q) flush(n_q);
            if (n_mem > RS_MAX) {  // more than the table takes in one share (it holds 320 on average): the window sort's business
                for (uint64_t i = lo + lane; i < hi; i += 64) {
                    const unsigned long long val = A.ent[i];
                    const uint64_t pos = val & POS_M;
                    if (val != ENT_INVALID && mine(val, round) && pos < A.text_bytes) set_flag(A.flag, pos);
                }
                lds_sync();
                continue;
            }
        }
        lds_sync();
        // ---- 2. members with an earlier anchor in their group, 64 at a time with every lane busy
        uint32_t n_list = 0;
        auto drain = [&](bool all) {
            lds_sync();
            uint32_t base = 0;
            for (; all ? base < n_list : base + 64 <= n_list; base += 64) {
                // ---- A: this member against the group's lowest anchor
                const bool active = base + lane < n_list;
                unsigned long long val = 0;
                uint32_t slot = 0, idx = 0;
                if (active) val = L.lval[base + lane], slot = L.lslot[base + lane], idx = L.lidx[base + lane];
                const uint64_t a = val & POS_M;
                const uint64_t b = active ? (L.tab[slot] & POS_M) : 0;  // (b < a: a member of the work list is not its group's lowest)
                // a group above max_group members is the window sort's business as a whole (one member with an unexplained window
                // hands it over anyway, and in a group that large one always has): nothing is loaded for it
                const uint32_t members = active ? L.head[slot] >> 9 : 0u;
                bool vouch = active && members <= A.max_group && a >= 20 && b >= 20 && a + 52 <= A.text_bytes + MDD_TEXT_PAD;
                uint32_t Ra[18], left = 0, marks = 0;
                if (vouch) {
                    load_region(A.text, a, Ra);
                    const uint32_t va = vmask19(A.vmap, a - 18);
                    uint32_t hi_or = 0;
#pragma unroll
                    for (int k = 0; k < 18; k++) hi_or |= Ra[k];
                    vouch = (hi_or & 0x80808080u) == 0u;
                    left = va;
                    if (vouch && left) {
                        uint32_t Rb[18];
                        load_region(A.text, b, Rb);
                        const uint32_t vb = vmask19(A.vmap, b - 18);
                        const uint32_t rm = region_match(Ra, Rb);
                        marks = rm & vb & va;
                        left = va & ~marks;
                        // a member whose 68 bytes equal the lowest anchor's, with no hashed window the lowest lacks, can explain
                        // nothing the lowest cannot: the walks below skip it
                        if (rm == 0x7FFFFu && (va & ~vb) == 0u) atomicOr(&L.clone[idx >> 5], 1u << (idx & 31u));
                    }
                }
                lds_sync();  // (the clone bits of this turn's members are in place)
                // ---- B: what the lowest anchor does not explain is tried on the other earlier members
                if (vouch && left) {
                    const uint32_t va = left | marks;
                    for (uint32_t i = L.head[slot] & RS_NIL; left && i != RS_NIL; i = L.next[i]) {
                        if ((L.clone[i >> 5] >> (i & 31u)) & 1u) continue;
                        const uint64_t b2 = A.ent[lo + (BIG ? (uint32_t)L.midx[i] : i)] & POS_M;
                        if (b2 >= a || b2 == b) continue;
                        if (b2 < 20) {
                            vouch = false;
                            break;
                        }
                        uint32_t Rb[18];
                        load_region(A.text, b2, Rb);
                        const uint32_t m = region_match(Ra, Rb) & vmask19(A.vmap, b2 - 18) & va;
                        marks |= m;
                        left &= ~m;
                    }
                }
                // verified repeats are repeats whatever becomes of the group
                if (marks) {  // whole dwords where all four bytes are marked (the marks of a run are), bytes elsewhere:
                              // other waves only ever write ones too
                    const uint64_t p0 = a - 18;
                    const uint32_t sh = (uint32_t)p0 & 3u;
                    const uint32_t m22 = marks << sh;  // bit 4 k + j: byte j of dword (p0 >> 2) + k
                    uint32_t* dw = reinterpret_cast<uint32_t*>(A.dup) + (p0 >> 2);
#pragma unroll
                    for (uint32_t k = 0; k < 6; k++) {
                        const uint32_t nib = (m22 >> (4 * k)) & 0xFu;
                        if (nib == 0xFu && ((p0 >> 2) + k + 1) * 4 <= A.text_bytes) {
                            dw[k] = 0x01010101u;
                        } else if (nib) {
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++)
                                if ((nib >> j) & 1u) A.dup[(((p0 >> 2) + k) << 2) + j] = 1;
                        }
                    }
                }
                if (active && !vouch) atomicOr(&L.fb[slot >> 5], 1u << (slot & 31u));
            }
            const uint32_t rest = base < n_list ? n_list - base : 0u;
            if (rest && base) {
                unsigned long long v = 0;
                uint16_t sl = 0, ix = 0;
                if (lane < rest) v = L.lval[base + lane], sl = L.lslot[base + lane], ix = L.lidx[base + lane];
                lds_sync();
                if (lane < rest) L.lval[lane] = v, L.lslot[lane] = sl, L.lidx[lane] = ix;
            }
            lds_sync();
            n_list = rest;
        };
        // (BIG: the round's members one after the other, through midx -- every lane has one)
        const uint64_t n_walk = BIG ? (uint64_t)n_mem : hi - lo;
        for (uint64_t c0 = 0; c0 < n_walk; c0 += 64) {
            const uint64_t i = c0 + lane;
            bool work = false;
            unsigned long long val = ENT_INVALID;
            uint32_t slot = 0;
            if (i < n_walk) {
                val = A.ent[lo + (BIG ? (uint64_t)L.midx[i] : i)];  // (second read: from the L2)
                slot = find_slot(val);
                work = L.tab[slot] != val;  // not the group's lowest anchor (a lone anchor is its group's lowest)
            }
            const uint64_t wb = __ballot(work);
            const uint32_t member = (uint32_t)i;
            if (work) {
                const uint32_t at = n_list + below(wb);
                L.lval[at] = val;
                L.lslot[at] = (uint16_t)slot;
                L.lidx[at] = (uint16_t)member;
            }
            n_list += (uint32_t)__popcll(wb);
            if (n_list > RS_LIST - 64) drain(false);
        }
        drain(true);
        // ---- 3. groups this pass cannot vouch for: every member is flagged for the window sort
        bool any_fb = false;
        if (lane < RS_SLOTS / 32) any_fb = L.fb[lane] != 0u;
        if (__ballot(any_fb)) {
            for (uint64_t i = lane; i < n_walk; i += 64) {
                const unsigned long long val = A.ent[lo + (BIG ? (uint64_t)L.midx[i] : i)];
                const uint32_t slot = find_slot(val);
                const uint64_t pos = val & POS_M;
                if (((L.fb[slot >> 5] >> (slot & 31u)) & 1u) && pos < A.text_bytes) set_flag(A.flag, pos);
            }
        }
        lds_sync();
      }  // rounds
    }
}

// cand bit p = 1 when an anchor in [p, p + 18] is flagged (window p selects one of these positions; flagging all of them is
// a superset).  Both are bitmaps: a thread makes 64 positions of cand from 83 bits of flag (OR over 19 shifts by doubling:
// 1, 2, 4, 8 give 16, one more by 3 gives 19); flag has two readable zero words past the last position.  *overflow != 0
// (anchors without room in the entry array): every position is a candidate.  *n_cand += candidate positions.
__global__ void __launch_bounds__(256) smear_kernel(const uint64_t* __restrict__ flag, uint64_t text_bytes, uint64_t* __restrict__ cand,
                                                    unsigned long long* n_cand, const unsigned long long* overflow) {
    __shared__ unsigned int wsum[4];
    const bool all = *overflow != 0;
    uint32_t c = 0;
    const uint64_t n_words = (text_bytes + 63) >> 6;
    // (grid-stride: one atomic per workgroup at the end -- same-address atomics retire at ~11 ns each)
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * 256) {
        uint64_t out = ~0ull;
        if (!all) {
            uint64_t lo = flag[w], hi = flag[w + 1];
            out = 0;
            if (lo | hi) {
#pragma unroll
                for (int k = 1; k <= 8; k <<= 1) {  // X |= X >> k on the 128-bit value hi:lo
                    lo |= (lo >> k) | (hi << (64 - k));
                    hi |= hi >> k;
                }
                out = lo | (lo >> 3) | (hi << 61);
            }
        }
        const uint64_t q = w << 6;
        if (q + 64 > text_bytes) out &= ~0ull >> (64 - (text_bytes - q));  // (text_bytes - q is 1 .. 63 here)
        cand[w] = out;
        c += (uint32_t)__builtin_popcountll(out);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (wsum[0] | wsum[1] | wsum[2] | wsum[3]))
        atomicAdd(n_cand, (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// The caller's workspace `ws` of the sorted method, carved (u64 words): off1[BINS1+1] | cursor1[BINS1] | off2[n_leaves+1] |
// cursor2[n_leaves] | hist1 (u32 x BINS1) | cnt2 (u32 x n_leaves) | counters[8] | scan scratch[1024] | tile_off (u32 x (BINS1+2)) |
// big_list (u32 x BIG_LIST_CAP) | big_tab[2^BIG_TAB_BITS]
struct SortWs {
    static constexpr uint64_t NL = (uint64_t)BINS1 * BINS2;
    static constexpr uint64_t WORDS =
        (BINS1 + 1) + BINS1 + (NL + 1) + NL + BINS1 / 2 + NL / 2 + 8 + 1024 + (BINS1 + 2) / 2 + BIG_LIST_CAP / 2 + (1ull << BIG_TAB_BITS);
    uint64_t* off1;
    unsigned long long* cursor1;
    uint64_t* off2;
    unsigned long long* cursor2;
    unsigned int* hist1;
    unsigned int* cnt2;
    unsigned long long* counters;  // [0] entries handed out, [1..5] MDD_LEAF_PROF, [6] big leaves queued (u32), [7] anchors without room
    uint64_t* sums;
    unsigned int* tile_off;
    unsigned int* big_list;
    unsigned long long* big_tab;
    explicit SortWs(uint64_t* ws) {
        off1 = ws;
        cursor1 = reinterpret_cast<unsigned long long*>(off1 + BINS1 + 1);
        off2 = reinterpret_cast<uint64_t*>(cursor1 + BINS1);
        cursor2 = reinterpret_cast<unsigned long long*>(off2 + NL + 1);
        hist1 = reinterpret_cast<unsigned int*>(cursor2 + NL);
        cnt2 = hist1 + BINS1;
        counters = reinterpret_cast<unsigned long long*>(cnt2 + NL);
        sums = reinterpret_cast<uint64_t*>(counters + 8);
        tile_off = reinterpret_cast<unsigned int*>(sums + 1024);
        big_list = tile_off + BINS1 + 2;
        big_tab = reinterpret_cast<unsigned long long*>(big_list + BIG_LIST_CAP);
    }
    // clears the histograms and the counters (they are contiguous)
    int clear(hipStream_t st) const {
        HIP_TRY(hipMemsetAsync(hist1, 0, (BINS1 + NL) * sizeof(unsigned int) + 8 * sizeof(uint64_t), st));
        return MDD_OK;
    }
};
static_assert(SortWs::WORDS == MDD_SUBSTR_SORTED_WS_WORDS, "include/mdd.h: MDD_SUBSTR_SORTED_WS_WORDS");

// the entries of ent_a (level-1 digits in d1_ws, their histogram in W.hist1, their number in W.counters[0]) sorted into
// leaves: afterwards ent_a holds them leaf by leaf and W.off2[leaf .. leaf + 1] is a leaf's range; ent_b is scratch
int sort_into_leaves(hipStream_t st, const SortWs& W, uint64_t* ent_a, uint64_t* ent_b, uint64_t ent_cap, uint16_t* d1_ws) {
    // ---- level 1
    {
        int rc = exclusive_scan(st, W.hist1, (uint64_t)BINS1, W.off1, W.sums);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(W.cursor1, W.off1, BINS1 * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(scatter_kernel<1>, dim3((unsigned)((ent_cap + TILE1 - 1) / TILE1)), dim3(SC_THREADS), 0, st, ent_a, d1_ws, W.counters, ent_cap,
                       W.off1, W.tile_off, ent_b, W.cursor1);
    HIP_TRY(hipGetLastError());
    // ---- level 2
    const unsigned max_tiles2 = (unsigned)((ent_cap / TILE2 + BINS1 + XCDS - 1) / XCDS * XCDS);
    hipLaunchKernelGGL(tile_map_kernel, dim3(1), dim3(1024), 0, st, W.off1, W.tile_off);
    hipLaunchKernelGGL(count2_kernel, dim3(max_tiles2), dim3(256), 0, st, ent_b, W.off1, W.tile_off, W.cnt2);
    {
        int rc = exclusive_scan(st, W.cnt2, SortWs::NL, W.off2, W.sums);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(W.cursor2, W.off2, SortWs::NL * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(scatter_kernel<2>, dim3(max_tiles2), dim3(SC_THREADS), 0, st, ent_b, (const uint16_t*)nullptr, W.counters, ent_cap, W.off1,
                       W.tile_off, ent_a, W.cursor2);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

// segment plan shared by the two entry points: long documents are cut into segments queued in the context
int plan_segments(mdd_ctx* ctx, const char* who, const uint64_t* offsets, uint64_t n_docs, uint64_t text_bytes, uint32_t& seg_bytes,
                  uint64_t& max_items) {
    uint64_t seg = ctx->seg_bytes, need = text_bytes / SEG_ITEM_CAP + 1;
    if (seg < need) seg = need;
    seg = (seg + 255) / 256 * 256;
    REQUIRE(seg <= 0xFFFFFF00ull, "%s: text too large for the segment plan", who);
    max_items = text_bytes / seg;
    if (max_items > SEG_ITEM_CAP) max_items = SEG_ITEM_CAP;
    seg_bytes = (uint32_t)seg;
    HIP_TRY(hipMemsetAsync(ctx->seg_n_items, 0, 2 * sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(seg_plan_kernel, dim3(grid_for(n_docs, 256)), dim3(256), 0, ctx->stream, offsets, n_docs, text_bytes, seg_bytes,
                       ctx->seg_items, reinterpret_cast<unsigned long long*>(ctx->seg_n_items), max_items, (uint32_t*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

}  // namespace

extern "C" int mdd_substr_candidates(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                     int window, uint64_t* ent_a, uint64_t* ent_b, uint64_t ent_cap, uint16_t* d1_ws, uint64_t* ws,
                                     uint64_t ws_words, uint8_t* cand, uint64_t* n_cand, uint8_t* dup) {
    REQUIRE(ctx, "mdd_substr_candidates: ctx is NULL");
    REQUIRE(window >= (int)AN_MINWIN && window <= 128, "mdd_substr_candidates: window=%d outside [%u,128] (smaller windows: no filter, pass cand = NULL to mdd_substr_mark_sorted)",
            window, AN_MINWIN);
    REQUIRE(text_bytes < (1ULL << POS_BITS), "mdd_substr_candidates: text_bytes must be below 2^36");
    REQUIRE(n_cand, "mdd_substr_candidates: n_cand is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(n_cand, 0, sizeof(uint64_t), st));
    if (n_docs == 0 || text_bytes == 0) return MDD_OK;
    REQUIRE(text && offsets && ent_a && ent_b && d1_ws && ws && cand && dup, "mdd_substr_candidates: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(text) & 3u) == 0, "mdd_substr_candidates: text must be 4-byte aligned");
    REQUIRE(ws_words >= SortWs::WORDS, "mdd_substr_candidates: workspace of %llu u64 words needed (MDD_SUBSTR_SORTED_WS_WORDS)",
            (unsigned long long)SortWs::WORDS);
    REQUIRE((reinterpret_cast<uintptr_t>(ent_a) & 7u) == 0 && (reinterpret_cast<uintptr_t>(ent_b) & 7u) == 0 &&
                (reinterpret_cast<uintptr_t>(d1_ws) & 1u) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0 &&
                (reinterpret_cast<uintptr_t>(cand) & 7u) == 0,
            "mdd_substr_candidates: ent_a, ent_b, ws, cand must be 8-byte aligned, d1_ws 2-byte aligned");
    const uint64_t waves = (uint64_t)ctx->cus * 4 * 7;
    // ent_b doubles as the anchor flags (a bit per text position, (text_bytes >> 6) + 3 words) once the sort is done with it;
    // the bound below is the one-byte-per-position layout's and stays: callers size ent_b by it
    REQUIRE(ent_cap >= waves * EMIT_CHUNK + TILE1 && ent_cap < (1ULL << 40) && ent_cap >= (text_bytes + 64 + 7) / 8,
            "mdd_substr_candidates: ent_cap=%llu too small (need max(%llu, (text_bytes + 71) / 8); anchors beyond it only switch the filter off)",
            (unsigned long long)ent_cap, (unsigned long long)(waves * EMIT_CHUNK + TILE1));
    const SortWs W(ws);
    {
        int rc = W.clear(st);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(dup, 0, text_bytes, st));
    const bool resolve = window == (int)AN_MINWIN;  // repeated anchors are settled in their leaves (anchor_resolve_kernel)
    if (resolve) {
        // cand holds the bitmap of hashed window starts (the word a 19-bit read may reach past the last position included)
        // until smear_kernel overwrites it
        HIP_TRY(hipMemsetAsync(cand, 0, MDD_SUBSTR_CAND_BYTES(text_bytes), st));
        uint64_t blocks = (n_docs + 3) / 4, cap = (uint64_t)ctx->cus * 64;
        hipLaunchKernelGGL(vmap_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, st, text, text_bytes, offsets, n_docs,
                           (uint32_t)window, reinterpret_cast<uint32_t*>(cand));
        HIP_TRY(hipGetLastError());
    }
    AnchorArgs E{};
    E.text = text;
    E.text_bytes = text_bytes;
    E.offsets = offsets;
    E.n_docs = n_docs;
    uint64_t max_items = 0;
    {
        int rc = plan_segments(ctx, "mdd_substr_candidates", offsets, n_docs, text_bytes, E.seg_bytes, max_items);
        if (rc) return rc;
    }
    E.items = ctx->seg_items;
    E.n_items = ctx->seg_n_items;
    E.max_items = max_items;
    E.work_counter = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 1);
    E.ent = ent_a;
    E.d1 = d1_ws;
    E.ent_cap = ent_cap;
    E.n_ent = W.counters;
    E.hist1 = W.hist1;
    E.overflow = W.counters + 7;
    E.d1_bits = an_d1_bits(text_bytes);
    uint64_t w = waves < n_docs + max_items ? waves : n_docs + max_items;
    while ((1ull << E.batch_shift) < w / 4) E.batch_shift++;  // (batches of 32 until ~8 items per wave are left: see emit_kernel)
    hipLaunchKernelGGL(anchor_kernel, dim3((unsigned)((w + EMIT_WAVES - 1) / EMIT_WAVES)), dim3(64 * EMIT_WAVES), 0, st, E);
    HIP_TRY(hipGetLastError());
    {
        int rc = sort_into_leaves(st, W, ent_a, ent_b, ent_cap, d1_ws);
        if (rc) return rc;
    }
    uint32_t* flag = reinterpret_cast<uint32_t*>(ent_b);  // a bit per position + the two zero words smear_kernel reads past them
    HIP_TRY(hipMemsetAsync(flag, 0, ((text_bytes >> 6) + 3) * 8, st));
    const uint32_t n_leaves_a = (1u << E.d1_bits) * (uint32_t)BINS2;  // anchors carry a 9- to 11-bit level-1 digit
    if (resolve && !getenv("MDD_SUBSTR_NO_RESOLVE")) {     // (the knob: A/B runs and tests of the filter-only path)
        ResolveArgs R{};
        R.ent = ent_a;
        R.off2 = W.off2;
        R.n_leaves = n_leaves_a;
        R.text = text;
        R.text_bytes = text_bytes;
        R.vmap = reinterpret_cast<const uint32_t*>(cand);
        R.dup = dup;
        R.flag = flag;
        R.max_group = RS_GROUP;
        if (const char* knob = getenv("MDD_SUBSTR_GROUP")) {
            const int g = atoi(knob);
            if (g >= 1 && g <= (int)RS_MAX) R.max_group = (uint32_t)g;
        }
        // expected anchors per leaf (~one per 10 bytes): up to ~250 every leaf fits the table and one above RS_MAX is heavy
        // repetition; beyond (more than ~9 GB of text: 2^22 leaves is all there is) leaves up to 4 x the expected size are
        // settled in rounds by the second instance
        const uint64_t per_leaf = text_bytes / 10 / n_leaves_a;
        R.big_hi = RS_MAX;
        if (per_leaf > 250 || getenv("MDD_SUBSTR_D1_BITS")) {      // (the test hook forces few leaves on small inputs)
            const uint64_t hi4 = 4 * per_leaf > 1024 ? 4 * per_leaf : 1024;
            R.big_hi = (uint32_t)(hi4 < RS_BIG_MAX ? hi4 : RS_BIG_MAX);
        }
        hipLaunchKernelGGL(anchor_resolve_kernel<false>, dim3((unsigned)ctx->cus * RS_WAVES_PER_CU), dim3(64), 0, st, R);
        if (R.big_hi > RS_MAX)
            hipLaunchKernelGGL(anchor_resolve_kernel<true>, dim3((unsigned)ctx->cus * RS_WAVES_PER_CU_BIG), dim3(64), 0, st, R);
    } else {
        hipLaunchKernelGGL(anchor_leaf_kernel, dim3((unsigned)ctx->cus * LEAF_WAVES_PER_CU), dim3(64), 0, st, ent_a, W.off2, n_leaves_a, flag,
                           text_bytes);
    }
    const uint64_t smear_blocks = (((text_bytes + 63) >> 6) + 255) / 256, smear_cap = (uint64_t)ctx->cus * 16;
    hipLaunchKernelGGL(smear_kernel, dim3((unsigned)(smear_blocks < smear_cap ? smear_blocks : smear_cap)), dim3(256), 0, st,
                       reinterpret_cast<const uint64_t*>(flag), text_bytes, reinterpret_cast<uint64_t*>(cand),
                       reinterpret_cast<unsigned long long*>(n_cand), W.counters + 7);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_substr_mark_sorted(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* offsets, uint64_t n_docs,
                                      int window, int part, int n_parts, uint64_t* ent_a, uint64_t* ent_b, uint64_t ent_cap,
                                      uint16_t* d1_ws, uint64_t* ws, uint64_t ws_words, uint8_t* dup, const uint8_t* cand) {
    REQUIRE(ctx, "mdd_substr_mark_sorted: ctx is NULL");
    REQUIRE(window >= 1 && window <= 128, "mdd_substr_mark_sorted: window=%d outside [1,128]", window);
    REQUIRE(text_bytes < (1ULL << POS_BITS), "mdd_substr_mark_sorted: text_bytes must be below 2^36 (use mdd_substr_mark beyond)");
    REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "mdd_substr_mark_sorted: part=%d outside [0,%d)", part, n_parts);
    REQUIRE(!cand || window >= (int)AN_MINWIN, "mdd_substr_mark_sorted: the candidate filter needs window >= %u", AN_MINWIN);
    REQUIRE((reinterpret_cast<uintptr_t>(cand) & 7u) == 0, "mdd_substr_mark_sorted: cand must be 8-byte aligned");
    if (n_docs == 0 || text_bytes == 0) return MDD_OK;
    REQUIRE(text && offsets && ent_a && ent_b && d1_ws && ws && dup, "mdd_substr_mark_sorted: NULL buffer");
    REQUIRE((reinterpret_cast<uintptr_t>(text) & 3u) == 0, "mdd_substr_mark_sorted: text must be 4-byte aligned");
    const uint64_t n_leaves = SortWs::NL;
    REQUIRE(ws_words >= SortWs::WORDS, "mdd_substr_mark_sorted: workspace of %llu u64 words needed (MDD_SUBSTR_SORTED_WS_WORDS)",
            (unsigned long long)SortWs::WORDS);
    REQUIRE((reinterpret_cast<uintptr_t>(ent_a) & 7u) == 0 && (reinterpret_cast<uintptr_t>(ent_b) & 7u) == 0 &&
                (reinterpret_cast<uintptr_t>(d1_ws) & 1u) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0,
            "mdd_substr_mark_sorted: ent_a, ent_b, ws must be 8-byte aligned, d1_ws 2-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t waves = (uint64_t)ctx->cus * 4 * 7;
    REQUIRE(ent_cap >= waves * EMIT_CHUNK + TILE1 && ent_cap < (1ULL << 40),
            "mdd_substr_mark_sorted: ent_cap=%llu too small (need the windows of the pass + %llu)", (unsigned long long)ent_cap,
            (unsigned long long)(waves * EMIT_CHUNK + TILE1));
    const SortWs W(ws);
    hipStream_t st = ctx->stream;
    {
        int rc = W.clear(st);
        if (rc) return rc;
    }
    if (part == 0 && !cand) HIP_TRY(hipMemsetAsync(dup, 0, text_bytes, st));  // (with cand: mdd_substr_candidates cleared dup and may have marked repeats)
    // ---- emit
    EmitArgs E{};
    E.text = text;
    E.text_bytes = text_bytes;
    E.offsets = offsets;
    E.n_docs = n_docs;
    E.window = (uint32_t)window;
    uint64_t max_items = 0;
    {
        int rc = plan_segments(ctx, "mdd_substr_mark_sorted", offsets, n_docs, text_bytes, E.seg_bytes, max_items);
        if (rc) return rc;
    }
    E.items = ctx->seg_items;
    E.n_items = ctx->seg_n_items;
    E.max_items = max_items;
    E.work_counter = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 1);
    E.ent = ent_a;
    E.ent_cap = ent_cap;
    E.n_ent = W.counters;
    E.hist1 = W.hist1;
    E.overflow = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 2);
    E.part = (uint32_t)part;
    E.n_parts = (uint32_t)n_parts;
    E.d1 = d1_ws;
    E.cand = reinterpret_cast<const uint64_t*>(cand);
    uint64_t w = waves < n_docs + max_items ? waves : n_docs + max_items;
    // guided self-scheduling: a claim takes (items left) >> batch_shift items, at most max_batch.  Without the filter an item
    // is ~30 rounds of hashing and single items at the end balance the waves; with it an item is a few microseconds and the
    // claims themselves (11 ns each, serialised on one address) were 2 of the kernel's 5 ms
    E.max_batch = cand ? 64u : 16u;
    while ((1ull << E.batch_shift) < (cand ? w / 4 : w * 4)) E.batch_shift++;
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((w + EMIT_WAVES - 1) / EMIT_WAVES)), dim3(64 * EMIT_WAVES), 0, st, E);
    HIP_TRY(hipGetLastError());
    {
        int rc = sort_into_leaves(st, W, ent_a, ent_b, ent_cap, d1_ws);
        if (rc) return rc;
    }
    // ---- leaves
    LeafArgs L{};
    L.ent = ent_a;
    L.off2 = W.off2;
    L.n_leaves = n_leaves;
    L.text = text;
    L.text_bytes = text_bytes;
    L.window = (uint32_t)window;
    L.dup = dup;
    L.overflow = reinterpret_cast<unsigned long long*>(ctx->seg_n_items + 2);
    L.prof = W.counters + 1;
    L.n_big = reinterpret_cast<unsigned int*>(W.counters + 6);  // zeroed with the counters above
    L.big_list = W.big_list;
    L.big_tab = W.big_tab;
    L.one_round = LEAF_ONE_ROUND;
    L.per_round = LEAF_PER_ROUND;
    L.n_ent = W.counters;
    if (const char* knob = getenv("MDD_SUBSTR_LEAF_ROUND")) {  // testing: "one_round,per_round" below the defaults forces the multi-round path
        unsigned a = 0, b = 0;
        if (sscanf(knob, "%u,%u", &a, &b) == 2 && a >= 1 && a <= LEAF_ONE_ROUND && b >= 1 && b <= LEAF_PER_ROUND) {
            L.one_round = a;
            L.per_round = b;
        }
    }
    HIP_TRY(hipMemsetAsync(W.big_tab, 0xFF, sizeof(uint64_t) << BIG_TAB_BITS, st));
    hipLaunchKernelGGL(leaf_kernel, dim3((unsigned)ctx->cus * LEAF_WAVES_PER_CU), dim3(64), 0, st, L);
    hipLaunchKernelGGL(bigleaf_kernel, dim3((unsigned)ctx->cus * LEAF_WAVES_PER_CU), dim3(64), 0, st, L);  // returns at once when nothing was queued
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}
