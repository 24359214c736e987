// Shingle enumeration shared by the translation units that walk documents unit by unit
// (mdd_hip.hip: sig_kernel; mdd_substr.hip: the sliding-window kernels of the exact substring stage).
#ifndef MDD_SHINGLE_H
#define MDD_SHINGLE_H

#include "mdd_common.h"

namespace {
using namespace mddi;

// Text access: `docw` is the (wave-uniform) aligned dword that contains the first byte of the
// document; a shingle is addressed by p = byte offset from docw (32 bit).  Loads run up to
// 80 bytes past the shingle start and are NOT clamped: the caller's allocation extends
// MDD_TEXT_PAD bytes past text_bytes (include/mdd.h).
// Region markers for tools/isa_hist.py: an assembler COMMENT in the kernel's instruction stream (no instruction, no
// operand) naming the source region the following code belongs to.  With -DMDD_PATHCOUNT (tools/sig_bench.hip only)
// the same macro also counts how often the region is entered, per wave, so that static per-region instruction
// histograms x dynamic entry counts give the executed instruction mix (profiles/r2_issue/).
enum PathRegion { PR_ITEM, PR_ROUND, PR_SCAN, PR_SHA_A13, PR_SHA4, PR_SHA10, PR_SHA14, PR_SHA_MULTI, PR_STAGE, PR_FULL, PR_FSETUP,
                  PR_FGROUP, PR_FDETAIL, PR_FFULL, PR_EXACT, PR_ITEM_END, PR_ASCIICHK, PR_RING, PR_COUNT };
#ifdef MDD_PATHCOUNT
static __device__ unsigned long long g_pathcnt[PR_COUNT];
#define MDD_REGION(name, id)                                                                      \
    do {                                                                                          \
        asm volatile("; MDD_REGION " name);                                                       \
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&g_pathcnt[id], 1ull); \
    } while (0)
#define MDD_REGION_CONT(name) asm volatile("; MDD_REGION_CONT " name)  // code that continues a region already counted
#elif defined(MDD_NO_REGION_MARKERS)  // A/B check that the comments do not change the generated code's speed
#define MDD_REGION(name, id) ((void)0)
#define MDD_REGION_CONT(name) ((void)0)
#else
#define MDD_REGION(name, id) asm volatile("; MDD_REGION " name)
#define MDD_REGION_CONT(name) asm volatile("; MDD_REGION_CONT " name)
#endif

struct TextView {
    const uint32_t* docw;
};

__device__ __forceinline__ uint32_t ld_dword(const TextView& tv, uint32_t idx) { return tv.docw[idx]; }

// ------------------------------------------------------------------------------------------
// device: wave helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)x, o, 64);
        x = x > y ? x : y;
    }
    return x;
}

__device__ __forceinline__ void wave_sync() {  // order this wave's LDS traffic (DS ops issue in order; this pins the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef MDD_WAVES
#define MDD_WAVES 1
#endif
constexpr int WAVES = MDD_WAVES;  // waves (= documents in flight) per workgroup; 1 frees a wave slot as soon as its document ends
constexpr int RING = 512;        // code-point / token boundary ring per wave (entries)
constexpr uint64_t SEG_ITEM_CAP = 1u << 20;     // extra segments one mdd_signatures call can queue
constexpr uint32_t SEG_BYTES_DEFAULT = 16384;   // documents longer than this are cut into segments
constexpr uint32_t FAST_GUARD = 8u;

struct __attribute__((aligned(16))) WaveLds {
    uint32_t hbuf[64];     // shingle hashes of the current round
    uint32_t ringS[RING];  // unit start offsets (doc-relative)
    uint32_t ringE[RING];  // unit end offsets (word unit only)
};

// ------------------------------------------------------------------------------------------
// Shingle enumeration shared by sig_kernel and the sliding-window kernels of the exact
// substring stage: walks the units (code points / words) of segment [seg_beg, seg_end) of one
// document in rounds of up to 64 shingles and calls visit(s, L, nvalid, n_done) once per
// round; lane < nvalid holds shingle n_done + lane: s = its byte offset from tv.docw, L = its
// length in bytes (lanes >= nvalid mirror a valid shingle).  Returns the number of shingles.
// ------------------------------------------------------------------------------------------
template <class Visit>
__device__ __forceinline__ uint32_t for_each_round(WaveLds& lds, const uint8_t* text, uint64_t beg, uint32_t len, const TextView& tv,
                                                   uint32_t boff, int unit, uint32_t width, bool ascii, bool split, uint32_t seg_beg,
                                                   uint32_t seg_end, uint32_t lane, Visit&& visit) {
    // ring state (wave-uniform)
    uint32_t n_entS = 0, n_entE = 0, pos = seg_beg;
    bool scan_done = false, seg_done = false;
    // word unit: a segment that begins inside a token sees that token's END but not its
    // START; drop that first end so that ringS[k] / ringE[k] keep describing the same token
    bool skip_end = unit != MDD_UNIT_CHAR && seg_beg > 0 && seg_beg < len && text[beg + seg_beg - 1] != 0x20 &&
                    text[beg + seg_beg] != 0x20;
    const uint32_t look = unit == MDD_UNIT_CHAR ? width : width - 1;
    uint32_t open_w = skip_end ? 1u : 0u;  // word unit: a token spans the first byte of the next scan window (ends come one short)
    uint32_t n_done = 0;
    uint32_t ascii_total = 0;  // shingles of this segment when boundaries are implicit
    if (ascii && len >= width) {
        uint32_t j_end = len - width + 1 < seg_end ? len - width + 1 : seg_end;
        ascii_total = j_end > seg_beg ? j_end - seg_beg : 0;
    }
    // boundary scan input: the scan window is 256 bytes, lane l owns the 4 bytes at pos + 4l, i.e. one
    // unaligned dword = two coalesced aligned loads.  The next TWO windows are kept in flight in
    // registers (a round of 64 word shingles consumes ~1.3 windows), so their latency hides behind the
    // SHA-1 + permute work of the current round instead of stalling the ring fill.
    uint32_t pfa_lo = 0, pfa_hi = 0, pfb_lo = 0, pfb_hi = 0;
    auto window_load = [&](uint32_t wpos, uint32_t& lo, uint32_t& hi) {
        uint32_t p0 = wpos + 4u * lane;
        // bytes at or after len are masked out by the scan (inm), so a lane past the end may load anything that is
        // mapped: it re-reads the document's first dwords.  No branch and no select on the loaded value: the load
        // then writes the prefetch registers directly and is only waited for where the scan consumes it.
        uint32_t i = p0 < len ? (boff + p0) >> 2 : 0u;  // reads stay within 8 bytes of the document's end
        lo = tv.docw[i];
        hi = tv.docw[i + 1];
    };
    uint32_t carry_sp = 0x80000000u;  // word unit, bit 31: the byte before the scan window is a space (or the document starts)
    if (!ascii) {
        window_load(pos, pfa_lo, pfa_hi);
        window_load(pos + 256u, pfb_lo, pfb_hi);
        if (unit != MDD_UNIT_CHAR && pos > 0 && pos <= len) carry_sp = text[beg + pos - 1] == 0x20 ? 0x80000000u : 0u;
    }

    for (;;) {
        uint32_t nvalid;
        uint32_t s = 0;  // byte offset of the lane's shingle from tv.docw
        uint32_t L = 0;
        MDD_REGION("round", PR_ROUND);
        if (ascii) {
            uint32_t rest = ascii_total - n_done;
            nvalid = rest < 64u ? rest : 64u;
            if (nvalid == 0) break;
            uint32_t j = seg_beg + n_done + (lane < nvalid ? lane : 0u);
            s = boff + j;
            L = width;
        } else {
            // ---- fill the boundary ring until a full round is available or the doc ends
            for (;;) {
                uint32_t n_ent = unit == MDD_UNIT_CHAR ? n_entS : n_entE;
                if (scan_done || n_ent - n_done >= 64u + look) break;
                MDD_REGION("scan", PR_SCAN);
                // window = doc-relative bytes [pos, pos+256); lane handles 4 of them
                uint32_t p0 = pos + 4u * lane;
                uint32_t w4 = __builtin_amdgcn_alignbyte(pfa_hi, pfa_lo, boff);  // bytes p0 .. p0+3
                const uint32_t pfn_hi = pfa_hi;  // byte p0+4 is its byte boff
                pfa_lo = pfb_lo;
                pfa_hi = pfb_hi;
                asm volatile("" : "+v"(pfa_lo), "+v"(pfa_hi));  // rotate NOW, so that the load below can write pfb itself
                window_load(pos + 512u, pfb_lo, pfb_hi);
                uint32_t nval = p0 < len ? (len - p0 < 4u ? len - p0 : 4u) : 0u;
                // Flags live in bit 7 of byte k (SWAR over the lane's dword): inm = byte p0+k lies inside the document
                const uint32_t inm = nval >= 4u ? 0x80808080u : 0x00808080u >> (8u * (3u - nval));
                uint32_t fs, fe = 0;  // byte k starts / ends a unit
                uint32_t open_l = 0, open_next = 0;
                if (unit == MDD_UNIT_CHAR) {
                    fs = inm & (~w4 | (w4 << 1));  // not a continuation byte 10xxxxxx: bit 7 clear or bit 6 set
                } else {
                    const uint32_t x = w4 ^ 0x20202020u;
                    const uint32_t nsp = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & inm;  // byte is inside and no space
                    const uint32_t sp = ~nsp & 0x80808080u;                            // a space, or past the end
                    // the neighbours' flags come over DPP wave shifts (lane l reads lane l -/+ 1; the end lanes keep
                    // `old`): lane 0's left neighbour is the carry of the previous window, lane 63's right neighbour
                    // is byte pos + 256, taken from lane 63's second dword
                    const uint32_t nb63 = (__builtin_amdgcn_readlane(pfn_hi, 63) >> (8u * boff)) & 0xFFu;
                    const uint32_t right = (pos + 256u >= len || nb63 == 0x20u) ? 0x80u : 0u;
                    const uint32_t sp_prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_sp, (int)sp, 0x138, 0xf, 0xf, false);
                    const uint32_t sp_next = (uint32_t)__builtin_amdgcn_update_dpp((int)right, (int)sp, 0x130, 0xf, 0xf, false);
                    carry_sp = __builtin_amdgcn_readlane(sp, 63);
                    fs = nsp & __builtin_amdgcn_alignbit(sp, sp_prev, 24);  // byte k-1 is a space: (sp << 8) | (sp_prev >> 24)
                    fe = nsp & __builtin_amdgcn_alignbit(sp_next, sp, 8);   // byte k+1 is a space: (sp >> 8) | (sp_next << 24)
                    open_l = ((nsp ^ fs) >> 7) & 1u;  // the lane's first byte continues a word begun in a lower lane
                    open_next = ((carry_sp >> 31) == 0u && right == 0u) ? 1u : 0u;  // a word runs on into the next window
                }
                // ranks in byte order from the bit planes of the lanes' counts: a dword holds at most 4 code-point
                // starts (3 planes), and at most 2 word starts and 2 word ends (2 planes each)
                const uint32_t cs = (uint32_t)__popc(fs);
                const uint64_t s0 = __ballot(cs & 1u), s1 = __ballot(cs & 2u);
                uint32_t rank_s = below(s0) + 2u * below(s1);
                uint32_t tot_s = (uint32_t)__popcll(s0) + 2u * (uint32_t)__popcll(s1), tot_e = 0;
                if (unit == MDD_UNIT_CHAR) {
                    const uint64_t s2 = __ballot(cs & 4u);
                    rank_s += 4u * below(s2);
                    tot_s += 4u * (uint32_t)__popcll(s2);
                    uint32_t rs = n_entS + rank_s;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (fs & (0x80u << (8 * k))) lds.ringS[(rs++) & (RING - 1)] = p0 + k;
                } else {
                    // Starts and ends alternate, so the ends below a lane follow from the starts below it, the word
                    // that spans the lane's first byte and the word that spans the window's first byte (open_w).
                    // (starts below byte p) - (ends below byte p) = [a word spans the boundary before byte p], so:
                    const uint32_t rank_e = rank_s - open_l + open_w;
                    tot_e = tot_s - open_next + open_w;
                    open_w = open_next;
                    // first / last flagged byte: bit 8k+7 -> k
                    if (fs) {
                        const uint32_t rs = n_entS + rank_s;
                        lds.ringS[rs & (RING - 1)] = p0 + ((uint32_t)__builtin_ctz(fs) >> 3);
                        if (cs == 2u) lds.ringS[(rs + 1u) & (RING - 1)] = p0 + ((31u - (uint32_t)__builtin_clz(fs)) >> 3);
                    }
                    if (fe) {  // skip_end drops the window's (= the segment's) very first end
                        const uint32_t sk = skip_end ? 1u : 0u;
                        const uint32_t re = n_entE + rank_e - sk;
                        if (!(sk && rank_e == 0u)) lds.ringE[re & (RING - 1)] = p0 + ((uint32_t)__builtin_ctz(fe) >> 3) + 1u;
                        if (fe & (fe - 1u)) lds.ringE[(re + 1u) & (RING - 1)] = p0 + ((31u - (uint32_t)__builtin_clz(fe)) >> 3) + 1u;
                    }
                }
                n_entS += tot_s;
                if (skip_end && tot_e) {
                    tot_e -= 1;
                    skip_end = false;
                }
                n_entE += tot_e;
                pos += 256u;
                if (pos >= len) {
                    scan_done = true;
                    if (unit == MDD_UNIT_CHAR) {  // sentinel: end of the last code point
                        if (lane == 0) lds.ringS[n_entS & (RING - 1)] = len;
                        n_entS += 1;
                    }
                }
            }
            MDD_REGION("ring", PR_RING);
            wave_sync();
            uint32_t n_ent = unit == MDD_UNIT_CHAR ? n_entS : n_entE;
            uint32_t rest = n_ent > n_done + look ? n_ent - look - n_done : 0u;
            nvalid = rest < 64u ? rest : 64u;
            if (nvalid == 0 || seg_done) break;
            uint32_t j = n_done + (lane < nvalid ? lane : 0u);
            uint32_t sb = lds.ringS[j & (RING - 1)];
            uint32_t eb = unit == MDD_UNIT_CHAR ? lds.ringS[(j + look) & (RING - 1)]
                                                : lds.ringE[(j + look) & (RING - 1)];
            if (split) {  // units that start at or after seg_end belong to the next segment (starts ascend)
                uint32_t nin = (uint32_t)__popcll(__ballot(lane < nvalid && sb < seg_end));
                if (nin < nvalid) {
                    seg_done = true;
                    nvalid = nin;
                    if (nvalid == 0) break;
                    if (lane >= nvalid) {
                        sb = lds.ringS[n_done & (RING - 1)];
                        eb = unit == MDD_UNIT_CHAR ? lds.ringS[(n_done + look) & (RING - 1)]
                                                   : lds.ringE[(n_done + look) & (RING - 1)];
                    }
                }
            }
            // the ring only ever holds offsets sb < eb <= len; a clamp keeps every text read inside the document even
            // if that invariant were broken (two VALU per round of 64 shingles)
            sb = sb < len ? sb : len;
            eb = eb < sb ? sb : (eb < len ? eb : len);
            s = boff + sb;
            L = eb - sb;
        }

        MDD_REGION_CONT("round");
        visit(s, L, nvalid, n_done);
        n_done += nvalid;
    }
    return n_done;
}

// true when no byte the segment's shingles can touch is >= 0x80 (whole dwords are tested, so a neighbour's
// non-ASCII byte may say "no": conservative)
__device__ __forceinline__ bool segment_is_ascii(const TextView& tv, uint32_t boff, uint32_t len, uint32_t seg_beg, uint32_t seg_end,
                                                 uint32_t width, uint32_t lane) {
    uint32_t acc = 0;
    if (len) {
        uint32_t hi = (uint64_t)seg_end + width < len ? seg_end + width : len;
        uint32_t d0 = (boff + seg_beg) >> 2, d1 = (boff + hi - 1u) >> 2;
        for (uint32_t i = d0 + lane; i <= d1; i += 64) {
            MDD_REGION("asciichk", PR_ASCIICHK);
            acc |= tv.docw[i];
        }
        MDD_REGION_CONT("item");
    }
    return __ballot((acc & 0x80808080u) != 0u) == 0ull;
}

// One thread per document: documents longer than seg_bytes get their extra segments queued
// and (sig != NULL) their signature row preset to 0xFFFFFFFF (their segments combine with atomicMin).
__global__ void __launch_bounds__(256) seg_plan_kernel(const uint64_t* offsets, uint64_t n_docs, uint64_t text_bytes, uint32_t seg_bytes,
                                                       uint64_t* items, unsigned long long* n_items, uint64_t max_items, uint32_t* sig,
                                                       int K) {
    uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    uint64_t beg, end;
    doc_range(offsets, d, text_bytes, beg, end);
    uint64_t len = end - beg;
    if (len <= seg_bytes) return;
    uint64_t nseg = (len + seg_bytes - 1) / seg_bytes;
    unsigned long long base = atomicAdd(n_items, (unsigned long long)(nseg - 1));
    for (uint64_t k = 1; k < nseg; k++)
        if (base + k - 1 < max_items) items[base + k - 1] = (d << 24) | k;
    if (sig)
        for (int q = 0; q < K; q++) sig[d * (uint64_t)K + q] = 0xFFFFFFFFu;
}

}  // namespace

#endif  // MDD_SHINGLE_H
