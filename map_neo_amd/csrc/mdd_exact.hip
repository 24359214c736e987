// Exact grouping of byte ranges ("units") of the packed text: for every unit the smallest
// unit index with IDENTICAL bytes.  One primitive behind the exact-match stages that bracket
// MinHash in the reference's EN pipeline (paths relative to
// /root/reference/Matrix/english_data_clean_pipeline/dedup):
//   Exact_Document_Deduplication/exact_document_deduplication.py:16-18  units = documents
//   Paragraph_Deduplication/paragraph.py:15-30                          units = "\n"-separated paragraphs
//   Exact_Substring_Deduplication/exact_substr_dedup.py:34-47           units = 50-character windows
// The reference groups on SHA-256 of the unit; equal SHA-256 <=> equal bytes (no collision is
// known), so grouping on the bytes themselves gives the same groups.  A 64-bit content hash only
// selects the hash-table slot and a 32-bit tag; membership is decided by comparing the bytes inside
// the kernels, so the result is exact and nothing is left to the caller.
// HBM-bound: every unit's bytes are read once to hash (duplicates once more to verify) plus one
// random 8-byte atomic and one random 8-byte read per unit.
#include "mdd_common.h"

namespace {
using namespace mddi;

constexpr uint64_t C1 = 0x9E3779B97F4A7C15ULL, C2 = 0xD6E8FEB86659FD93ULL;

// Unit u = [b, b + len) of the text, clamped into [0, text_bytes]: ranges come from device memory and cannot be
// validated on the host without a synchronisation; malformed ones must not reach outside the allocation.
__device__ __forceinline__ void unit_range(const uint64_t* ubeg, const uint64_t* uend, uint64_t u, uint64_t text_bytes, uint64_t& b,
                                           uint64_t& len) {
    uint64_t e = uend[u];
    b = ubeg[u];
    e = e < text_bytes ? e : text_bytes;
    b = b < e ? b : e;
    len = e - b;
}

// 8 bytes at absolute byte address a (any alignment) from an 8-byte aligned base; the allocation
// extends MDD_TEXT_PAD bytes past the text, so the second load is always in bounds
__device__ __forceinline__ uint64_t load8(const uint64_t* t64, uint64_t a) {
    uint64_t i = a >> 3;
    uint32_t sh = (uint32_t)(a & 7) * 8;
    uint64_t lo = t64[i];
    if (sh == 0) return lo;
    return (lo >> sh) | (t64[i + 1] << (64 - sh));
}

// order-sensitive, position-mixed SUM over 8-byte chunks: chunks can be hashed in any order and
// by any number of lanes (thread-per-unit and wave-per-unit kernels give the same value)
__device__ __forceinline__ uint64_t chunk_term(const uint64_t* t64, uint64_t b, uint64_t len, uint64_t c, uint32_t& cps) {
    uint64_t v = load8(t64, b + 8 * c);
    uint64_t rem = len - 8 * c;
    if (rem < 8) v &= (1ULL << (8 * rem)) - 1;  // rem >= 1 here
    // code points that start in the chunk: its bytes minus the utf-8 continuation bytes 10xxxxxx (the masked-off bytes are 0)
    cps += (uint32_t)(rem < 8 ? rem : 8) - (uint32_t)__builtin_popcountll(v & ~(v << 1) & 0x8080808080808080ULL);
    return mix64(v + C1 * (c + 1));
}
__device__ __forceinline__ uint64_t finish_key(uint64_t sum, uint64_t len, int key_bits) {
    uint64_t h = mix64(sum ^ (len * C2) ^ C1);
    if (key_bits < 64) h &= (1ULL << key_bits) - 1;
    return h ? h : 1ULL;
}

__global__ void __launch_bounds__(256) exact_hash_thread_kernel(const uint64_t* t64, uint64_t text_bytes, const uint64_t* ubeg,
                                                                const uint64_t* uend, uint64_t n, int key_bits, uint64_t* key,
                                                                uint32_t* ucps) {
    uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    uint64_t b, len;
    unit_range(ubeg, uend, u, text_bytes, b, len);
    uint64_t nch = (len + 7) >> 3, sum = 0;
    uint32_t cps = 0;
    for (uint64_t c = 0; c < nch; c++) sum += chunk_term(t64, b, len, c, cps);
    key[u] = finish_key(sum, len, key_bits);
    if (ucps) ucps[u] = cps;
}

// one wave per unit: for long units (documents, paragraphs).  Grid-stride: a launch may not exceed 2^32
// work-items in total, and 64 lanes per unit would pass that at 67 M units.
__global__ void __launch_bounds__(256) exact_hash_wave_kernel(const uint64_t* t64, uint64_t text_bytes, const uint64_t* ubeg,
                                                              const uint64_t* uend, uint64_t n, int key_bits, uint64_t* key,
                                                              uint32_t* ucps) {
    const uint32_t lane = lane_id();
    const uint64_t nw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t u = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < n; u += nw) {
        uint64_t b, len;
        unit_range(ubeg, uend, u, text_bytes, b, len);
        uint64_t nch = (len + 7) >> 3, sum = 0;
        uint32_t cps = 0;
        for (uint64_t c = lane; c < nch; c += 64) sum += chunk_term(t64, b, len, c, cps);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, o, 64);
            uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), o, 64);
            sum += ((uint64_t)hi << 32) | lo;
            cps += (uint32_t)__shfl_xor((int)cps, o, 64);
        }
        if (lane == 0) {
            key[u] = finish_key(sum, len, key_bits);
            if (ucps) ucps[u] = cps;
        }
    }
}

// eight lanes per unit, eight units per wave: for units of a few hundred bytes (paragraphs), where a whole wave per unit
// leaves most lanes idle and pays the unit's range loads and the reduction once per unit (2.9 ms for 8.3 M paragraphs of
// 263 bytes against 1.2 ms here).  A group's eight lanes read 64 contiguous bytes per step; the key is the same sum.
__global__ void __launch_bounds__(256) exact_hash_group8_kernel(const uint64_t* t64, uint64_t text_bytes, const uint64_t* ubeg,
                                                                const uint64_t* uend, uint64_t n, int key_bits, uint64_t* key,
                                                                uint32_t* ucps) {
    const uint32_t sub = threadIdx.x & 7u;
    const uint64_t ng = (uint64_t)gridDim.x * (blockDim.x >> 3);
    for (uint64_t u0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;; u0 += ng) {
        if (__ballot(u0 < n) == 0ull) break;  // whole waves stay in the loop (shuffles); groups past the end redo the last unit
        const uint64_t u = u0 < n ? u0 : n - 1;
        uint64_t b, len;
        unit_range(ubeg, uend, u, text_bytes, b, len);
        uint64_t nch = (len + 7) >> 3, sum = 0;
        uint32_t cps = 0;
        for (uint64_t c = sub; c < nch; c += 8) sum += chunk_term(t64, b, len, c, cps);
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sum, o, 64);
            uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), o, 64);
            sum += ((uint64_t)hi << 32) | lo;
            cps += (uint32_t)__shfl_xor((int)cps, o, 64);
        }
        if (sub == 0 && u0 < n) {
            key[u] = finish_key(sum, len, key_bits);
            if (ucps) ucps[u] = cps;
        }
    }
}

// One 8-byte table entry = (tag << 32 | smallest unit index seen), tag = high half of the key, slot from its low
// bits.  A tag match is confirmed by comparing the BYTES with the entry's unit: different -> just another occupant,
// keep probing.  So the table is exact (nothing is left to the caller), every slot holds one content for good, and
// only its unit index shrinks.  ONE random atomic per unit in the common case (the claim) and one random read per
// unit in the link pass; every device atomic is a full memory transaction, so the table passes are bound by the
// device's random-access rate, not by bytes.
constexpr unsigned long long EMPTY = ~0ULL;

__device__ __forceinline__ bool same_unit(const uint64_t* t64, uint64_t b, uint64_t len, uint64_t rb, uint64_t rlen) {
    if (len != rlen) return false;
    uint64_t nch = (len + 7) >> 3;
    for (uint64_t c = 0; c < nch; c++) {
        uint64_t x = load8(t64, b + 8 * c) ^ load8(t64, rb + 8 * c);
        uint64_t rem = len - 8 * c;
        if (rem < 8) x &= (1ULL << (8 * rem)) - 1;
        if (x) return false;
    }
    return true;
}

__global__ void __launch_bounds__(256) exact_insert_kernel(const uint64_t* t64, uint64_t text_bytes, const uint64_t* ubeg,
                                                           const uint64_t* uend, const uint64_t* key, uint64_t n,
                                                           unsigned long long* tab, uint64_t tab_mask, unsigned long long* n_coll) {
    uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    uint64_t k = key[u];
    unsigned long long val = (k & 0xFFFFFFFF00000000ULL) | (uint32_t)u;
    uint64_t slot = k & tab_mask;
    uint64_t b = 0, len = 0;
    bool have_range = false;
    for (uint64_t probe = 0; probe <= tab_mask; probe++) {  // bounded: the table is at most half full
        unsigned long long cur = atomicCAS(&tab[slot], EMPTY, val);
        if (cur == EMPTY) return;
        if ((cur >> 32) == (val >> 32)) {
            if (!have_range) {
                unit_range(ubeg, uend, u, text_bytes, b, len);
                have_range = true;
            }
            uint64_t rb, rlen;
            unit_range(ubeg, uend, (uint32_t)cur, text_bytes, rb, rlen);
            if (same_unit(t64, b, len, rb, rlen)) {
                if (cur > val) atomicMin(&tab[slot], val);
                return;
            }
            atomicAdd(n_coll, 1ULL);  // same tag, different bytes: statistics only
        }
        slot = (slot + 1) & tab_mask;
    }
}

__global__ void __launch_bounds__(256) exact_link_kernel(const uint64_t* t64, uint64_t text_bytes, const uint64_t* ubeg,
                                                         const uint64_t* uend, const uint64_t* key, uint64_t n,
                                                         const unsigned long long* tab, uint64_t tab_mask, uint32_t* first_of) {
    uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    uint64_t k = key[u];
    uint64_t slot = k & tab_mask;
    uint32_t out = (uint32_t)u;
    uint64_t b = 0, len = 0;
    bool have_range = false;
    for (uint64_t probe = 0; probe <= tab_mask; probe++) {
        unsigned long long cur = tab[slot];
        if (cur == EMPTY) break;  // cannot happen after the insert pass; defensive
        if ((cur >> 32) == (k >> 32)) {
            if ((uint32_t)cur == (uint32_t)u) break;  // this unit is the first of its group
            if (!have_range) {
                unit_range(ubeg, uend, u, text_bytes, b, len);
                have_range = true;
            }
            uint64_t rb, rlen;
            unit_range(ubeg, uend, (uint32_t)cur, text_bytes, rb, rlen);
            if (same_unit(t64, b, len, rb, rlen)) {
                out = (uint32_t)cur;
                break;
            }
        }
        slot = (slot + 1) & tab_mask;
    }
    first_of[u] = out;
}

}  // namespace

static int exact_group_impl(const char* who, mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* ubeg, const uint64_t* uend,
                            uint64_t n_units, uint64_t avg_unit_bytes, int key_bits, uint64_t* key_scratch, uint64_t* table, uint64_t tab_size,
                            uint32_t* first_of, uint64_t* n_coll, uint32_t* unit_cps) {
    REQUIRE(ctx && n_coll, "%s: NULL argument", who);
    REQUIRE(key_bits >= 1 && key_bits <= 64, "%s: key_bits=%d outside [1,64]", who, key_bits);
    REQUIRE(n_units <= MAX_ELEMS_PER_LAUNCH, "%s: n_units=%llu exceeds 2^32-256", who, (unsigned long long)n_units);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(n_coll, 0, sizeof(uint64_t), ctx->stream));
    if (n_units == 0) return MDD_OK;
    REQUIRE(text && ubeg && uend && key_scratch && table && first_of, "%s: NULL buffer", who);
    REQUIRE((reinterpret_cast<uintptr_t>(text) & 7u) == 0, "%s: text must be 8-byte aligned", who);
    REQUIRE(tab_size >= 2 && (tab_size & (tab_size - 1)) == 0 && tab_size >= 2 * n_units,
            "%s: tab_size=%llu must be a power of two >= 2*n_units", who, (unsigned long long)tab_size);
    const uint64_t* t64 = reinterpret_cast<const uint64_t*>(text);
    if (avg_unit_bytes >= 64 && avg_unit_bytes < 2048) {
        uint64_t blocks = (n_units + 31) / 32;
        if (blocks > (1u << 20)) blocks = 1u << 20;
        hipLaunchKernelGGL(exact_hash_group8_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t64, text_bytes, ubeg,
                           uend, n_units, key_bits, key_scratch, unit_cps);
    } else if (avg_unit_bytes >= 256) {
        uint64_t blocks = (n_units + 3) / 4;
        if (blocks > (1u << 22)) blocks = 1u << 22;  // 2^30 work-items: well inside the 2^32 per-launch limit
        hipLaunchKernelGGL(exact_hash_wave_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t64, text_bytes, ubeg,
                           uend, n_units, key_bits, key_scratch, unit_cps);
    } else {
        hipLaunchKernelGGL(exact_hash_thread_kernel, dim3(grid_for(n_units, 256)), dim3(256), 0, ctx->stream, t64, text_bytes, ubeg,
                           uend, n_units, key_bits, key_scratch, unit_cps);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(table, 0xFF, sizeof(uint64_t) * tab_size, ctx->stream));
    unsigned g = grid_for(n_units, 256);
    hipLaunchKernelGGL(exact_insert_kernel, dim3(g), dim3(256), 0, ctx->stream, t64, text_bytes, ubeg, uend, key_scratch, n_units,
                       reinterpret_cast<unsigned long long*>(table), tab_size - 1, reinterpret_cast<unsigned long long*>(n_coll));
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(exact_link_kernel, dim3(g), dim3(256), 0, ctx->stream, t64, text_bytes, ubeg, uend, key_scratch, n_units,
                       reinterpret_cast<const unsigned long long*>(table), tab_size - 1, first_of);
    HIP_TRY(hipGetLastError());
    return MDD_OK;
}

extern "C" int mdd_exact_group(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* ubeg, const uint64_t* uend,
                               uint64_t n_units, uint64_t avg_unit_bytes, int key_bits, uint64_t* key_scratch, uint64_t* table,
                               uint64_t tab_size, uint32_t* first_of, uint64_t* n_coll) {
    return exact_group_impl("mdd_exact_group", ctx, text, text_bytes, ubeg, uend, n_units, avg_unit_bytes, key_bits, key_scratch, table, tab_size,
                            first_of, n_coll, nullptr);
}

extern "C" int mdd_exact_group_cps(mdd_ctx* ctx, const uint8_t* text, uint64_t text_bytes, const uint64_t* ubeg, const uint64_t* uend,
                                   uint64_t n_units, uint64_t avg_unit_bytes, int key_bits, uint64_t* key_scratch, uint64_t* table,
                                   uint64_t tab_size, uint32_t* first_of, uint64_t* n_coll, uint32_t* unit_cps) {
    REQUIRE(unit_cps || n_units == 0, "mdd_exact_group_cps: unit_cps is NULL");
    return exact_group_impl("mdd_exact_group_cps", ctx, text, text_bytes, ubeg, uend, n_units, avg_unit_bytes, key_bits, key_scratch, table,
                            tab_size, first_of, n_coll, unit_cps);
}
