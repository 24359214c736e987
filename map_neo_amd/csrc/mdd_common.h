// Shared by the translation units of libmdd_hip.so (mdd_hip.hip, mdd_exact.hip).
#ifndef MDD_COMMON_H
#define MDD_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mdd.h"

struct ReadyNote {
    volatile uint64_t* flag;
    uint64_t value;
};

struct mdd_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    int cus;
    // long-document segment plan (filled by seg_plan_kernel on every mdd_signatures call)
    uint64_t* seg_items;    // device, SEG_ITEM_CAP entries of (doc << 24 | segment)
    uint64_t* seg_n_items;  // device: [0] queued segments, [1] work counter of sig_kernel, [2] windows whose probe loop ran out (mdd_substr_check)
    // streamed input (mdd_signatures_from_host): device words the kernel polls / reports through, and the pinned host
    // table the "bytes ready" values are copied from (one 8-byte H2D copy after each text chunk, on the copy stream)
    uint64_t* stream_words;     // device: [0] = highest count a wave has read from the host flag, [1] = status (non-zero: a wave gave up waiting), [2] = poller lock
    volatile uint64_t* ready_host;  // pinned host flag word: stream_base of the call + bytes of its text that have landed (only ever grows)
    uint64_t* ready_flag_dev;   // device-side address of that word
    uint64_t stream_base;       // flag value at which the NEXT call's text has 0 bytes
    ReadyNote* notes;           // host: arguments of the stream-ordered host functions (2 * READY_SLOTS, alternating halves per call)
    hipEvent_t stream_ev[2];    // end of the copy-stream work of the call that last used each half
    hipEvent_t order_ev;        // orders the copy stream behind the compute stream at the start of a call
    hipStream_t note_stream;    // the host functions that publish the byte counts run here, BEHIND the copy stream (an event per
                                // chunk), not on it: a late host thread then delays a count, never the next copy
    hipEvent_t* chunk_ev;       // 2 * READY_SLOTS events, created on first use (n_chunk_ev[half] of each half exist)
    uint32_t n_chunk_ev[2];
    uint32_t stream_calls;      // calls so far (selects the half)
    bool stream_pending;        // a streamed call was enqueued since the last mdd_ctx_sync
    // multi-GPU exchange (mdd_comm.hip): RCCL communicator (ncclComm_t) + staging for the counts
    void* comm;
    int comm_rank, comm_world;
    uint64_t* comm_words;       // device, 128 words
    uint64_t* comm_words_host;  // pinned host mirror
    uint64_t* static_scratch;  // device address of this GPU's file-scope scratch array (mdd_signatures_noalloc)
    uint32_t seg_bytes;     // segment length in bytes (multiple of 256)
    uint32_t filter_from;   // sig_kernel filters its permute phase after this many shingles of a work item (0 = per-K default)
};

namespace mddi {
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// The low nb (0..4) bytes of v to out[pos ..), nothing at or beyond cap.  Four bytes go out as ONE dword store at a byte
// address (the target runs with unaligned global access enabled -- the compiler itself relies on it and emits a single
// global_store_dword here); the write kernels with a dword per lane depend on this: 64 lanes x 4 consecutive bytes.
typedef uint32_t mdd_u32_a1 __attribute__((aligned(1)));
__device__ __forceinline__ void store_low_bytes(uint8_t* __restrict__ out, uint64_t pos, uint64_t cap, uint32_t v, uint32_t nb) {
    if (nb == 4u && pos + 4 <= cap) {
        *reinterpret_cast<mdd_u32_a1*>(out + pos) = v;
        return;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
        if (k < nb && pos + k < cap) out[pos + k] = (uint8_t)(v >> (8 * k));
}
__device__ __forceinline__ uint32_t below(uint64_t m) {  // set bits of m in lanes below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// Document d = [beg, end) of the text.  offsets live in device memory and cannot be validated on the host
// without a synchronisation, so every kernel clamps them instead: end <= text_bytes, beg <= end.  Malformed
// offsets then give meaningless signatures but never an access outside the caller's allocation.
__device__ __forceinline__ void doc_range(const uint64_t* offsets, uint64_t d, uint64_t text_bytes, uint64_t& beg, uint64_t& end) {
    beg = offsets[d];
    end = offsets[d + 1];
    end = end < text_bytes ? end : text_bytes;
    beg = beg < end ? beg : end;
}

// one launch holds at most 2^32 - 1 work-items: thread-per-element launches (256-thread blocks) are checked against this
constexpr uint64_t MAX_ELEMS_PER_LAUNCH = 0xFFFFFF00ull;
inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }
}  // namespace mddi

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return mddi::fail(MDD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define REQUIRE(cond, ...)                                            \
    do {                                                              \
        if (!(cond)) return mddi::fail(MDD_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#endif  // MDD_COMMON_H
