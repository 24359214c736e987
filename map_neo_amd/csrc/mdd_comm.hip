// Multi-GPU exchange of libmdd_hip.so: the band-record all-to-all and the edge all-gather,
// issued by the library itself against RCCL on the context's stream.  The reference has no counterpart: it leaves the
// shuffle to Spark (EN:136 groupBy, EN:140-142) or to per-band directories on a shared filesystem (ZH-1:58-61).
//
// RCCL is bound at run time (dlopen), not at link time: a process that also uses torch must use the RCCL (and through
// it the HIP runtime) torch has already loaded -- a second copy of either would not share devices and streams.  A
// single-GPU user never loads it.
//
// One host synchronisation per exchange is inherent: ncclSend / ncclRecv take their sizes on the host, and the sizes
// (how many records hash to each owner, how many edges each rank found) are computed on the device.  The two *_counts
// calls move those few numbers (grouped 8-byte send/recv or one ncclAllGather, one small D2H copy, one stream sync) and
// are the only synchronous steps; the payload calls are asynchronous on the context's stream.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "mdd_common.h"

namespace {
using namespace mddi;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

// No single message exceeds this many bytes: RCCL on this stack was seen to drop the tail of an exchange beyond 1 GiB
// (tools/diag_a2a.py); larger transfers go in rounds.  Both ends of a pair know the pair's size, so they cut it alike.
// TEST HOOK: MDD_COMM_MSG_BYTES (>= 64) lowers the limit so that small exchanges run in several rounds.
uint64_t g_msg_bytes = 512ull << 20;

int load_rccl() {
    if (g_rccl.handle) return MDD_OK;
    void* h = nullptr;
    if (const char* e = getenv("MDD_COMM_MSG_BYTES")) {
        unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 64 && v <= (512ull << 20)) g_msg_bytes = v;
    }
    // TEST HOOK: MDD_RCCL_LIB names the library to bind instead of RCCL (tests/native/fake_rccl.cpp: a loop-back
    // transport that lets W > 1 ranks run as threads on one GPU).  Not for production use.
    const char* forced = getenv("MDD_RCCL_LIB");
    if (forced && forced[0]) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(MDD_ERR_UNSUPPORTED, "mdd_comm: cannot load MDD_RCCL_LIB=%s (%s)", forced, dlerror());
    }
    // an already loaded RCCL first (torch's), then the usual names
    if (!h)
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h)
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) return fail(MDD_ERR_UNSUPPORTED, "mdd_comm: cannot load librccl.so (%s)", dlerror());
    bool ok = true;
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        if (!p) ok = false;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) return fail(MDD_ERR_UNSUPPORTED, "mdd_comm: librccl.so lacks a required symbol");
    g_rccl.handle = h;
    return MDD_OK;
}

#define NCCL_TRY(expr)                                                                                               \
    do {                                                                                                             \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess)                                                                                       \
            return mddi::fail(MDD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Inside an open ncclGroupStart: a failing call must not leave the communicator in group mode (later collectives of
// this process -- torch's included, it is the same RCCL -- would queue behind a group that never ends).  ncclGroupEnd
// after a failed call drops the group and reports the error again; its result is ignored here.
#define NCCL_GROUP_TRY(expr)                                                                                         \
    do {                                                                                                             \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess) {                                                                                     \
            (void)g_rccl.GroupEnd();                                                                                 \
            return mddi::fail(MDD_ERR_HIP, "%s failed inside a group: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
        }                                                                                                            \
    } while (0)

int need_comm(const char* fn, mdd_ctx* ctx) {
    REQUIRE(ctx != nullptr, "%s: ctx is NULL", fn);
    REQUIRE(ctx->comm != nullptr, "%s: call mdd_comm_init first", fn);
    return MDD_OK;
}

// grouped variable all-to-all of `narr` arrays that share their per-peer element counts (scaled by width[a] elements of
// elem[a] bytes): array a of peer p starts at (exclusive prefix of counts)[p] * width[a] elements
int alltoallv(mdd_ctx* ctx, int narr, const void* const* send, void* const* recv, const uint32_t* width, const uint32_t* elem,
              const uint64_t* send_counts, const uint64_t* recv_counts) {
    const int W = ctx->comm_world, me = ctx->comm_rank;
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    std::vector<uint64_t> soff(W + 1, 0), roff(W + 1, 0);
    for (int p = 0; p < W; p++) {
        soff[p + 1] = soff[p] + send_counts[p];
        roff[p + 1] = roff[p] + recv_counts[p];
    }
    REQUIRE(send_counts[me] == recv_counts[me], "mdd_comm: a rank's count of records for itself differs between send and recv");
    uint32_t rec_bytes = 0;
    for (int a = 0; a < narr; a++) rec_bytes = rec_bytes > width[a] * elem[a] ? rec_bytes : width[a] * elem[a];
    const uint64_t per_round = g_msg_bytes / rec_bytes ? g_msg_bytes / rec_bytes : 1;  // records of one pair per round
    uint64_t most = 0;
    for (int p = 0; p < W; p++) {
        if (p == me) continue;
        most = most > send_counts[p] ? most : send_counts[p];
        most = most > recv_counts[p] ? most : recv_counts[p];
    }
    // this rank's own records never leave the device
    for (int a = 0; a < narr; a++) {
        uint64_t bytes = send_counts[me] * width[a] * elem[a];
        if (bytes)
            HIP_TRY(hipMemcpyAsync(static_cast<char*>(recv[a]) + roff[me] * width[a] * elem[a],
                                   static_cast<const char*>(send[a]) + soff[me] * width[a] * elem[a], bytes, hipMemcpyDeviceToDevice,
                                   ctx->stream));
    }
    for (uint64_t done = 0; done < most; done += per_round) {
        NCCL_TRY(g_rccl.GroupStart());
        for (int p = 0; p < W; p++) {
            if (p == me) continue;
            uint64_t ns = send_counts[p] > done ? (send_counts[p] - done < per_round ? send_counts[p] - done : per_round) : 0;
            uint64_t nr = recv_counts[p] > done ? (recv_counts[p] - done < per_round ? recv_counts[p] - done : per_round) : 0;
            for (int a = 0; a < narr; a++) {
                const uint64_t eb = (uint64_t)width[a] * elem[a];
                if (ns) NCCL_GROUP_TRY(g_rccl.Send(static_cast<const char*>(send[a]) + (soff[p] + done) * eb, ns * eb, ncclUint8, p, comm, ctx->stream));
                if (nr) NCCL_GROUP_TRY(g_rccl.Recv(static_cast<char*>(recv[a]) + (roff[p] + done) * eb, nr * eb, ncclUint8, p, comm, ctx->stream));
            }
        }
        NCCL_TRY(g_rccl.GroupEnd());
    }
    return MDD_OK;
}

}  // namespace

extern "C" {

int mdd_comm_unique_id(uint8_t* id_host) {
    REQUIRE(id_host != nullptr, "mdd_comm_unique_id: NULL output");
    static_assert(sizeof(ncclUniqueId) == MDD_COMM_ID_BYTES, "MDD_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_host, &id, sizeof id);
    return MDD_OK;
}

int mdd_comm_init(mdd_ctx* ctx, const uint8_t* id_host, int rank, int world) {
    REQUIRE(ctx && id_host, "mdd_comm_init: NULL argument");
    REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "mdd_comm_init: rank %d / world %d out of range (world <= 64)", rank, world);
    REQUIRE(ctx->comm == nullptr, "mdd_comm_init: the context already has a communicator");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof id);
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    // staging for the few numbers the *_counts calls move: device words + their pinned host mirror
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->comm_words), 2 * 64 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->comm_words_host), 2 * 64 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)g_rccl.CommDestroy(comm);
        if (ctx->comm_words) (void)hipFree(ctx->comm_words);
        ctx->comm_words = nullptr;
        return fail(MDD_ERR_HIP, "mdd_comm_init: staging allocation failed: %s", hipGetErrorString(e));
    }
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return MDD_OK;
}

int mdd_comm_destroy(mdd_ctx* ctx) {
    REQUIRE(ctx != nullptr, "mdd_comm_destroy: ctx is NULL");
    if (!ctx->comm) return MDD_OK;
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    if (ctx->comm_words) (void)hipFree(ctx->comm_words);
    if (ctx->comm_words_host) (void)hipHostFree(ctx->comm_words_host);
    ctx->comm_words = ctx->comm_words_host = nullptr;
    return MDD_OK;
}

int mdd_exchange_counts(mdd_ctx* ctx, const uint64_t* send_counts, uint64_t* send_counts_host, uint64_t* recv_counts_host) {
    int rc = need_comm("mdd_exchange_counts", ctx);
    if (rc) return rc;
    REQUIRE(send_counts && send_counts_host && recv_counts_host, "mdd_exchange_counts: NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int W = ctx->comm_world, me = ctx->comm_rank;
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    uint64_t* rdev = ctx->comm_words;  // [0, W): what the peers will send here
    HIP_TRY(hipMemcpyAsync(rdev + me, send_counts + me, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    if (W > 1) {
        NCCL_TRY(g_rccl.GroupStart());
        for (int p = 0; p < W; p++) {
            if (p == me) continue;
            NCCL_GROUP_TRY(g_rccl.Send(send_counts + p, 1, ncclUint64, p, comm, ctx->stream));
            NCCL_GROUP_TRY(g_rccl.Recv(rdev + p, 1, ncclUint64, p, comm, ctx->stream));
        }
        NCCL_TRY(g_rccl.GroupEnd());
    }
    uint64_t* h = ctx->comm_words_host;
    HIP_TRY(hipMemcpyAsync(h, rdev, W * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(h + 64, send_counts, W * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the one host synchronisation of the exchange
    for (int p = 0; p < W; p++) {
        recv_counts_host[p] = h[p];
        send_counts_host[p] = h[64 + p];
    }
    return MDD_OK;
}

int mdd_exchange_bands(mdd_ctx* ctx, const uint32_t* tuples_send, const uint64_t* meta_send, const uint64_t* send_counts_host,
                       const uint64_t* recv_counts_host, int rows, uint32_t* tuples_recv, uint64_t* meta_recv) {
    int rc = need_comm("mdd_exchange_bands", ctx);
    if (rc) return rc;
    REQUIRE(send_counts_host && recv_counts_host && rows >= 1 && rows <= 256, "mdd_exchange_bands: bad argument");
    uint64_t ns = 0, nr = 0;
    for (int p = 0; p < ctx->comm_world; p++) {
        ns += send_counts_host[p];
        nr += recv_counts_host[p];
    }
    REQUIRE((ns == 0 || (tuples_send && meta_send)) && (nr == 0 || (tuples_recv && meta_recv)), "mdd_exchange_bands: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    const void* send[2] = {tuples_send, meta_send};
    void* recv[2] = {tuples_recv, meta_recv};
    const uint32_t width[2] = {(uint32_t)rows, 1u}, elem[2] = {4u, 8u};
    return alltoallv(ctx, 2, send, recv, width, elem, send_counts_host, recv_counts_host);  // tuples + meta: one group per round
}

int mdd_allgather_counts(mdd_ctx* ctx, const uint64_t* n_local, uint64_t* counts_host) {
    int rc = need_comm("mdd_allgather_counts", ctx);
    if (rc) return rc;
    REQUIRE(n_local && counts_host, "mdd_allgather_counts: NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int W = ctx->comm_world;
    uint64_t* dev = ctx->comm_words + 64;
    NCCL_TRY(g_rccl.AllGather(n_local, dev, 1, ncclUint64, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->comm_words_host, dev, W * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < W; p++) counts_host[p] = ctx->comm_words_host[p];
    return MDD_OK;
}

int mdd_allgather_edges(mdd_ctx* ctx, const uint32_t* edge_u, const uint32_t* edge_v, const uint64_t* counts_host, uint32_t* all_u,
                        uint32_t* all_v) {
    int rc = need_comm("mdd_allgather_edges", ctx);
    if (rc) return rc;
    REQUIRE(counts_host != nullptr, "mdd_allgather_edges: counts_host is NULL");
    const int W = ctx->comm_world, me = ctx->comm_rank;
    uint64_t total = 0;
    for (int p = 0; p < W; p++) total += counts_host[p];
    REQUIRE((counts_host[me] == 0 || (edge_u && edge_v)) && (total == 0 || (all_u && all_v)), "mdd_allgather_edges: NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    // an all-gather of unequal pieces = an all-to-all in which every peer gets the same piece: the send side repeats this
    // rank's edges for every peer (offset 0 each time), which alltoallv's contiguous layout cannot express -- so the
    // groups are issued here
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    std::vector<uint64_t> off(W + 1, 0);
    for (int p = 0; p < W; p++) off[p + 1] = off[p] + counts_host[p];
    const uint64_t mine = counts_host[me];
    if (mine) {
        HIP_TRY(hipMemcpyAsync(all_u + off[me], edge_u, mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(all_v + off[me], edge_v, mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    const uint64_t per_round = g_msg_bytes / 4;
    uint64_t most = 0;
    for (int p = 0; p < W; p++) most = most > counts_host[p] ? most : counts_host[p];
    for (uint64_t done = 0; W > 1 && done < most; done += per_round) {
        NCCL_TRY(g_rccl.GroupStart());
        uint64_t ns = mine > done ? (mine - done < per_round ? mine - done : per_round) : 0;
        for (int p = 0; p < W; p++) {
            if (p == me) continue;
            uint64_t nr = counts_host[p] > done ? (counts_host[p] - done < per_round ? counts_host[p] - done : per_round) : 0;
            if (ns) {
                NCCL_GROUP_TRY(g_rccl.Send(edge_u + done, ns, ncclUint32, p, comm, ctx->stream));
                NCCL_GROUP_TRY(g_rccl.Send(edge_v + done, ns, ncclUint32, p, comm, ctx->stream));
            }
            if (nr) {
                NCCL_GROUP_TRY(g_rccl.Recv(all_u + off[p] + done, nr, ncclUint32, p, comm, ctx->stream));
                NCCL_GROUP_TRY(g_rccl.Recv(all_v + off[p] + done, nr, ncclUint32, p, comm, ctx->stream));
            }
        }
        NCCL_TRY(g_rccl.GroupEnd());
    }
    return MDD_OK;
}

}  // extern "C"
