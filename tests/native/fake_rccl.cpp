// TEST INFRASTRUCTURE -- a loop-back stand-in for the nine RCCL entry points csrc/mdd_comm.hip binds with dlopen
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclAllGather,
// ncclGetErrorString), so that the library's multi-rank exchange code can run with W > 1 peers on a ONE-GPU box:
// every "rank" is a thread of one process with its own mdd context (and stream) on the same device; a send meets its
// recv in a rendez-vous table and the payload moves with a device-to-device copy.  Selected with the documented test
// hook MDD_RCCL_LIB=<path to this library> (mdd_comm.hip: load_rccl).  Nothing under map_neo_amd/ refers to it.
//
// It is deliberately STRICTER than RCCL, so that index bugs in the caller fail loudly instead of corrupting silently:
//   * the k-th recv a rank posts from peer p is matched with the k-th send p posted to that rank, and their byte counts
//     must be equal (RCCL: undefined behaviour) -> ncclInvalidArgument;
//   * a wait that lasts longer than FAKE_RCCL_TIMEOUT_S (default 30) returns ncclSystemError with a diagnostic instead of
//     hanging;
//   * ncclSend / ncclRecv outside any group behave like a group of one; nested groups are counted; a thread that
//     leaves a group open is visible through fake_rccl_group_depth();
//   * an operation that fails inside a group poisons the group: ncclGroupEnd discards what was queued and returns the
//     error (what NCCL >= 2.14 does); fake_rccl_fail_send(n) makes the n-th ncclSend of EVERY thread fail, for tests of
//     the caller's error path.
// Semantics kept from RCCL: operations are ordered with the stream they are given (the stream is synchronised before a
// send buffer is published and after a recv buffer is filled -- a host-blocking simplification of stream ordering), and
// a rank's own ncclGroupEnd returns only when its sends have been consumed, so send buffers may be reused afterwards.
// With FAKE_RCCL_HOSTCOPY=1 the payload moves with memcpy and no HIP call is made (self-test of this file on a machine
// without a GPU: tests/test_fake_rccl.py).
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int RING = 256;

struct Posted {
    const void* ptr;
    size_t bytes;
};

struct Pair {  // the sends rank s has posted for rank d, in order
    Posted ring[RING];
    uint64_t posted = 0, consumed = 0;
};

struct World {
    int nranks = 0, joined = 0, left = 0;
    std::vector<Pair> pairs;  // [s * nranks + d]
    std::mutex mu;
    std::condition_variable cv;
};

struct Comm {
    World* world;
    int rank;
};

struct Op {
    bool is_send;
    const void* sptr;
    void* rptr;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};

std::mutex g_mu;
std::map<uint64_t, World*> g_worlds;
uint64_t g_next_id = 1;

thread_local int t_depth = 0;
thread_local ncclResult_t t_poison = ncclSuccess;
thread_local std::vector<Op> t_ops;
thread_local uint64_t t_sends_seen = 0;

// statistics / fault injection (process-wide)
std::mutex g_stat_mu;
uint64_t g_stats[6] = {0, 0, 0, 0, 0, 0};  // groups with work, sends, recvs, bytes sent, largest message, allgathers
uint64_t g_fail_send = 0;                   // 0 = never

int g_hostcopy = -1;  // -1: ask the environment every time; fake_rccl_set_hostcopy overrides
bool host_copy() {
    if (g_hostcopy >= 0) return g_hostcopy == 1;
    const char* e = getenv("FAKE_RCCL_HOSTCOPY");
    return e && e[0] == '1';
}

double timeout_s() {
    const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
    double t = e ? atof(e) : 30.0;
    return t > 0 ? t : 30.0;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: case ncclFloat8e4m3: case ncclFloat8e5m2: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t sync_stream(hipStream_t s) {
    if (host_copy()) return ncclSuccess;
    return hipStreamSynchronize(s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t copy_payload(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return ncclSuccess;
    if (host_copy()) {
        memcpy(dst, src, bytes);
        return ncclSuccess;
    }
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return hipStreamSynchronize(s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

template <class Pred>
bool wait_for(World* w, std::unique_lock<std::mutex>& lk, Pred pred) {
    return w->cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), pred);
}

// run the queued operations of this thread: publish every send, then fill every recv, then wait for the sends to be taken
ncclResult_t run_ops(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    uint64_t n_s = 0, n_r = 0, bytes = 0, biggest = 0;
    for (const Op& o : ops) {
        ncclResult_t r = sync_stream(o.stream);
        if (r != ncclSuccess) return r;
    }
    for (const Op& o : ops) {
        if (!o.is_send) continue;
        World* w = o.comm->world;
        std::unique_lock<std::mutex> lk(w->mu);
        Pair& p = w->pairs[(size_t)o.comm->rank * w->nranks + o.peer];
        if (p.posted - p.consumed >= RING) {
            fprintf(stderr, "fake_rccl: rank %d has %d unconsumed sends for rank %d\n", o.comm->rank, RING, o.peer);
            return ncclInternalError;
        }
        p.ring[p.posted % RING] = Posted{o.sptr, o.bytes};
        p.posted++;
        n_s++;
        bytes += o.bytes;
        biggest = biggest > o.bytes ? biggest : o.bytes;
        w->cv.notify_all();
    }
    for (const Op& o : ops) {
        if (o.is_send) continue;
        World* w = o.comm->world;
        Posted got;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            Pair& p = w->pairs[(size_t)o.peer * w->nranks + o.comm->rank];
            if (!wait_for(w, lk, [&] { return p.posted > p.consumed; })) {
                fprintf(stderr, "fake_rccl: rank %d waited %.0f s for a send of rank %d (%zu bytes expected): unmatched recv\n",
                        o.comm->rank, timeout_s(), o.peer, o.bytes);
                return ncclSystemError;
            }
            got = p.ring[p.consumed % RING];
        }
        if (got.bytes != o.bytes) {
            fprintf(stderr, "fake_rccl: rank %d posted a recv of %zu bytes from rank %d whose matching send has %zu bytes\n",
                    o.comm->rank, o.bytes, o.peer, got.bytes);
            rc = ncclInvalidArgument;
        } else {
            ncclResult_t r = copy_payload(o.rptr, got.ptr, o.bytes, o.stream);
            if (r != ncclSuccess) rc = r;
        }
        {
            std::unique_lock<std::mutex> lk(w->mu);
            Pair& p = w->pairs[(size_t)o.peer * w->nranks + o.comm->rank];
            p.consumed++;  // taken (or refused): the sender may go on either way
            w->cv.notify_all();
        }
        n_r++;
    }
    for (const Op& o : ops) {
        if (!o.is_send) continue;
        World* w = o.comm->world;
        std::unique_lock<std::mutex> lk(w->mu);
        Pair& p = w->pairs[(size_t)o.comm->rank * w->nranks + o.peer];
        if (!wait_for(w, lk, [&] { return p.consumed == p.posted; })) {
            fprintf(stderr, "fake_rccl: rank %d waited %.0f s for rank %d to take a send (%zu bytes): unmatched send\n", o.comm->rank,
                    timeout_s(), o.peer, o.bytes);
            return ncclSystemError;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_stat_mu);
        g_stats[0] += 1;
        g_stats[1] += n_s;
        g_stats[2] += n_r;
        g_stats[3] += bytes;
        g_stats[4] = g_stats[4] > biggest ? g_stats[4] : biggest;
    }
    return rc;
}

ncclResult_t enqueue(const Op& o) {
    if (t_depth > 0) {
        if (t_poison == ncclSuccess) t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_ops(one);
}

ncclResult_t check_p2p(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !c->world || type_bytes(dt) == 0) return ncclInvalidArgument;
    if (peer < 0 || peer >= c->world->nranks) return ncclInvalidArgument;
    if (count && !buf) return ncclInvalidArgument;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof *id);
    uint64_t v = g_next_id++;
    memcpy(id->internal, "FAKERCCL", 8);
    memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8) != 0) return ncclInvalidArgument;
    uint64_t v;
    memcpy(&v, id.internal + 8, 8);
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_worlds.find(v);
        if (it == g_worlds.end()) {
            w = new World;
            w->nranks = nranks;
            w->pairs.resize((size_t)nranks * nranks);
            g_worlds[v] = w;
        } else {
            w = it->second;
        }
    }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->nranks != nranks) return ncclInvalidArgument;
    w->joined++;
    w->cv.notify_all();
    if (!wait_for(w, lk, [&] { return w->joined >= w->nranks; })) {  // collective, like the real one
        fprintf(stderr, "fake_rccl: rank %d waited %.0f s in ncclCommInitRank: %d of %d ranks joined\n", rank, timeout_s(), w->joined, nranks);
        return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    delete c;  // the World stays in the table: a few hundred bytes per test communicator
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    ncclResult_t poison = t_poison;
    t_poison = ncclSuccess;
    if (poison != ncclSuccess) return poison;  // the group is dropped as a whole
    return run_ops(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    ncclResult_t r = check_p2p(buf, count, dt, peer, comm);
    uint64_t fail_at;
    {
        std::lock_guard<std::mutex> lk(g_stat_mu);
        fail_at = g_fail_send;
    }
    if (r == ncclSuccess && fail_at && ++t_sends_seen == fail_at) r = ncclInternalError;
    if (r != ncclSuccess) {
        if (t_depth > 0) t_poison = r;
        return r;
    }
    return enqueue(Op{true, buf, nullptr, count * type_bytes(dt), peer, reinterpret_cast<Comm*>(comm), stream});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    ncclResult_t r = check_p2p(buf, count, dt, peer, comm);
    if (r != ncclSuccess) {
        if (t_depth > 0) t_poison = r;
        return r;
    }
    return enqueue(Op{false, nullptr, buf, count * type_bytes(dt), peer, reinterpret_cast<Comm*>(comm), stream});
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !c->world || type_bytes(dt) == 0 || (sendcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    const size_t bytes = sendcount * type_bytes(dt);
    const int W = c->world->nranks;
    ncclResult_t r = sync_stream(stream);
    if (r != ncclSuccess) return r;
    char* out = static_cast<char*>(recvbuff);
    if (out + (size_t)c->rank * bytes != sendbuff) {  // in-place when equal
        r = copy_payload(out + (size_t)c->rank * bytes, sendbuff, bytes, stream);
        if (r != ncclSuccess) return r;
    }
    std::vector<Op> ops;
    for (int p = 0; p < W; p++)
        if (p != c->rank) ops.push_back(Op{true, sendbuff, nullptr, bytes, p, c, stream});
    for (int p = 0; p < W; p++)
        if (p != c->rank) ops.push_back(Op{false, nullptr, out + (size_t)p * bytes, bytes, p, c, stream});
    {
        std::lock_guard<std::mutex> lk(g_stat_mu);
        g_stats[5] += 1;
    }
    if (t_depth > 0) {
        if (t_poison == ncclSuccess) t_ops.insert(t_ops.end(), ops.begin(), ops.end());
        return ncclSuccess;
    }
    return run_ops(ops);
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake_rccl: no error";
        case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
        case ncclSystemError: return "fake_rccl: timed out waiting for a peer";
        case ncclInternalError: return "fake_rccl: internal error (or injected fault)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument (or send/recv size mismatch)";
        case ncclInvalidUsage: return "fake_rccl: invalid usage";
        default: return "fake_rccl: error";
    }
}

// ---- test-side helpers (not RCCL symbols) ----
void fake_rccl_stats(uint64_t* out6, int reset) {
    std::lock_guard<std::mutex> lk(g_stat_mu);
    if (out6) memcpy(out6, g_stats, sizeof g_stats);
    if (reset) memset(g_stats, 0, sizeof g_stats);
}
int fake_rccl_group_depth(void) { return t_depth; }
// the n-th ncclSend (counted per thread, from this call on in every thread that has not sent yet) fails; 0 switches it off
void fake_rccl_fail_send(uint64_t n) {
    std::lock_guard<std::mutex> lk(g_stat_mu);
    g_fail_send = n;
}
void fake_rccl_reset_thread(void) { t_sends_seen = 0; }
void fake_rccl_set_hostcopy(int v) { g_hostcopy = v; }

}  // extern "C"
