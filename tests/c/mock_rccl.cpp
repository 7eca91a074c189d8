// tests/c/mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose ranks are threads of ONE process on ONE device.
//
// RCCL refuses two ranks on the same GPU, and the test box has one GPU, so the RCCL branch of oatk_amd/csrc/api_multi.inc (grouped
// ncclSend / ncclRecv, grouped ncclBroadcast, ncclAllGather, ncclAllReduce with the offsets and counts the library computes) would
// otherwise only ever run with a world of one.  This library implements the handful of entry points that branch uses with the semantics
// the RCCL headers document -- operations inside ncclGroupStart / ncclGroupEnd take effect at ncclGroupEnd, the k-th send from a to b
// matches the k-th receive at b from a, in-place all-reduce -- over device-to-device copies and a barrier.  It is loaded through
// OATK_RCCL_LIB (a test hook of rccl_load) by tests/test_gpu_multi_c.py; nothing in the product links or loads it.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
}

namespace {

struct Op { int kind; const void *send; void *recv; size_t bytes; int peer; int dtype, red; size_t count; };      // kind: 0 send, 1 recv, 2 bcast, 3 allgather, 4 allreduce
struct Group {
    int n = 0, joined = 0, arrived = 0, phase = 0;
    bool aborted = false;                   // ncclCommAbort on any rank's communicator: everybody waiting returns an error, nobody waits again
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::vector<Op>> ops;       // per rank, the operations of the group being executed
    bool barrier()                          // false: the group was aborted
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const int ph = phase;
        if (++arrived == n) { arrived = 0, ++phase; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph || aborted; });
        return !aborted;
    }
    void abort()
    {
        std::unique_lock<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};
struct Comm { Group *g; int rank; };
std::mutex g_mu;
std::map<uint64_t, Group *> g_groups;
uint64_t g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local Comm *t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
thread_local Comm *t_mine = nullptr;          // the communicator of this thread (a rank is a thread here): an EMPTY group still meets the others' barriers

size_t width(int dt) { return dt <= 1? 1 : (dt <= 3? 4 : 8); }

ncclResult_t run(Comm *c, hipStream_t st, std::vector<Op> &mine)
{
    Group *g = c->g;
    if (st && hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;      // what the others are about to read is complete
    if (!st && hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    g->ops[c->rank] = mine;
    if (!g->barrier()) return ncclSystemError;
    ncclResult_t rc = ncclSuccess;
    std::vector<size_t> recv_seen(g->n, 0);
    for (size_t e = 0; e < mine.size() && rc == ncclSuccess; ++e) {
        const Op &o = mine[e];
        if (o.kind == 1) {                   // the k-th receive from `peer` takes the k-th send of `peer` to this rank
            size_t k = recv_seen[o.peer]++, seen = 0;
            const Op *match = nullptr;
            for (const Op &p : g->ops[o.peer]) if (p.kind == 0 && p.peer == c->rank && seen++ == k) { match = &p; break; }
            if (!match || match->bytes != o.bytes) { rc = ncclInvalidUsage; break; }
            if (o.bytes && hipMemcpy(o.recv, match->send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        } else if (o.kind == 2) {            // every rank lists the same broadcasts in the same order
            const Op &root = g->ops[o.peer][e];
            if (root.kind != 2 || root.peer != o.peer || root.bytes != o.bytes) { rc = ncclInvalidUsage; break; }
            if (o.bytes && o.recv != root.send && hipMemcpy(o.recv, root.send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        } else if (o.kind == 3) {
            for (int r = 0; r < g->n && rc == ncclSuccess; ++r) {
                const Op &p = g->ops[r][e];
                if (p.kind != 3 || p.bytes != o.bytes) { rc = ncclInvalidUsage; break; }
                if (o.bytes && hipMemcpy((uint8_t *) o.recv + (size_t) r * o.bytes, p.send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
            }
        } else if (o.kind == 4) {
            std::vector<uint8_t> acc(o.bytes), tmp(o.bytes);
            for (int r = 0; r < g->n && rc == ncclSuccess; ++r) {
                const Op &p = g->ops[r][e];
                if (p.kind != 4 || p.bytes != o.bytes || p.dtype != o.dtype || p.red != o.red) { rc = ncclInvalidUsage; break; }
                if (hipMemcpy(r == 0? acc.data() : tmp.data(), p.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
                if (r == 0) continue;
                if (width(o.dtype) == 8) {
                    uint64_t *a = (uint64_t *) acc.data(), *b = (uint64_t *) tmp.data();
                    for (size_t i = 0; i < o.count; ++i) a[i] = o.red == ncclMin? (b[i] < a[i]? b[i] : a[i]) : a[i] + b[i];
                } else {
                    uint32_t *a = (uint32_t *) acc.data(), *b = (uint32_t *) tmp.data();
                    for (size_t i = 0; i < o.count; ++i) a[i] = o.red == ncclMin? (b[i] < a[i]? b[i] : a[i]) : a[i] + b[i];
                }
            }
            if (!g->barrier()) return ncclSystemError;      // in place: nobody's input changes before everybody has read it
            if (rc == ncclSuccess && o.bytes && hipMemcpy(o.recv, acc.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        }
    }
    // (round 6) The copies above are hipMemcpy device-to-device on the NULL stream: asynchronous to the host, and not ordered with the callers' streams, which are created
    // hipStreamNonBlocking -- a kernel the caller launches on its stream right after this call could read a buffer the copy had not filled yet.  It did, once, in the
    // 20th run of tests/test_gpu_multi_c.py this round (one rank's refreshed `del` flags): the mock now drains the null stream before anybody goes on.  Real RCCL
    // orders a collective on the stream it is given.
    if (hipStreamSynchronize(nullptr) != hipSuccess && rc == ncclSuccess) rc = ncclUnhandledCudaError;
    if (!g->barrier()) return ncclSystemError;       // nobody's buffers go away before everybody has read them
    return rc;
}

ncclResult_t submit(Comm *c, hipStream_t st, const Op &o)
{
    if (t_depth > 0) {
        if (t_comm && t_comm != c) return ncclInvalidUsage;
        t_comm = c, t_stream = st;
        t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one(1, o);
    return run(c, st, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, 8);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(Comm **out, int n, ncclUniqueId id, int rank)
{
    uint64_t key;
    memcpy(&key, id.internal, 8);
    Group *g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Group *&slot = g_groups[key];
        if (!slot) { slot = new Group(); slot->n = n; slot->ops.resize(n); }
        g = slot;
    }
    if (g->n != n || rank < 0 || rank >= n) return ncclInvalidArgument;
    *out = new Comm{g, rank};
    t_mine = *out;
    if (!g->barrier()) return ncclSystemError;       // ncclCommInitRank returns when every rank has joined
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(Comm *c) { delete c; return ncclSuccess; }
// (real RCCL: the aborting rank's peers find out through ncclCommGetAsyncError or a time-out -- oatk's comm_wait polls both; here the ranks are
//  threads at a barrier, so the abort wakes them directly)
ncclResult_t ncclCommAbort(Comm *c) { c->g->abort(); delete c; return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(Comm *c, ncclResult_t *e) { *e = c->g->aborted? ncclSystemError : ncclSuccess; return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    // (real RCCL pairs sends and receives rank by rank; this mock meets at a barrier, so a rank whose group is empty -- an empty shard has
    //  nothing to send or receive -- takes part all the same: api_multi.inc's collectives are entered by every rank)
    Comm *c = t_comm? t_comm : t_mine;
    if (c) rc = run(c, t_stream, t_ops);
    t_ops.clear(), t_comm = nullptr;
    return rc;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, Comm *c, hipStream_t st)
{
    if (t_depth <= 0) return ncclInvalidUsage;                       // (this mock pairs sends and receives at ncclGroupEnd only)
    return submit(c, st, Op{0, buf, nullptr, count * width(dt), peer, dt, 0, count});
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, Comm *c, hipStream_t st)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    return submit(c, st, Op{1, nullptr, buf, count * width(dt), peer, dt, 0, count});
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, Comm *c, hipStream_t st)
{
    return submit(c, st, Op{2, send, recv, count * width(dt), root, dt, 0, count});
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, Comm *c, hipStream_t st)
{
    return submit(c, st, Op{3, send, recv, count * width(dt), 0, dt, 0, count});
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, Comm *c, hipStream_t st)
{
    return submit(c, st, Op{4, send, recv, count * width(dt), 0, dt, (int) op, count});
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess? "no error" : (r == ncclInvalidUsage? "mock rccl: invalid usage (unmatched operation)" : "mock rccl: error"); }

}
