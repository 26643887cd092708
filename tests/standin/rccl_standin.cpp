// rccl_standin.cpp -- TEST INFRASTRUCTURE ONLY: an in-process stand-in for the nine RCCL entry points csrc/exchange.cpp calls,
// so that the library's multi-GPU join (join_frame, the submit thread, the second gather round, the send / receive group) can run
// with W = 2, 3, 8 ranks on ONE GPU: ranks are threads of one process, all on device 0, and the "wire" is a device-to-device copy.
// Linked into tests/standin/lib/libcwipc_util.so INSTEAD of librccl (tests/standin/build_standin.py); the shipped
// cwipc_util_amd/lib/libcwipc_util.so never contains it.  RCCL itself refuses two ranks on one device, and the builder has one.
//
// What it checks that a test on one rank cannot: every receive meets exactly one send of the same length and type from the peer
// it names, in the order the pair issued them; every send is consumed; every rank takes part in every collective.  A rank that
// waits for a partner that never comes gets ncclInternalError after RENDEZVOUS_SECONDS (and says which rendezvous it was) instead
// of passing -- round 2's deadlock (one rank left the frame before the group) would end every other rank's call that way.
//
// It is stricter than RCCL in one respect: calls block the host until the data has moved (RCCL enqueues and returns).  Code that
// is correct against RCCL's stream semantics is correct against this; the reverse does not hold for timing, only for matching.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#define STANDIN_EXPORT extern "C" __attribute__((visibility("hidden")))

namespace {

constexpr int RENDEZVOUS_SECONDS = 20;

struct Msg {
    const void *ptr;
    size_t count;
    ncclDataType_t type;
    bool consumed = false;
};

struct World {
    int nranks = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    // all-gather rendezvous (one at a time per world: collectives are issued in one order)
    unsigned long long ag_gen = 0;
    int ag_arrived = 0, ag_done = 0;
    std::vector<const void *> ag_send;
    std::vector<size_t> ag_bytes;
    // point to point: (src, dst) -> messages posted and not yet consumed, in order of issue
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> posted;
    unsigned long long p2p_messages = 0, allgathers = 0;
};

std::mutex g_worlds_mutex;
std::map<std::string, std::shared_ptr<World>> g_worlds;
unsigned long long g_next_id = 1;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

struct Op {
    bool send;
    const void *sendbuf;
    void *recvbuf;
    size_t count;
    ncclDataType_t type;
    int peer;
    struct ncclComm *comm;
    hipStream_t stream;
};
thread_local int tl_group_depth = 0;
thread_local std::vector<Op> tl_ops;

bool wait_until(World &w, std::unique_lock<std::mutex> &g, const std::function<bool()> &pred, const char *what, int rank) {
    if (w.cv.wait_for(g, std::chrono::seconds(RENDEZVOUS_SECONDS), pred)) return true;
    fprintf(stderr, "rccl stand-in: rank %d gave up after %d s waiting for: %s\n", rank, RENDEZVOUS_SECONDS, what);
    return false;
}

}  // namespace

struct ncclComm {
    std::shared_ptr<World> world;
    int rank = 0;
};

namespace {

ncclResult_t run_ops(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    ncclComm *cm = ops[0].comm;
    World &w = *cm->world;
    const int me = cm->rank;
    // the send buffers hold what the stream has written so far
    for (const Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<std::shared_ptr<Msg>> mine, taken;
    std::unique_lock<std::mutex> g(w.m);
    for (const Op &o : ops) {
        if (!o.send) continue;
        auto msg = std::make_shared<Msg>(Msg{o.sendbuf, o.count, o.type});
        w.posted[{me, o.peer}].push_back(msg);
        mine.push_back(msg);
        w.p2p_messages++;
    }
    w.cv.notify_all();
    for (const Op &o : ops) {
        if (o.send) continue;
        auto &q = w.posted[{o.peer, me}];
        char what[128];
        snprintf(what, sizeof(what), "a send of rank %d that matches this receive (%zu elements)", o.peer, o.count);
        if (!wait_until(w, g, [&] { return !q.empty(); }, what, me)) return ncclInternalError;
        std::shared_ptr<Msg> msg = q.front();
        q.pop_front();
        if (msg->count != o.count || type_size(msg->type) != type_size(o.type)) {
            fprintf(stderr, "rccl stand-in: rank %d receives %zu x %zu bytes from rank %d, which sent %zu x %zu\n", me, o.count, type_size(o.type), o.peer, msg->count,
                    type_size(msg->type));
            return ncclInvalidArgument;
        }
        g.unlock();
        const hipError_t e = hipMemcpyAsync(o.recvbuf, msg->ptr, o.count * type_size(o.type), hipMemcpyDeviceToDevice, o.stream);
        g.lock();
        if (e != hipSuccess) return ncclUnhandledCudaError;
        taken.push_back(msg);
    }
    g.unlock();
    for (const Op &o : ops)
        if (!o.send && hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    g.lock();
    for (auto &msg : taken) msg->consumed = true;
    w.cv.notify_all();
    for (auto &msg : mine)
        if (!wait_until(w, g, [&] { return msg->consumed; }, "a receive that matches one of this rank's sends", me)) return ncclInternalError;
    return ncclSuccess;
}

}  // namespace

STANDIN_EXPORT const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (stand-in)";
        case ncclInternalError: return "internal error (stand-in: a rendezvous was not met)";
        case ncclInvalidArgument: return "invalid argument (stand-in: a send and its receive do not match)";
        case ncclInvalidUsage: return "invalid usage (stand-in)";
        default: return "error (stand-in)";
    }
}

STANDIN_EXPORT ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_worlds_mutex);
    memset(id, 0, sizeof(*id));
    const unsigned long long n = g_next_id++;
    memcpy(id->internal, "standin!", 8);
    memcpy(id->internal + 8, &n, sizeof(n));
    return ncclSuccess;
}

STANDIN_EXPORT ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> g(g_worlds_mutex);
        auto &slot = g_worlds[std::string(id.internal, sizeof(id.internal))];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->nranks = nranks;
            slot->ag_send.assign(nranks, nullptr);
            slot->ag_bytes.assign(nranks, 0);
        }
        w = slot;
    }
    if (w->nranks != nranks) return ncclInvalidArgument;
    {   // creation is collective
        std::unique_lock<std::mutex> g(w->m);
        w->joined++;
        w->cv.notify_all();
        if (!wait_until(*w, g, [&] { return w->joined >= nranks; }, "the other ranks' ncclCommInitRank", rank)) return ncclInternalError;
    }
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

STANDIN_EXPORT ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    {
        std::lock_guard<std::mutex> g(comm->world->m);
        comm->world->left++;
        if (comm->world->left == comm->world->nranks && getenv("CWIPC_STANDIN_VERBOSE"))
            fprintf(stderr, "rccl stand-in: world of %d ranks done: %llu all-gathers, %llu point-to-point messages\n", comm->world->nranks, comm->world->allgathers,
                    comm->world->p2p_messages);
    }
    delete comm;
    return ncclSuccess;
}

STANDIN_EXPORT ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    World &w = *comm->world;
    const int me = comm->rank, W = w.nranks;
    const size_t bytes = sendcount * type_size(datatype);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> g(w.m);
    const unsigned long long gen = w.ag_gen;
    w.ag_send[me] = sendbuff;
    w.ag_bytes[me] = bytes;
    w.ag_arrived++;
    if (me == 0) w.allgathers++;
    w.cv.notify_all();
    if (!wait_until(w, g, [&] { return w.ag_gen != gen || w.ag_arrived >= W; }, "the other ranks' ncclAllGather", me)) return ncclInternalError;
    std::vector<const void *> from(w.ag_send);
    for (int r = 0; r < W; r++)
        if (w.ag_bytes[r] != bytes) { fprintf(stderr, "rccl stand-in: all-gather of %zu bytes on rank %d meets %zu on rank %d\n", bytes, me, w.ag_bytes[r], r); return ncclInvalidArgument; }
    g.unlock();
    for (int r = 0; r < W; r++)
        if (hipMemcpyAsync((char *)recvbuff + (size_t)r * bytes, from[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g.lock();
    w.ag_done++;
    if (w.ag_done == W) {   // everybody has read everybody's buffer: the next collective may reuse them
        w.ag_arrived = 0;
        w.ag_done = 0;
        w.ag_gen++;
        w.cv.notify_all();
    } else if (!wait_until(w, g, [&] { return w.ag_gen != gen; }, "the other ranks to finish their all-gather", me)) {
        return ncclInternalError;
    }
    return ncclSuccess;
}

STANDIN_EXPORT ncclResult_t ncclGroupStart() {
    tl_group_depth++;
    return ncclSuccess;
}

STANDIN_EXPORT ncclResult_t ncclGroupEnd() {
    if (tl_group_depth <= 0) return ncclInvalidUsage;
    if (--tl_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run_ops(ops);
}

STANDIN_EXPORT ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->nranks || (count && !sendbuff)) return ncclInvalidArgument;
    tl_ops.push_back(Op{true, sendbuff, nullptr, count, datatype, peer, comm, stream});
    if (tl_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run_ops(ops);
}

STANDIN_EXPORT ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->nranks || (count && !recvbuff)) return ncclInvalidArgument;
    tl_ops.push_back(Op{false, nullptr, recvbuff, count, datatype, peer, comm, stream});
    if (tl_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run_ops(ops);
}
