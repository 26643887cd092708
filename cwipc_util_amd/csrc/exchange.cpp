// exchange.cpp -- the multi-GPU join inside the library: one process per GPU, RCCL over xGMI, one C call per frame.
//
// What it replaces: the reference fuses the camera tiles of a frame in ONE process by folding cwipc_join pairwise
// (reference python/cwipc/net/source_synchronizer.py:175-188, python/cwipc/util.py:1330-1332, src/cwipc_filters.cpp:388-418).
// Here every rank has filtered its own tile on its own GPU; the fused cloud (rank order = tile order = the reference's
// fold order; timestamp and cellsize = the minimum over the contributing clouds, src/cwipc_filters.cpp:411-414) is put
// together on every rank by
//   1. one ncclAllGather of a 32-byte record per rank (count, has-cloud flag, cellsize bits, timestamp), read back by the host:
//      the one wait of the call -- the counts size the result and the receives;
//   2. one group of ncclSend / ncclRecv: this rank's four planes to every other rank, every other rank's planes from it,
//      received straight into the result's planes at the prefix-sum displacement.  The clouds are SoA on both sides, so
//      there is no pack or unpack kernel and no padding on the wire; the rank's own part is one copy kernel.
// The call returns when the group has been enqueued: the result carries a `ready` event like every asynchronous filter
// result, the input is kept until the sends have read it.
//
// cwipc_util_amd/multigpu.py holds the same protocol on torch.distributed (all_gather of padded slots); it runs on gloo
// without a GPU, which is how the world-size-2 and -3 tests cover the host logic.  The two are compared on the device.
#include "internal.hpp"

#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

namespace cwipc_amd {

namespace {

struct FrameMeta {          // what every rank tells the others about its part of the frame: 8 words
    uint32_t count;
    uint32_t has_cloud;
    uint32_t cellsize_bits;
    uint32_t pad0;
    uint32_t ts_lo, ts_hi;
    uint32_t pad1, pad2;
};
static_assert(sizeof(FrameMeta) == 32, "FrameMeta travels as 8 uint32");

bool nccl_failed(ncclResult_t r, const char *what, char **errorMessage) {
    std::string msg = std::string(what) + ": " + ncclGetErrorString(r);
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_comm", msg);
    cwipc_log_set_errorbuf(nullptr);
    return false;
}

}  // namespace

}  // namespace cwipc_amd

using namespace cwipc_amd;

namespace {

// The fused cloud of a frame handed out before the exchange has happened (cwipc_hip_comm_submit): the communicator's own
// thread fills it in; the cloud that holds it settles on first use.
struct PendingJoin : cwipc_amd::DeferredResult {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    std::shared_ptr<cwipc_amd::DeviceSoA> planes;   // nullptr: the exchange failed on this rank (logged)
    uint64_t timestamp = 0;
    float cellsize = 0;
    void fulfil(std::shared_ptr<cwipc_amd::DeviceSoA> p, uint64_t ts, float cs) {
        { std::lock_guard<std::mutex> g(m); planes = std::move(p); timestamp = ts; cellsize = cs; done = true; }
        cv.notify_all();
    }
    std::shared_ptr<cwipc_amd::DeviceSoA> settle() override {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return done; });
        return planes;
    }
    bool late_metadata(uint64_t *ts, float *cs) override {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return done; });
        *ts = timestamp; *cs = cellsize;
        return true;
    }
};

struct JoinJob {
    cwipc_amd::cwipc_hip_pointcloud::Snapshot input;   // has_data false + no planes: no tile this frame
    bool has_cloud = false, bad_input = false, loopback = false, stop = false;
    std::shared_ptr<PendingJoin> result;
};

}  // namespace

struct cwipc_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    FrameMeta *meta_host = nullptr;     // pinned: [0] = ours going out, [1 .. nranks] = everybody's coming back
    FrameMeta *meta_dev = nullptr;      // the same layout in device memory
    std::mutex lock;                    // one frame at a time per communicator (collectives must be issued in one order)
    unsigned long long frames = 0;
    // cwipc_hip_comm_submit: frames wait here for the communicator's thread, which takes them in the order they came
    std::mutex queue_lock;
    std::condition_variable queue_cv;
    std::deque<JoinJob> queue;
    std::thread worker;
    bool worker_running = false;
};

static_assert(CWIPC_HIP_COMM_ID_BYTES == sizeof(ncclUniqueId), "CWIPC_HIP_COMM_ID_BYTES is RCCL's ncclUniqueId");

extern "C" int cwipc_hip_comm_unique_id(void *id, char **errorMessage) {
    if (id == nullptr) return -1;
    ncclUniqueId uid;
    ncclResult_t r = ncclGetUniqueId(&uid);
    if (r != ncclSuccess) { nccl_failed(r, "ncclGetUniqueId", errorMessage); return -1; }
    memcpy(id, &uid, sizeof(uid));
    return 0;
}

extern "C" cwipc_hip_comm *cwipc_hip_comm_create(const void *id, int rank, int nranks, char **errorMessage) {
    auto refuse = [&](const std::string &why) -> cwipc_hip_comm * {
        cwipc_log_set_errorbuf(errorMessage);
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_comm_create", why);
        cwipc_log_set_errorbuf(nullptr);
        return nullptr;
    };
    if (id == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return refuse("bad arguments");
    if (nranks > 1) {
        // Between processes RCCL hands device buffers over with HIP IPC; this image's host driver supports only the dmabuf
        // flavour, and without this setting the first exchange dies in hipIpcGetMemHandle ("invalid argument") -- after the
        // other ranks have already entered the collective.  Refuse here, where every rank still can.
        const char *e = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        if (e == nullptr || strcmp(e, "0") != 0)
            return refuse("a join between processes needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every rank (set before the process starts)");
    }
    if (!device_available("cwipc_hip_comm_create")) return refuse("no GPU");
    ThreadCtx &c = tctx();
    if (!c.ensure()) return refuse("no device context");
    std::unique_ptr<cwipc_hip_comm> cm(new cwipc_hip_comm());
    cm->rank = rank;
    cm->nranks = nranks;
    cm->device = current_device();
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = ncclCommInitRank(&cm->comm, nranks, uid, rank);
    if (r != ncclSuccess) { nccl_failed(r, "ncclCommInitRank", errorMessage); return nullptr; }
    bool ok = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc((void **)&cm->meta_host, sizeof(FrameMeta) * (nranks + 1), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&cm->meta_dev, sizeof(FrameMeta) * (nranks + 1)) == hipSuccess;
    if (!ok) {
        hip_failed(hipGetLastError(), "cwipc_hip_comm_create", __FILE__, __LINE__);
        cwipc_hip_comm_free(cm.release());
        return refuse("out of memory");
    }
    return cm.release();
}

extern "C" void cwipc_hip_comm_free(cwipc_hip_comm *cm) {
    if (cm == nullptr) return;
    if (cm->worker_running) {   // frames still queued are exchanged first (the other ranks count on them)
        { std::lock_guard<std::mutex> g(cm->queue_lock); JoinJob stop; stop.stop = true; cm->queue.push_back(std::move(stop)); }
        cm->queue_cv.notify_all();
        cm->worker.join();
        cm->worker_running = false;
    }
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->comm) (void)ncclCommDestroy(cm->comm);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    if (cm->meta_host) (void)hipHostFree(cm->meta_host);
    if (cm->meta_dev) (void)hipFree(cm->meta_dev);
    delete cm;
}

extern "C" int cwipc_hip_comm_rank(cwipc_hip_comm *cm) { return cm ? cm->rank : -1; }
extern "C" int cwipc_hip_comm_nranks(cwipc_hip_comm *cm) { return cm ? cm->nranks : -1; }

namespace {

struct JoinOutcome {
    std::shared_ptr<DeviceSoA> planes;   // nullptr: failed (logged)
    uint64_t timestamp = 0;
    float cellsize = 0;
};

// One frame's exchange, once this rank's part is known: `src` are its planes (nullptr: no tile this frame, or a tile that
// could not be read -- bad_input -- which the other ranks see as "no tile" while this rank's call fails).  The caller holds
// cm->lock.
JoinOutcome join_frame(cwipc_hip_comm *cm, std::shared_ptr<DeviceSoA> src, uint64_t src_ts, float src_cs, bool bad_input, bool loopback) {
    const char *who = "cwipc_hip_comm_join";
    JoinOutcome none;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return none;
    if (current_device() != cm->device) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "the communicator was made for another device");
        return none;
    }
    if (src && src->npoints >= ((size_t)1 << 32)) { bad_input = true; src = nullptr; }
    FrameMeta mine{};
    if (src) {
        const float cs = src_cs;
        const uint64_t ts = src_ts;
        mine.count = (uint32_t)src->npoints;
        mine.has_cloud = 1;
        memcpy(&mine.cellsize_bits, &cs, 4);
        mine.ts_lo = (uint32_t)ts;
        mine.ts_hi = (uint32_t)(ts >> 32);
    }

    // 1. everybody's record
    const int W = cm->nranks;
    FrameMeta *all = cm->meta_host + 1;
    if (W == 1 && !loopback) {
        all[0] = mine;
    } else {
        cm->meta_host[0] = mine;
        bool ok = hipMemcpyAsync(cm->meta_dev, cm->meta_host, sizeof(FrameMeta), hipMemcpyHostToDevice, cm->stream) == hipSuccess;
        ncclResult_t r = ok ? ncclAllGather(cm->meta_dev, cm->meta_dev + 1, sizeof(FrameMeta) / 4, ncclUint32, cm->comm, cm->stream) : ncclSuccess;
        if (r != ncclSuccess) { nccl_failed(r, "ncclAllGather", nullptr); return none; }
        ok = ok && hipMemcpyAsync(all, cm->meta_dev + 1, sizeof(FrameMeta) * W, hipMemcpyDeviceToHost, cm->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(cm->stream) == hipSuccess;
        if (!ok) { hip_failed(hipGetLastError(), who, __FILE__, __LINE__); return none; }
    }
    cm->frames++;

    // 2. what the fused cloud looks like
    std::vector<size_t> disp(W + 1, 0);
    bool any = false;
    uint64_t ts_min = 0;
    float cs_min = 0;
    for (int r = 0; r < W; r++) {
        disp[r + 1] = disp[r] + all[r].count;
        if (!all[r].has_cloud) continue;
        const uint64_t ts = ((uint64_t)all[r].ts_hi << 32) | all[r].ts_lo;
        float cs;
        memcpy(&cs, &all[r].cellsize_bits, 4);
        if (!any || ts < ts_min) ts_min = ts;
        if (!any || cs < cs_min) cs_min = cs;
        any = true;
    }
    const size_t total = disp[W];
    if (total >= ((size_t)1 << 32)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "the fused cloud would hold 2^32 points or more");
        return none;   // every rank sees the same counts and takes the same way out: nobody is left waiting
    }
    auto finish = [&](std::shared_ptr<DeviceSoA> planes) -> JoinOutcome {
        if (bad_input) {
            cwipc_log(CWIPC_LOG_LEVEL_WARNING, who, "cannot read the point data of the argument (the other ranks got a frame without this tile)");
            return none;
        }
        JoinOutcome out;
        out.planes = std::move(planes);
        out.timestamp = ts_min;
        out.cellsize = cs_min;
        return out;
    };
    const size_t n_me = all[cm->rank].count;
    // all points are this rank's own: the result holds its planes, nothing moves (cwipc_hip_join_multi's rule)
    if (total > 0 && n_me == total && !loopback) return finish(src);

    auto dst = soa_alloc(total);
    if (!dst) {
        // no result here, but the others will send: take what they send into nothing?  There is no such thing; a rank that
        // cannot allocate its result cannot stay in step, and saying so loudly is all that is left.
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "out of device memory for the fused cloud: this rank leaves the exchange");
        return none;
    }
    if (total == 0) return finish(dst);

    // 3. the planes, in one group
    if (src) src->wait_on(cm->stream);
    ncclResult_t r = ncclGroupStart();
    for (int peer = 0; peer < W && r == ncclSuccess; peer++) {
        if (peer == cm->rank && !loopback) continue;
        if (n_me) {
            r = ncclSend(src->x(), n_me, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->y(), n_me, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->z(), n_me, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->rgbt(), n_me, ncclUint32, peer, cm->comm, cm->stream);
        }
        const size_t n = all[peer].count, at = disp[peer];
        if (n && r == ncclSuccess) {
            r = ncclRecv(dst->x() + at, n, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->y() + at, n, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->z() + at, n, ncclFloat32, peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->rgbt() + at, n, ncclUint32, peer, cm->comm, cm->stream);
        }
    }
    ncclResult_t r_end = ncclGroupEnd();
    if (r == ncclSuccess) r = r_end;
    if (r != ncclSuccess) {
        nccl_failed(r, "ncclSend/ncclRecv group", nullptr);
        (void)hipStreamSynchronize(cm->stream);   // `dst` goes back to the pool
        return none;
    }
    if (n_me && !loopback) {
        k::JoinPart part{src->x(), src->y(), src->z(), src->rgbt(), n_me, disp[cm->rank]};
        k::join_copy(part, *dst, cm->stream);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            hip_failed(e, who, __FILE__, __LINE__);
            (void)hipStreamSynchronize(cm->stream);
            return none;
        }
    }
    if (src) src->note_reader(cm->stream);   // the sends (and the copy) are still reading the input
    dst->mark_pending(cm->stream);
    return finish(dst);
}

// This rank's part of a frame as a snapshot: planes, or the pending result that will become them (see internal.hpp).
JoinJob job_of(cwipc_pointcloud *pc, int flags) {
    JoinJob job;
    job.loopback = (flags & CWIPC_HIP_JOIN_LOOPBACK) != 0;
    if (pc == nullptr) return job;
    job.has_cloud = true;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    cwipc_hip_pointcloud *ours = as_ours(pc);
    if (!ours) { keep = import_foreign(pc); ours = keep.get(); }
    if (!ours || !ours->has_data()) { job.bad_input = true; return job; }
    job.input = ours->snapshot();
    if (!job.input.pending && !job.input.dev) job.bad_input = true;
    return job;
}

JoinOutcome run_job(cwipc_hip_comm *cm, JoinJob &job) {
    std::shared_ptr<DeviceSoA> src;
    if (job.has_cloud && !job.bad_input) {
        src = job.input.pending ? job.input.pending->settle() : job.input.dev;
        if (!src) job.bad_input = true;   // (a filter that failed after its call had returned: logged there)
    }
    std::lock_guard<std::mutex> guard(cm->lock);
    return join_frame(cm, src, job.input.timestamp, job.input.cellsize, job.bad_input, job.loopback);
}

void comm_worker(cwipc_hip_comm *cm) {
    for (;;) {
        JoinJob job;
        {
            std::unique_lock<std::mutex> g(cm->queue_lock);
            cm->queue_cv.wait(g, [&] { return !cm->queue.empty(); });
            job = std::move(cm->queue.front());
            cm->queue.pop_front();
        }
        if (job.stop) return;
        JoinOutcome out = run_job(cm, job);
        job.result->fulfil(out.planes, out.timestamp, out.cellsize);
    }
}

}  // namespace

extern "C" cwipc_pointcloud *cwipc_hip_comm_join(cwipc_hip_comm *cm, cwipc_pointcloud *pc, int flags) {
    if (cm == nullptr) return nullptr;
    if (!device_available("cwipc_hip_comm_join")) return nullptr;
    if (cm->worker_running) {
        // frames submitted before this one are exchanged first: through the queue, and wait
        cwipc_pointcloud *later = cwipc_hip_comm_submit(cm, pc, flags);
        if (later == nullptr) return nullptr;
        if (!as_ours(later)->has_device()) { later->free(); return nullptr; }   // (settles; a failed exchange leaves no planes)
        return later;
    }
    JoinJob job = job_of(pc, flags);
    JoinOutcome out = run_job(cm, job);
    if (!out.planes) return nullptr;
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_device(out.planes, out.timestamp, out.cellsize);
    return rv;
}

extern "C" cwipc_pointcloud *cwipc_hip_comm_submit(cwipc_hip_comm *cm, cwipc_pointcloud *pc, int flags) {
    if (cm == nullptr) return nullptr;
    if (!device_available("cwipc_hip_comm_submit")) return nullptr;
    JoinJob job = job_of(pc, flags);
    job.result = std::make_shared<PendingJoin>();
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_deferred(job.result, 0, 0.0f, true);
    {
        std::lock_guard<std::mutex> g(cm->queue_lock);
        if (!cm->worker_running) {
            cm->worker = std::thread(comm_worker, cm);
            cm->worker_running = true;
        }
        cm->queue.push_back(std::move(job));
    }
    cm->queue_cv.notify_one();
    return rv;
}
