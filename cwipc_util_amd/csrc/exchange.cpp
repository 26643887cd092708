// exchange.cpp -- the multi-GPU join inside the library: one process per GPU, RCCL over xGMI, one C call per frame.
//
// What it replaces: the reference fuses the camera tiles of a frame in ONE process by folding cwipc_join pairwise
// (reference python/cwipc/net/source_synchronizer.py:175-188, python/cwipc/util.py:1330-1332, src/cwipc_filters.cpp:388-418).
// Here every rank has filtered its own tile on its own GPU; the fused cloud (rank order = tile order = the reference's
// fold order; timestamp and cellsize = the minimum over the contributing clouds, src/cwipc_filters.cpp:411-414) is put
// together on every rank by
//   1. one ncclAllGather of a 32-byte record per rank (count, has-cloud flag, cellsize bits, timestamp, what the rank can do
//      this frame, and how much room for the fused cloud it already holds), read back by the host: the one wait of the call --
//      the counts size the result and the receives.  Only when some rank has to allocate its result now (the first frame, a
//      frame that outgrew the last one by more than 25 %) the ranks meet once more, 4 bytes each, so that a rank that could
//      not is known to all BEFORE payload moves;
//   2. one group of ncclSend / ncclRecv as exchange_plan.hpp lays it out from the gathered records alone: this rank's four
//      planes to every other rank that builds a fused cloud, every other rank's planes from it, received straight into the
//      result's planes at the prefix-sum displacement.  The clouds are SoA on both sides, so there is no pack or unpack
//      kernel and no padding on the wire; the rank's own part is one copy kernel.  A rank whose tile is the whole frame hands
//      its input on as the result -- and still sends it to the others.
// The call returns when the group has been enqueued: the result carries a `ready` event like every asynchronous filter
// result, the input is kept until the sends have read it.
//
// cwipc_util_amd/multigpu.py holds the same protocol on torch.distributed (all_gather of padded slots); it runs on gloo
// without a GPU, which is how the world-size-2 and -3 tests cover the host logic.  The two are compared on the device.
#include "internal.hpp"
#include "exchange_plan.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <thread>

namespace cwipc_amd {

namespace {

using xplan::FrameMeta;

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
    FrameMeta *meta_host = nullptr;     // pinned: [0] = ours going out, [1 .. nranks] = everybody's coming back; then nranks + 1 status words likewise
    FrameMeta *meta_dev = nullptr;      // the same layout in device memory
    std::mutex lock;                    // one frame at a time per communicator (collectives must be issued in one order)
    unsigned long long frames = 0;
    size_t expect_points = 0;           // room taken for the next frame's fused cloud before the ranks meet (last frame's size + 25 %)
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
              hipHostMalloc((void **)&cm->meta_host, sizeof(FrameMeta) * (nranks + 1) + 4 * (nranks + 1), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&cm->meta_dev, sizeof(FrameMeta) * (nranks + 1) + 4 * (nranks + 1)) == hipSuccess;
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

// The communicator's device for the duration of a call, whatever the calling thread had selected.
struct DeviceGuard {
    int before = -1;
    bool switched = false;
    explicit DeviceGuard(int want) {
        if (hipGetDevice(&before) == hipSuccess && before != want) switched = hipSetDevice(want) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(before); }
};

// One gather of `words` uint32 per rank through the communicator's stream, read back by the host (the wait of the call).
// `mine_host` / `all_host` are pinned, `mine_dev` / `all_dev` device memory.
bool gather_words(cwipc_hip_comm *cm, const void *mine_host, void *mine_dev, void *all_dev, void *all_host, size_t words, const char *who) {
    const int W = cm->nranks;
    bool ok = hipMemcpyAsync(mine_dev, mine_host, words * 4, hipMemcpyHostToDevice, cm->stream) == hipSuccess;
    if (ok) {
        ncclResult_t r = ncclAllGather(mine_dev, all_dev, words, ncclUint32, cm->comm, cm->stream);
        if (r != ncclSuccess) { nccl_failed(r, "ncclAllGather", nullptr); return false; }
    }
    ok = ok && hipMemcpyAsync(all_host, all_dev, words * 4 * W, hipMemcpyDeviceToHost, cm->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(cm->stream) == hipSuccess;
    if (!ok) hip_failed(hipGetLastError(), who, __FILE__, __LINE__);
    return ok;
}

#ifdef CWIPC_EXCHANGE_TEST_HOOKS
// TEST BUILD ONLY (tests/standin/build_standin.py compiles this file a second time with the macro; the shipped library is built
// without it): faults on chosen frames of chosen ranks, so that the branches of join_frame that need a rank in trouble -- a rank
// that takes no part (ST_ABSENT), a rank that cannot allocate its fused cloud between the gathers (ST_NO_RECV), a rank that holds
// no room yet (a second gather round where only some ranks allocate) -- issue their real sends and receives.
// CWIPC_TEST_EXCHANGE_FAULTS="rank:frame:bits,..." with bits 1: no device context, 2: the allocation of round 1b fails, 4: no room
// is taken before the gather.
uint32_t test_fault(int rank, unsigned long long frame) {
    const char *e = getenv("CWIPC_TEST_EXCHANGE_FAULTS");
    uint32_t bits = 0;
    while (e && *e) {
        int r = -1; unsigned long long f = 0; unsigned b = 0; int used = 0;
        if (sscanf(e, "%d:%llu:%u%n", &r, &f, &b, &used) >= 3 && r == rank && f == frame) bits |= b;
        const char *comma = strchr(e, ',');
        e = comma ? comma + 1 : nullptr;
    }
    return bits;
}
#else
inline uint32_t test_fault(int, unsigned long long) { return 0; }
#endif

// One frame's exchange, once this rank's part is known: `src` are its planes (nullptr: no tile this frame, or a tile that
// could not be read -- bad_input -- which the other ranks see as "no tile" while this rank's call fails).  The caller holds
// cm->lock.
//
// Every exit before the payload group is COLLECTIVE: what only this rank knows (no usable device context, no memory for the
// fused cloud) goes into its record and the plan (exchange_plan.hpp), a function of the gathered records alone, leaves the
// rank out on every side.  A rank never leaves between the gather and the group on grounds the others cannot see.
JoinOutcome join_frame(cwipc_hip_comm *cm, std::shared_ptr<DeviceSoA> src, uint64_t src_ts, float src_cs, bool bad_input, bool loopback) {
    const char *who = "cwipc_hip_comm_join";
    JoinOutcome none;
    const int W = cm->nranks;
    const bool wire = W > 1 || loopback;
    ThreadCtx &c = tctx();
    uint32_t my_status = xplan::ST_OK;
    const uint32_t fault = test_fault(cm->rank, cm->frames);   // (0 in the shipped library)
    if (!c.ensure() || (fault & 1u)) {
        my_status = xplan::ST_ABSENT;   // (logged by ensure)
    } else if (current_device() != cm->device) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "the communicator was made for another device: this rank's tile is left out of the frame");
        my_status = xplan::ST_ABSENT;
    }
    if (my_status != xplan::ST_OK && !wire) return none;
    DeviceGuard on_device(cm->device);
    if (src && src->npoints >= ((size_t)1 << 32)) { bad_input = true; src = nullptr; }
    if (my_status != xplan::ST_OK) src = nullptr;

    // room for the fused cloud, taken BEFORE the ranks meet: sized from the last frame, so that in a stream of frames nobody
    // has to allocate between the gather and the payload (and nobody can fail there unseen)
    std::shared_ptr<DeviceSoA> room;
    size_t room_points = 0;
    if (wire && my_status == xplan::ST_OK && cm->expect_points && !(fault & 4u)) {
        room = soa_alloc(cm->expect_points);
        if (room) room_points = cm->expect_points;
    }

    FrameMeta mine{};
    mine.status = my_status;
    mine.capacity = (uint32_t)std::min<size_t>(room_points, 0xffffffffu);
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
    FrameMeta *all = cm->meta_host + 1;
    if (!wire) {
        all[0] = mine;
    } else {
        cm->meta_host[0] = mine;
        if (!gather_words(cm, cm->meta_host, cm->meta_dev, cm->meta_dev + 1, all, sizeof(FrameMeta) / 4, who)) return none;
    }
    cm->frames++;

    // 1b. only if some rank has to allocate now: one more meeting, so that a rank that cannot is known to all
    if (wire && xplan::needs_second_round(W, all, loopback)) {
        const size_t total = xplan::frame_total(W, all);
        uint32_t word = my_status;
        if (xplan::needs_buffer(cm->rank, W, all, loopback) && room_points < total) {
            room = (fault & 2u) ? nullptr : soa_alloc(total + total / 4 + 1024);
            room_points = room ? total + total / 4 + 1024 : 0;
            if (!room) {
                cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "out of device memory for the fused cloud: this rank sends its tile and gets no result");
                word = xplan::ST_NO_RECV;
            }
        }
        uint32_t *status_host = reinterpret_cast<uint32_t *>(cm->meta_host + 1 + W);   // [0] ours, [1 .. W] everybody's
        uint32_t *status_dev = reinterpret_cast<uint32_t *>(cm->meta_dev + 1 + W);
        status_host[0] = word;
        // (the one exit between the gathers that is not a collective decision: a HIP or RCCL call of the second gather itself failed on
        // this rank -- a lost device, a broken communicator.  The other ranks are then inside a collective that cannot complete either;
        // nothing this rank could send would reach them.  Every other way out of a frame is taken by all ranks alike, from the records.)
        if (!gather_words(cm, status_host, status_dev, status_dev + 1, status_host + 1, 1, who)) return none;
        for (int r = 0; r < W; r++)
            if (all[r].status == xplan::ST_OK) all[r].status = status_host[1 + r];
    }

    // 2. what this rank does (a function of the records alone: the ranks' plans fit together)
    const xplan::FramePlan plan = xplan::plan_frame(cm->rank, W, all, loopback);
    if (plan.too_big) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "the fused cloud would hold 2^32 points or more");
        return none;   // every rank sees the same counts and takes the same way out: nobody is left waiting
    }
    auto finish = [&](std::shared_ptr<DeviceSoA> planes) -> JoinOutcome {
        if (bad_input) {
            cwipc_log(CWIPC_LOG_LEVEL_WARNING, who, "cannot read the point data of the argument (the other ranks got a frame without this tile)");
            return none;
        }
        if (plan.no_result) return none;
        JoinOutcome out;
        out.planes = std::move(planes);
        out.timestamp = plan.ts_min;
        out.cellsize = plan.cs_min;
        return out;
    };
    if (plan.total == 0) return finish(plan.no_result ? nullptr : soa_alloc(0));   // nothing moves for an empty frame

    std::shared_ptr<DeviceSoA> dst;
    if (!plan.no_result && !plan.share_input) {
        dst = room;     // big enough by construction: planned as a receiver only with capacity >= total or a fresh allocation
        if (!dst || room_points < plan.total) {
            // cannot happen while plan and records agree; if it does, the group below would write out of bounds: do not issue it
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, "internal error: no room for the fused cloud after the ranks agreed there was");
            return none;
        }
        dst->npoints = plan.total;   // (the planes keep their spacing, as after a compaction)
        cm->expect_points = plan.total + plan.total / 4 + 1024;
    }

    // 3. the planes, in one group
    if (!plan.sends.empty() || !plan.recvs.empty()) {
        if (src) src->wait_on(cm->stream);
        ncclResult_t r = ncclGroupStart();
        // per pair of ranks the messages match in order: x, y, z, rgbt
        for (const xplan::Transfer &t : plan.sends) {
            if (r == ncclSuccess) r = ncclSend(src->x(), t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->y(), t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->z(), t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclSend(src->rgbt(), t.n, ncclUint32, t.peer, cm->comm, cm->stream);
        }
        for (const xplan::Transfer &t : plan.recvs) {
            if (r == ncclSuccess) r = ncclRecv(dst->x() + t.offset, t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->y() + t.offset, t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->z() + t.offset, t.n, ncclFloat32, t.peer, cm->comm, cm->stream);
            if (r == ncclSuccess) r = ncclRecv(dst->rgbt() + t.offset, t.n, ncclUint32, t.peer, cm->comm, cm->stream);
        }
        ncclResult_t r_end = ncclGroupEnd();
        if (r == ncclSuccess) r = r_end;
        if (r != ncclSuccess) {
            nccl_failed(r, "ncclSend/ncclRecv group", nullptr);
            (void)hipStreamSynchronize(cm->stream);   // `dst` goes back to the pool
            return none;
        }
    }
    if (plan.own_copy) {
        k::JoinPart part{src->x(), src->y(), src->z(), src->rgbt(), src->npoints, plan.disp[cm->rank]};
        k::join_copy(part, *dst, cm->stream);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            hip_failed(e, who, __FILE__, __LINE__);
            (void)hipStreamSynchronize(cm->stream);
            return none;
        }
    }
    if (src && (!plan.sends.empty() || plan.own_copy)) src->note_reader(cm->stream);   // the sends (and the copy) are still reading the input
    if (plan.share_input) return finish(src);   // all points are this rank's own: the result holds its planes (cwipc_hip_join_multi's rule)
    if (dst) dst->mark_pending(cm->stream);
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

// Test hook (pure host code, no device): the plan of one rank for a frame, see hip_ext.h.
extern "C" int cwipc_hip_exchange_plan(int rank, int nranks, const uint32_t *metas, int loopback, uint64_t *summary, uint64_t *sends, uint64_t *recvs, int cap) {
    if (metas == nullptr || summary == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    static_assert(sizeof(xplan::FrameMeta) == 8 * sizeof(uint32_t), "records are 8 words");
    const xplan::FrameMeta *all = reinterpret_cast<const xplan::FrameMeta *>(metas);
    const xplan::FramePlan plan = xplan::plan_frame(rank, nranks, all, loopback != 0);
    uint32_t cs_bits;
    memcpy(&cs_bits, &plan.cs_min, 4);
    summary[0] = plan.total;
    summary[1] = (plan.too_big ? 1u : 0u) | (plan.no_result ? 2u : 0u) | (plan.share_input ? 4u : 0u) | (plan.own_copy ? 8u : 0u) | (plan.any ? 16u : 0u) |
                 (xplan::needs_second_round(nranks, all, loopback != 0) ? 32u : 0u) | (xplan::needs_buffer(rank, nranks, all, loopback != 0) ? 64u : 0u);
    summary[2] = plan.ts_min;
    summary[3] = cs_bits;
    summary[4] = plan.disp[rank];
    summary[5] = plan.sends.size();
    summary[6] = plan.recvs.size();
    summary[7] = 0;
    if ((int)plan.sends.size() > cap || (int)plan.recvs.size() > cap) return -2;
    for (size_t i = 0; i < plan.sends.size() && sends; i++) { sends[3 * i] = plan.sends[i].peer; sends[3 * i + 1] = plan.sends[i].n; sends[3 * i + 2] = plan.sends[i].offset; }
    for (size_t i = 0; i < plan.recvs.size() && recvs; i++) { recvs[3 * i] = plan.recvs[i].peer; recvs[3 * i + 1] = plan.recvs[i].n; recvs[3 * i + 2] = plan.recvs[i].offset; }
    return 0;
}
