// filters.cpp -- C entry points of the per-point filter hot path and their host
// orchestration.  Reference: src/cwipc_filters.cpp (wrapping logic, NULL/ERROR
// conventions, timestamp/cellsize propagation) -- the per-point work itself is in
// kernels_*.hip.  There is NO CPU fallback: without a usable GPU every filter
// logs an ERROR and returns NULL.
#include "internal.hpp"

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cstring>

namespace cwipc_amd {

namespace {

// Resolve the argument to one of our clouds with device-resident planes.
// `keep` owns a temporary when the cloud came from another implementation.
std::shared_ptr<DeviceSoA> device_input(const char *who, cwipc_pointcloud *pc, std::unique_ptr<cwipc_hip_pointcloud> &keep) {
    cwipc_hip_pointcloud *ours = as_ours(pc);
    if (!ours) {
        keep = import_foreign(pc);
        ours = keep.get();
        if (!ours) {
            cwipc_log(CWIPC_LOG_LEVEL_WARNING, who, "cannot read the point data of the argument");
            return nullptr;
        }
    }
    if (!ours->has_data()) {
        // the reference sees a NULL pcl cloud here (src/cwipc_filters.cpp:37-40 and alike)
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, who, "pcl_pointcloud is NULL");
        return nullptr;
    }
    if (!device_available(who)) return nullptr;
    return ours->device_points();
}

// Filters that keep coordinates and order keep the first point (the octree anchor of a later
// cwipc_downsample): no device read-back needed for it.
void inherit_first(DeviceSoA &dst, const DeviceSoA &src) {
    if (src.has_first && dst.npoints == src.npoints) {
        dst.first[0] = src.first[0]; dst.first[1] = src.first[1]; dst.first[2] = src.first[2];
        dst.has_first = true;
    }
}

cwipc_pointcloud *wrap(std::shared_ptr<DeviceSoA> planes, uint64_t timestamp, float cellsize) {
    if (!planes) return nullptr;
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_device(planes, timestamp, cellsize);
    return rv;
}

}  // namespace

// Stable compaction driver: count -> scan -> scatter, back to back with one wait at the end.  The
// output planes have room for every input point (the kept count is known only when the kernels are
// done; the scan kernel writes it into the thread's pinned words); a result that uses less than a
// sixteenth of that room is copied into a buffer of its own size.
std::shared_ptr<DeviceSoA> compact(const DeviceSoA &src, const k::Predicate &p, bool may_return_early) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    size_t n = src.npoints;
    if (n == 0) return soa_alloc(0);
    size_t nb = k::compact_blocks(n);
    uint32_t *counts = (uint32_t *)c.device_scratch((nb + 1) * sizeof(uint32_t));   // (+ the total, for the scatter kernel)
    auto dst = soa_alloc(n);
    if (!counts || !dst) return nullptr;
    // the scan kernel publishes the kept count with this tag in the upper half of the first 64-bit pinned word
    const uint32_t tag = ++c.tag ? c.tag : ++c.tag;
    volatile unsigned long long *word = reinterpret_cast<volatile unsigned long long *>(c.host_words);
    *word = 0ull;
    if (k::compact_count_scan(src, p, counts, c.tickets, reinterpret_cast<unsigned long long *>(c.host_words), tag, c.stream)) {
        k::compact_scatter(src, p, counts, *dst, c.stream);
    } else {
        // big clouds (r4): two launches, not three -- the scatter kernel's workgroups add up the counts in front of them themselves
        k::compact_count(src, p, counts, c.stream);
        k::compact_scatter(src, p, counts, *dst, c.stream, reinterpret_cast<unsigned long long *>(c.host_words), tag);
    }
    bool ok = hipGetLastError() == hipSuccess;
    // The count is there when the scan kernel is done; the scatter kernel behind it needs no more attention
    // from the host, so a caller that allows it gets the result back with that kernel still running.
    bool seen = false;
    if (ok && may_return_early && !profiling_enabled()) {
        const auto t_give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(poll_budget_us());
        for (int spin = 0;; spin++) {
            if ((uint32_t)(*word >> 32) == tag) { seen = true; break; }
            if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_give_up) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) ok = c.sync() && ok;
    if (!ok || (uint32_t)(*word >> 32) != tag) {
        (void)c.sync();   // `dst` goes back to the pool: nothing may still be writing it
        hip_failed(hipGetLastError(), "compaction", __FILE__, __LINE__);
        return nullptr;
    }
    const size_t kept = (uint32_t)*word;
    if (kept > n) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip", "compaction: inconsistent count");
        if (seen) (void)c.sync();
        return nullptr;
    }
    if (kept == n) {
        // every point kept: the result holds the input's planes (clouds are immutable); the scatter kernel saw the same total
        // and copied nothing.  (The input is complete: the count kernel, ordered behind its producer, has run.)
        auto same = std::make_shared<DeviceSoA>();
        same->xyz_block = src.xyz_block;
        same->rgbt_block = src.rgbt_block;
        same->npoints = src.npoints;
        same->stride = src.stride;
        same->device = src.device;
        if (src.has_first) { same->first[0] = src.first[0]; same->first[1] = src.first[1]; same->first[2] = src.first[2]; same->has_first = true; }
        // (`dst` goes back to the pool untouched: the scatter kernel writes nothing)
        return same;
    }
    if (kept * 16 >= n) {
        dst->npoints = kept;   // the planes keep their spacing (stride), only the count shrinks
        if (seen) {
            dst->mark_pending(c.stream);
            src.note_reader(c.stream);   // the scatter kernel is still reading the input
        }
        return dst;
    }
    if (seen && !c.sync()) return nullptr;   // the copy below reads what the scatter kernel writes, then `dst` goes back to the pool
    auto small = soa_alloc(kept);
    if (!small) return nullptr;
    if (kept) {
        bool copied = hipMemcpyAsync(small->x(), dst->x(), kept * 4, hipMemcpyDeviceToDevice, c.stream) == hipSuccess &&
                      hipMemcpyAsync(small->y(), dst->y(), kept * 4, hipMemcpyDeviceToDevice, c.stream) == hipSuccess &&
                      hipMemcpyAsync(small->z(), dst->z(), kept * 4, hipMemcpyDeviceToDevice, c.stream) == hipSuccess &&
                      hipMemcpyAsync(small->rgbt(), dst->rgbt(), kept * 4, hipMemcpyDeviceToDevice, c.stream) == hipSuccess;
        copied = c.sync() && copied;
        if (!copied) { hip_failed(hipGetLastError(), "compaction", __FILE__, __LINE__); return nullptr; }
    }
    return small;
}

}  // namespace cwipc_amd

using namespace cwipc_amd;

// ---------------------------------------------------------------------------
// reference src/cwipc_filters.cpp:281-306
// ---------------------------------------------------------------------------
extern "C" cwipc_pointcloud *cwipc_tilefilter(cwipc_pointcloud *pc, int tile) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_tilefilter", pc, keep);
    if (!src) return nullptr;
    // tile 0 keeps every point (reference :296): the result is the same cloud -- clouds are immutable, so it holds
    // the very same planes instead of a copy of them
    if (tile == 0) return wrap(src, pc->timestamp(), pc->cellsize());
    // what is known about the cloud's tiles without looking at a point (DeviceSoA::tiles): a camera's own tile through its own
    // filter -- the per-tile chain of the reference's registration tooling -- is the cloud itself; a tile that cannot occur is empty
    if (src->npoints && src->only_tile((unsigned)tile)) return wrap(src, pc->timestamp(), pc->cellsize());
    if (src->npoints && !src->may_have_tile((unsigned)tile)) {
        auto none = soa_alloc(0);
        if (none) none->set_one_tile((unsigned)tile);
        return wrap(none, pc->timestamp(), pc->cellsize());
    }
    k::Predicate p{};
    p.mode = 0;
    p.tile = tile;
    auto dst = compact(*src, p, true);
    if (dst) dst->set_one_tile((unsigned)tile);   // (tile > 255 keeps nothing, reference :296; a set of "value 255 & ..." never matters for an empty cloud)
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference python/cwipc/registration/util.py:98-112 (numpy boolean-mask selection)
extern "C" cwipc_pointcloud *cwipc_hip_tilefilter_masked(cwipc_pointcloud *pc, int mask) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_tilefilter_masked", pc, keep);
    if (!src) return nullptr;
    k::Predicate p{};
    p.mode = 2;
    p.tile = mask & 0xff;
    auto dst = compact(*src, p, true);
    if (dst) dst->set_tiles_from(*src);   // (a subset of the points: what could not occur still cannot)
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference src/cwipc_filters.cpp:333-360
extern "C" cwipc_pointcloud *cwipc_crop(cwipc_pointcloud *pc, float bbox[6]) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_crop", pc, keep);
    if (!src) return nullptr;
    k::Predicate p{};
    p.mode = 1;
    memcpy(p.bbox, bbox, 6 * sizeof(float));
    auto dst = compact(*src, p, true);
    if (dst) dst->set_tiles_from(*src);   // (a subset of the points: what could not occur still cannot)
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference src/cwipc_filters.cpp:308-331
extern "C" cwipc_pointcloud *cwipc_tilemap(cwipc_pointcloud *pc, uint8_t map[256]) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_tilemap", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_rgbt(src);   // the coordinates do not change: the result holds the very same planes
    if (!dst) return nullptr;
    // the 256-byte table travels through the per-thread pinned words
    memcpy(c.host_words, map, 256);
    bool ok = hipMemcpyAsync(c.dev_words, c.host_words, 256, hipMemcpyHostToDevice, c.stream) == hipSuccess;
    if (ok) k::map_tile(*src, *dst, (const uint8_t *)c.dev_words, c.stream);
    ok = c.sync() && ok;
    if (!ok) return nullptr;
    inherit_first(*dst, *src);
    {   // the tiles that may occur afterwards: the images of those that may occur now (of all 256 values if nothing is known)
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned t = 0; t < 256; t++)
            if (src->may_have_tile(t)) w[map[t] >> 5] |= 1u << (map[t] & 31u);
        dst->set_tiles(w);
    }
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference src/cwipc_filters.cpp:362-386
extern "C" cwipc_pointcloud *cwipc_colormap(cwipc_pointcloud *pc, uint32_t clearBits, uint32_t setBits) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_colormap", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_rgbt(src);   // the coordinates do not change: the result holds the very same planes
    if (!dst) return nullptr;
    k::map_color_bits(*src, *dst, clearBits, setBits, c.stream);
    if (!c.sync()) return nullptr;
    inherit_first(*dst, *src);
    if (src->has_tiles) {   // the tile is bits 24-31 of the word the masks work on (:377-378)
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned t = 0; t < 256; t++)
            if (src->may_have_tile(t)) { const unsigned u = ((t & ~(clearBits >> 24)) | (setBits >> 24)) & 255u; w[u >> 5] |= 1u << (u & 31u); }
        dst->set_tiles(w);
    }
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference python/cwipc/registration/util.py:295-309 (cwipc_transform: numpy R @ p + t in float64, stored as float32)
extern "C" cwipc_pointcloud *cwipc_hip_transform(cwipc_pointcloud *pc, const double *matrix4x4) {
    if (pc == nullptr || matrix4x4 == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_transform", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_xyz(src);   // colours and tiles do not change: the result holds the very same words
    if (!dst) return nullptr;
    double m[12];
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++) m[r * 4 + col] = matrix4x4[r * 4 + col];
    k::map_affine(*src, *dst, m, 0, c.stream);
    if (!c.sync()) return nullptr;
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference python/cwipc/filters/transform.py:38-52 (TransformFilter: (p + offset) * scale in Python floats; cellsize * scale)
extern "C" cwipc_pointcloud *cwipc_hip_offset_scale(cwipc_pointcloud *pc, double x, double y, double z, double scale) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_offset_scale", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_xyz(src);   // colours and tiles do not change: the result holds the very same words
    if (!dst) return nullptr;
    double m[12] = {scale, 0, 0, x, 0, 0, 0, y, 0, 0, 0, z};
    k::map_affine(*src, *dst, m, 1, c.stream);
    if (!c.sync()) return nullptr;
    return wrap(dst, pc->timestamp(), (float)((double)pc->cellsize() * scale));
}

// reference python/cwipc/registration/util.py:285-293 (get_tiles_used): used[t] = 1 for every tile value that occurs
extern "C" int cwipc_hip_tiles_used(cwipc_pointcloud *pc, uint8_t *used256) {
    if (pc == nullptr || used256 == nullptr) return -1;
    memset(used256, 0, 256);
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_tiles_used", pc, keep);
    if (!src) return -1;
    if (src->npoints == 0) return 0;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return -1;
    uint32_t *dev = (uint32_t *)c.dev_words;
    bool ok = hipMemsetAsync(dev, 0, 32, c.stream) == hipSuccess;
    if (ok) k::tiles_used(*src, dev, c.stream);
    ok = ok && hipMemcpyAsync(c.host_words, dev, 32, hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    ok = c.sync() && ok;
    if (!ok) return -1;
    int count = 0;
    for (int t = 0; t < 256; t++) {
        used256[t] = (c.host_words[t >> 5] >> (t & 31)) & 1u;
        count += used256[t];
    }
    // a census: remembered on the cloud.  The cloud may be in other threads' hands (clouds are immutable but for this note): one
    // census at a time writes the words, and only a cloud that has no set yet gets one (a set a producer left is at least as good)
    {
        static std::mutex census_mutex;
        std::lock_guard<std::mutex> lock(census_mutex);
        if (!src->has_tiles.load(std::memory_order_acquire)) src->set_tiles(c.host_words);
    }
    return count;
}

// reference python/cwipc/filters/simulatecams.py:44-70 (hard = True): the per-point loop; the centroid is the caller's
extern "C" cwipc_pointcloud *cwipc_hip_simulatecams(cwipc_pointcloud *pc, int ncamera, float centroid_x, float centroid_z, const double *camera_dirs) {
    if (pc == nullptr || camera_dirs == nullptr || ncamera < 1 || ncamera > 32) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_simulatecams", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_rgbt(src);   // the coordinates do not change: the result holds the very same planes
    if (!dst) return nullptr;
    k::map_cameras(*src, *dst, ncamera, centroid_x, centroid_z, camera_dirs, c.stream);
    if (!c.sync()) return nullptr;
    inherit_first(*dst, *src);
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference python/cwipc/filters/colorize.py:100-119
extern "C" cwipc_pointcloud *cwipc_hip_colorize(cwipc_pointcloud *pc, double weight, const double *lut, const uint8_t *valid) {
    if (pc == nullptr || lut == nullptr || valid == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_colorize", pc, keep);
    if (!src) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_with_new_rgbt(src);   // the coordinates do not change: the result holds the very same planes
    if (!dst) return nullptr;
    const int ndoubles = 1025 + 256;
    // (r4) A stream of frames colours every tile with the same weight and map: the table of a (device, weight, map) stays on the
    // device -- up to sixteen of them for the life of the process -- and a call that finds its table there is one kernel launch that
    // nobody waits for (the result carries an event, the input remembers its reader).  Round 3's call built the table, copied it
    // and waited for the kernel: 35 us per camera tile in config 5's chain, most of it the wait.
    struct CachedTable { int device; double weight; double lut[768]; uint8_t valid[256]; double *dev; };
    static std::mutex cache_mutex;
    static std::vector<CachedTable *> cache;
    double *dev_table = nullptr;
    {
        std::lock_guard<std::mutex> lock(cache_mutex);
        for (CachedTable *t : cache)
            if (t->device == src->device && memcmp(&t->weight, &weight, sizeof(double)) == 0 && memcmp(t->lut, lut, sizeof(t->lut)) == 0 &&
                memcmp(t->valid, valid, sizeof(t->valid)) == 0) { dev_table = t->dev; break; }
    }
    if (dev_table && !profiling_enabled()) {
        // (on the thread's second stream: cwipc_downsample takes "the thread's first stream is busy" for calls that come faster than
        // its workspace turns around and answers with another workspace, 0.3 GB of leaf grids -- a kernel of this filter in flight
        // there made every thread of config 5's chain hold three, test_config5_eight_threads_workspace_footprint)
        hipStream_t s = c.stream_alt ? c.stream_alt : c.stream;
        if (s != c.stream) src->wait_on(s);   // (device_input has ordered the first stream behind the input's producer, not this one)
        k::map_colorize(*src, *dst, dev_table, s);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            hip_failed(e, "cwipc_hip_colorize", __FILE__, __LINE__);
            (void)c.sync();
            return nullptr;
        }
        dst->mark_pending(s);
        src->note_reader(s);
        inherit_first(*dst, *src);
        dst->set_tiles_from(*src);   // (colours change, tiles do not)
        return wrap(dst, pc->timestamp(), pc->cellsize());
    }
    const bool cached = dev_table != nullptr;
    double *host_table = (double *)c.staging(ndoubles * sizeof(double));
    if (!host_table) return nullptr;
    // the same IEEE double operations Python performs, in the same order
    for (int t = 0; t < 256; t++)
        for (int ch = 0; ch < 3; ch++) host_table[t * 3 + ch] = lut[t * 3 + ch] * weight;
    for (int v = 0; v < 256; v++) host_table[768 + v] = v / 255.0;
    host_table[1024] = 1 - weight;
    for (int t = 0; t < 256; t++) host_table[1025 + t] = valid[t] ? 1.0 : 0.0;
    bool ok = true;
    if (!cached) {
        dev_table = (double *)pool_alloc(ndoubles * sizeof(double));
        if (!dev_table) return nullptr;
        ok = hipMemcpyAsync(dev_table, host_table, ndoubles * sizeof(double), hipMemcpyHostToDevice, c.stream) == hipSuccess;
    }
    if (ok) k::map_colorize(*src, *dst, dev_table, c.stream);
    ok = c.sync() && ok;
    if (!cached) {
        bool kept = false;
        if (ok) {
            std::lock_guard<std::mutex> lock(cache_mutex);
            if (cache.size() < 16) {
                auto *t = new CachedTable();
                t->device = src->device; t->weight = weight; t->dev = dev_table;
                memcpy(t->lut, lut, sizeof(t->lut)); memcpy(t->valid, valid, sizeof(t->valid));
                cache.push_back(t);   // (the block stays out of the pool from here on)
                kept = true;
            }
        }
        if (!kept) pool_free(dev_table);
    }
    if (!ok) return nullptr;
    inherit_first(*dst, *src);
    dst->set_tiles_from(*src);   // (colours change, tiles do not)
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

// reference src/cwipc_filters.cpp:388-418; n-ary form = left fold (python/cwipc/util.py:1330-1332)
extern "C" cwipc_pointcloud *cwipc_hip_join_multi(cwipc_pointcloud **pcs, int npc) {
    if (pcs == nullptr || npc <= 0) return nullptr;
    for (int i = 0; i < npc; i++) if (pcs[i] == nullptr) return nullptr;
    std::vector<std::unique_ptr<cwipc_hip_pointcloud>> keep(npc);
    std::vector<std::shared_ptr<DeviceSoA>> src(npc);
    size_t total = 0;
    for (int i = 0; i < npc; i++) {
        src[i] = device_input("cwipc_join", pcs[i], keep[i]);
        if (!src[i]) {
            cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_join", "some pcl_pointcloud is NULL");
            return nullptr;
        }
        total += src[i]->npoints;
    }
    uint64_t ts = pcs[0]->timestamp();
    float cellsize = pcs[0]->cellsize();
    for (int i = 1; i < npc; i++) {
        ts = std::min(ts, pcs[i]->timestamp());
        cellsize = std::min(cellsize, pcs[i]->cellsize());
    }
    // all points in one of the inputs (the others are empty): the result holds that input's planes, nothing is copied
    for (int i = 0; i < npc; i++) {
        if (src[i]->npoints == total && total > 0) return wrap(src[i], ts, cellsize);
    }
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto dst = soa_alloc(total);
    if (!dst) return nullptr;
    size_t off = 0;
    for (int i = 0; i < npc; i++) {
        k::JoinPart part{src[i]->x(), src[i]->y(), src[i]->z(), src[i]->rgbt(), src[i]->npoints, off};
        k::join_copy(part, *dst, c.stream);
        off += src[i]->npoints;
    }
    if (profiling_enabled()) {
        if (!c.sync()) return nullptr;
        return wrap(dst, ts, cellsize);
    }
    // the copies need no more attention from the host: the result goes out with them still running (it carries an event),
    // the inputs stay until they have been read
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
        hip_failed(e, "cwipc_join", __FILE__, __LINE__);
        (void)c.sync();
        return nullptr;
    }
    dst->mark_pending(c.stream);
    for (int i = 0; i < npc; i++) if (src[i]->npoints) src[i]->note_reader(c.stream);
    {   // the tiles that may occur: the union over the parts that hold points (unknown as soon as one of them is)
        bool known = true;
        uint32_t u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < npc; i++) {
            if (!src[i]->npoints) continue;
            known = known && src[i]->has_tiles;
            for (int w = 0; w < 8; w++) u[w] |= src[i]->tiles[w];
        }
        if (known) dst->set_tiles(u);
    }
    return wrap(dst, ts, cellsize);
}

extern "C" cwipc_pointcloud *cwipc_join(cwipc_pointcloud *pc1, cwipc_pointcloud *pc2) {
    if (pc1 == nullptr || pc2 == nullptr) return nullptr;
    cwipc_pointcloud *both[2] = {pc1, pc2};
    return cwipc_hip_join_multi(both, 2);
}

// ---------------------------------------------------------------------------
// reference src/cwipc_filters.cpp:30-172
// ---------------------------------------------------------------------------
extern "C" cwipc_pointcloud *cwipc_downsample(cwipc_pointcloud *pc, float cellsize) {
    bool leaf_split = true;
    const char *who = "cwipc_downsample";
    if (cellsize < 0) {          // :90-92 -> cwipc_downsample_voxelgrid(pc, -cellsize)
        cellsize = -cellsize;
        leaf_split = false;
        who = "cwipc_downsample_voxelgrid";
    }
    if (pc == nullptr) return nullptr;
    if (cwipc_hip_device_count() > current_device()) voxel_sample_streams();   // (before the input puts a wait into this thread's stream)
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input(who, pc, keep);
    if (!src) return nullptr;
    float oldcellsize = pc->cellsize();   // :42-46, :103-107
    if (oldcellsize >= cellsize) cellsize = oldcellsize;
    if (src->npoints == 0) {
        if (leaf_split) return wrap(soa_alloc(0), pc->timestamp(), cellsize);   // zero leaves -> empty cloud
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "VoxelGrid filter produced empty pointcloud");   // :58-62
        return nullptr;
    }
    int err = 0;
    std::shared_ptr<DeferredResult> pending;
    auto dst = voxel_downsample(src, cellsize, leaf_split, &err, &pending);
    if (pending) {
        // a stream of frames: the result is handed out while its kernels run (it settles when somebody asks for its points)
        auto *rv = new cwipc_hip_pointcloud();
        rv->adopt_deferred(pending, pc->timestamp(), cellsize);
        return rv;
    }
    if (!dst) return nullptr;
    return wrap(dst, pc->timestamp(), cellsize);
}

// ---------------------------------------------------------------------------
// reference src/cwipc_filters.cpp:181-278
// ---------------------------------------------------------------------------
namespace {

// The inner overload (:181-211): SOR over one cloud's planes.
std::shared_ptr<DeviceSoA> sor_once(const DeviceSoA &src, int k, float stddev_mul) {
    if (src.npoints == 0) return soa_alloc(0);
    // d_i, then the threshold (on the device: nothing is read back in between), then the compaction, whose
    // wait is the only one after the k-NN grid has been set up
    float *dist = (float *)pool_alloc(src.npoints * sizeof(float) + 256);
    if (!dist) return nullptr;
    double *thr_dev = reinterpret_cast<double *>(reinterpret_cast<char *>(dist) + ((src.npoints * sizeof(float) + 127) & ~(size_t)127));
    std::shared_ptr<DeviceSoA> out;
    if (sor_mean_distances(src, k, dist)) out = sor_threshold_and_select(src, dist, stddev_mul, thr_dev);
    // every failure exit: kernels that read or write `dist` may still be in flight, and the block goes back to a pool
    // other threads allocate from
    if (!out) (void)tctx().sync();
    pool_free(dist);
    return out;
}

}  // namespace

extern "C" cwipc_pointcloud *cwipc_remove_outliers(cwipc_pointcloud *pc, int kNeighbors, float stddevMulThresh, bool perTile) {
    if (pc == nullptr) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_remove_outliers", pc, keep);
    if (!src) return nullptr;
    if (!perTile) {   // :262-268
        return wrap(sor_once(*src, kNeighbors, stddevMulThresh), pc->timestamp(), pc->cellsize());
    }
    // :238-261 -- distinct tiles in first-appearance order: a kernel leaves the index of every tile value's first point
    // in 256 words, the host sorts the tiles that occur by it (1 KB read back, the tile plane stays where it is).
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    size_t n = src->npoints;
    std::vector<int> tiles;
    if (n) {
        uint32_t *dev_first = (uint32_t *)c.device_scratch(256 * sizeof(uint32_t));
        uint32_t *first = (uint32_t *)c.staging(256 * sizeof(uint32_t));
        if (!dev_first || !first) return nullptr;
        k::tile_first_index(*src, dev_first, c.stream);
        bool ok = hipMemcpyAsync(first, dev_first, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
        ok = c.sync() && ok;
        if (!ok) return nullptr;
        std::vector<std::pair<uint32_t, int>> order;
        for (int t = 0; t < 256; t++) if (first[t] != 0xffffffffu) order.emplace_back(first[t], t);
        std::sort(order.begin(), order.end());
        for (auto &o : order) tiles.push_back(o.second);
    }
    std::vector<std::shared_ptr<DeviceSoA>> parts;
    size_t total = 0;
    for (int tile : tiles) {
        k::Predicate p{};
        p.mode = 0;          // cwipc_tilefilter semantics, including tile 0 = wildcard (:252, :296)
        p.tile = tile;
        auto sub = compact(*src, p);
        if (!sub) return nullptr;
        auto cleaned = sor_once(*sub, kNeighbors, stddevMulThresh);
        if (!cleaned) return nullptr;
        total += cleaned->npoints;
        parts.push_back(cleaned);
    }
    auto dst = soa_alloc(total);
    if (!dst) return nullptr;
    size_t off = 0;
    for (auto &part : parts) {
        k::JoinPart jp{part->x(), part->y(), part->z(), part->rgbt(), part->npoints, off};
        k::join_copy(jp, *dst, c.stream);
        off += part->npoints;
    }
    if (!c.sync()) return nullptr;
    return wrap(dst, pc->timestamp(), pc->cellsize());
}

extern "C" int cwipc_hip_knn_mean_dist(cwipc_pointcloud *pc, int kNeighbors, float *mean_dist, size_t cap, double *threshold, float stddevMulThresh) {
    if (pc == nullptr) return -1;
    std::unique_ptr<cwipc_hip_pointcloud> keep;
    auto src = device_input("cwipc_hip_knn_mean_dist", pc, keep);
    if (!src) return -1;
    size_t n = src->npoints;
    if (cap < n) return -1;
    if (n == 0) return 0;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return -1;
    float *dist = (float *)pool_alloc(n * sizeof(float));
    if (!dist) return -1;
    bool ok = sor_mean_distances(*src, kNeighbors, dist);
    double thr = 0;
    if (ok && threshold) ok = sor_threshold(dist, n, stddevMulThresh, &thr);
    if (ok) {
        void *stage = c.staging(n * sizeof(float));
        ok = stage && hipMemcpyAsync(stage, dist, n * sizeof(float), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
        ok = c.sync() && ok;
        if (ok) memcpy(mean_dist, stage, n * sizeof(float));
    }
    pool_free(dist);
    if (ok && threshold) *threshold = thr;
    return ok ? 0 : -1;
}
