// internal.hpp -- shared declarations of the MI355X libcwipc_util implementation.
//
// Layering:
//   logging.cpp     cwipc_log & friends            (reference src/logging.cpp)
//   device.cpp      device selection, per-thread stream, memory pool, profiling
//   pointcloud.cpp  cwipc_hip_pointcloud container, C accessors, packet / dump I/O
//   synthetic.cpp   cwipc_synthetic source          (reference src/cwipc_synthetic.cpp)
//   stubs.cpp       out-of-scope constructors that fail loudly
//   filters.cpp     C entry points of the hot path, host orchestration
//   kernels_*.hip   the HIP kernels (gfx950)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstddef>
#include <atomic>
#include <memory>
#include <utility>
#include <mutex>
#include <string>
#include <vector>

#include "cwipc_util/api.h"
#include "cwipc_util_amd/hip_ext.h"

// ---------------------------------------------------------------------------
// logging (reference include/cwipc_util/internal/logging.hpp:11-21)
// ---------------------------------------------------------------------------
extern "C" {
_CWIPC_UTIL_EXPORT void cwipc_log(cwipc_log_level level, std::string module, std::string message);
_CWIPC_UTIL_EXPORT void cwipc_log_set_errorbuf(char **errorbuf);
_CWIPC_UTIL_EXPORT cwipc_log_level cwipc_log_get_level();
}

namespace cwipc_amd {

// Reject a constructor call whose apiVersion is outside the accepted window
// (pattern of reference src/cwipc_util.cpp:663-670).  Returns true if rejected.
bool api_version_rejected(const char *fname, uint64_t apiVersion, char **errorMessage);

// ---------------------------------------------------------------------------
// device context
// ---------------------------------------------------------------------------

// Record a HIP failure: thread-local text + cwipc_log(ERROR).  Returns false.
bool hip_failed(hipError_t err, const char *what, const char *file, int line);
#define CW_HIP_OK(expr) ((expr) == hipSuccess ? true : ::cwipc_amd::hip_failed(hipGetLastError(), #expr, __FILE__, __LINE__))
#define CW_HIP_TRY(expr)                                                            \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            ::cwipc_amd::hip_failed(_e, #expr, __FILE__, __LINE__);                 \
            return false;                                                           \
        }                                                                           \
    } while (0)

// True when a usable GPU exists; logs an ERROR (once per call) otherwise.  The
// product has NO CPU fallback for the filters: callers return NULL when false.
bool device_available(const char *who);
int current_device();

// Device memory pool: size-classed free lists, never shrinks unless trimmed.
// Library calls synchronise their stream before returning, so a block can be reused by any
// stream as soon as it has been released.  The one exception is a filter result whose last
// kernel is still in flight when the call returns: it carries a `ready` event (DeviceSoA),
// consumers make their stream wait for it, and its block is released only after the event.
void *pool_alloc(size_t bytes);
void pool_free(void *ptr);

struct PoolDeleter {
    void operator()(void *p) const { pool_free(p); }
};

// Per-thread execution context: one non-blocking stream, pinned staging, scratch.
struct ThreadCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    // A second stream: cwipc_downsample alternates between the two (with a workspace for each), so that the
    // finalize kernel of one call, in flight when the call returns, does not hold up the first kernel of
    // the next.  Everything else runs on `stream`; sync() waits for both.
    hipStream_t stream_alt = nullptr;
    // ... and a third and fourth (r4), created when a thread's downsample calls come faster than two workspaces turn around
    static constexpr int EXTRA_STREAMS = 2;
    hipStream_t stream_extra[EXTRA_STREAMS] = {nullptr, nullptr};
    hipStream_t extra_stream(int i);   // stream_extra[i], created on first use (nullptr on failure)
    void *pinned = nullptr;       // staging for H2D/D2H of AoS points
    size_t pinned_bytes = 0;
    uint32_t *host_words = nullptr;   // 64 pinned 32-bit words for small read-backs
    void *dev_words = nullptr;        // 64 device words
    uint32_t *tickets = nullptr;      // 16 device words, zero between kernels: "last workgroup done" counters (each kernel that uses one sets it back)
    void *scratch = nullptr;          // device scratch of this thread's calls (block counts of a compaction): work on one
    size_t scratch_bytes = 0;         //   stream is ordered, so the next call may overwrite it while nobody else can
    uint32_t tag = 0;                 // sequence number of the last value a kernel published into host_words
    void *device_scratch(size_t bytes);
    std::vector<void *> deferred;     // pool blocks whose last kernel may still run: given back at the next sync()
    void free_later(void *pool_block) { if (pool_block) deferred.push_back(pool_block); }
    bool ensure();                    // create the stream etc. for the current device
    void *staging(size_t bytes);      // pinned buffer of at least `bytes`
    bool sync();
    ~ThreadCtx();
};
ThreadCtx &tctx();

// How long a call polls pinned memory for a kernel's report before it falls back to an ordinary stream wait
// (microseconds; CWIPC_POLL_US overrides, 0 = never poll: the test suite runs both ways).
inline long poll_budget_us() {
    const char *e = getenv("CWIPC_POLL_US");
    return e ? atol(e) : 2000;
}

// Cached hipEvents without timing, for the `ready` marks of asynchronous results.
hipEvent_t event_get();
void event_put(hipEvent_t e);

// Profiling: kernels are launched through CW_LAUNCH so that per-kernel device
// time can be collected with hipEvents on the launching stream.
void profile_begin(const char *name, hipStream_t s);
void profile_end(hipStream_t s);
void profile_collect();   // after a stream sync: fold finished event pairs into totals
bool profiling_enabled();

#define CW_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                               \
    do {                                                                                       \
        if (::cwipc_amd::profiling_enabled()) ::cwipc_amd::profile_begin(name, stream);        \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                   \
        if (::cwipc_amd::profiling_enabled()) ::cwipc_amd::profile_end(stream);                \
    } while (0)

// ---------------------------------------------------------------------------
// point storage
// ---------------------------------------------------------------------------

// A pool block that several clouds may hold (clouds are immutable: a filter that changes colours or tiles only
// gives its result the input's coordinate planes instead of copying them).
struct PlaneBlock {
    void *ptr = nullptr;
    explicit PlaneBlock(void *p) : ptr(p) {}
    PlaneBlock(const PlaneBlock &) = delete;
    PlaneBlock &operator=(const PlaneBlock &) = delete;
    ~PlaneBlock() { if (ptr) pool_free(ptr); }
};

// Device representation: four planes, each `stride` elements long (stride = npoints rounded up to 256, so every
// plane starts on a 1 KiB boundary and a full wave step of 256 points can always be loaded): x, y, z in one pool
// block, rgbt in another.  rgbt = r | g<<8 | b<<16 | tile<<24, i.e. the last four bytes of a cwipc_point read as
// one little-endian word.
struct DeviceSoA {
    std::shared_ptr<PlaneBlock> xyz_block, rgbt_block;
    size_t npoints = 0;
    size_t stride = 0;
    int device = 0;
    // The cloud's first point, when the host knows it (the octree lattice of cwipc_downsample is
    // anchored there); fetched from the device on demand otherwise.
    mutable bool has_first = false;
    mutable float first[3] = {0, 0, 0};
    // Which tile values MAY occur in the cloud (bit t of 256), when a producer knows: a tilemap with one target value, a tile
    // filter's result, the synthetic source, a census (cwipc_hip_tiles_used); filters that keep the tile words hand it on,
    // a join takes the union.  A one-element set on a cloud with points says that EVERY point has that tile: cwipc_tilefilter
    // for it hands the planes on without looking at a point (a camera's tile is filtered by its own mask in the per-tile chain
    // of the reference, python/cwipc/registration/util.py:170-182), and a filter for a value outside the set is empty.
    // Producers fill the set before they publish the cloud; a census (cwipc_hip_tiles_used) adds it to a cloud other threads may be
    // filtering at that moment: the words first, then the flag with release order, and readers take the flag with acquire order.
    mutable std::atomic<bool> has_tiles{false};
    mutable uint32_t tiles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void set_tiles(const uint32_t words[8]) const { for (int i = 0; i < 8; i++) tiles[i] = words[i]; has_tiles.store(true, std::memory_order_release); }
    void set_tiles_from(const DeviceSoA &o) const { if (o.has_tiles.load(std::memory_order_acquire)) set_tiles(o.tiles); else has_tiles.store(false, std::memory_order_release); }
    void set_one_tile(unsigned t) const { uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0}; w[(t & 255u) >> 5] = 1u << (t & 31u); set_tiles(w); }
    bool may_have_tile(unsigned t) const { return !has_tiles.load(std::memory_order_acquire) || t > 255u || ((tiles[t >> 5] >> (t & 31u)) & 1u) != 0u; }
    bool only_tile(unsigned t) const {
        if (!has_tiles.load(std::memory_order_acquire) || t > 255u) return false;
        for (int i = 0; i < 8; i++) if (tiles[i] != (i == (int)(t >> 5) ? 1u << (t & 31u) : 0u)) return false;
        return true;
    }
    // Set (before the cloud is published) when the producing call returned with its last kernel
    // still running: every consumer orders its stream after this event; never changed afterwards.
    hipEvent_t ready = nullptr;
    void mark_pending(hipStream_t producer);            // record `ready` on the producer's stream
    void wait_on(hipStream_t consumer) const {          // device-side wait, no host blocking
        if (ready && hipStreamWaitEvent(consumer, ready, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventSynchronize(ready);           // order on the host instead
        }
    }
    void wait_host() const {                            // for consumers outside the library's streams
        if (ready) (void)hipEventSynchronize(ready);
    }
    // The other direction: a call that returned with a kernel still READING these planes leaves an event
    // here (one per stream, the latest), and the planes are not given back to the pool before it.
    mutable std::mutex readers_mutex;
    mutable std::vector<std::pair<hipStream_t, hipEvent_t>> readers;
    void note_reader(hipStream_t consumer) const;
    float *x() const { return (float *)xyz_block->ptr; }
    float *y() const { return (float *)xyz_block->ptr + stride; }
    float *z() const { return (float *)xyz_block->ptr + 2 * stride; }
    uint32_t *rgbt() const { return (uint32_t *)rgbt_block->ptr; }
    ~DeviceSoA() {
        if (ready) {
            (void)hipEventSynchronize(ready);   // normally long complete
            event_put(ready);
        }
        for (auto &r : readers) {
            (void)hipEventSynchronize(r.second);
            event_put(r.second);
        }
        // (the blocks go back to the pool when the last cloud that holds them has come this far)
    }
};
std::shared_ptr<DeviceSoA> soa_alloc(size_t npoints);
// A cloud with `src`'s coordinates (the very planes) and colour / tile words of its own, still to be written.
std::shared_ptr<DeviceSoA> soa_with_new_rgbt(const std::shared_ptr<DeviceSoA> &src);
// ... and the other way round: `src`'s colour / tile words, coordinate planes of its own.
std::shared_ptr<DeviceSoA> soa_with_new_xyz(const std::shared_ptr<DeviceSoA> &src);

// A filter result that is still being computed when the call returns: its planes exist, its point count does not yet.
// cwipc_downsample (octree path) hands these out in a stream of frames, so that the host does not stand between one
// frame's kernels and the next one's; the first accessor that needs the points settles it (waits, and runs the pass
// again the slow way in the rare case that its assumptions did not hold).
struct DeferredResult {
    virtual ~DeferredResult() {}
    // Blocks until the result is there; nullptr = the filter failed (logged).  May be called from any thread, any
    // number of times.
    virtual std::shared_ptr<DeviceSoA> settle() = 0;
    // Results whose timestamp and cellsize are themselves part of what is still being computed (the multi-GPU join: the
    // minimum over the ranks' clouds) hand them over here, after settle(); false = they were known when the result was made.
    virtual bool late_metadata(uint64_t *timestamp, float *cellsize) { (void)timestamp; (void)cellsize; return false; }
};

// Host memory for point buffers: page-locked and pooled when a GPU is there (the DMA engines then
// read and write it directly, no staging copy), plain malloc otherwise.
void *host_alloc(size_t bytes, bool *pinned);
void host_free(void *ptr, bool pinned);
// memcpy split over a few threads for buffers of a megabyte and more (one core copies at 10-20 GB/s)
void parallel_memcpy(void *dst, const void *src, size_t bytes);
// [ptr, ptr + bytes) in page-locked memory the CALLER holds (cwipc_hip_host_alloc / cwipc_hip_host_register): the address a kernel
// reaches it under; nullptr for any other memory
void *host_range_device_alias(const void *ptr, size_t bytes);

// Host representation: AoS exactly as handed in through the C-ABI.
struct HostAoS {
    cwipc_point *points = nullptr;
    size_t npoints = 0;
    bool pinned = false;
    ~HostAoS() { host_free(points, pinned); }
};

// The cwipc_pointcloud implementation.  Either representation may be missing;
// the other one is materialised on demand (device: filters; host: accessors).
class cwipc_hip_pointcloud : public cwipc_pointcloud {
public:
    cwipc_hip_pointcloud();
    ~cwipc_hip_pointcloud() override;

    // cwipc_pointcloud interface
    void free() override;
    cwipc_pointcloud *_shallowcopy() override;
    uint64_t timestamp() override;
    float cellsize() override;
    void _set_cellsize(float cellsize) override;
    void _set_timestamp(uint64_t timestamp) override;
    int count() override;
    size_t get_uncompressed_size() override;
    int copy_uncompressed(struct cwipc_point *pointbuf, size_t size) override;
    size_t copy_packet(uint8_t *packet, size_t size) override;
    cwipc_pcl_pointcloud access_pcl_pointcloud() override;
    cwipc_metadata *access_metadata() override;

    // construction helpers
    int from_points(const cwipc_point *points, size_t size, int npoint, uint64_t timestamp, bool exact_size = true);
    void adopt_device(std::shared_ptr<DeviceSoA> dev, uint64_t timestamp, float cellsize, bool exact_size = false);
    void adopt_deferred(std::shared_ptr<DeferredResult> pending, uint64_t timestamp, float cellsize, bool late_metadata = false);
    // What a library thread needs of this cloud to work on it after the caller has gone on (or freed the cloud): the planes, or
    // the pending result that will become them.  A cloud that lives in host memory only is uploaded first (by the caller).
    struct Snapshot {
        std::shared_ptr<DeferredResult> pending;
        std::shared_ptr<DeviceSoA> dev;
        uint64_t timestamp = 0;
        float cellsize = 0;
        bool has_data = false;
    };
    Snapshot snapshot();

    // residency
    std::shared_ptr<DeviceSoA> device_points();   // uploads if needed; nullptr on failure
    std::shared_ptr<HostAoS> host_points();       // downloads if needed; nullptr on failure
    bool has_device() { settle(); return (bool)m_dev; }
    bool has_host() { settle(); return (bool)m_host; }
    bool drop_host();
    size_t npoints() { settle(); return m_npoints; }
    bool has_data() const { return m_has_data; }

private:
    int copy_impl(struct cwipc_point *pointbuf, size_t size, bool exact, bool dst_pinned = false);
    void settle();                // a deferred result becomes an ordinary device cloud (no-op otherwise)
    std::shared_ptr<DeferredResult> m_pending;
    std::mutex m_lock;
    uint64_t m_timestamp = 0;
    float m_cellsize = 0;
    size_t m_npoints = 0;
    bool m_has_data = false;      // counts towards cwipc_dangling_allocations
    bool m_exact_size = false;    // from_points flavour: copy_uncompressed wants size == exact
    bool m_late_metadata = false; // timestamp and cellsize come with the pending result
    std::shared_ptr<HostAoS> m_host;
    std::shared_ptr<DeviceSoA> m_dev;
    cwipc_metadata *m_metadata = nullptr;
};

// A cwipc_pointcloud that is not ours (another library's implementation) is
// read through its virtual interface into a temporary of ours.
std::unique_ptr<cwipc_hip_pointcloud> import_foreign(cwipc_pointcloud *pc);
cwipc_hip_pointcloud *as_ours(cwipc_pointcloud *pc);

void count_alloc();
void count_dealloc();

// ---------------------------------------------------------------------------
// kernel launchers (kernels_basic.hip, kernels_voxel.hip, kernels_sor.hip)
// All work on the calling thread's stream; none synchronises unless stated.
// ---------------------------------------------------------------------------
namespace k {

// AoS (device) <-> SoA
void aos_to_soa(const cwipc_point *aos, const DeviceSoA &dst, size_t n, hipStream_t s);
constexpr int MAX_SLOTS = 64;
void slots_to_soa(const void *slots, int nslots, size_t slot_rows, size_t header_rows, const uint32_t *counts, const DeviceSoA &dst, hipStream_t s);
void soa_to_aos(const DeviceSoA &src, cwipc_point *aos, size_t n, hipStream_t s);

// Stable compaction.  mode 0: tile == 0 || tile == pt.tile ; mode 1: half-open bbox ;
// mode 2: (pt.tile & tile) != 0 ; mode 3: !(dist[i] > thr)  (outlier removal).
struct Predicate {
    int mode;
    int tile;
    float bbox[6];
    const float *dist;
    double thr;
    const double *thr_dev;   // mode 3: device address of the threshold when a kernel computed it (thr is ignored then)
    // mode 3, small clouds (r4): the threshold is still 1024 pairs of partial sums (sor_stats_partial_device): the count kernel's workgroups run the
    // statistics' tree themselves -- each the same tree -- and the first one writes the threshold to thr_out (= thr_dev) for the scatter kernel
    const double *stat_partial;
    size_t stat_n;
    float stat_mul;
    double *thr_out;
};
// Pass 1: per-block keep counts into block_counts[nblocks]; returns nblocks through the argument.
size_t compact_blocks(size_t n);
size_t compact_small_cloud_limit();   // clouds up to here go through compact_count_scan
void compact_count(const DeviceSoA &src, const Predicate &p, uint32_t *block_counts, hipStream_t s);
// count and scan in one launch (the last workgroup to finish scans the block counts): for clouds of up to a million points,
// where a launch costs the host more than the kernel costs the device; `ticket`: a zeroed device word, left zeroed
bool compact_count_scan(const DeviceSoA &src, const Predicate &p, uint32_t *block_counts, uint32_t *ticket, unsigned long long *total_host, uint32_t tag,
                        hipStream_t s);
// Exclusive scan of block_counts in place; total written to *total_dev.
void compact_scan(uint32_t *block_counts, size_t nblocks, unsigned long long *total_host, uint32_t tag, hipStream_t s);
// Pass 2: scatter kept points to dst in input order.
// total_host != nullptr: block_offsets are the count kernel's COUNTS; every workgroup sums those in front of it, the first one publishes the total
void compact_scatter(const DeviceSoA &src, const Predicate &p, const uint32_t *block_offsets, const DeviceSoA &dst, hipStream_t s,
                     unsigned long long *total_host = nullptr, uint32_t tag = 0);

// Per-point maps on the rgbt word (x,y,z copied).
void map_tile(const DeviceSoA &src, const DeviceSoA &dst, const uint8_t *dev_map256, hipStream_t s);
void map_color_bits(const DeviceSoA &src, const DeviceSoA &dst, uint32_t clearBits, uint32_t setBits, hipStream_t s);
// mode 0: p' = R p + t with m = rows of [R | t]; mode 1: p' = (p + (m[3], m[7], m[11])) * m[0]; f64 arithmetic, one rounding to fp32
void map_affine(const DeviceSoA &src, const DeviceSoA &dst, const double m[12], int mode, hipStream_t s);
// the synthetic source's cloud written straight into device planes; the tables are device memory (see synthetic.cpp)
void synthetic_fill(const DeviceSoA &dst, int hsteps, int asteps, float m_angle, bool eyes_white, const float *radius, const float *height,
                    const float *angle, const double *sin_a, const double *cos_a, hipStream_t s);
// simulatecams (hard): tile = 1 << index of the camera direction (dirs: cos, sin per camera) nearest to the centred position
void map_cameras(const DeviceSoA &src, const DeviceSoA &dst, int ncam, float cen_x, float cen_z, const double *dirs, hipStream_t s);
// ORs the set of tile values that occur into 8 device words
void tiles_used(const DeviceSoA &src, uint32_t *dev_bits8, hipStream_t s);
// first256[t] = index of the first point with tile value t, 0xffffffff if there is none (256 device words)
void tile_first_index(const DeviceSoA &src, uint32_t *dev_first256, hipStream_t s);
// colorize: dev_table = 256 entries of {double cw[3]; double valid;} followed by 256 doubles old/255.0, then (1-w).
void map_colorize(const DeviceSoA &src, const DeviceSoA &dst, const double *dev_table, hipStream_t s);

// Concatenate nsrc clouds (device array of plane pointers) into dst.
struct JoinPart {
    const float *x, *y, *z;
    const uint32_t *rgbt;
    size_t n;
    size_t dst_offset;
};
void join_copy(const JoinPart &part, const DeviceSoA &dst, hipStream_t s);

}  // namespace k

// Voxel-grid downsample (kernels_voxel.hip).  Returns the new cloud's planes or
// nullptr (error already logged).  leaf_split = positive-cellsize path.
// With `deferred` (octree path only) the call may come back before its kernels are done: it then returns nullptr and
// leaves the pending result in *deferred.
void voxel_sample_streams();   // which of the calling thread's workspace streams are at work: call before the input is resolved (kernels_voxel.hip)
std::shared_ptr<DeviceSoA> voxel_downsample(const std::shared_ptr<DeviceSoA> &src, float cellsize, bool leaf_split, int *error_code,
                                            std::shared_ptr<DeferredResult> *deferred = nullptr);

// Statistical outlier removal (kernels_sor.hip).  Computes d_i into dev_dist
// (n floats, device), returns false on failure.
bool sor_mean_distances(const DeviceSoA &src, int k, float *dev_dist);
// mean/stddev threshold from d_i exactly as pcl::StatisticalOutlierRemoval; result in *thr.
bool sor_threshold(const float *dev_dist, size_t n, float stddev_mul, double *thr);
// The same on the device, nothing read back: *thr_dev (device memory) receives the threshold.
bool sor_threshold_device(const float *dev_dist, size_t n, float stddev_mul, double *thr_dev);
// Stable compaction of points with !(d_i > thr).
std::shared_ptr<DeviceSoA> sor_select(const DeviceSoA &src, const float *dev_dist, double thr, const double *thr_dev = nullptr);
// The outlier filter's last steps on the device: statistics, threshold, compaction.  Small clouds: the statistics' second kernel rides with the compaction's count.
std::shared_ptr<DeviceSoA> sor_threshold_and_select(const DeviceSoA &src, const float *dev_dist, float stddev_mul, double *thr_dev);

// Generic stable compaction driver used by tilefilter / crop / masked filter.
// may_return_early: the call may come back with the scatter kernel still running (the result carries a
// `ready` event); only for predicates that refer to nothing the caller frees afterwards.
std::shared_ptr<DeviceSoA> compact(const DeviceSoA &src, const k::Predicate &p, bool may_return_early = false);

}  // namespace cwipc_amd
