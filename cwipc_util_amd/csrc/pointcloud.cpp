// pointcloud.cpp -- the cwipc_pointcloud container and the point-buffer copy path.
//
// Reference: src/cwipc_util.cpp.  The reference keeps two host representations
// (malloc'd AoS in cwipc_uncompressed_impl, :312-410, and a PCL cloud in
// cwipc_impl, :94-303) and converts lazily between them.  Here the pair is
// host AoS <-> device SoA: clouds handed in through the C-ABI stay in host
// memory until a filter needs them (one H2D + de-interleave kernel), filter
// results stay in HBM until a host accessor needs them (interleave kernel +
// one D2H).  A chain of filters therefore crosses PCIe once in and once out.
#include "internal.hpp"

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace cwipc_amd {

// ---------------------------------------------------------------------------
// metadata collection (reference src/cwipc_util.cpp:24-87)
// ---------------------------------------------------------------------------
class metadata_impl : public cwipc_metadata {
    struct item {
        std::string name, description;
        void *pointer;
        size_t size;
        deallocfunc dealloc;
    };
    std::vector<item> m_items;

public:
    ~metadata_impl() override {
        for (auto &it : m_items) if (it.dealloc) it.dealloc(it.pointer);
    }
    int count() override { return (int)m_items.size(); }
    const std::string &name(int idx) override { return m_items[idx].name; }
    const std::string &description(int idx) override { return m_items[idx].description; }
    void *pointer(int idx) override { return m_items[idx].pointer; }
    size_t size(int idx) override { return m_items[idx].size; }
    void _add(const std::string &name, const std::string &description, void *pointer, size_t size, deallocfunc dealloc) override {
        m_items.push_back(item{name, description, pointer, size, dealloc});
    }
    // Moves OUR items to `other` (reference :78-86 appends to other_impl and clears this).
    void _move(cwipc_metadata *other) override {
        for (auto &it : m_items) other->_add(it.name, it.description, it.pointer, it.size, it.dealloc);
        m_items.clear();
    }
};

// ---------------------------------------------------------------------------
// allocation accounting (reference src/cwipc_util.cpp:89-93, 420-430)
// ---------------------------------------------------------------------------
static std::mutex g_alloc_mutex;
static int g_alloc = 0, g_dealloc = 0;

void count_alloc() { std::lock_guard<std::mutex> l(g_alloc_mutex); g_alloc++; }
void count_dealloc() { std::lock_guard<std::mutex> l(g_alloc_mutex); g_dealloc++; }

std::shared_ptr<DeviceSoA> soa_alloc(size_t npoints) {
    auto soa = std::make_shared<DeviceSoA>();
    soa->npoints = npoints;
    // Planes are padded to a multiple of 256 points so that a wave's 4-points-per-lane vector
    // loads of the last, partial step stay inside the plane without per-lane bounds checks.
    soa->stride = ((npoints + 255) / 256) * 256;
    if (soa->stride == 0) soa->stride = 256;
    soa->device = current_device();
    void *xyz = pool_alloc(soa->stride * 12), *rgbt = pool_alloc(soa->stride * 4);
    if (xyz) soa->xyz_block = std::make_shared<PlaneBlock>(xyz);
    if (rgbt) soa->rgbt_block = std::make_shared<PlaneBlock>(rgbt);
    if (!xyz || !rgbt) return nullptr;
    return soa;
}

std::shared_ptr<DeviceSoA> soa_with_new_xyz(const std::shared_ptr<DeviceSoA> &src) {
    auto soa = std::make_shared<DeviceSoA>();
    soa->npoints = src->npoints;
    soa->stride = src->stride;
    soa->device = src->device;
    soa->rgbt_block = src->rgbt_block;
    soa->set_tiles_from(*src);   // (the very same tile words)
    void *xyz = pool_alloc(soa->stride * 12);
    if (!xyz) return nullptr;
    soa->xyz_block = std::make_shared<PlaneBlock>(xyz);
    return soa;
}

std::shared_ptr<DeviceSoA> soa_with_new_rgbt(const std::shared_ptr<DeviceSoA> &src) {
    auto soa = std::make_shared<DeviceSoA>();
    soa->npoints = src->npoints;
    soa->stride = src->stride;
    soa->device = src->device;
    soa->xyz_block = src->xyz_block;
    void *rgbt = pool_alloc(soa->stride * 4);
    if (!rgbt) return nullptr;
    soa->rgbt_block = std::make_shared<PlaneBlock>(rgbt);
    return soa;
}

// ---------------------------------------------------------------------------
// cwipc_hip_pointcloud
// ---------------------------------------------------------------------------
cwipc_hip_pointcloud::cwipc_hip_pointcloud() {}
cwipc_hip_pointcloud::~cwipc_hip_pointcloud() { free(); }

// reference :149-163, :356-366 -- idempotent, releases data and metadata, keeps the shell object.
void cwipc_hip_pointcloud::free() {
    std::lock_guard<std::mutex> lock(m_lock);
    m_pending.reset();   // (a pass still in flight looks after its input, its output and its workspace itself)
    if (m_has_data) {
        count_dealloc();
        m_has_data = false;
    }
    m_host.reset();
    m_dev.reset();
    m_npoints = 0;
    delete m_metadata;
    m_metadata = nullptr;
}

// reference :114-118, :323-327 -- shares the point data, counts as one more allocation, metadata not copied.
cwipc_pointcloud *cwipc_hip_pointcloud::_shallowcopy() {
    settle();
    std::lock_guard<std::mutex> lock(m_lock);
    auto *rv = new cwipc_hip_pointcloud();
    rv->m_timestamp = m_timestamp;
    rv->m_cellsize = m_cellsize;
    rv->m_npoints = m_npoints;
    rv->m_host = m_host;
    rv->m_dev = m_dev;
    rv->m_exact_size = false;   // the reference's copy is a cwipc_impl (size >= needed accepted)
    if (m_has_data) {
        rv->m_has_data = true;
        count_alloc();
    }
    return rv;
}

uint64_t cwipc_hip_pointcloud::timestamp() {
    if (m_late_metadata) settle();
    return m_timestamp;
}
float cwipc_hip_pointcloud::cellsize() {
    if (m_late_metadata) settle();
    return m_cellsize;
}
void cwipc_hip_pointcloud::_set_timestamp(uint64_t timestamp) {
    if (m_late_metadata) settle();   // (what is being set must not be overwritten when the pending result arrives)
    m_timestamp = timestamp;
}

// reference :173-204 -- a negative value asks for the heuristic: minimum fp32
// distance between every point and the FIRST point (prevPoint never advances).
// Rare and O(N): done on the host copy.
void cwipc_hip_pointcloud::_set_cellsize(float cellsize) {
    if (cellsize < 0 || m_late_metadata) settle();
    if (cellsize < 0 && m_has_data) {
        auto host = host_points();
        float minDistance = std::numeric_limits<float>::infinity();
        if (host) {
            const cwipc_point *p = host->points;
            for (size_t i = 1; i < host->npoints; i++) {
                float dx = p[i].x - p[0].x, dy = p[i].y - p[0].y, dz = p[i].z - p[0].z;
                float d2 = dx * dx;
                d2 += dy * dy;
                d2 += dz * dz;
                float d = sqrtf(d2);
                if (d < minDistance) minDistance = d;
            }
        }
        if (minDistance == std::numeric_limits<float>::infinity()) minDistance = 0;
        cellsize = minDistance;
    }
    m_cellsize = cellsize;
}

int cwipc_hip_pointcloud::count() {
    settle();
    if (!m_has_data) {
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_util", "count: NULL pointcloud");
        return 0;
    }
    return (int)m_npoints;
}

size_t cwipc_hip_pointcloud::get_uncompressed_size() {
    settle();
    if (!m_has_data) {
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_util", "get_uncompressed_size: NULL pointcloud");
        return 0;
    }
    return m_npoints * sizeof(cwipc_point);
}

// reference :226-250 (size >= needed) and :393-401 (size == exact for from_points clouds).
int cwipc_hip_pointcloud::copy_uncompressed(struct cwipc_point *pointbuf, size_t size) {
    return copy_impl(pointbuf, size, m_exact_size);
}

int cwipc_hip_pointcloud::copy_impl(struct cwipc_point *pointbuf, size_t size, bool exact, bool dst_pinned) {
    settle();
    if (!m_has_data) {
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_util", "copy_uncompressed: NULL pointcloud");
        return 0;
    }
    size_t need = m_npoints * sizeof(cwipc_point);
    if (exact ? size != need : size < need) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "copy_uncompressed: buffer too small");
        return -1;
    }
    if (need == 0) return 0;
    {
        std::lock_guard<std::mutex> lock(m_lock);
        if (m_host) {
            parallel_memcpy(pointbuf, m_host->points, need);
            return (int)m_npoints;
        }
    }
    // Device-only cloud: interleave on the GPU and copy straight into the caller's buffer.
    std::shared_ptr<DeviceSoA> dev;
    {
        std::lock_guard<std::mutex> lock(m_lock);
        dev = m_dev;
    }
    if (!dev || !device_available("copy_uncompressed")) return -1;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return -1;
    // r4: a destination in page-locked memory of the caller is written by the interleave kernel itself, over PCIe
    if (void *alias = dst_pinned ? nullptr : host_range_device_alias(pointbuf, need)) {
        dev->wait_on(c.stream);
        k::soa_to_aos(*dev, (cwipc_point *)alias, m_npoints, c.stream);
        if (!c.sync()) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "copy_uncompressed: device to host copy failed");
            return -1;
        }
        return (int)m_npoints;
    }
    void *aos = pool_alloc(need);
    if (!aos) return -1;
    dev->wait_on(c.stream);
    k::soa_to_aos(*dev, (cwipc_point *)aos, m_npoints, c.stream);
    // a page-locked destination (our own host copy) is written by the DMA engine directly
    void *stage = dst_pinned ? (void *)pointbuf : c.staging(need);
    bool ok = stage != nullptr;
    if (ok) ok = hipMemcpyAsync(stage, aos, need, hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    ok = c.sync() && ok;
    pool_free(aos);
    if (!ok) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "copy_uncompressed: device to host copy failed");
        return -1;
    }
    if (!dst_pinned) parallel_memcpy(pointbuf, stage, need);
    return (int)m_npoints;
}

// reference :252-290 -- NULL packet asks for the size; otherwise size must match exactly.
size_t cwipc_hip_pointcloud::copy_packet(uint8_t *packet, size_t size) {
    if (!m_has_data) {
        cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_util", "copy_packet: NULL pointcloud");
        return 0;
    }
    size_t dataSize = get_uncompressed_size();
    size_t sizeNeeded = sizeof(cwipc_cwipcdump_header) + dataSize;
    if (packet == nullptr) return sizeNeeded;
    if (size != sizeNeeded) return 0;
    cwipc_cwipcdump_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    memcpy(hdr.hdr, CWIPC_CWIPCDUMP_HEADER, 4);
    hdr.magic = CWIPC_CWIPCDUMP_VERSION;
    hdr.timestamp = timestamp();
    hdr.cellsize = cellsize();
    hdr.unused = 0;
    hdr.size = dataSize;
    memcpy(packet, &hdr, sizeof(hdr));
    int n = copy_impl((cwipc_point *)(packet + sizeof(hdr)), dataSize, false);
    if (n < 0) return 0;
    return sizeNeeded;
}

// No PCL in this build: PCL-aware callers get an empty shared_ptr image (see api.h).
cwipc_pcl_pointcloud cwipc_hip_pointcloud::access_pcl_pointcloud() { return cwipc_pcl_pointcloud(); }

cwipc_metadata *cwipc_hip_pointcloud::access_metadata() {
    std::lock_guard<std::mutex> lock(m_lock);
    if (!m_metadata) m_metadata = new metadata_impl();
    return m_metadata;
}

// reference :329-354 -- validates npoint*16 == size, owns a copy.
int cwipc_hip_pointcloud::from_points(const cwipc_point *points, size_t size, int npoint, uint64_t timestamp, bool exact_size) {
    if (npoint < 0 || (size_t)npoint * sizeof(cwipc_point) != size) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "from_points: size and npoint inconsistent");
        return -1;
    }
    // r4: a buffer in page-locked memory the caller holds (cwipc_hip_host_alloc / cwipc_hip_host_register) is read by the
    // de-interleave kernel where it lies, over PCIe: the owned copy is the SoA planes in HBM, there is no copy on the host (one is
    // made from the device if somebody asks for host bytes).  The kernel is done when the call returns: the buffer is the caller's again.
    std::shared_ptr<DeviceSoA> dev;
    const cwipc_point *alias = size >= ((size_t)1 << 16) ? (const cwipc_point *)host_range_device_alias(points, size) : nullptr;
    if (alias && cwipc_hip_device_count() > 0 && current_device() < cwipc_hip_device_count()) {
        ThreadCtx &c = tctx();
        if (c.ensure()) {
            dev = soa_alloc((size_t)npoint);
            // The records through a DMA engine into device memory, then the de-interleave kernel (eight 4.8 MB camera tiles: 43.6 GB/s;
            // one 160 MB cloud: 55.7); CWIPC_PINNED_UPLOAD=kernel: the kernel reads the host buffer itself over PCIe (40.4 / 55.2 GB/s,
            // no device staging).  Ordinary memory, through the staging copy: 22-25 / 25-31 GB/s (scratch/pinned_upload.py).
            static const bool by_dma = []() { const char *e = getenv("CWIPC_PINNED_UPLOAD"); return !e || strcmp(e, "kernel") != 0; }();
            if (dev && by_dma) {
                void *aos = pool_alloc(size);
                bool ok = aos != nullptr && hipMemcpyAsync(aos, points, size, hipMemcpyHostToDevice, c.stream) == hipSuccess;
                if (ok) k::aos_to_soa((const cwipc_point *)aos, *dev, (size_t)npoint, c.stream);
                ok = c.sync() && ok;
                pool_free(aos);
                if (!ok) { (void)hipGetLastError(); dev.reset(); }
            } else if (dev) {
                k::aos_to_soa(alias, *dev, (size_t)npoint, c.stream);
                if (!c.sync()) { (void)hipGetLastError(); dev.reset(); }
            }
            if (dev) {
                dev->first[0] = points[0].x; dev->first[1] = points[0].y; dev->first[2] = points[0].z;
                dev->has_first = true;
            }
        }
    }
    std::shared_ptr<HostAoS> host;
    if (!dev) {
        host = std::make_shared<HostAoS>();
        host->npoints = (size_t)npoint;
        host->points = (cwipc_point *)host_alloc(size, &host->pinned);
        if (!host->points) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "from_points: could not allocate memory for points, size=" + std::to_string(size));
            return -1;
        }
        if (size) parallel_memcpy(host->points, points, size);
    }
    std::lock_guard<std::mutex> lock(m_lock);
    m_timestamp = timestamp;
    m_npoints = (size_t)npoint;
    m_host = host;
    m_dev = dev;
    m_exact_size = exact_size;   // (the reference's from_points clouds insist on their exact size in copy_uncompressed, its PCL-backed ones take any buffer that is large enough)
    if (!m_has_data) {
        m_has_data = true;
        count_alloc();
    }
    return npoint;
}

void cwipc_hip_pointcloud::adopt_device(std::shared_ptr<DeviceSoA> dev, uint64_t timestamp, float cellsize, bool exact_size) {
    std::lock_guard<std::mutex> lock(m_lock);
    m_timestamp = timestamp;
    m_cellsize = cellsize;
    m_npoints = dev ? dev->npoints : 0;
    m_dev = dev;
    m_host.reset();
    m_exact_size = exact_size;
    if (!m_has_data) {
        m_has_data = true;
        count_alloc();
    }
}

void cwipc_hip_pointcloud::adopt_deferred(std::shared_ptr<DeferredResult> pending, uint64_t timestamp, float cellsize, bool late_metadata) {
    std::lock_guard<std::mutex> lock(m_lock);
    m_timestamp = timestamp;
    m_cellsize = cellsize;
    m_late_metadata = late_metadata;
    m_npoints = 0;
    m_dev.reset();
    m_host.reset();
    m_pending = pending;
    m_exact_size = false;
    if (!m_has_data) {
        m_has_data = true;
        count_alloc();
    }
}

void cwipc_hip_pointcloud::settle() {
    std::shared_ptr<DeferredResult> p;
    {
        std::lock_guard<std::mutex> lock(m_lock);
        p = m_pending;
    }
    if (!p) return;
    std::shared_ptr<DeviceSoA> r = p->settle();   // (idempotent: two threads settling at once get the same planes)
    uint64_t late_ts = 0;
    float late_cs = 0;
    const bool late = p->late_metadata(&late_ts, &late_cs);
    std::lock_guard<std::mutex> lock(m_lock);
    if (m_pending != p) return;
    m_pending.reset();
    if (late && m_late_metadata) { m_timestamp = late_ts; m_cellsize = late_cs; }
    m_late_metadata = false;
    if (r) {
        m_dev = r;
        m_npoints = r->npoints;
    } else {
        // the filter failed after the call had returned (already logged): what the caller holds is an empty cloud
        m_host = std::make_shared<HostAoS>();
        m_npoints = 0;
    }
}

cwipc_hip_pointcloud::Snapshot cwipc_hip_pointcloud::snapshot() {
    Snapshot snap;
    {
        std::lock_guard<std::mutex> lock(m_lock);
        snap.pending = m_pending;
        if (!snap.pending) snap.dev = m_dev;
        snap.has_data = m_has_data;
    }
    if (!snap.pending && !snap.dev && snap.has_data) snap.dev = device_points();   // host copy only: upload now
    if (!snap.pending || !m_late_metadata) {
        snap.timestamp = timestamp();
        snap.cellsize = cellsize();
    } else {
        // a pending join as the input of another: its metadata come with it (settles here, rare)
        snap.timestamp = timestamp();
        snap.cellsize = cellsize();
        std::lock_guard<std::mutex> lock(m_lock);
        snap.pending.reset();
        snap.dev = m_dev;
    }
    return snap;
}

// H2D: pinned staging -> device AoS -> de-interleave kernel -> SoA planes.
std::shared_ptr<DeviceSoA> cwipc_hip_pointcloud::device_points() {
    settle();
    std::lock_guard<std::mutex> lock(m_lock);
    if (m_dev && m_dev->device == current_device()) {
        if (m_dev->ready) {   // result of a call whose last kernel may still be running: order this thread's stream after it
            ThreadCtx &c = tctx();
            if (!c.ensure()) return nullptr;
            m_dev->wait_on(c.stream);
        }
        return m_dev;
    }
    if (!m_has_data) return nullptr;
    if (!device_available("cwipc_pointcloud")) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    std::shared_ptr<HostAoS> host = m_host;
    if (!host) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_pointcloud", "cloud is resident on another device and has no host copy");
        return nullptr;
    }
    auto soa = soa_alloc(m_npoints);
    if (!soa) return nullptr;
    size_t bytes = m_npoints * sizeof(cwipc_point);
    if (bytes) {
        void *aos = pool_alloc(bytes);
        if (!aos) return nullptr;
        // Large uploads go through pinned staging in chunks so that the host memcpy of
        // chunk i+1 overlaps the DMA of chunk i.
        const size_t chunk = (size_t)16 << 20;
        bool ok = true;
        if (host->pinned) {
            // page-locked already: the DMA engine reads the cloud where it lies
            ok = hipMemcpyAsync(aos, host->points, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess;
        } else if (bytes <= chunk) {
            void *stage = c.staging(bytes);
            ok = stage != nullptr;
            if (ok) {
                memcpy(stage, host->points, bytes);
                ok = hipMemcpyAsync(aos, stage, bytes, hipMemcpyHostToDevice, c.stream) == hipSuccess;
            }
        } else {
            char *stage = (char *)c.staging(2 * chunk);
            ok = stage != nullptr;
            hipEvent_t done[2] = {nullptr, nullptr};
            if (ok) ok = hipEventCreateWithFlags(&done[0], hipEventDisableTiming) == hipSuccess &&
                         hipEventCreateWithFlags(&done[1], hipEventDisableTiming) == hipSuccess;
            size_t off = 0;
            int slot = 0;
            bool used[2] = {false, false};
            while (ok && off < bytes) {
                size_t len = bytes - off < chunk ? bytes - off : chunk;
                if (used[slot]) ok = hipEventSynchronize(done[slot]) == hipSuccess;
                if (!ok) break;
                memcpy(stage + slot * chunk, (const char *)host->points + off, len);
                ok = hipMemcpyAsync((char *)aos + off, stage + slot * chunk, len, hipMemcpyHostToDevice, c.stream) == hipSuccess &&
                     hipEventRecord(done[slot], c.stream) == hipSuccess;
                used[slot] = true;
                off += len;
                slot ^= 1;
            }
            if (done[0]) { (void)hipStreamSynchronize(c.stream); (void)hipEventDestroy(done[0]); }
            if (done[1]) (void)hipEventDestroy(done[1]);
        }
        if (ok) k::aos_to_soa((const cwipc_point *)aos, *soa, m_npoints, c.stream);
        ok = c.sync() && ok;
        pool_free(aos);
        if (!ok) {
            hip_failed(hipGetLastError(), "upload of point data", __FILE__, __LINE__);
            return nullptr;
        }
    }
    if (m_npoints) {
        soa->first[0] = host->points[0].x;
        soa->first[1] = host->points[0].y;
        soa->first[2] = host->points[0].z;
        soa->has_first = true;
    }
    m_dev = soa;
    return m_dev;
}

std::shared_ptr<HostAoS> cwipc_hip_pointcloud::host_points() {
    settle();
    {
        std::lock_guard<std::mutex> lock(m_lock);
        if (m_host) return m_host;
        if (!m_has_data) return nullptr;
    }
    auto host = std::make_shared<HostAoS>();
    host->npoints = m_npoints;
    size_t bytes = m_npoints * sizeof(cwipc_point);
    host->points = (cwipc_point *)host_alloc(bytes, &host->pinned);
    if (!host->points) return nullptr;
    if (bytes) {
        int n = copy_impl(host->points, bytes, false, host->pinned);
        if (n < 0) return nullptr;
    }
    std::lock_guard<std::mutex> lock(m_lock);
    if (!m_host) m_host = host;
    return m_host;
}

bool cwipc_hip_pointcloud::drop_host() {
    settle();
    std::lock_guard<std::mutex> lock(m_lock);
    if (!m_dev) return false;
    m_host.reset();
    return true;
}

// ---------------------------------------------------------------------------
// foreign clouds
// ---------------------------------------------------------------------------
cwipc_hip_pointcloud *as_ours(cwipc_pointcloud *pc) { return dynamic_cast<cwipc_hip_pointcloud *>(pc); }

std::unique_ptr<cwipc_hip_pointcloud> import_foreign(cwipc_pointcloud *pc) {
    size_t bytes = pc->get_uncompressed_size();
    std::vector<cwipc_point> tmp(bytes / sizeof(cwipc_point) + 1);
    int n = bytes ? pc->copy_uncompressed(tmp.data(), bytes) : 0;
    if (n < 0) return nullptr;
    std::unique_ptr<cwipc_hip_pointcloud> rv(new cwipc_hip_pointcloud());
    if (rv->from_points(tmp.data(), (size_t)n * sizeof(cwipc_point), n, pc->timestamp()) < 0) return nullptr;
    rv->_set_cellsize(pc->cellsize());
    return rv;
}

}  // namespace cwipc_amd

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------
using namespace cwipc_amd;

#define CW_STR2(x) #x
#define CW_STR(x) CW_STR2(x)

extern "C" const char *cwipc_get_version() {
#ifdef CWIPC_VERSION
    return CW_STR(CWIPC_VERSION);
#else
    return "unknown";
#endif
}

// reference src/cwipc_util.cpp:420-430
extern "C" int cwipc_dangling_allocations(bool log) {
    int alloc, dealloc;
    {
        std::lock_guard<std::mutex> l(g_alloc_mutex);
        alloc = g_alloc;
        dealloc = g_dealloc;
    }
    int dangling = alloc - dealloc;
    if (log && dangling != 0) {
        std::string msg = std::to_string(dangling) + " free() mismatch. nAlloc=" + std::to_string(alloc) + ", nFree=" + std::to_string(dealloc);
        _cwipc_log_emit(CWIPC_LOG_LEVEL_WARNING, "cwipc_pointcloud", msg.c_str());
    }
    return dangling < 0 ? -dangling : dangling;
}

// reference src/cwipc_util.cpp:662-683
extern "C" cwipc_pointcloud *cwipc_from_points(struct cwipc_point *points, size_t size, int npoint, uint64_t timestamp, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_from_points", apiVersion, errorMessage)) return nullptr;
    cwipc_log_set_errorbuf(errorMessage);
    auto *rv = new cwipc_hip_pointcloud();
    if (rv->from_points(points, size, npoint, timestamp) < 0) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_from_points", "cannot load points (size error?)");
        cwipc_log_set_errorbuf(nullptr);
        delete rv;
        return nullptr;
    }
    cwipc_log_set_errorbuf(nullptr);
    return rv;
}

// reference src/cwipc_util.cpp:685-729
extern "C" cwipc_pointcloud *cwipc_from_packet(uint8_t *packet, size_t size, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_from_packet", apiVersion, errorMessage)) return nullptr;
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_cwipcdump_header header;
    if (packet == nullptr || size < sizeof(header)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "cwipc_from_packet: incorrect packet header or version");
        cwipc_log_set_errorbuf(nullptr);
        return nullptr;
    }
    memcpy(&header, packet, sizeof(header));
    if (memcmp(header.hdr, CWIPC_CWIPCDUMP_HEADER, 4) != 0 || header.magic != CWIPC_CWIPCDUMP_VERSION) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "cwipc_from_packet: incorrect packet header or version");
        cwipc_log_set_errorbuf(nullptr);
        return nullptr;
    }
    size_t dataSize = size - sizeof(header);
    int npoint = (int)(header.size / sizeof(cwipc_point));
    if ((size_t)npoint * sizeof(cwipc_point) != dataSize) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_util", "cwipc_from_packet: inconsistent dataSize");
        cwipc_log_set_errorbuf(nullptr);
        return nullptr;
    }
    auto *rv = new cwipc_hip_pointcloud();
    if (rv->from_points((const cwipc_point *)(packet + sizeof(header)), dataSize, npoint, header.timestamp) < 0) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_from_packet", "cannot load points (size error?)");
        cwipc_log_set_errorbuf(nullptr);
        delete rv;
        return nullptr;
    }
    rv->_set_cellsize(header.cellsize);
    cwipc_log_set_errorbuf(nullptr);
    return rv;
}

// ---------------------------------------------------------------------------
// The proxy's wire format as a packet codec (no sockets: the transport stays the application's).
// reference src/cwipc_proxy.cpp:179-216 (what the server reads: a 24-byte cwipc_point_packetheader, api.h:100-110, then
// dataCount bytes of cwipc_point records; it answers with the 8 bytes of the timestamp) and
// python/cwipc/scripts/cwipc_toproxy.py:51-57 (what the sender writes: struct.pack("<iiqfi", magic, len, timestamp, cellsize, 0)).
// The two do not agree on the magic number -- C: 0x20201016 (api.h:110), Python: 0x20210208 (util.py:346) -- so a reference
// sender cannot talk to a reference server (its proxy tests are skipped, test_cwipc_util.py:617-633).  Not "fixed" here: the
// writer takes the magic it is told (default: the C one), the reader accepts the C one and, only when asked to, the Python one.
// ---------------------------------------------------------------------------
static_assert(sizeof(cwipc_point_packetheader) == 24, "the proxy header is 24 bytes on the wire");

extern "C" size_t cwipc_hip_proxy_packet(cwipc_pointcloud *pc, uint8_t *packet, size_t size, uint32_t magic) {
    if (pc == nullptr) return 0;
    const size_t data = pc->get_uncompressed_size();
    const size_t need = sizeof(cwipc_point_packetheader) + data;
    if (packet == nullptr) return need;   // (how much room the packet takes)
    if (size < need || data > 0xffffffffull) return 0;
    cwipc_point_packetheader h;
    memset(&h, 0, sizeof(h));
    h.magic = magic ? magic : (uint32_t)CWIPC_POINT_PACKETHEADER_MAGIC;
    h.dataCount = (uint32_t)data;
    h.timestamp = pc->timestamp();
    h.cellsize = pc->cellsize();
    memcpy(packet, &h, sizeof(h));
    if (data && pc->copy_uncompressed(reinterpret_cast<cwipc_point *>(packet + sizeof(h)), data) < 0) return 0;
    return need;
}

extern "C" cwipc_pointcloud *cwipc_hip_from_proxy_packet(const uint8_t *packet, size_t size, int accept_python_magic, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_hip_from_proxy_packet", apiVersion, errorMessage)) return nullptr;
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_pointcloud *out = nullptr;
    do {
        cwipc_point_packetheader h;
        if (packet == nullptr || size < sizeof(h)) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_proxy", "short packet header");
            break;
        }
        memcpy(&h, packet, sizeof(h));
        if (h.magic != (uint32_t)CWIPC_POINT_PACKETHEADER_MAGIC && !(accept_python_magic && h.magic == 0x20210208u)) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_proxy", "invalid magic number in packet header");   // (the server's words, cwipc_proxy.cpp:188)
            break;
        }
        if ((size_t)h.dataCount != size - sizeof(h)) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_proxy", "packet length does not match dataCount");
            break;
        }
        // as the server: cwipc_from_points(points, dataCount, dataCount / sizeof(cwipc_point), timestamp), then the cellsize (:204-213)
        auto *rv = new cwipc_hip_pointcloud();
        if (rv->from_points(reinterpret_cast<const cwipc_point *>(packet + sizeof(h)), h.dataCount, (int)(h.dataCount / sizeof(cwipc_point)), h.timestamp) < 0) {
            delete rv;   // (from_points has logged "incorrect size" / ...)
            break;
        }
        rv->_set_cellsize(h.cellsize);
        out = rv;
    } while (false);
    cwipc_log_set_errorbuf(nullptr);
    return out;
}

// reference src/cwipc_util.cpp:499-580 -- same 32-byte header + AoS payload as a packet.
extern "C" cwipc_pointcloud *cwipc_read_debugdump(const char *filename, char **errorMessage, uint64_t apiVersion) {
    if (api_version_rejected("cwipc_read_debugdump", apiVersion, errorMessage)) return nullptr;
    cwipc_log_set_errorbuf(errorMessage);
    cwipc_pointcloud *rv = nullptr;
    FILE *fp = fopen(filename, "rb");
    cwipc_point *data = nullptr;
    do {
        if (!fp) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", std::string("Cannot open file: ") + filename);
            break;
        }
        cwipc_cwipcdump_header hdr;
        if (fread(&hdr, 1, sizeof(hdr), fp) != sizeof(hdr)) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", std::string("Cannot read pointcloud dumpfile header: ") + filename);
            break;
        }
        if (memcmp(hdr.hdr, CWIPC_CWIPCDUMP_HEADER, 4) != 0) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", std::string("Pointcloud dumpfile header incorrect: ") + filename);
            break;
        }
        if (hdr.magic != CWIPC_CWIPCDUMP_VERSION) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", std::string("Pointcloud dumpfile version incorrect: ") + filename);
            break;
        }
        size_t dataSize = hdr.size;
        int npoint = (int)(dataSize / sizeof(cwipc_point));
        if ((size_t)npoint * sizeof(cwipc_point) != dataSize) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", "Pointcloud dumpfile datasize inconsistent");
            break;
        }
        data = (cwipc_point *)malloc(dataSize ? dataSize : 1);
        if (!data) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", "Could not allocate memory for point data");
            break;
        }
        if (fread(data, 1, dataSize, fp) != dataSize) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_read_debugdump", "Could not read point data of correct size");
            break;
        }
        auto *pc = new cwipc_hip_pointcloud();
        if (pc->from_points(data, dataSize, npoint, hdr.timestamp) < 0) {
            delete pc;
            break;
        }
        pc->_set_cellsize(hdr.cellsize);
        rv = pc;
    } while (0);
    if (fp) fclose(fp);
    ::free(data);
    cwipc_log_set_errorbuf(nullptr);
    return rv;
}

// reference src/cwipc_util.cpp:582-641
extern "C" int cwipc_write_debugdump(const char *filename, cwipc_pointcloud *pointcloud, char **errorMessage) {
    cwipc_log_set_errorbuf(errorMessage);
    int status = -1;
    size_t dataSize = pointcloud->get_uncompressed_size();
    cwipc_point *buf = (cwipc_point *)malloc(dataSize ? dataSize : 1);
    FILE *fp = nullptr;
    do {
        if (!buf) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_write_debugdump", "Cannot allocate memory, size=" + std::to_string(dataSize));
            break;
        }
        int nPoint = pointcloud->copy_uncompressed(buf, dataSize);
        if (nPoint < 0) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_write_debugdump", "Cannot copy points, size=" + std::to_string(dataSize));
            break;
        }
        fp = fopen(filename, "wb");
        if (!fp) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_write_debugdump", std::string("Cannot open output file: ") + filename);
            break;
        }
        cwipc_cwipcdump_header hdr;
        memset(&hdr, 0, sizeof(hdr));
        memcpy(hdr.hdr, CWIPC_CWIPCDUMP_HEADER, 4);
        hdr.magic = CWIPC_CWIPCDUMP_VERSION;
        hdr.timestamp = pointcloud->timestamp();
        hdr.cellsize = pointcloud->cellsize();
        hdr.size = dataSize;
        fwrite(&hdr, sizeof(hdr), 1, fp);
        if (fwrite(buf, sizeof(cwipc_point), (size_t)nPoint, fp) != (size_t)nPoint) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_write_debugdump", "Cannot write point data, nPoint=" + std::to_string(nPoint));
            break;
        }
        status = 0;
    } while (0);
    if (fp) fclose(fp);
    ::free(buf);
    cwipc_log_set_errorbuf(nullptr);
    return status;
}

// reference src/cwipc_util.cpp:731-797 -- thin forwards to the virtuals
extern "C" void cwipc_pointcloud_free(cwipc_pointcloud *pc) { pc->free(); }
extern "C" cwipc_pointcloud *cwipc_pointcloud__shallowcopy(cwipc_pointcloud *pc) { return pc->_shallowcopy(); }
extern "C" uint64_t cwipc_pointcloud_timestamp(cwipc_pointcloud *pc) { return pc->timestamp(); }
extern "C" float cwipc_pointcloud_cellsize(cwipc_pointcloud *pc) { return pc->cellsize(); }
extern "C" void cwipc_pointcloud__set_cellsize(cwipc_pointcloud *pc, float cellsize) { pc->_set_cellsize(cellsize); }
extern "C" void cwipc_pointcloud__set_timestamp(cwipc_pointcloud *pc, uint64_t timestamp) { pc->_set_timestamp(timestamp); }
extern "C" int cwipc_pointcloud_count(cwipc_pointcloud *pc) { return pc->count(); }
extern "C" size_t cwipc_pointcloud_get_uncompressed_size(cwipc_pointcloud *pc) { return pc->get_uncompressed_size(); }
extern "C" int cwipc_pointcloud_copy_uncompressed(cwipc_pointcloud *pc, struct cwipc_point *points, size_t size) { return pc->copy_uncompressed(points, size); }
extern "C" size_t cwipc_pointcloud_copy_packet(cwipc_pointcloud *pc, uint8_t *packet, size_t size) { return pc->copy_packet(packet, size); }
extern "C" cwipc_metadata *cwipc_pointcloud_access_metadata(cwipc_pointcloud *pc) { return pc->access_metadata(); }

extern "C" void cwipc_metadata__move(cwipc_metadata *src, cwipc_metadata *dest) { src->_move(dest); }
extern "C" int cwipc_metadata_count(cwipc_metadata *collection) { return collection->count(); }
extern "C" const char *cwipc_metadata_name(cwipc_metadata *collection, int idx) { return collection->name(idx).c_str(); }
extern "C" const char *cwipc_metadata_description(cwipc_metadata *collection, int idx) { return collection->description(idx).c_str(); }
extern "C" void *cwipc_metadata_pointer(cwipc_metadata *collection, int idx) { return collection->pointer(idx); }
extern "C" size_t cwipc_metadata_size(cwipc_metadata *collection, int idx) { return collection->size(idx); }

extern "C" bool cwipc_activesource_start(cwipc_activesource *src) { return src->start(); }
extern "C" void cwipc_activesource_stop(cwipc_activesource *src) { src->stop(); }
extern "C" cwipc_pointcloud *cwipc_source_get(cwipc_source *src) { return src->get(); }
extern "C" void cwipc_source_free(cwipc_source *src) { src->free(); }
extern "C" bool cwipc_source_eof(cwipc_source *src) { return src->eof(); }
extern "C" bool cwipc_source_available(cwipc_source *src, bool wait) { return src->available(wait); }
extern "C" void cwipc_activesource_request_metadata(cwipc_activesource *src, const char *name) { src->request_metadata(name); }
extern "C" bool cwipc_activesource_is_metadata_requested(cwipc_activesource *src, const char *name) { return src->is_metadata_requested(name); }
extern "C" bool cwipc_activesource_reload_config(cwipc_activesource *src, const char *configFile) { return src->reload_config(configFile); }
extern "C" size_t cwipc_activesource_get_config(cwipc_activesource *src, char *buffer, size_t size) { return src->get_config(buffer, size); }
extern "C" bool cwipc_activesource_seek(cwipc_activesource *src, uint64_t timestamp) { return src->seek(timestamp); }
extern "C" int cwipc_activesource_maxtile(cwipc_activesource *src) { return src->maxtile(); }
extern "C" bool cwipc_activesource_get_tileinfo(cwipc_activesource *src, int tilenum, struct cwipc_tileinfo *tileinfo) { return src->get_tileinfo(tilenum, tileinfo); }
extern "C" bool cwipc_activesource_auxiliary_operation(cwipc_activesource *src, const char *op, const void *inbuf, size_t insize, void *outbuf, size_t outsize) {
    return src->auxiliary_operation(std::string(op), inbuf, insize, outbuf, outsize);
}
extern "C" void cwipc_sink_free(cwipc_sink *sink) { sink->free(); }
extern "C" bool cwipc_sink_feed(cwipc_sink *sink, cwipc_pointcloud *pc, bool clear) { return sink->feed(pc, clear); }
extern "C" bool cwipc_sink_caption(cwipc_sink *sink, const char *caption) { return sink->caption(caption); }
extern "C" char cwipc_sink_interact(cwipc_sink *sink, const char *prompt, const char *responses, int32_t millis) { return sink->interact(prompt, responses, millis); }

// ---------------------------------------------------------------------------
// residency extensions
// ---------------------------------------------------------------------------
extern "C" int cwipc_hip_upload(cwipc_pointcloud *pc) {
    auto *ours = as_ours(pc);
    if (!ours) return -1;
    return ours->device_points() ? 0 : -1;
}

extern "C" int cwipc_hip_drop_host_copy(cwipc_pointcloud *pc) {
    auto *ours = as_ours(pc);
    if (!ours) return -1;
    return ours->drop_host() ? 0 : -1;
}

extern "C" int cwipc_hip_is_device_resident(cwipc_pointcloud *pc) {
    auto *ours = as_ours(pc);
    return ours && ours->has_device() ? 1 : 0;
}

extern "C" int cwipc_hip_device_planes(cwipc_pointcloud *pc, const float **x, const float **y, const float **z, const uint32_t **rgbt, size_t *npoint) {
    auto *ours = as_ours(pc);
    if (!ours) return -1;
    auto dev = ours->device_points();
    if (!dev) return -1;
    dev->wait_host();   // the caller's streams are not ours: hand out finished data only
    if (x) *x = dev->x();
    if (y) *y = dev->y();
    if (z) *z = dev->z();
    if (rgbt) *rgbt = dev->rgbt();
    if (npoint) *npoint = dev->npoints;
    return 0;
}

extern "C" long cwipc_hip_copy_device_aos(cwipc_pointcloud *pc, void *dev_points, size_t size) {
    auto *ours = as_ours(pc);
    if (!ours) return -1;
    auto dev = ours->device_points();
    if (!dev) return -1;
    if (size < dev->npoints * sizeof(cwipc_point)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_copy_device_aos", "buffer too small");
        return -1;
    }
    ThreadCtx &c = tctx();
    if (!c.ensure()) return -1;
    if (dev->npoints) k::soa_to_aos(*dev, (cwipc_point *)dev_points, dev->npoints, c.stream);
    if (!c.sync()) return -1;
    return (long)dev->npoints;
}

// The same as a step of the CALLER's stream (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream): the
// kernel is ordered behind whatever that stream holds and in front of what the caller puts there next (a
// collective that sends the buffer), and the call does not wait for it.
extern "C" long cwipc_hip_copy_device_aos_on_stream(cwipc_pointcloud *pc, void *dev_points, size_t size, void *stream) {
    auto *ours = as_ours(pc);
    if (!ours) return -1;
    auto dev = ours->device_points();
    if (!dev) return -1;
    if (size < dev->npoints * sizeof(cwipc_point)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_copy_device_aos_on_stream", "buffer too small");
        return -1;
    }
    ThreadCtx &c = tctx();
    if (!c.ensure()) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dev->npoints) {
        dev->wait_on(s);                                       // the cloud may still be in the making on another stream
        k::soa_to_aos(*dev, (cwipc_point *)dev_points, dev->npoints, s);
        if (hipGetLastError() != hipSuccess) return -1;
        dev->note_reader(s);                                   // ... and must stay until this kernel has read it
    }
    return (long)dev->npoints;
}

extern "C" cwipc_pointcloud *cwipc_hip_from_device_slots(const void *dev_slots, int nslots, size_t slot_rows, size_t header_rows,
                                                        const uint32_t *counts, uint64_t timestamp, float cellsize) {
    if (!device_available("cwipc_hip_from_device_slots")) return nullptr;
    if (nslots < 0 || nslots > k::MAX_SLOTS || (nslots > 0 && (dev_slots == nullptr || counts == nullptr))) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_from_device_slots", "bad arguments (at most 64 slots)");
        return nullptr;
    }
    size_t total = 0;
    for (int i = 0; i < nslots; i++) {
        if (header_rows + counts[i] > slot_rows) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_from_device_slots", "a slot's count exceeds its rows");
            return nullptr;
        }
        total += counts[i];
    }
    if (total >= ((size_t)1 << 32)) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto soa = soa_alloc(total);
    if (!soa) return nullptr;
    if (total) k::slots_to_soa(dev_slots, nslots, slot_rows, header_rows, counts, *soa, c.stream);
    if (!c.sync()) return nullptr;   // the caller may reuse its receive buffer
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_device(soa, timestamp, cellsize);
    return rv;
}

// The same as a step of the caller's stream (see cwipc_hip_copy_device_aos_on_stream): behind the collective that fills
// the receive buffer, in front of the next use the caller makes of that buffer on this stream; the new cloud carries a
// `ready` event, the call does not wait.
extern "C" cwipc_pointcloud *cwipc_hip_from_device_slots_on_stream(const void *dev_slots, int nslots, size_t slot_rows, size_t header_rows,
                                                                  const uint32_t *counts, uint64_t timestamp, float cellsize, void *stream) {
    if (!device_available("cwipc_hip_from_device_slots_on_stream")) return nullptr;
    if (nslots < 0 || nslots > k::MAX_SLOTS || (nslots > 0 && (dev_slots == nullptr || counts == nullptr))) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_from_device_slots_on_stream", "bad arguments (at most 64 slots)");
        return nullptr;
    }
    size_t total = 0;
    for (int i = 0; i < nslots; i++) {
        if (header_rows + counts[i] > slot_rows) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_from_device_slots_on_stream", "a slot's count exceeds its rows");
            return nullptr;
        }
        total += counts[i];
    }
    if (total >= ((size_t)1 << 32)) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto soa = soa_alloc(total);
    if (!soa) return nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (total) {
        k::slots_to_soa(dev_slots, nslots, slot_rows, header_rows, counts, *soa, s);
        if (hipGetLastError() != hipSuccess) return nullptr;
        soa->mark_pending(s);
    }
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_device(soa, timestamp, cellsize);
    return rv;
}

extern "C" cwipc_pointcloud *cwipc_hip_from_device_aos(const void *dev_points, size_t npoint, uint64_t timestamp, float cellsize) {
    if (!device_available("cwipc_hip_from_device_aos")) return nullptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    auto soa = soa_alloc(npoint);
    if (!soa) return nullptr;
    if (npoint) k::aos_to_soa((const cwipc_point *)dev_points, *soa, npoint, c.stream);
    if (!c.sync()) return nullptr;
    auto *rv = new cwipc_hip_pointcloud();
    rv->adopt_device(soa, timestamp, cellsize);
    return rv;
}
