// device.cpp -- device selection, per-thread stream context, device memory pool,
// per-kernel profiling.  No reference counterpart: the reference is CPU-only.
//
// Re-entrancy: the reference is called from several threads at once (ctypes
// releases the GIL; reference python/cwipc/net/source_synchronizer.py:17,184),
// so every thread gets its own HIP stream, pinned staging buffer and scratch;
// the pool and the profile table are the only shared state and are locked.
#include "internal.hpp"

#include <pthread.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>
#include <unordered_map>

namespace cwipc_amd {

namespace {

std::atomic<int> g_device{-2};   // -2: not decided yet
std::atomic<int> g_device_count{-1};
thread_local std::string t_last_error;

int device_count() {
    int n = g_device_count.load();
    if (n >= 0) return n;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) {
        (void)hipGetLastError();
        cnt = 0;
    }
    g_device_count.store(cnt);
    return cnt;
}

}  // namespace

bool hip_failed(hipError_t err, const char *what, const char *file, int line) {
    std::string msg = std::string(what) + " failed: " + hipGetErrorString(err) + " (" + file + ":" + std::to_string(line) + ")";
    t_last_error = msg;
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip", msg);
    return false;
}

int current_device() {
    int d = g_device.load();
    if (d == -2) {
        const char *env = getenv("CWIPC_HIP_DEVICE");
        d = env ? atoi(env) : 0;
        g_device.store(d);
    }
    return d;
}

bool device_available(const char *who) {
    if (device_count() > 0 && current_device() < device_count()) return true;
    std::string msg = "no usable HIP device (hipGetDeviceCount=" + std::to_string(device_count()) +
                      ", selected " + std::to_string(current_device()) + "); the filters have no CPU fallback";
    t_last_error = msg;
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, who, msg);
    return false;
}

// ---------------------------------------------------------------------------
// memory pool
// ---------------------------------------------------------------------------
namespace {

std::mutex g_pool_mutex;
std::map<std::pair<int, size_t>, std::vector<void *>> g_pool_free;   // (device, class) -> blocks
std::unordered_map<void *, std::pair<int, size_t>> g_pool_live;      // block -> (device, class)
size_t g_pool_bytes = 0;

// Classes: powers of two up to 1 MiB, then eighths of the enclosing power of two (<= 12.5 % slack).
size_t size_class(size_t bytes) {
    if (bytes < 256) bytes = 256;
    size_t p = 256;
    while (p < bytes && p < (1u << 20)) p <<= 1;
    if (p >= bytes) return p;
    p = (size_t)1 << 20;
    while ((p << 1) <= bytes) p <<= 1;     // p <= bytes < 2p
    size_t step = p >> 3;
    return ((bytes + step - 1) / step) * step;
}

}  // namespace

void *pool_alloc(size_t bytes) {
    int dev = current_device();
    size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        auto it = g_pool_free.find({dev, cls});
        if (it != g_pool_free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pool_live[p] = {dev, cls};
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {
        // give cached blocks back and retry once
        cwipc_hip_pool_trim();
        e = hipMalloc(&p, cls);
        if (e != hipSuccess) {
            hip_failed(e, "hipMalloc", __FILE__, __LINE__);
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool_live[p] = {dev, cls};
    g_pool_bytes += cls;
    return p;
}

// ---------------------------------------------------------------------------
// pinned host pool, parallel memcpy
// ---------------------------------------------------------------------------
namespace {
std::mutex g_host_mutex;
std::map<size_t, std::vector<void *>> g_host_free;     // class -> page-locked blocks
std::unordered_map<void *, size_t> g_host_live;        // block -> class
size_t g_host_cached_bytes = 0;
constexpr size_t HOST_CACHE_LIMIT = (size_t)4 << 30;   // page-locked memory kept for reuse
}  // namespace

void *host_alloc(size_t bytes, bool *pinned) {
    *pinned = false;
    if (bytes < ((size_t)1 << 16) || device_count() < 1) return malloc(bytes ? bytes : 1);   // small buffers: pinning does not pay
    const size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        auto it = g_host_free.find(cls);
        if (it != g_host_free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_host_cached_bytes -= cls;
            g_host_live[p] = cls;
            *pinned = true;
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return malloc(bytes ? bytes : 1);
    }
    std::lock_guard<std::mutex> lock(g_host_mutex);
    g_host_live[p] = cls;
    *pinned = true;
    return p;
}

void host_free(void *ptr, bool pinned) {
    if (!ptr) return;
    if (!pinned) { ::free(ptr); return; }
    size_t cls = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        auto it = g_host_live.find(ptr);
        if (it == g_host_live.end()) return;
        cls = it->second;
        g_host_live.erase(it);
        if (g_host_cached_bytes + cls <= HOST_CACHE_LIMIT) {
            g_host_free[cls].push_back(ptr);
            g_host_cached_bytes += cls;
            keep = true;
        }
    }
    if (!keep) (void)hipHostFree(ptr);
}

// A few threads that live as long as the process copy big buffers in pieces (round 4: a thread per piece and call, as in rounds
// 1-3, cost more than a 5 MB camera tile's copy took; clouds of 1 MB and more now go in pieces of at least 256 KB).
namespace {
struct CopyPool {
    struct Task { char *dst; const char *src; size_t len; };
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<Task> tasks;
    size_t pending = 0;
    std::vector<std::thread> workers;
    explicit CopyPool(unsigned n) {
        for (unsigned i = 0; i < n; i++)
            workers.emplace_back([this]() {
                for (;;) {
                    Task t;
                    {
                        std::unique_lock<std::mutex> g(m);
                        cv_work.wait(g, [&] { return !tasks.empty(); });
                        t = tasks.front();
                        tasks.pop_front();
                    }
                    memcpy(t.dst, t.src, t.len);
                    {
                        std::lock_guard<std::mutex> g(m);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
        for (auto &w : workers) w.detach();   // (they sleep on the queue; the process may end under them)
    }
};
// (a process that forks takes the pool's memory along but not its threads: the child starts a pool of its own at its first big copy)
std::atomic<CopyPool *> g_copy_pool{nullptr};
std::mutex g_copy_pool_make;
std::mutex *g_copy_calls = new std::mutex();   // one parallel copy at a time (the callers' own threads are the parallelism beyond that)
void copy_pool_after_fork() {
    g_copy_pool.store(nullptr);                 // the parent's pool object is abandoned in the child (its threads do not exist here)
    new (&g_copy_pool_make) std::mutex();       // either mutex may have been held by a thread that is not in the child
    g_copy_calls = new std::mutex();
}
CopyPool *copy_pool() {
    CopyPool *p = g_copy_pool.load(std::memory_order_acquire);
    if (p) return p;
    std::lock_guard<std::mutex> g(g_copy_pool_make);
    p = g_copy_pool.load(std::memory_order_acquire);
    if (!p) {
        static bool registered = false;
        if (!registered) { (void)pthread_atfork(nullptr, nullptr, copy_pool_after_fork); registered = true; }
        const unsigned hw = std::thread::hardware_concurrency();
        p = new CopyPool(hw >= 8 ? 3u : hw >= 4 ? 2u : 1u);   // never destroyed: threads may outlive the statics
        g_copy_pool.store(p, std::memory_order_release);
    }
    return p;
}
}  // namespace

void parallel_memcpy(void *dst, const void *src, size_t bytes) {
    const size_t min_piece = (size_t)256 << 10;
    std::mutex *const calls = g_copy_calls;
    if (bytes < ((size_t)1 << 20) || std::thread::hardware_concurrency() < 2 || !calls->try_lock()) { memcpy(dst, src, bytes); return; }
    std::lock_guard<std::mutex> one(*calls, std::adopt_lock);
    CopyPool *pool = copy_pool();
    const size_t parts = std::min<size_t>(pool->workers.size() + 1, bytes / min_piece);
    const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
    size_t queued = 0;
    {
        std::lock_guard<std::mutex> g(pool->m);
        for (size_t off = per; off < bytes; off += per) {
            pool->tasks.push_back({(char *)dst + off, (const char *)src + off, std::min(per, bytes - off)});
            pool->pending++;
            queued++;
        }
    }
    if (queued) pool->cv_work.notify_all();
    memcpy(dst, src, std::min(per, bytes));   // this thread takes the first piece
    if (queued) {
        std::unique_lock<std::mutex> g(pool->m);
        pool->cv_done.wait(g, [&] { return pool->pending == 0; });
    }
}

// ---------------------------------------------------------------------------
// page-locked buffers of the CALLER (round 4): memory from cwipc_hip_host_alloc, or the caller's own registered with
// cwipc_hip_host_register.  cwipc_from_points on such a buffer reads it straight from the device (no staging copy on the host),
// copy_uncompressed into one lets the DMA engine write it directly.
// ---------------------------------------------------------------------------
namespace {
std::mutex g_user_pinned_mutex;
struct UserPinned { size_t bytes; bool ours; uintptr_t dev_base; };   // ours: to free with hipHostFree / else registered by the caller; dev_base: the device's address of the first byte
std::map<uintptr_t, UserPinned> g_user_pinned;   // by start address
}  // namespace

void *host_range_device_alias(const void *ptr, size_t bytes) {
    if (!ptr || !bytes) return nullptr;
    std::lock_guard<std::mutex> lock(g_user_pinned_mutex);
    if (g_user_pinned.empty()) return nullptr;
    auto it = g_user_pinned.upper_bound((uintptr_t)ptr);
    if (it == g_user_pinned.begin()) return nullptr;
    --it;
    if ((uintptr_t)ptr < it->first || (uintptr_t)ptr + bytes > it->first + it->second.bytes) return nullptr;
    return (void *)(it->second.dev_base + ((uintptr_t)ptr - it->first));
}

extern "C" void *cwipc_hip_host_alloc(size_t bytes) {
    if (!bytes || device_count() < 1) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) { (void)hipGetLastError(); d = p; }
    std::lock_guard<std::mutex> lock(g_user_pinned_mutex);
    g_user_pinned[(uintptr_t)p] = UserPinned{bytes, true, (uintptr_t)d};
    return p;
}

extern "C" void cwipc_hip_host_free(void *ptr) {
    if (!ptr) return;
    {
        std::lock_guard<std::mutex> lock(g_user_pinned_mutex);
        auto it = g_user_pinned.find((uintptr_t)ptr);
        if (it == g_user_pinned.end() || !it->second.ours) return;
        g_user_pinned.erase(it);
    }
    (void)hipHostFree(ptr);
}

extern "C" int cwipc_hip_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes || device_count() < 1) return -1;
    if (hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return -1; }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, ptr, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostUnregister(ptr); return -1; }
    std::lock_guard<std::mutex> lock(g_user_pinned_mutex);
    g_user_pinned[(uintptr_t)ptr] = UserPinned{bytes, false, (uintptr_t)d};
    return 0;
}

extern "C" int cwipc_hip_host_unregister(void *ptr) {
    if (!ptr) return -1;
    {
        std::lock_guard<std::mutex> lock(g_user_pinned_mutex);
        auto it = g_user_pinned.find((uintptr_t)ptr);
        if (it == g_user_pinned.end() || it->second.ours) return -1;
        g_user_pinned.erase(it);
    }
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}

void pool_free(void *ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto it = g_pool_live.find(ptr);
    if (it == g_pool_live.end()) return;
    g_pool_free[it->second].push_back(ptr);
    g_pool_live.erase(it);
}

// ---------------------------------------------------------------------------
// per-thread context
// ---------------------------------------------------------------------------
ThreadCtx &tctx() {
    static thread_local ThreadCtx ctx;
    return ctx;
}

namespace {
// Streams of threads that have ended.  They are handed to the next thread instead of being destroyed:
// the `ready` event of a cloud that outlives its producing thread still refers to the stream it was
// recorded on, and so does that event when it comes back through the event cache.
struct RetiredStream { int device; hipStream_t stream; };
std::mutex g_retired_mutex;
std::vector<RetiredStream> *g_retired = new std::vector<RetiredStream>();   // never destroyed: threads may end after the statics

hipStream_t retired_stream_take(int device) {
    std::lock_guard<std::mutex> lock(g_retired_mutex);
    for (size_t i = 0; i < g_retired->size(); i++) {
        if ((*g_retired)[i].device == device) {
            hipStream_t s = (*g_retired)[i].stream;
            g_retired->erase(g_retired->begin() + (long)i);
            return s;
        }
    }
    return nullptr;
}

void retire_stream(int device, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_retired_mutex);
    g_retired->push_back(RetiredStream{device, s});
}
}  // namespace

bool ThreadCtx::ensure() {
    int dev = current_device();
    CW_HIP_TRY(hipSetDevice(dev));
    if (stream && device == dev) return true;
    if (stream) {
        (void)hipStreamSynchronize(stream);
        retire_stream(device, stream);
        stream = nullptr;
        if (stream_alt) {
            (void)hipStreamSynchronize(stream_alt);
            retire_stream(device, stream_alt);
            stream_alt = nullptr;
        }
        for (hipStream_t &s : stream_extra) {
            if (!s) continue;
            (void)hipStreamSynchronize(s);
            retire_stream(device, s);
            s = nullptr;
        }
        if (dev_words) { (void)hipFree(dev_words); dev_words = nullptr; }
        if (tickets) { (void)hipFree(tickets); tickets = nullptr; }
    }
    device = dev;
    stream = retired_stream_take(dev);
    if (!stream) CW_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    stream_alt = retired_stream_take(dev);
    if (!stream_alt) CW_HIP_TRY(hipStreamCreateWithFlags(&stream_alt, hipStreamNonBlocking));
    if (!host_words) CW_HIP_TRY(hipHostMalloc((void **)&host_words, 64 * sizeof(uint32_t), hipHostMallocDefault));
    CW_HIP_TRY(hipMalloc(&dev_words, 64 * sizeof(uint32_t)));
    CW_HIP_TRY(hipMalloc((void **)&tickets, 16 * sizeof(uint32_t)));
    CW_HIP_TRY(hipMemset(tickets, 0, 16 * sizeof(uint32_t)));   // (once per thread; done before any of the thread's streams uses them)
    return true;
}

hipStream_t ThreadCtx::extra_stream(int i) {
    if (i < 0 || i >= EXTRA_STREAMS) return nullptr;
    if (!stream_extra[i]) {
        stream_extra[i] = retired_stream_take(device);
        if (!stream_extra[i] && hipStreamCreateWithFlags(&stream_extra[i], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            stream_extra[i] = nullptr;
        }
    }
    return stream_extra[i];
}

void *ThreadCtx::staging(size_t bytes) {
    if (bytes <= pinned_bytes) return pinned;
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_bytes = 0;
    size_t want = size_class(bytes);
    if (hipHostMalloc(&pinned, want, hipHostMallocDefault) != hipSuccess) {
        hip_failed(hipGetLastError(), "hipHostMalloc(staging)", __FILE__, __LINE__);
        pinned = nullptr;
        return nullptr;
    }
    pinned_bytes = want;
    return pinned;
}

void *ThreadCtx::device_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return scratch;
    if (scratch) {
        (void)hipStreamSynchronize(stream);   // kernels of earlier calls may still use the old block
        (void)hipFree(scratch);
    }
    scratch = nullptr;
    scratch_bytes = 0;
    const size_t want = size_class(bytes < 65536 ? 65536 : bytes);
    if (hipMalloc(&scratch, want) != hipSuccess) {
        hip_failed(hipGetLastError(), "hipMalloc(scratch)", __FILE__, __LINE__);
        scratch = nullptr;
        return nullptr;
    }
    scratch_bytes = want;
    return scratch;
}

bool ThreadCtx::sync() {
    hipError_t e = hipStreamSynchronize(stream);
    if (stream_alt) {
        const hipError_t e2 = hipStreamSynchronize(stream_alt);
        if (e == hipSuccess) e = e2;
    }
    for (hipStream_t s : stream_extra) {
        if (!s) continue;
        const hipError_t e2 = hipStreamSynchronize(s);
        if (e == hipSuccess) e = e2;
    }
    for (void *p : deferred) pool_free(p);   // (their kernels are done, or the stream is beyond help)
    deferred.clear();
    if (e != hipSuccess) return hip_failed(e, "hipStreamSynchronize(stream)", __FILE__, __LINE__);
    if (profiling_enabled()) profile_collect();
    return true;
}

ThreadCtx::~ThreadCtx() {
    // Runs at thread exit; the runtime may already be shutting down, so errors are ignored.
    if (stream) {
        (void)hipStreamSynchronize(stream);
        retire_stream(device, stream);
    }
    if (stream_alt) {
        (void)hipStreamSynchronize(stream_alt);
        retire_stream(device, stream_alt);
    }
    for (hipStream_t s : stream_extra) {
        if (!s) continue;
        (void)hipStreamSynchronize(s);
        retire_stream(device, s);
    }
    if (pinned) (void)hipHostFree(pinned);
    if (host_words) (void)hipHostFree(host_words);
    if (dev_words) (void)hipFree(dev_words);
    if (tickets) (void)hipFree(tickets);
    if (scratch) (void)hipFree(scratch);
    for (void *p : deferred) pool_free(p);
}

// ---------------------------------------------------------------------------
// event cache
// ---------------------------------------------------------------------------
namespace {
std::mutex g_event_mutex;
std::vector<hipEvent_t> g_event_cache;
}  // namespace

hipEvent_t event_get() {
    {
        std::lock_guard<std::mutex> lock(g_event_mutex);
        if (!g_event_cache.empty()) {
            hipEvent_t e = g_event_cache.back();
            g_event_cache.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

void event_put(hipEvent_t e) {
    if (!e) return;
    std::lock_guard<std::mutex> lock(g_event_mutex);
    g_event_cache.push_back(e);
}

void DeviceSoA::mark_pending(hipStream_t producer) {
    ready = event_get();
    // without an event the only safe thing is to finish the work now
    if (!ready || hipEventRecord(ready, producer) != hipSuccess) {
        (void)hipGetLastError();   // not left behind for the next launch check of this thread
        (void)hipStreamSynchronize(producer);
        if (ready) { event_put(ready); ready = nullptr; }
    }
}

void DeviceSoA::note_reader(hipStream_t consumer) const {
    hipEvent_t e = event_get();
    if (!e || hipEventRecord(e, consumer) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(consumer);   // without an event: let the reader finish now
        if (e) event_put(e);
        return;
    }
    std::lock_guard<std::mutex> lock(readers_mutex);
    for (auto &r : readers) {
        if (r.first == consumer) {   // a later event on the same stream covers the earlier one
            event_put(r.second);
            r.second = e;
            return;
        }
    }
    readers.emplace_back(consumer, e);
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
namespace {

std::atomic<int> g_profiling{0};
struct ProfEntry { std::string name; double ms = 0; long launches = 0; };
std::mutex g_prof_mutex;
std::vector<ProfEntry> g_prof;

struct Pending { const char *name; hipEvent_t start, stop; };
thread_local std::vector<Pending> t_pending;
thread_local std::vector<hipEvent_t> t_event_cache;

hipEvent_t get_event() {
    if (!t_event_cache.empty()) {
        hipEvent_t e = t_event_cache.back();
        t_event_cache.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

}  // namespace

bool profiling_enabled() { return g_profiling.load(std::memory_order_relaxed) != 0; }

void profile_begin(const char *name, hipStream_t s) {
    Pending p{name, get_event(), get_event()};
    (void)hipEventRecord(p.start, s);
    t_pending.push_back(p);
}

void profile_end(hipStream_t s) {
    if (t_pending.empty()) return;
    (void)hipEventRecord(t_pending.back().stop, s);
}

void profile_collect() {
    if (t_pending.empty()) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto &p : t_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            ProfEntry *e = nullptr;
            for (auto &x : g_prof) if (x.name == p.name) { e = &x; break; }
            if (!e) { g_prof.push_back(ProfEntry{p.name, 0, 0}); e = &g_prof.back(); }
            e->ms += ms;
            e->launches++;
        } else {
            (void)hipGetLastError();
        }
        t_event_cache.push_back(p.start);
        t_event_cache.push_back(p.stop);
    }
    t_pending.clear();
}

}  // namespace cwipc_amd

// ---------------------------------------------------------------------------
// C extension entry points
// ---------------------------------------------------------------------------
using namespace cwipc_amd;

extern "C" int cwipc_hip_device_count(void) { return device_count(); }

extern "C" int cwipc_hip_set_device(int device) {
    if (device < 0 || device >= device_count()) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_hip_set_device", "no such device " + std::to_string(device));
        return -1;
    }
    g_device.store(device);
    return 0;
}

extern "C" int cwipc_hip_get_device(void) { return current_device(); }

extern "C" const char *cwipc_hip_last_error(void) { return t_last_error.c_str(); }

extern "C" void cwipc_hip_synchronize(void) {
    ThreadCtx &c = tctx();
    if (c.stream) c.sync();
}

extern "C" size_t cwipc_hip_pool_bytes(void) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    return g_pool_bytes;
}

extern "C" void cwipc_hip_pool_trim(void) {
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        for (auto &kv : g_host_free) {
            for (void *p : kv.second) (void)hipHostFree(p);
            kv.second.clear();
        }
        g_host_cached_bytes = 0;
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &kv : g_pool_free) {
        for (void *p : kv.second) {
            (void)hipFree(p);
            g_pool_bytes -= kv.first.second;
        }
        kv.second.clear();
    }
}

extern "C" void cwipc_hip_profile_enable(int on) { g_profiling.store(on ? 1 : 0); }

extern "C" void cwipc_hip_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof.clear();
}

extern "C" int cwipc_hip_profile_count(void) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    return (int)g_prof.size();
}

extern "C" int cwipc_hip_profile_get(int i, const char **name, double *total_ms, long *launches) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (i < 0 || i >= (int)g_prof.size()) return -1;
    if (name) *name = g_prof[i].name.c_str();
    if (total_ms) *total_ms = g_prof[i].ms;
    if (launches) *launches = g_prof[i].launches;
    return 0;
}
