/*
 * cwipc_util/api.h -- drop-in boundary of the MI355X build of libcwipc_util.
 *
 * This header RE-DECLARES the binary interface of the reference's
 * include/cwipc_util/api.h (API version 0x20260129); it is not a copy of it.
 * Every declaration cites the reference line it has to stay ABI-compatible with.
 * What matters for drop-in use:
 *   - struct layouts (cwipc_point 16 B, cwipcdump header 32 B, tileinfo, vector);
 *   - the order of the virtual functions of the abstract C++ classes (sibling
 *     cwipc libraries call them directly, so vtable slot order is ABI);
 *   - the extern "C" symbols the ctypes wrapper binds eagerly
 *     (reference python/cwipc/util.py:387-550).
 *
 * Hot-path entry points implemented with HIP kernels for gfx950:
 *   cwipc_downsample, cwipc_remove_outliers, cwipc_tilefilter, cwipc_tilemap,
 *   cwipc_crop, cwipc_colormap, cwipc_join   (reference api.h:1063-1131)
 * plus the point-buffer copy path cwipc_from_points / cwipc_from_packet /
 * cwipc_pointcloud_copy_uncompressed / cwipc_pointcloud_copy_packet.
 * Device-side extensions (no reference counterpart) live in
 * cwipc_util_amd/hip_ext.h.
 */
#ifndef CWIPC_UTIL_AMD_API_H
#define CWIPC_UTIL_AMD_API_H

#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
#include <string>
#include <set>
#endif

#ifndef _CWIPC_UTIL_EXPORT
#define _CWIPC_UTIL_EXPORT __attribute__((visibility("default")))
#endif

/* reference api.h:33,39 -- accepted API version window */
#define CWIPC_API_VERSION ((uint64_t)0x20260129)
#define CWIPC_API_VERSION_OLD ((uint64_t)0x20260129)

/* reference api.h:43,47,51 */
#define CWIPC_CWIPCDUMP_HEADER "cpcd"
#define CWIPC_CWIPCDUMP_VERSION ((uint32_t)0x20210208)
#define CWIPC_FLAG_BINARY 1

/* reference api.h:59-66 -- 32-byte header of a .cwipcdump file / packet, followed by `size` bytes of cwipc_point */
struct cwipc_cwipcdump_header {
    char hdr[4];
    uint32_t magic;
    uint64_t timestamp;
    float cellsize;
    uint32_t unused;
    size_t size;
};

/* reference api.h:77-81 */
struct cwipc_vector {
    double x, y, z;
};

/* reference api.h:88-96 -- the external point record, 16 bytes, no padding */
struct cwipc_point {
    float x, y, z;
    uint8_t r, g, b;
    uint8_t tile;
};

/* reference api.h:100-110 -- proxy wire header (packet codec: cwipc_hip_proxy_packet / cwipc_hip_from_proxy_packet in cwipc_util_amd/hip_ext.h; the TCP transport is out of scope) */
struct cwipc_point_packetheader {
    uint32_t magic;
    uint32_t dataCount;
    uint64_t timestamp;
    float cellsize;
    uint32_t unused;
};
#define CWIPC_POINT_PACKETHEADER_MAGIC 0x20201016

/* reference api.h:150-155 */
struct cwipc_tileinfo {
    struct cwipc_vector normal;
    char *cameraName;
    uint8_t ncamera;
    uint8_t cameraMask;
};

/* reference api.h:159,162 */
enum cwipc_log_level {
    CWIPC_LOG_LEVEL_NONE = 0,
    CWIPC_LOG_LEVEL_ERROR = 1,
    CWIPC_LOG_LEVEL_WARNING = 2,
    CWIPC_LOG_LEVEL_TRACE = 3,
    CWIPC_LOG_LEVEL_DEBUG = 4
};
typedef void (*cwipc_log_callback_t)(int level, const char *message);

#ifdef __cplusplus

static_assert(sizeof(struct cwipc_cwipcdump_header) == 32, "cwipcdump header must be 32 bytes");
static_assert(sizeof(struct cwipc_point) == 16, "cwipc_point must be 16 bytes");

class cwipc_metadata;

/*
 * The reference returns `cwipc_pcl_pointcloud` BY VALUE from a virtual
 * (reference api.h:169-172 placeholder void*, api_pcl.h:74 a pcl::shared_ptr).
 * PCL-aware sibling libraries therefore pass a hidden return slot and expect a
 * 16-byte shared_ptr image in it.  This build has no PCL: it returns an empty
 * two-pointer object with a non-trivial destructor, which has exactly the
 * calling convention and the layout of an empty libstdc++ shared_ptr, so such
 * callers see a NULL cloud (which the reference code paths already handle,
 * e.g. src/cwipc_filters.cpp:37-40).
 */
#ifndef _CWIPC_PCL_POINTCLOUD_DEFINED
struct cwipc_pcl_pointcloud {
    void *ptr;
    void *ctrl;
    cwipc_pcl_pointcloud() : ptr(nullptr), ctrl(nullptr) {}
    cwipc_pcl_pointcloud(const cwipc_pcl_pointcloud &) : ptr(nullptr), ctrl(nullptr) {}
    ~cwipc_pcl_pointcloud() {}
    bool operator==(decltype(nullptr)) const { return ptr == nullptr; }
};
#define _CWIPC_PCL_POINTCLOUD_DEFINED
#endif

/* reference api.h:184-284 -- slot order: dtor, free, _shallowcopy, timestamp, cellsize, _set_cellsize,
 * _set_timestamp, count, get_uncompressed_size, copy_uncompressed, copy_packet, access_pcl_pointcloud, access_metadata */
class cwipc_pointcloud {
public:
    virtual ~cwipc_pointcloud() {}
    virtual void free() = 0;
    virtual cwipc_pointcloud *_shallowcopy() = 0;
    virtual uint64_t timestamp() = 0;
    virtual float cellsize() = 0;
    virtual void _set_cellsize(float cellsize) = 0;
    virtual void _set_timestamp(uint64_t timestamp) = 0;
    virtual int count() = 0;
    virtual size_t get_uncompressed_size() = 0;
    virtual int copy_uncompressed(struct cwipc_point *pointbuf, size_t size) = 0;
    virtual size_t copy_packet(uint8_t *packet, size_t size) = 0;
    virtual cwipc_pcl_pointcloud access_pcl_pointcloud() = 0;
    virtual cwipc_metadata *access_metadata() = 0;
};

/* reference api.h:291-335 */
class cwipc_source {
public:
    virtual ~cwipc_source() {}
    virtual void free() = 0;
    virtual bool seek(uint64_t timestamp) = 0;
    virtual bool eof() = 0;
    virtual bool available(bool wait) = 0;
    virtual cwipc_pointcloud *get() = 0;
};

/* reference api.h:345-444 -- note the std::set data member (:443) is part of the object layout */
class cwipc_activesource : public cwipc_source {
public:
    virtual ~cwipc_activesource() {}
    virtual bool reload_config(const char *configFile) = 0;
    virtual size_t get_config(char *buffer, size_t size) = 0;
    virtual bool start() = 0;
    virtual void stop() = 0;
    virtual bool seek(uint64_t timestamp) = 0;
    virtual int maxtile() = 0;
    virtual bool get_tileinfo(int tilenum, struct cwipc_tileinfo *tileinfo) = 0;
    virtual void request_metadata(const std::string &name) { metadata_wanted.insert(name); }
    bool is_metadata_requested(const std::string &name) { return metadata_wanted.find(name) != metadata_wanted.end(); }
    virtual bool auxiliary_operation(const std::string op, const void *inbuf, size_t insize, void *outbuf, size_t outsize) {
        (void)op; (void)inbuf; (void)insize; (void)outbuf; (void)outsize;
        return false;
    }

private:
    std::set<std::string> metadata_wanted;
};

/* reference api.h:452-500 */
class cwipc_sink {
public:
    virtual ~cwipc_sink() {}
    virtual void free() = 0;
    virtual bool feed(cwipc_pointcloud *pc, bool clear) = 0;
    virtual bool caption(const char *caption) = 0;
    virtual char interact(const char *prompt, const char *responses, int32_t millis) = 0;
};

/* reference api.h:508-562 */
class cwipc_metadata {
public:
    typedef void (*deallocfunc)(void *);
    virtual ~cwipc_metadata() {}
    virtual int count() = 0;
    virtual const std::string &name(int idx) = 0;
    virtual const std::string &description(int idx) = 0;
    virtual void *pointer(int idx) = 0;
    virtual size_t size(int idx) = 0;
    virtual void _add(const std::string &name, const std::string &description, void *pointer, size_t size, deallocfunc dealloc) = 0;
    virtual void _move(cwipc_metadata *other) = 0;
};

#else /* C view: opaque handles (reference api.h:566-588) */

typedef struct _cwipc_pointcloud { int _dummy; } cwipc_pointcloud;
typedef struct _cwipc_source { int _dummy; } cwipc_source;
typedef struct cwipc_activesource { struct _cwipc_source source; } cwipc_activesource;
typedef struct _cwipc_sink { int _dummy; } cwipc_sink;
typedef struct _cwipc_metadata { int _dummy; } cwipc_metadata;

#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / logging (reference api.h:598-620, src/logging.cpp) ---- */
_CWIPC_UTIL_EXPORT const char *cwipc_get_version();
_CWIPC_UTIL_EXPORT void cwipc_log_configure(int level, cwipc_log_callback_t callback);
_CWIPC_UTIL_EXPORT void _cwipc_log_emit(int level, const char *module, const char *message);
_CWIPC_UTIL_EXPORT int cwipc_dangling_allocations(bool log);

/* ---- constructors (reference api.h:632-709) ---- */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_read(const char *filename, uint64_t timestamp, char **errorMessage, uint64_t apiVersion);               /* PLY (x, y, z, rgba), ASCII or binary: csrc/ply.cpp */
_CWIPC_UTIL_EXPORT int cwipc_write(const char *filename, cwipc_pointcloud *pc, char **errorMessage);                                              /* ASCII PLY in the layout PCL writes */
_CWIPC_UTIL_EXPORT int cwipc_write_ext(const char *filename, cwipc_pointcloud *pc, int flag, char **errorMessage);                                /* flag & CWIPC_FLAG_BINARY: binary PLY */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_from_points(struct cwipc_point *points, size_t size, int npoint, uint64_t timestamp, char **errorMessage, uint64_t apiVersion);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_from_packet(uint8_t *packet, size_t size, char **errorMessage, uint64_t apiVersion);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_read_debugdump(const char *filename, char **errorMessage, uint64_t apiVersion);
_CWIPC_UTIL_EXPORT int cwipc_write_debugdump(const char *filename, cwipc_pointcloud *pc, char **errorMessage);

/* ---- cwipc_pointcloud accessors (reference api.h:723-800, src/cwipc_util.cpp:731-773) ---- */
_CWIPC_UTIL_EXPORT void cwipc_pointcloud_free(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_pointcloud__shallowcopy(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT uint64_t cwipc_pointcloud_timestamp(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT float cwipc_pointcloud_cellsize(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT void cwipc_pointcloud__set_cellsize(cwipc_pointcloud *pc, float cellsize);
_CWIPC_UTIL_EXPORT void cwipc_pointcloud__set_timestamp(cwipc_pointcloud *pc, uint64_t timestamp);
_CWIPC_UTIL_EXPORT int cwipc_pointcloud_count(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT size_t cwipc_pointcloud_get_uncompressed_size(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT int cwipc_pointcloud_copy_uncompressed(cwipc_pointcloud *pc, struct cwipc_point *pointbuf, size_t size);
_CWIPC_UTIL_EXPORT size_t cwipc_pointcloud_copy_packet(cwipc_pointcloud *pc, uint8_t *packet, size_t size);
_CWIPC_UTIL_EXPORT cwipc_metadata *cwipc_pointcloud_access_metadata(cwipc_pointcloud *pc);

/* ---- sources / sinks (reference api.h:807-964, src/cwipc_util.cpp:799-870) ---- */
_CWIPC_UTIL_EXPORT bool cwipc_activesource_start(cwipc_activesource *src);
_CWIPC_UTIL_EXPORT void cwipc_activesource_stop(cwipc_activesource *src);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_source_get(cwipc_source *src);
_CWIPC_UTIL_EXPORT void cwipc_source_free(cwipc_source *src);
_CWIPC_UTIL_EXPORT bool cwipc_source_eof(cwipc_source *src);
_CWIPC_UTIL_EXPORT bool cwipc_source_available(cwipc_source *src, bool wait);
_CWIPC_UTIL_EXPORT void cwipc_activesource_request_metadata(cwipc_activesource *src, const char *name);
_CWIPC_UTIL_EXPORT bool cwipc_activesource_is_metadata_requested(cwipc_activesource *src, const char *name);
_CWIPC_UTIL_EXPORT bool cwipc_activesource_reload_config(cwipc_activesource *src, const char *configFile);
_CWIPC_UTIL_EXPORT size_t cwipc_activesource_get_config(cwipc_activesource *src, char *buffer, size_t size);
_CWIPC_UTIL_EXPORT bool cwipc_activesource_seek(cwipc_activesource *src, uint64_t timestamp);
_CWIPC_UTIL_EXPORT int cwipc_activesource_maxtile(cwipc_activesource *src);
_CWIPC_UTIL_EXPORT bool cwipc_activesource_get_tileinfo(cwipc_activesource *src, int tilenum, struct cwipc_tileinfo *tileinfo);
_CWIPC_UTIL_EXPORT bool cwipc_activesource_auxiliary_operation(cwipc_activesource *src, const char *op, const void *inbuf, size_t insize, void *outbuf, size_t outsize);
_CWIPC_UTIL_EXPORT void cwipc_sink_free(cwipc_sink *sink);
_CWIPC_UTIL_EXPORT bool cwipc_sink_feed(cwipc_sink *sink, cwipc_pointcloud *pc, bool clear);
_CWIPC_UTIL_EXPORT bool cwipc_sink_caption(cwipc_sink *sink, const char *caption);
_CWIPC_UTIL_EXPORT char cwipc_sink_interact(cwipc_sink *sink, const char *prompt, const char *responses, int32_t millis);

/* ---- metadata (reference api.h:970-1008) ---- */
_CWIPC_UTIL_EXPORT void cwipc_metadata__move(cwipc_metadata *src, cwipc_metadata *dest);
_CWIPC_UTIL_EXPORT int cwipc_metadata_count(cwipc_metadata *collection);
_CWIPC_UTIL_EXPORT const char *cwipc_metadata_name(cwipc_metadata *collection, int idx);
_CWIPC_UTIL_EXPORT const char *cwipc_metadata_description(cwipc_metadata *collection, int idx);
_CWIPC_UTIL_EXPORT void *cwipc_metadata_pointer(cwipc_metadata *collection, int idx);
_CWIPC_UTIL_EXPORT size_t cwipc_metadata_size(cwipc_metadata *collection, int idx);

/* ---- generators (reference api.h:1020-1050, 1143) ---- */
_CWIPC_UTIL_EXPORT cwipc_activesource *cwipc_synthetic(int fps, int npoints, char **errorMessage, uint64_t apiVersion);
/* cwipc_capturer dispatches on the "type" of the camera configuration ("auto", a .json file name or a JSON literal)
 * to the factory a camera plugin registered (reference src/cwipc_capturer.cpp:31-150).  No camera plugin is part of
 * this library: without a registered plugin the call fails with a message. */
_CWIPC_UTIL_EXPORT cwipc_activesource *cwipc_capturer(const char *configFilename, char **errorMessage, uint64_t apiVersion);
/* Registration hook of the camera plugins (reference include/cwipc_util/internal/capturers.hpp:508-516, src/cwipc_capturer.cpp:152-160). */
typedef int _cwipc_functype_count_devices(void);
typedef cwipc_activesource *_cwipc_func_capturer_factory(const char *configFilename, char **errorMessage, uint64_t apiVersion);
_CWIPC_UTIL_EXPORT int _cwipc_register_capturer(const char *name, _cwipc_functype_count_devices *countFunc, _cwipc_func_capturer_factory *factoryFunc);
_CWIPC_UTIL_EXPORT cwipc_sink *cwipc_window(const char *title, char **errorMessage, uint64_t apiVersion);                    /* GUI: out of scope */
_CWIPC_UTIL_EXPORT cwipc_activesource *cwipc_proxy(const char *host, int port, char **errorMessage, uint64_t apiVersion);    /* TCP transport: out of scope */

/* ---- THE HOT PATH: per-point filters (reference api.h:1063-1131, src/cwipc_filters.cpp) ----
 * Every filter returns a NEW cloud owned by the caller, never consumes its input,
 * returns NULL for NULL input or internal failure (after cwipc_log(ERROR)). */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_downsample(cwipc_pointcloud *pc, float voxelsize);                                        /* src/cwipc_filters.cpp:30-172 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_remove_outliers(cwipc_pointcloud *pc, int kNeighbors, float stddevMulThresh, bool perTile); /* :181-278 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_tilefilter(cwipc_pointcloud *pc, int tile);                                               /* :281-306 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_tilemap(cwipc_pointcloud *pc, uint8_t map[256]);                                          /* :308-331 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_crop(cwipc_pointcloud *pc, float bbox[6]);                                                /* :333-360 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_colormap(cwipc_pointcloud *pc, uint32_t clearBits, uint32_t setBits);                     /* :362-386 */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_join(cwipc_pointcloud *pc1, cwipc_pointcloud *pc2);                                       /* :388-418 */

#ifdef __cplusplus
}
#endif
#endif /* CWIPC_UTIL_AMD_API_H */
