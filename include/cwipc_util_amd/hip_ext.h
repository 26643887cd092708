/*
 * cwipc_util_amd/hip_ext.h -- device-side extensions of the MI355X libcwipc_util.
 *
 * None of these symbols exist in the reference; they expose what a GPU-resident
 * pipeline needs on top of the drop-in C-ABI of cwipc_util/api.h: device
 * selection, explicit residency control, zero-copy access to the SoA planes
 * (so torch.distributed / RCCL can move them), two filters whose reference
 * implementation is Python-side, and per-kernel timing for bench.py.
 * Plain C: pointers and sizes only, no torch types.
 */
#ifndef CWIPC_UTIL_AMD_HIP_EXT_H
#define CWIPC_UTIL_AMD_HIP_EXT_H

#include "cwipc_util/api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device management ---- */
_CWIPC_UTIL_EXPORT int cwipc_hip_device_count(void);          /* 0 when no GPU is visible */
_CWIPC_UTIL_EXPORT int cwipc_hip_set_device(int device);      /* process-wide device for all later calls; 0 ok, -1 error */
_CWIPC_UTIL_EXPORT int cwipc_hip_get_device(void);
_CWIPC_UTIL_EXPORT const char *cwipc_hip_last_error(void);    /* thread-local text of the last HIP failure ("" if none) */
_CWIPC_UTIL_EXPORT void cwipc_hip_synchronize(void);          /* wait for the calling thread's stream */
_CWIPC_UTIL_EXPORT size_t cwipc_hip_pool_bytes(void);         /* bytes currently held by the device memory pool */
_CWIPC_UTIL_EXPORT size_t cwipc_hip_workspace_bytes(void);    /* device bytes held by the voxel filter's workspaces (leaf grids; two per thread that downsamples) */
_CWIPC_UTIL_EXPORT void cwipc_hip_pool_trim(void);            /* return cached device memory to the driver */
_CWIPC_UTIL_EXPORT size_t cwipc_hip_workspace_trim(void);     /* give back the voxel workspaces that ended threads left for the next ones (at most 8); returns how many */

/* ---- page-locked host buffers of the caller (round 4) ----
 * The copy path of the reference (src/cwipc_util.cpp:329-354 from_points, :226-250 copy_uncompressed) moves bytes between the
 * caller's buffer and a buffer the cloud owns.  Here the owned copy lives in HBM, and a buffer the DMA engines cannot reach costs
 * one more copy on the host on the way.  A caller that keeps its point buffers in page-locked memory -- allocated here, or its own
 * memory registered once (a numpy array that is reused frame after frame) -- skips it: cwipc_from_points / cwipc_from_packet read
 * such a buffer straight from the device (and have finished reading when they return, as the reference's copy has),
 * cwipc_pointcloud_copy_uncompressed / copy_packet into one are written by the DMA engine directly.  Nothing else changes. */
_CWIPC_UTIL_EXPORT void *cwipc_hip_host_alloc(size_t bytes);                 /* NULL without a GPU or memory */
_CWIPC_UTIL_EXPORT void cwipc_hip_host_free(void *ptr);
_CWIPC_UTIL_EXPORT int cwipc_hip_host_register(void *ptr, size_t bytes);     /* 0 ok; the memory stays the caller's */
_CWIPC_UTIL_EXPORT int cwipc_hip_host_unregister(void *ptr);

/* ---- residency ----
 * A cloud made by cwipc_from_points lives in host memory until a filter needs
 * it; filter results live in HBM (SoA planes x,y,z:f32[n], rgbt:u32[n] with
 * r | g<<8 | b<<16 | tile<<24) until a host accessor needs them. */
_CWIPC_UTIL_EXPORT int cwipc_hip_upload(cwipc_pointcloud *pc);            /* make the SoA copy now; 0 ok */
_CWIPC_UTIL_EXPORT int cwipc_hip_drop_host_copy(cwipc_pointcloud *pc);    /* free the host AoS copy (device copy must exist) */
_CWIPC_UTIL_EXPORT int cwipc_hip_is_device_resident(cwipc_pointcloud *pc);
_CWIPC_UTIL_EXPORT int cwipc_hip_device_planes(cwipc_pointcloud *pc, const float **x, const float **y, const float **z,
                                               const uint32_t **rgbt, size_t *npoint);
/* Interleave the planes into a DEVICE buffer of npoint*16 bytes (cwipc_point records). Returns npoint or -1. */
_CWIPC_UTIL_EXPORT long cwipc_hip_copy_device_aos(cwipc_pointcloud *pc, void *dev_points, size_t size);
/* New cloud from cwipc_point records that already are in device memory. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_from_device_aos(const void *dev_points, size_t npoint, uint64_t timestamp, float cellsize);
/* The same from the receive buffer of an all-gather (the multi-GPU join, reference cwipc_join folded over the tiles,
   src/cwipc_filters.cpp:388-418): nslots (<= 64) slots of slot_rows 16-byte rows each in device memory, the records of
   slot s in rows [header_rows, header_rows + counts[s]); the new cloud holds them in slot order.  counts is a host array. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_from_device_slots(const void *dev_slots, int nslots, size_t slot_rows, size_t header_rows,
                                                                 const uint32_t *counts, uint64_t timestamp, float cellsize);
/* Both as steps of the CALLER's stream (hipStream_t; e.g. the stream its collectives are ordered on): no wait inside, the
   kernel runs behind what the stream holds and in front of what the caller enqueues next; the new cloud carries an event. */
_CWIPC_UTIL_EXPORT long cwipc_hip_copy_device_aos_on_stream(cwipc_pointcloud *pc, void *dev_points, size_t size, void *stream);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_from_device_slots_on_stream(const void *dev_slots, int nslots, size_t slot_rows, size_t header_rows,
                                                                           const uint32_t *counts, uint64_t timestamp, float cellsize, void *stream);

/* ---- the multi-GPU join: one process per GPU, RCCL inside the library, one call per frame ----
 * Replaces the reference's in-process fold of cwipc_join over a frame's tiles (python/cwipc/net/source_synchronizer.py:175-188,
 * src/cwipc_filters.cpp:388-418) when the tiles live on different GPUs.  Bootstrap as with any RCCL communicator: one rank
 * asks for an id, the application hands its bytes to the other ranks (MPI, a socket, torch.distributed's store ...), every
 * rank creates its end on the device chosen with cwipc_hip_set_device.  With nranks > 1 the call REFUSES to start unless
 * HSA_ENABLE_IPC_MODE_LEGACY=0 is in the environment (this image's driver: RCCL's buffer hand-over between processes fails
 * without it, inside the first collective). */
#define CWIPC_HIP_COMM_ID_BYTES 128
typedef struct cwipc_hip_comm cwipc_hip_comm;
_CWIPC_UTIL_EXPORT int cwipc_hip_comm_unique_id(void *id /* CWIPC_HIP_COMM_ID_BYTES */, char **errorMessage);              /* 0 ok */
_CWIPC_UTIL_EXPORT cwipc_hip_comm *cwipc_hip_comm_create(const void *id, int rank, int nranks, char **errorMessage);       /* collective */
_CWIPC_UTIL_EXPORT void cwipc_hip_comm_free(cwipc_hip_comm *comm);
_CWIPC_UTIL_EXPORT int cwipc_hip_comm_rank(cwipc_hip_comm *comm);
_CWIPC_UTIL_EXPORT int cwipc_hip_comm_nranks(cwipc_hip_comm *comm);
/* Every rank calls this once per frame with its cloud (NULL: no tile this frame) and gets the fused cloud: the ranks' points in
 * rank order, timestamp and cellsize the minimum over the clouds that took part (src/cwipc_filters.cpp:411-414).  One
 * ncclAllGather of 32 bytes per rank, then one group of ncclSend/ncclRecv that moves the planes straight into the result; the
 * call returns when the group is enqueued (the result carries an event).  NULL on error (logged).  What a single rank finds
 * out on its own (no usable device, no memory for the fused cloud) travels in its record, so the others leave it out instead
 * of waiting for it; a rank whose tile is the whole frame gets its own planes back and still sends them to the others. */
#define CWIPC_HIP_JOIN_LOOPBACK 1   /* this rank's own part travels through RCCL too (send/recv to itself): exercises the exchange on one GPU */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_comm_join(cwipc_hip_comm *comm, cwipc_pointcloud *pc, int flags);
/* The same for a stream of frames: returns at once with a cloud that stands for the fused cloud of this frame; the exchange
 * itself (waiting for this rank's filter results, the record all-gather, the send/recv group) is done by a thread of the
 * communicator, frame after frame in the order of the calls, and the cloud settles when it is first used (count, a filter, a
 * copy; its timestamp and cellsize too: they are the minimum over the ranks).  Every rank submits every frame, in the same
 * order.  The argument may be freed as soon as the call returns.  A failed exchange shows as an empty cloud plus the logged
 * error.  cwipc_hip_comm_free waits for the frames still queued. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_comm_submit(cwipc_hip_comm *comm, cwipc_pointcloud *pc, int flags);

/* Test hook, pure host code (works without a GPU): the plan rank `rank` of `nranks` follows for a frame whose gathered records are
 * `metas` (nranks x 8 uint32: count, has_cloud, cellsize bits, status [0 ok, 1 absent, 2 sends but cannot receive], ts_lo, ts_hi,
 * capacity, 0) -- the very function cwipc_hip_comm_join issues its ncclSend / ncclRecv from (csrc/exchange_plan.hpp).
 * summary[8] = total points, flags (1 too big, 2 no result on this rank, 4 result = this rank's input, 8 own part by copy kernel,
 * 16 some tile arrived, 32 the ranks meet a second time before payload moves, 64 this rank needs a result buffer), ts_min,
 * cellsize bits, this rank's displacement, number of sends, number of receives, 0; sends / recvs: cap x {peer, points, offset}.
 * Returns 0, -1 bad arguments, -2 cap too small. */
_CWIPC_UTIL_EXPORT int cwipc_hip_exchange_plan(int rank, int nranks, const uint32_t *metas, int loopback, uint64_t *summary,
                                               uint64_t *sends, uint64_t *recvs, int cap);

/* ---- the proxy's wire format as a packet codec (reference src/cwipc_proxy.cpp:179-216, include/cwipc_util/api.h:100-110) ----
 * A packet = the 24-byte cwipc_point_packetheader + dataCount bytes of cwipc_point records; the receiver's answer is the 8 bytes of
 * the timestamp.  No sockets here: the application moves the bytes.
 * cwipc_hip_proxy_packet: writes the packet of `pc` (magic 0 = CWIPC_POINT_PACKETHEADER_MAGIC); packet NULL: returns the size it
 * takes; returns the bytes written, 0 on error.  cwipc_hip_from_proxy_packet: the cloud of a packet (timestamp and cellsize from
 * the header); the C magic is always accepted, the reference's Python sender's (0x20210208, python/cwipc/util.py:346 -- the two
 * disagree upstream) only when accept_python_magic is set. */
_CWIPC_UTIL_EXPORT size_t cwipc_hip_proxy_packet(cwipc_pointcloud *pc, uint8_t *packet, size_t size, uint32_t magic);
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_from_proxy_packet(const uint8_t *packet, size_t size, int accept_python_magic, char **errorMessage, uint64_t apiVersion);

/* ---- filters whose reference implementation is Python-side ---- */
/* ColorizeFilter._mapcolor (reference python/cwipc/filters/colorize.py:100-119): lut = 256x3 doubles, valid = 256 flags. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_colorize(cwipc_pointcloud *pc, double weight, const double *lut, const uint8_t *valid);
/* SimulatecamsFilter, hard assignment (reference python/cwipc/filters/simulatecams.py:44-70): tile = 1 << c for the camera direction
 * (camera_dirs: cos, sin of 2 pi c / ncamera, as doubles) with the largest dot product with the point's position minus the
 * centroid, y ignored.  The centroid (numpy.mean of the float32 coordinates) is computed by the caller, as the reference does. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_simulatecams(cwipc_pointcloud *pc, int ncamera, float centroid_x, float centroid_z, const double *camera_dirs);
/* cwipc_join_multi (reference python/cwipc/util.py:1330-1332): same result as the left fold of cwipc_join, one pass. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_join_multi(cwipc_pointcloud **pcs, int npc);
/* p' = R p + t for a row-major 4x4 matrix (last row ignored), in f64, stored as float: what the reference's
 * cwipc_transform does through numpy (python/cwipc/registration/util.py:295-309). */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_transform(cwipc_pointcloud *pc, const double *matrix4x4);
/* p' = (p + (x, y, z)) * scale in f64, cellsize * scale: the reference's TransformFilter loop (python/cwipc/filters/transform.py:38-52). */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_offset_scale(cwipc_pointcloud *pc, double x, double y, double z, double scale);
/* used256[t] = 1 for every tile value t that occurs (python/cwipc/registration/util.py:285-293, get_tiles_used); returns how many, -1 on error. */
_CWIPC_UTIL_EXPORT int cwipc_hip_tiles_used(cwipc_pointcloud *pc, uint8_t *used256);
/* cwipc_tilefilter_masked (reference python/cwipc/registration/util.py:98-112): keep points with (tile & mask) != 0. */
_CWIPC_UTIL_EXPORT cwipc_pointcloud *cwipc_hip_tilefilter_masked(cwipc_pointcloud *pc, int mask);

/* ---- intermediate results for parity tests ---- */
/* Mean k-NN distance d_i of every point (the quantity pcl::StatisticalOutlierRemoval thresholds) into host memory; 0 ok. */
_CWIPC_UTIL_EXPORT int cwipc_hip_knn_mean_dist(cwipc_pointcloud *pc, int kNeighbors, float *mean_dist, size_t cap, double *threshold, float stddevMulThresh);

/* ---- per-kernel device timing (hipEvents on the calling thread's stream) ---- */
_CWIPC_UTIL_EXPORT void cwipc_hip_profile_enable(int on);
_CWIPC_UTIL_EXPORT void cwipc_hip_profile_reset(void);
/* Number of distinct kernels seen; name/total milliseconds/launch count of entry i. */
_CWIPC_UTIL_EXPORT int cwipc_hip_profile_count(void);
_CWIPC_UTIL_EXPORT int cwipc_hip_profile_get(int i, const char **name, double *total_ms, long *launches);

#ifdef __cplusplus
}
#endif
#endif
