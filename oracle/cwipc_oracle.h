/*
 * cwipc_oracle.h -- CPU restatement of the cwipc_util per-point filter path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS
 *   - tilefilter / tilemap / crop / colormap / join / colorize / container /
 *     synthetic: restated from the reference sources cited per function and
 *     pinned by the reference's own known-answer tests (tests/test_oracle_kat.py).
 *   - voxel grid, octree split, statistical outlier removal: the arithmetic
 *     lives in PCL, which is NOT in /root/reference and is not pinned there
 *     (src/CMakeLists.txt:52 links ${PCL_LIBRARIES}).  Those parts restate the
 *     published upstream PCL algorithms (pcl::VoxelGrid, pcl::octree::
 *     OctreePointCloud, pcl::StatisticalOutlierRemoval, pcl::CentroidPoint).
 *     The reference's tests hold only count invariants for them
 *     (python/test_cwipc_util.py:528-594), so value parity for downsample and
 *     remove_outliers is "PARITY UNPINNED".
 */
#ifndef CWIPC_ORACLE_H
#define CWIPC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* External point record: include/cwipc_util/api.h:88-96 (16 bytes, no padding). */
typedef struct {
    float x, y, z;
    uint8_t r, g, b, tile;
} oracle_point;

/* ---- synthetic source: src/cwipc_synthetic.cpp:32-49, 131, 182-222 ---- */
int   oracle_synthetic_count(int npoints);
float oracle_synthetic_cellsize(int npoints);
void  oracle_synthetic(int npoints, float angle, oracle_point *out);

/* ---- container helpers: src/cwipc_util.cpp:173-204 ---- */
float oracle_guess_cellsize(const oracle_point *pts, size_t n);

/* ---- exact per-point filters: src/cwipc_filters.cpp:281-418 ---- */
size_t oracle_tilefilter(const oracle_point *in, size_t n, int tile, oracle_point *out);
void   oracle_tilemap(const oracle_point *in, size_t n, const uint8_t map[256], oracle_point *out);
size_t oracle_crop(const oracle_point *in, size_t n, const float bbox[6], oracle_point *out);
void   oracle_colormap(const oracle_point *in, size_t n, uint32_t clearBits, uint32_t setBits, oracle_point *out);
size_t oracle_join(const oracle_point *a, size_t na, const oracle_point *b, size_t nb, oracle_point *out);

/* ---- colorize: python/cwipc/filters/colorize.py:100-119 ----
 * lut[t*3+c] = colour component c of tile t (as Python float), valid[t] != 0
 * where the colour map has an entry for t (None otherwise). */
void oracle_colorize(const oracle_point *in, size_t n, double weight,
                     const double *lut, const uint8_t *valid, oracle_point *out);

/* ---- voxel grid (negative cellsize): src/cwipc_filters.cpp:30-87 + pcl::VoxelGrid ----
 * Returns the number of output points written to out (capacity cap), or
 *   -1  empty result / NULL input cloud  (reference logs ERROR, returns NULL)
 *   -2  grid index overflow (VoxelGrid copies input; getCentroidIndex then throws; NULL)
 *   -3  out capacity too small
 * *out_cellsize receives max(cellsize, pc_cellsize). */
long oracle_downsample_voxelgrid(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                                 oracle_point *out, size_t cap, float *out_cellsize);

/* ---- octree-split voxel grid (positive cellsize): src/cwipc_filters.cpp:89-172 ----
 * Same return convention; empty input yields 0 (empty cloud, not an error).
 * If n_leaves / depth are non-NULL they receive octree statistics. */
long oracle_downsample(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                       oracle_point *out, size_t cap, float *out_cellsize,
                       int *n_leaves, int *depth);

/* Test aid, not part of the reference: oracle_downsample plus, per output, the mean of its contributors in double
 * (mean64: cap x 3) and their number (count: cap). */
long oracle_downsample_audit(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                             oracle_point *out, size_t cap, float *out_cellsize, double *mean64, uint32_t *count);

/* ---- statistical outlier removal: src/cwipc_filters.cpp:181-278 + pcl::StatisticalOutlierRemoval ----
 * Returns number of kept points (written to out in order), or -1 on error.
 * mean_dist (optional, n floats; perTile=0 only) receives d_i; thr (optional) the threshold. */
long oracle_remove_outliers(const oracle_point *in, size_t n, int k, float stddev_mul, int per_tile,
                            oracle_point *out, float *mean_dist, double *thr);

/* Mean distance to the k nearest neighbours (excluding self) for every point,
 * exact, as pcl::StatisticalOutlierRemoval computes it.  Used by tests directly. */
int oracle_knn_mean_dist(const oracle_point *in, size_t n, int k, float *mean_dist);

#ifdef __cplusplus
}
#endif
#endif
