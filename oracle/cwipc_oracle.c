/*
 * cwipc_oracle.c -- CPU restatement of the cwipc_util per-point filter path.
 *
 * TEST INFRASTRUCTURE ONLY (see cwipc_oracle.h).  Plain C99, no dependencies
 * beyond libm.  Build with -ffp-contract=off so that every fp32 operation is
 * rounded separately, as a baseline x86-64 build of the reference would do.
 *
 * Every function cites the reference file:line it follows.  Where the reference
 * delegates to PCL (absent from /root/reference, version not pinned) the
 * published upstream algorithm is restated and marked [PCL upstream].
 */
#include "cwipc_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Synthetic source                                                           */
/* ------------------------------------------------------------------------- */

/* src/cwipc_synthetic.cpp:41-47 -- hsteps = asteps = int(sqrt(npoints)); 0 means 160000. */
static int synthetic_steps(int npoints) {
    if (npoints == 0) npoints = 160000;
    return (int)sqrt((double)npoints);
}

int oracle_synthetic_count(int npoints) {
    int s = synthetic_steps(npoints);
    return s * s;
}

/* src/cwipc_synthetic.cpp:131 -- rv->_set_cellsize(2.0 / m_hsteps) (double expression, float parameter). */
float oracle_synthetic_cellsize(int npoints) {
    return (float)(2.0 / synthetic_steps(npoints));
}

/* src/cwipc_synthetic.cpp:182-222 (generate_points).  `angle` replaces the
 * wall-clock m_angle (:120,126) so that colours are reproducible.  libm calls
 * are evaluated in double as in C; this is the INPUT generator, shared by the
 * oracle and the HIP path, so its last-bit behaviour is not a parity question. */
void oracle_synthetic(int npoints, float m_angle, oracle_point *out) {
    const int hsteps = synthetic_steps(npoints);
    const int asteps = hsteps;
    const float pi = 3.14159265358979f;
    const float max_height = 2.0;
    const float delta_h = max_height / hsteps;
    const float delta_a = 2 * pi / asteps;
    oracle_point *p = out;

    for (int hi = 0; hi < hsteps; hi++) {
        float height = hi * delta_h;
        for (int ai = 0; ai < asteps; ai++) {
            float angle = ai * delta_a;
            float radius = 0.3 * pow(cos(height * pi / 3 - pi / 6), 0.71);
            float x = radius * sin(angle);
            float y = radius * cos(angle);
            float r = (1 + sin(2 * pi * height + m_angle + angle)) / 2;
            float g = (1 + sin(3 * pi * height + m_angle + angle)) / 2;
            float b = (1 + sin(4 * pi * height + m_angle + angle)) / 2;
            int rr = (int)(r * 255.0);
            int gg = (int)(g * 255.0);
            int bb = (int)(b * 255.0);
            /* "Eyes", :206-210 */
            if (height > 1.7 && height < 1.8 &&
                ((angle > pi * 0.083 && angle < pi * 0.1667) || (angle > pi * 1.833 && angle < pi * 1.917))) {
                if (fmod(m_angle, pi / 2) > 0.08) {
                    rr = gg = bb = 255;
                }
            }
            p->x = -x;
            p->y = height;
            p->z = y;
            p->r = (uint8_t)rr;
            p->g = (uint8_t)gg;
            p->b = (uint8_t)bb;
            p->tile = y < 0 ? 1 : 2;
            p++;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Container helper                                                           */
/* ------------------------------------------------------------------------- */

/* src/cwipc_util.cpp:173-204 -- _set_cellsize(negative): minimum distance between
 * every point and the FIRST point (prevPoint is never advanced), 0 if none.
 * pcl::geometry::distance = (a - b).norm() in fp32 [PCL upstream]. */
float oracle_guess_cellsize(const oracle_point *pts, size_t n) {
    float minDistance = INFINITY;
    for (size_t i = 1; i < n; i++) {
        float dx = pts[i].x - pts[0].x;
        float dy = pts[i].y - pts[0].y;
        float dz = pts[i].z - pts[0].z;
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        float d = sqrtf(d2);
        if (d < minDistance) minDistance = d;
    }
    if (minDistance == INFINITY) minDistance = 0;
    return minDistance;
}

/* ------------------------------------------------------------------------- */
/* Exact per-point filters                                                    */
/* ------------------------------------------------------------------------- */

/* src/cwipc_filters.cpp:295-299 -- keep iff tile == 0 || tile == pt.a (u8 promoted to int). */
size_t oracle_tilefilter(const oracle_point *in, size_t n, int tile, oracle_point *out) {
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        if (tile == 0 || tile == (int)in[i].tile) out[m++] = in[i];
    }
    return m;
}

/* src/cwipc_filters.cpp:322-325 */
void oracle_tilemap(const oracle_point *in, size_t n, const uint8_t map[256], oracle_point *out) {
    for (size_t i = 0; i < n; i++) {
        out[i] = in[i];
        out[i].tile = map[in[i].tile];
    }
}

/* src/cwipc_filters.cpp:347-354 -- half-open box, fp32 compares. */
size_t oracle_crop(const oracle_point *in, size_t n, const float bbox[6], oracle_point *out) {
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const oracle_point *pt = &in[i];
        if (bbox[0] <= pt->x && pt->x < bbox[1] &&
            bbox[2] <= pt->y && pt->y < bbox[3] &&
            bbox[4] <= pt->z && pt->z < bbox[5]) {
            out[m++] = *pt;
        }
    }
    return m;
}

/* src/cwipc_filters.cpp:376-380 -- operates on PCL's packed word
 * rgba = a<<24 | r<<16 | g<<8 | b  (include/cwipc_util/api_pcl.h:20-70, PCL_ADD_RGB),
 * where a is the tile (src/cwipc_util.cpp:138,245). */
void oracle_colormap(const oracle_point *in, size_t n, uint32_t clearBits, uint32_t setBits, oracle_point *out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t rgba = ((uint32_t)in[i].tile << 24) | ((uint32_t)in[i].r << 16) |
                        ((uint32_t)in[i].g << 8) | (uint32_t)in[i].b;
        rgba &= ~clearBits;
        rgba |= setBits;
        out[i] = in[i];
        out[i].tile = (uint8_t)(rgba >> 24);
        out[i].r = (uint8_t)(rgba >> 16);
        out[i].g = (uint8_t)(rgba >> 8);
        out[i].b = (uint8_t)rgba;
    }
}

/* src/cwipc_filters.cpp:403-409 -- all of pc1 then all of pc2. */
size_t oracle_join(const oracle_point *a, size_t na, const oracle_point *b, size_t nb, oracle_point *out) {
    if (na) memcpy(out, a, na * sizeof(oracle_point));
    if (nb) memcpy(out + na, b, nb * sizeof(oracle_point));
    return na + nb;
}

/* python/cwipc/filters/colorize.py:100-119 -- Python float (= IEEE double) blend,
 * int() truncation toward zero, stored into a c_ubyte field (low 8 bits). */
void oracle_colorize(const oracle_point *in, size_t n, double weight,
                     const double *lut, const uint8_t *valid, oracle_point *out) {
    for (size_t i = 0; i < n; i++) {
        out[i] = in[i];
        unsigned t = in[i].tile;
        if (!valid[t]) continue;
        const double *c = &lut[t * 3];
        double old_r = in[i].r / 255.0, old_g = in[i].g / 255.0, old_b = in[i].b / 255.0;
        double new_r = c[0] * weight + old_r * (1 - weight);
        double new_g = c[1] * weight + old_g * (1 - weight);
        double new_b = c[2] * weight + old_b * (1 - weight);
        out[i].r = (uint8_t)(long long)(new_r * 255);
        out[i].g = (uint8_t)(long long)(new_g * 255);
        out[i].b = (uint8_t)(long long)(new_b * 255);
    }
}

/* ------------------------------------------------------------------------- */
/* Stable LSD radix sort of (key, value) pairs on a 32-bit key                 */
/* ------------------------------------------------------------------------- */

typedef struct { uint32_t key; uint32_t val; } kv32;

static int radix_sort_kv32(kv32 *a, size_t n) {
    if (n < 2) return 0;
    kv32 *tmp = (kv32 *)malloc(n * sizeof(kv32));
    if (!tmp) return -1;
    kv32 *src = a, *dst = tmp;
    for (int pass = 0; pass < 4; pass++) {
        size_t count[257];
        memset(count, 0, sizeof(count));
        int shift = pass * 8;
        for (size_t i = 0; i < n; i++) count[((src[i].key >> shift) & 0xff) + 1]++;
        if (count[1] == n && pass > 0) continue; /* this digit is all zero: nothing moves */
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (size_t i = 0; i < n; i++) dst[count[(src[i].key >> shift) & 0xff]++] = src[i];
        kv32 *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(kv32));
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* pcl::VoxelGrid  [PCL upstream: filters/impl/voxel_grid.hpp applyFilter,     */
/* downsample_all_data_ = true, min_points_per_voxel_ = 0, save_leaf_layout_]  */
/* followed by the tile clear/OR of src/cwipc_filters.cpp:64-74 / :145-155.    */
/* ------------------------------------------------------------------------- */

/* Returns #outputs (>=0), -2 on index overflow, -3 if cap is too small, -4 on OOM.
 * `sel` (optional) selects and orders the input points (a leaf's index vector). */
static long voxelgrid_filter(const oracle_point *pts, const uint32_t *sel, size_t n, float leaf,
                             oracle_point *out, size_t cap, double *audit_mean, uint32_t *audit_count) {
    if (n == 0) return 0;
    /* setLeafSize: inverse_leaf_size_ = 1 / leaf_size_ (fp32). */
    const float inv = 1.0f / leaf;

    /* getMinMax3D */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; i++) {
        const oracle_point *p = &pts[sel ? sel[i] : i];
        if (p->x < mn[0]) mn[0] = p->x;
        if (p->y < mn[1]) mn[1] = p->y;
        if (p->z < mn[2]) mn[2] = p->z;
        if (p->x > mx[0]) mx[0] = p->x;
        if (p->y > mx[1]) mx[1] = p->y;
        if (p->z > mx[2]) mx[2] = p->z;
    }
    /* "Leaf size is too small for the input dataset. Integer indices would overflow." */
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
    int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
    int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) return -2;

    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; a++) {
        min_b[a] = (int)floorf(mn[a] * inv);
        max_b[a] = (int)floorf(mx[a] * inv);
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];

    /* First pass: (idx, cloud_point_index).  Note the fp32 subtraction of min_b. */
    kv32 *iv = (kv32 *)malloc(n * sizeof(kv32));
    if (!iv) return -4;
    for (size_t i = 0; i < n; i++) {
        uint32_t pi = sel ? sel[i] : (uint32_t)i;
        const oracle_point *p = &pts[pi];
        int ijk0 = (int)(floorf(p->x * inv) - (float)min_b[0]);
        int ijk1 = (int)(floorf(p->y * inv) - (float)min_b[1]);
        int ijk2 = (int)(floorf(p->z * inv) - (float)min_b[2]);
        int idx = ijk0 + ijk1 * mul1 + ijk2 * mul2;
        iv[i].key = (uint32_t)idx;
        iv[i].val = pi;
    }
    /* Second pass: sort on idx.  Upstream uses an UNSTABLE sort (std::sort /
     * boost spreadsort), so the order of points inside one voxel -- and with it
     * the fp32 summation order -- is unspecified there; the oracle keeps input
     * order (stable).  This is why xyz parity is a tolerance, not bit-exact. */
    if (radix_sort_kv32(iv, n) != 0) { free(iv); return -4; }

    /* Third/fourth pass: one CentroidPoint per run of equal idx. */
    long total = 0;
    size_t index = 0;
    while (index < n) {
        size_t i = index + 1;
        while (i < n && iv[i].key == iv[index].key) ++i;
        if ((size_t)total >= cap) { free(iv); return -3; }
        /* AccumulatorXYZ (Eigen::Vector3f sum), AccumulatorRGBA (float r,g,b,a). */
        float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
        unsigned tile_or = 0;
        for (size_t li = index; li < i; li++) {
            const oracle_point *p = &pts[iv[li].val];
            sx += p->x; sy += p->y; sz += p->z;
            sr += (float)p->r; sg += (float)p->g; sb += (float)p->b;
            /* cwipc step 3: the layout lookup of every source point lands on the
             * voxel of its own idx (getCentroidIndex recomputes the same cell with
             * an integer subtraction of min_b; identical while |idx| < 2^24). */
            tile_or |= p->tile;
        }
        size_t cnt = i - index;
        if (audit_mean) {   /* test aid, not part of the reference: the voxel's mean in double, and its population */
            double ax = 0, ay = 0, az = 0;
            for (size_t li = index; li < i; li++) {
                const oracle_point *p = &pts[iv[li].val];
                ax += (double)p->x; ay += (double)p->y; az += (double)p->z;
            }
            audit_mean[3 * total + 0] = ax / (double)cnt; audit_mean[3 * total + 1] = ay / (double)cnt; audit_mean[3 * total + 2] = az / (double)cnt;
        }
        if (audit_count) audit_count[total] = (uint32_t)cnt;
        oracle_point *o = &out[total];
        o->x = sx / (float)cnt;
        o->y = sy / (float)cnt;
        o->z = sz / (float)cnt;
        o->r = (uint8_t)(uint32_t)(sr / (float)cnt);
        o->g = (uint8_t)(uint32_t)(sg / (float)cnt);
        o->b = (uint8_t)(uint32_t)(sb / (float)cnt);
        o->tile = (uint8_t)tile_or;
        total++;
        index = i;
    }
    free(iv);
    return total;
}

/* src/cwipc_filters.cpp:30-87 */
/* Test aid (oracle_downsample_audit): where the per-output double means and populations go, if anybody wants them. */
static __thread double *g_audit_mean;
static __thread uint32_t *g_audit_count;

long oracle_downsample_voxelgrid(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                                 oracle_point *out, size_t cap, float *out_cellsize) {
    /* :42-46 */
    if (pc_cellsize >= cellsize) cellsize = pc_cellsize;
    if (out_cellsize) *out_cellsize = cellsize;
    long m = voxelgrid_filter(in, NULL, n, cellsize, out, cap, g_audit_mean, g_audit_count);
    if (m == 0) return -1; /* :58-62 "VoxelGrid filter produced empty pointcloud" -> NULL */
    return m;
}

/* ------------------------------------------------------------------------- */
/* pcl::octree::OctreePointCloud  [PCL upstream: octree_pointcloud.hpp          */
/* addPointIdx / adoptBoundingBoxToPoint / getKeyBitSize / genOctreeKeyforPoint */
/* and OctreeDepthFirstIterator child order]                                    */
/* ------------------------------------------------------------------------- */

typedef struct {
    double res;
    double mn[3], mx[3];
    int depth;
    int defined;
} obox;

static int obox_first_point(obox *b, const oracle_point *p) {
    const double eps = (double)FLT_EPSILON; /* minValue */
    const float c[3] = {p->x, p->y, p->z};
    for (int a = 0; a < 3; a++) {
        b->mn[a] = c[a] - b->res / 2;
        b->mx[a] = c[a] + b->res / 2;
    }
    /* getKeyBitSize() with leaf_count_ == 0 */
    uint32_t max_key = 0;
    for (int a = 0; a < 3; a++) {
        uint32_t mk = (uint32_t)ceil((b->mx[a] - b->mn[a] - eps) / b->res);
        if (mk > max_key) max_key = mk;
    }
    uint32_t max_voxels = max_key > 2 ? max_key : 2;
    double d = ceil(log2((double)max_voxels) - eps);
    if (d > 32) d = 32;
    if (d < 0) d = 0;
    b->depth = (int)d;
    double side = (double)(1u << b->depth) * b->res;
    for (int a = 0; a < 3; a++) {
        double oversize = (side - (b->mx[a] - b->mn[a])) / 2.0;
        if (oversize > eps) {
            b->mn[a] -= oversize;
            b->mx[a] += oversize;
        }
    }
    b->defined = 1;
    return 0;
}

/* Grow the box until p fits.  For every growth step, shift[a] receives the
 * key offset (2^old_depth) that existing keys gain on axes whose minimum moved. */
static int obox_adopt(obox *b, const oracle_point *p, uint32_t *keys, size_t nkeys) {
    const double eps = (double)FLT_EPSILON;
    const float c[3] = {p->x, p->y, p->z};
    for (;;) {
        if (!b->defined) { obox_first_point(b, p); continue; }
        int lo[3], up[3], any = 0;
        for (int a = 0; a < 3; a++) {
            lo[a] = c[a] < b->mn[a];
            up[a] = c[a] >= b->mx[a];
            any |= lo[a] | up[a];
        }
        if (!any) return 0;
        if (b->depth >= 31) return -1; /* upstream asserts on OctreeKey::maxDepth */
        double side = (double)(1u << b->depth) * b->res;
        for (int a = 0; a < 3; a++) {
            if (!up[a]) {
                b->mn[a] -= side;
                /* the old root becomes the upper child on this axis */
                for (size_t i = 0; i < nkeys; i++) keys[i * 3 + a] += (1u << b->depth);
            }
        }
        b->depth++;
        side = (double)(1u << b->depth) * b->res - eps;
        for (int a = 0; a < 3; a++) b->mx[a] = b->mn[a] + side;
    }
}

/* Depth-first leaf order: at every level the child index is
 * (xbit<<2)|(ybit<<1)|zbit, children visited in ascending index. */
static int morton_less(const uint32_t *a, const uint32_t *b) {
    uint32_t dx = a[0] ^ b[0], dy = a[1] ^ b[1], dz = a[2] ^ b[2];
    uint32_t m = dx | dy | dz;
    if (!m) return 0;
    /* highest differing bit level */
    int top = 31;
    while (!((m >> top) & 1u)) top--;
    uint32_t bit = 1u << top;
    unsigned ca = ((a[0] & bit) ? 4 : 0) | ((a[1] & bit) ? 2 : 0) | ((a[2] & bit) ? 1 : 0);
    unsigned cb = ((b[0] & bit) ? 4 : 0) | ((b[1] & bit) ? 2 : 0) | ((b[2] & bit) ? 1 : 0);
    return ca < cb;
}

typedef struct { uint32_t k[3]; uint32_t first; uint32_t count; uint32_t fill; } oleaf;

static const oleaf *g_sort_leaves;
static int leaf_cmp(const void *pa, const void *pb) {
    const oleaf *a = &g_sort_leaves[*(const uint32_t *)pa];
    const oleaf *b = &g_sort_leaves[*(const uint32_t *)pb];
    if (morton_less(a->k, b->k)) return -1;
    if (morton_less(b->k, a->k)) return 1;
    return 0;
}

/* src/cwipc_filters.cpp:89-172 */
long oracle_downsample(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                       oracle_point *out, size_t cap, float *out_cellsize, int *n_leaves, int *depth_out) {
    if (cellsize < 0) /* :90-92 */
        return oracle_downsample_voxelgrid(in, n, pc_cellsize, -cellsize, out, cap, out_cellsize);
    /* :103-107 */
    if (pc_cellsize >= cellsize) cellsize = pc_cellsize;
    if (out_cellsize) *out_cellsize = cellsize;
    if (n_leaves) *n_leaves = 0;
    if (depth_out) *depth_out = 0;
    if (n == 0) return 0; /* zero leaves -> empty cloud (test_downsample_empty) */

    /* :113-114 -- float product, widened to the octree's double resolution_ */
    const int octree_count = 8 * 8;
    float octree_cellsize = octree_count * cellsize;

    obox box;
    memset(&box, 0, sizeof(box));
    box.res = (double)octree_cellsize;

    /* addPointsFromInputCloud: insert finite points in index order. */
    uint32_t *keys = (uint32_t *)malloc(n * 3 * sizeof(uint32_t));
    uint32_t *pidx = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!keys || !pidx) { free(keys); free(pidx); return -4; }
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const oracle_point *p = &in[i];
        if (!isfinite(p->x) || !isfinite(p->y) || !isfinite(p->z)) continue;
        if (obox_adopt(&box, p, keys, m) != 0) { free(keys); free(pidx); return -2; }
        /* genOctreeKeyforPoint */
        keys[m * 3 + 0] = (uint32_t)((p->x - box.mn[0]) / box.res);
        keys[m * 3 + 1] = (uint32_t)((p->y - box.mn[1]) / box.res);
        keys[m * 3 + 2] = (uint32_t)((p->z - box.mn[2]) / box.res);
        pidx[m] = (uint32_t)i;
        m++;
    }
    if (depth_out) *depth_out = box.depth;

    /* Group points per leaf (insertion order inside a leaf) via a small hash map. */
    size_t hcap = 1024;
    uint32_t *htab = NULL;
    oleaf *leaves = NULL;
    size_t nleaf = 0, leafcap = 0;
    uint32_t *leaf_of = (uint32_t *)malloc((m ? m : 1) * sizeof(uint32_t));
    if (!leaf_of) { free(keys); free(pidx); return -4; }
rebuild:
    free(htab);
    htab = (uint32_t *)malloc(hcap * sizeof(uint32_t));
    if (!htab) { free(keys); free(pidx); free(leaf_of); free(leaves); return -4; }
    memset(htab, 0xff, hcap * sizeof(uint32_t));
    nleaf = 0;
    for (size_t i = 0; i < m; i++) {
        const uint32_t *k = &keys[i * 3];
        uint64_t h = (uint64_t)k[0] * 0x9E3779B97F4A7C15ull ^ (uint64_t)k[1] * 0xC2B2AE3D27D4EB4Full ^
                     (uint64_t)k[2] * 0x165667B19E3779F9ull;
        size_t s = (size_t)(h >> 17) & (hcap - 1);
        for (;;) {
            uint32_t id = htab[s];
            if (id == 0xffffffffu) {
                if ((nleaf + 1) * 2 > hcap) { hcap *= 4; goto rebuild; }
                if (nleaf == leafcap) {
                    leafcap = leafcap ? leafcap * 2 : 64;
                    leaves = (oleaf *)realloc(leaves, leafcap * sizeof(oleaf));
                    if (!leaves) { free(keys); free(pidx); free(leaf_of); free(htab); return -4; }
                }
                leaves[nleaf].k[0] = k[0]; leaves[nleaf].k[1] = k[1]; leaves[nleaf].k[2] = k[2];
                leaves[nleaf].count = 0;
                htab[s] = (uint32_t)nleaf;
                id = (uint32_t)nleaf++;
            } else if (leaves[id].k[0] != k[0] || leaves[id].k[1] != k[1] || leaves[id].k[2] != k[2]) {
                s = (s + 1) & (hcap - 1);
                continue;
            }
            leaf_of[i] = id;
            leaves[id].count++;
            break;
        }
    }
    free(htab);
    if (n_leaves) *n_leaves = (int)nleaf;

    /* bucket the point indices per leaf, stable */
    uint32_t first = 0;
    for (size_t l = 0; l < nleaf; l++) { leaves[l].first = first; leaves[l].fill = 0; first += leaves[l].count; }
    uint32_t *bucket = (uint32_t *)malloc((m ? m : 1) * sizeof(uint32_t));
    uint32_t *order = (uint32_t *)malloc((nleaf ? nleaf : 1) * sizeof(uint32_t));
    if (!bucket || !order) { free(keys); free(pidx); free(leaf_of); free(leaves); free(bucket); free(order); return -4; }
    for (size_t i = 0; i < m; i++) {
        oleaf *L = &leaves[leaf_of[i]];
        bucket[L->first + L->fill++] = pidx[i];
    }
    /* leaf_depth_begin() order */
    for (size_t l = 0; l < nleaf; l++) order[l] = (uint32_t)l;
    g_sort_leaves = leaves;
    qsort(order, nleaf, sizeof(uint32_t), leaf_cmp);

    /* :124-158 -- per leaf: gather, VoxelGrid, tile clear/OR, append */
    long total = 0;
    for (size_t o = 0; o < nleaf; o++) {
        const oleaf *L = &leaves[order[o]];
        long got = voxelgrid_filter(in, &bucket[L->first], L->count, cellsize, out + total, cap - (size_t)total,
                                    g_audit_mean ? g_audit_mean + 3 * total : NULL, g_audit_count ? g_audit_count + total : NULL);
        if (got < 0) { total = got; break; }
        total += got;
    }
    free(keys); free(pidx); free(leaf_of); free(leaves); free(bucket); free(order);
    return total;
}

/* cwipc_downsample (either sign of cellsize) plus, per output, the mean of its contributors in DOUBLE and their number.
 * Not part of the reference: it lets the tests say how far the fp32 running sums of pcl::VoxelGrid (what `out` holds) and the
 * HIP path each are from the exact mean.  mean64: cap x 3 doubles, count: cap words. */
long oracle_downsample_audit(const oracle_point *in, size_t n, float pc_cellsize, float cellsize,
                             oracle_point *out, size_t cap, float *out_cellsize, double *mean64, uint32_t *count) {
    g_audit_mean = mean64;
    g_audit_count = count;
    long m = oracle_downsample(in, n, pc_cellsize, cellsize, out, cap, out_cellsize, NULL, NULL);
    g_audit_mean = NULL;
    g_audit_count = NULL;
    return m;
}

/* ------------------------------------------------------------------------- */
/* pcl::StatisticalOutlierRemoval  [PCL upstream: filters/impl/                 */
/* statistical_outlier_removal.hpp applyFilterIndices; search = exact k-NN      */
/* (FLANN KDTreeSingleIndex, L2_Simple fp32 distance, eps = 0, sorted)]         */
/* ------------------------------------------------------------------------- */

typedef struct { uint64_t key; uint32_t idx; } cellent;

static int cellent_cmp(const void *a, const void *b) {
    const cellent *x = (const cellent *)a, *y = (const cellent *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

#define CELL_BITS 21
static inline uint64_t cell_key(int cx, int cy, int cz) {
    return ((uint64_t)(uint32_t)cz << (2 * CELL_BITS)) | ((uint64_t)(uint32_t)cy << CELL_BITS) | (uint64_t)(uint32_t)cx;
}

typedef struct {
    const oracle_point *pts;
    size_t n;
    double h;
    float mn[3];
    int dim[3];
    cellent *ents;       /* sorted by cell key */
    uint64_t *hkeys;     /* hash: cell key+1 -> first entry */
    uint32_t *hfirst, *hcount;
    size_t hcap;
} knn_grid;

static void grid_free(knn_grid *g) {
    free(g->ents); free(g->hkeys); free(g->hfirst); free(g->hcount);
}

static inline void point_cell(const knn_grid *g, const oracle_point *p, int c[3]) {
    const float v[3] = {p->x, p->y, p->z};
    for (int a = 0; a < 3; a++) {
        int k = (int)floor(((double)v[a] - (double)g->mn[a]) / g->h);
        if (k < 0) k = 0;
        if (k >= g->dim[a]) k = g->dim[a] - 1;
        c[a] = k;
    }
}

static inline size_t hslot(uint64_t key, size_t cap) {
    return (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
}

static int grid_build(knn_grid *g, const oracle_point *pts, size_t n, double h) {
    memset(g, 0, sizeof(*g));
    g->pts = pts; g->n = n; g->h = h;
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    g->mn[0] = g->mn[1] = g->mn[2] = FLT_MAX;
    for (size_t i = 0; i < n; i++) {
        const float v[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; a++) {
            if (v[a] < g->mn[a]) g->mn[a] = v[a];
            if (v[a] > mx[a]) mx[a] = v[a];
        }
    }
    for (int a = 0; a < 3; a++) {
        double d = floor(((double)mx[a] - (double)g->mn[a]) / h) + 1;
        if (d > (double)((1 << CELL_BITS) - 1)) return -1;
        g->dim[a] = (int)d;
    }
    g->ents = (cellent *)malloc((n ? n : 1) * sizeof(cellent));
    if (!g->ents) return -1;
    for (size_t i = 0; i < n; i++) {
        int c[3];
        point_cell(g, &pts[i], c);
        g->ents[i].key = cell_key(c[0], c[1], c[2]);
        g->ents[i].idx = (uint32_t)i;
    }
    qsort(g->ents, n, sizeof(cellent), cellent_cmp);
    size_t ncell = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || g->ents[i].key != g->ents[i - 1].key) ncell++;
    g->hcap = 16;
    while (g->hcap < ncell * 2 + 1) g->hcap <<= 1;
    g->hkeys = (uint64_t *)calloc(g->hcap, sizeof(uint64_t));
    g->hfirst = (uint32_t *)malloc(g->hcap * sizeof(uint32_t));
    g->hcount = (uint32_t *)malloc(g->hcap * sizeof(uint32_t));
    if (!g->hkeys || !g->hfirst || !g->hcount) return -1;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && g->ents[j].key == g->ents[i].key) j++;
        uint64_t k1 = g->ents[i].key + 1;
        size_t s = hslot(k1, g->hcap);
        while (g->hkeys[s]) s = (s + 1) & (g->hcap - 1);
        g->hkeys[s] = k1; g->hfirst[s] = (uint32_t)i; g->hcount[s] = (uint32_t)(j - i);
        i = j;
    }
    return 0;
}

static inline int grid_lookup(const knn_grid *g, int cx, int cy, int cz, uint32_t *first, uint32_t *count) {
    uint64_t k1 = cell_key(cx, cy, cz) + 1;
    size_t s = hslot(k1, g->hcap);
    while (g->hkeys[s]) {
        if (g->hkeys[s] == k1) { *first = g->hfirst[s]; *count = g->hcount[s]; return 1; }
        s = (s + 1) & (g->hcap - 1);
    }
    return 0;
}

/* FLANN L2_Simple<float>: result += diff*diff, dimension order x,y,z, fp32. */
static inline float l2_simple(const oracle_point *a, const oracle_point *b) {
    float result = 0;
    float d = a->x - b->x; result += d * d;
    d = a->y - b->y; result += d * d;
    d = a->z - b->z; result += d * d;
    return result;
}

/* keep the `want` smallest values in best[] (ascending) */
static inline void topk_insert(float *best, int *have, int want, float v) {
    int n = *have;
    if (n == want) {
        if (v >= best[n - 1]) return;
        n--;
    }
    int j = n;
    while (j > 0 && best[j - 1] > v) { best[j] = best[j - 1]; j--; }
    best[j] = v;
    *have = n + 1;
}

/* d_i = float( sum_{j=1..k} sqrt(dist2_j) / k ), dist2 ascending, index 0 (the
 * query itself, distance 0) skipped; sum in double.  `sqrt` on a float resolves
 * to the float overload with libstdc++, so sqrtf.  If fewer than k+1 points
 * exist upstream reads past the result arrays (undefined); the oracle DEFINES
 * the missing distances as 0. */
static float mean_dist_of(const float *best, int have, int k) {
    double dist_sum = 0.0;
    for (int j = 1; j < have; j++) dist_sum += sqrtf(best[j]);
    return (float)(dist_sum / k);
}

int oracle_knn_mean_dist(const oracle_point *in, size_t n, int k, float *mean_dist) {
    if (n == 0) return 0;
    if (k < 1) { for (size_t i = 0; i < n; i++) mean_dist[i] = 0; return 0; }
    const int want = k + 1;
    float *best = (float *)malloc((size_t)want * sizeof(float));
    if (!best) return -1;

    /* choose a cell size: probe the occupancy at extent/512, then rescale so that
     * an occupied cell holds about (k+1)/4 points assuming a surface (pts ~ h^2). */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; i++) {
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
    }
    double ext = 0;
    for (int a = 0; a < 3; a++) if ((double)mx[a] - mn[a] > ext) ext = (double)mx[a] - mn[a];
    if (!(ext > 0)) ext = 1.0;
    double h = ext / 512.0;
    knn_grid g;
    if (grid_build(&g, in, n, h) != 0) { grid_free(&g); free(best); return -1; }
    size_t occ = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || g.ents[i].key != g.ents[i - 1].key) occ++;
    double ppc = (double)n / (double)occ;
    double target = (double)want / 4.0;
    double h2 = h * sqrt(target / ppc);
    if (h2 < ext / 1.0e6) h2 = ext / 1.0e6;
    if (h2 > ext) h2 = ext;
    grid_free(&g);
    if (grid_build(&g, in, n, h2) != 0) { grid_free(&g); free(best); return -1; }
    h = h2;

    const int maxring = (g.dim[0] > g.dim[1] ? (g.dim[0] > g.dim[2] ? g.dim[0] : g.dim[2])
                                             : (g.dim[1] > g.dim[2] ? g.dim[1] : g.dim[2]));
    for (size_t i = 0; i < n; i++) {
        const oracle_point *q = &in[i];
        int c[3];
        point_cell(&g, q, c);
        int have = 0;
        for (int ring = 0; ring <= maxring; ring++) {
            /* visit the shell of Chebyshev radius `ring` */
            for (int dz = -ring; dz <= ring; dz++) {
                int cz = c[2] + dz;
                if (cz < 0 || cz >= g.dim[2]) continue;
                for (int dy = -ring; dy <= ring; dy++) {
                    int cy = c[1] + dy;
                    if (cy < 0 || cy >= g.dim[1]) continue;
                    int onface = (dz == -ring || dz == ring || dy == -ring || dy == ring);
                    int step = onface ? 1 : (2 * ring > 0 ? 2 * ring : 1);
                    for (int dx = -ring; dx <= ring; dx += step) {
                        int cx = c[0] + dx;
                        if (cx < 0 || cx >= g.dim[0]) continue;
                        uint32_t first, count;
                        if (!grid_lookup(&g, cx, cy, cz, &first, &count)) continue;
                        for (uint32_t e = first; e < first + count; e++)
                            topk_insert(best, &have, want, l2_simple(q, &in[g.ents[e].idx]));
                    }
                }
            }
            /* every unvisited point is farther than ring*h from q */
            if (have == want) {
                double reach = (double)ring * h;
                if ((double)best[want - 1] < reach * reach * (1.0 - 1e-6)) break;
            }
        }
        mean_dist[i] = mean_dist_of(best, have, k);
    }
    grid_free(&g);
    free(best);
    return 0;
}

/* The inner overload, src/cwipc_filters.cpp:181-211: SOR over one cloud. */
static long sor_filter(const oracle_point *in, size_t n, int k, float stddev_mul,
                       oracle_point *out, float *mean_dist, double *thr_out) {
    if (n == 0) { if (thr_out) *thr_out = NAN; return 0; }
    float *distances = mean_dist ? mean_dist : (float *)malloc(n * sizeof(float));
    if (!distances) return -1;
    if (oracle_knn_mean_dist(in, n, k, distances) != 0) { if (!mean_dist) free(distances); return -1; }
    /* mean / stddev of the distance vector: double accumulators, float square */
    double sum = 0, sq_sum = 0;
    for (size_t i = 0; i < n; i++) {
        float distance = distances[i];
        sum += distance;
        sq_sum += distance * distance;
    }
    double valid = (double)n;
    double mean = sum / valid;
    double variance = (sq_sum - sum * sum / valid) / (valid - 1);
    double stddev = sqrt(variance);
    double distance_threshold = mean + (double)stddev_mul * stddev;
    if (thr_out) *thr_out = distance_threshold;
    long kept = 0;
    for (size_t i = 0; i < n; i++) {
        if (distances[i] > distance_threshold) continue;
        out[kept++] = in[i];
    }
    if (!mean_dist) free(distances);
    return kept;
}

/* src/cwipc_filters.cpp:222-278 */
long oracle_remove_outliers(const oracle_point *in, size_t n, int k, float stddev_mul, int per_tile,
                            oracle_point *out, float *mean_dist, double *thr) {
    if (!per_tile) return sor_filter(in, n, k, stddev_mul, out, mean_dist, thr);
    /* :241-249 distinct tiles in first-appearance order */
    int tiles[256], ntiles = 0;
    uint8_t seen[256];
    memset(seen, 0, sizeof(seen));
    for (size_t i = 0; i < n; i++) {
        if (!seen[in[i].tile]) { seen[in[i].tile] = 1; tiles[ntiles++] = in[i].tile; }
    }
    oracle_point *aux = (oracle_point *)malloc((n ? n : 1) * sizeof(oracle_point));
    if (!aux) return -1;
    long total = 0;
    for (int t = 0; t < ntiles; t++) {
        /* :252 cwipc_tilefilter(pc, tile) -- tile 0 acts as a wildcard there */
        size_t na = oracle_tilefilter(in, n, tiles[t], aux);
        long got = sor_filter(aux, na, k, stddev_mul, out + total, NULL, NULL);
        if (got < 0) { free(aux); return -1; }
        total += got;
    }
    free(aux);
    return total;
}
