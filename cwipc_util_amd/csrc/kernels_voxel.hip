// kernels_voxel.hip -- voxel-grid downsample on gfx950.
//
// Reference: cwipc_downsample / cwipc_downsample_voxelgrid, src/cwipc_filters.cpp:30-172.
// The arithmetic the reference delegates to PCL is restated from the published
// upstream algorithms (see oracle/cwipc_oracle.c for the scalar restatement):
//   pcl::VoxelGrid            voxel (i,j,k) = floor(p * (1/leaf)) in fp32; one output per
//                             occupied voxel = mean xyz, truncated mean rgb; outputs in
//                             ascending (k,j,i); grids above 2^31 cells are refused.
//   pcl::octree::OctreePointCloud (positive cellsize only) leaves of side R = 64*leaf on a
//                             lattice anchored at the first point; each leaf is voxelised
//                             separately, so a voxel cut by a leaf face yields one output
//                             per side; leaves are emitted in depth-first (Morton) order of
//                             the final octree keys, which depend on how the bounding box
//                             grew while points were inserted in input order.
//   tile of an output         OR of the tiles of its contributors (src/cwipc_filters.cpp:64-74).
//
// Structure (all HBM-bound integer/byte work, no MFMA):
//   K1 voxel_accumulate   one pass over the planes (the 16 B/point of algorithmic traffic).
//                         Lane-local merge of consecutive equal voxels -> per-workgroup LDS
//                         hash table (64-bit keys, packed 64-bit integer sums, all LDS
//                         atomics) -> one flush per workgroup into a global open-addressing
//                         table with integer atomics.  Sums are fixed-point integers, so the
//                         result is bitwise reproducible run to run.  Also emits one
//                         bounding box per workgroup chunk.
//   K2 octree_replay      one workgroup replays the octree's bounding-box growth over the
//                         chunk boxes (re-reading only chunks that trigger a growth step),
//                         or, for the plain grid, reduces them to the global box.
//   K3 make_sort_keys     one lane per occupied voxel: 64-bit output-order key.
//   (radix sort of the ~40 k keys)
//   K4 emit_and_clean     gathers the sums in output order, writes the planes, and zeroes
//                         the table slots it read so the workspace is clean for the next call.
#include "internal.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cfloat>
#include <cmath>

namespace cwipc_amd {

namespace {

// ---------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------
constexpr int ACC_BLOCK = 256;
constexpr int ACC_ITEMS = 4;                                  // points per lane per step (dwordx4 per plane)
constexpr int ACC_STEPS = 4;
constexpr int ACC_CHUNK = ACC_BLOCK * ACC_ITEMS * ACC_STEPS;  // 4096 points per workgroup
constexpr int LDS_SLOTS = 1024;                               // per-workgroup table
constexpr int LDS_PROBES = 24;
constexpr int COORD_BITS = 20;                                // voxel offset from the first point's voxel
constexpr int COORD_BIAS = 1 << (COORD_BITS - 1);
constexpr double FIX_ONE = 1073741824.0;                      // 2^30 fixed-point units per voxel edge

enum : uint32_t {
    ERR_RANGE = 1,        // a voxel lies more than 2^19 cells from the first point
    ERR_TABLE_FULL = 2,
    ERR_DEPTH = 4,        // octree deeper than the sort key can express
    ERR_GRID_OVERFLOW = 8,   // pcl::VoxelGrid: "Leaf size is too small ... indices would overflow"
    ERR_LEAF_RANGE = 16,
    ERR_FIRST_POINT = 32,
    ERR_CELL_RANGE = 64,
};

// control block, 32-bit words in device memory
enum { C_ERR = 0, C_COUNT = 1, C_DEPTH = 2, C_EVENTS = 3, C_SHIFT = 4 /* 3 x int64 */, C_MINB = 10, C_DIVB = 13, C_WORDS = 32 };

struct VoxParams {
    size_t n;
    float inv_leaf;      // 1 / leaf in fp32, as pcl::VoxelGrid::setLeafSize
    float leaf;
    double leaf_d;       // (double)leaf
    double fix_scale;    // 2^30 / leaf
    double res;          // octree resolution (double)(float)(64 * leaf)
    int leaf_split;
    uint32_t table_mask; // global table capacity - 1
};

struct VoxTable {
    unsigned long long *keys;   // 0 = empty
    unsigned long long *sx, *sy, *sz;   // fixed-point coordinate sums (two's complement)
    unsigned long long *cr;     // count << 32 | sum r
    unsigned long long *gb;     // sum g << 32 | sum b
    uint32_t *tile;             // OR of tiles
    unsigned long long *leaf;   // packed leaf lattice coordinates, written by the claiming lane
    uint32_t *occupied;         // list of claimed slots
    uint32_t *ctrl;
};

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// Octree box after the first point: adoptBoundingBoxToPoint's "octree is empty"
// branch followed by getKeyBitSize() [PCL upstream octree_pointcloud.hpp].
__device__ __forceinline__ void first_box(const double p[3], double res, double mn[3], double mx[3], int &depth) {
    const double eps = (double)FLT_EPSILON;
    unsigned max_key = 0;
    for (int a = 0; a < 3; a++) {
        mn[a] = p[a] - res / 2;
        mx[a] = p[a] + res / 2;
        unsigned mk = (unsigned)ceil((mx[a] - mn[a] - eps) / res);
        max_key = mk > max_key ? mk : max_key;
    }
    unsigned max_voxels = max_key > 2 ? max_key : 2;
    double d = ceil(log2((double)max_voxels) - eps);
    d = d > 32 ? 32 : (d < 0 ? 0 : d);
    depth = (int)d;
    double side = (double)(1u << depth) * res;
    for (int a = 0; a < 3; a++) {
        double oversize = (side - (mx[a] - mn[a])) / 2.0;
        if (oversize > eps) {
            mn[a] -= oversize;
            mx[a] += oversize;
        }
    }
}

__device__ __forceinline__ unsigned long long pack_leaf(int lx, int ly, int lz) {
    return ((unsigned long long)(uint32_t)(lx & 0x1fffff)) | ((unsigned long long)(uint32_t)(ly & 0x1fffff) << 21) |
           ((unsigned long long)(uint32_t)(lz & 0x1fffff) << 42);
}
__device__ __forceinline__ int unpack_leaf(unsigned long long v, int axis) {
    int t = (int)((v >> (21 * axis)) & 0x1fffff);
    return (t << 11) >> 11;   // sign-extend 21 bits
}

struct Run {
    unsigned long long key;
    unsigned long long leaf;
    long long qx, qy, qz;
    unsigned long long cr, gb;
    uint32_t tile;
};

// Insert a run into the global table.  Integer atomics only.
__device__ __forceinline__ void global_insert(const VoxTable &t, uint32_t mask, const Run &r) {
    uint32_t slot = (uint32_t)mix64(r.key) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {
        unsigned long long old = atomicCAS(&t.keys[slot], 0ull, r.key);
        if (old == 0ull) {
            // this lane claimed the slot: it alone records the slot and its leaf
            uint32_t idx = atomicAdd(&t.ctrl[C_COUNT], 1u);
            t.occupied[idx] = slot;
            t.leaf[slot] = r.leaf;
            old = r.key;
        }
        if (old == r.key) {
            atomicAdd(&t.sx[slot], (unsigned long long)r.qx);
            atomicAdd(&t.sy[slot], (unsigned long long)r.qy);
            atomicAdd(&t.sz[slot], (unsigned long long)r.qz);
            atomicAdd(&t.cr[slot], r.cr);
            atomicAdd(&t.gb[slot], r.gb);
            atomicOr(&t.tile[slot], r.tile);
            return;
        }
        slot = (slot + 1) & mask;
        if (probe > 4096) break;
    }
    atomicOr(&t.ctrl[C_ERR], ERR_TABLE_FULL);
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
struct LdsTable {
    unsigned long long keys[LDS_SLOTS];
    unsigned long long leaf[LDS_SLOTS];
    unsigned long long sx[LDS_SLOTS], sy[LDS_SLOTS], sz[LDS_SLOTS];
    unsigned long long cr[LDS_SLOTS], gb[LDS_SLOTS];
    uint32_t tile[LDS_SLOTS];
};

__device__ __forceinline__ void lds_insert(LdsTable &l, const VoxTable &t, uint32_t gmask, const Run &r) {
    uint32_t slot = (uint32_t)(mix64(r.key) >> 32) & (LDS_SLOTS - 1);
    for (int probe = 0; probe < LDS_PROBES; probe++) {
        unsigned long long old = atomicCAS(&l.keys[slot], 0ull, r.key);
        if (old == 0ull) {
            l.leaf[slot] = r.leaf;
            old = r.key;
        }
        if (old == r.key) {
            atomicAdd(&l.sx[slot], (unsigned long long)r.qx);
            atomicAdd(&l.sy[slot], (unsigned long long)r.qy);
            atomicAdd(&l.sz[slot], (unsigned long long)r.qz);
            atomicAdd(&l.cr[slot], r.cr);
            atomicAdd(&l.gb[slot], r.gb);
            atomicOr(&l.tile[slot], r.tile);
            return;
        }
        slot = (slot + 1) & (LDS_SLOTS - 1);
    }
    global_insert(t, gmask, r);   // workgroup table saturated (incoherent input): go straight to HBM
}

__global__ void __launch_bounds__(ACC_BLOCK) voxel_accumulate_kernel(VoxParams P, const float *__restrict__ x, const float *__restrict__ y,
                                                                    const float *__restrict__ z, const uint32_t *__restrict__ rgbt,
                                                                    VoxTable T, float *__restrict__ chunk_bbox) {
    __shared__ LdsTable L;
    __shared__ float red[6][ACC_BLOCK / 64];

    for (int i = threadIdx.x; i < LDS_SLOTS; i += ACC_BLOCK) {
        L.keys[i] = 0; L.leaf[i] = 0; L.sx[i] = 0; L.sy[i] = 0; L.sz[i] = 0; L.cr[i] = 0; L.gb[i] = 0; L.tile[i] = 0;
    }

    // Anchor: the first point (octree lattice phase and the origin of the key offsets).
    const float p0x = x[0], p0y = y[0], p0z = z[0];
    const int i0 = (int)floorf(p0x * P.inv_leaf), j0 = (int)floorf(p0y * P.inv_leaf), k0 = (int)floorf(p0z * P.inv_leaf);
    double mn0[3] = {0, 0, 0};
    if (P.leaf_split) {
        double pp[3] = {(double)p0x, (double)p0y, (double)p0z}, mx0[3];
        int d0;
        first_box(pp, P.res, mn0, mx0, d0);
    }
    __syncthreads();

    float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const size_t chunk0 = (size_t)blockIdx.x * ACC_CHUNK;

#pragma unroll 1
    for (int s = 0; s < ACC_STEPS; s++) {
        const size_t base = chunk0 + (size_t)s * ACC_BLOCK * ACC_ITEMS + (size_t)threadIdx.x * ACC_ITEMS;
        if (base >= P.n) continue;
        float px[4], py[4], pz[4];
        uint32_t pw[4];
        int cnt = 4;
        if (base + 4 <= P.n) {
            float4 a = *(const float4 *)(x + base), b = *(const float4 *)(y + base), c = *(const float4 *)(z + base);
            uint4 w = *(const uint4 *)(rgbt + base);
            px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w;
            py[0] = b.x; py[1] = b.y; py[2] = b.z; py[3] = b.w;
            pz[0] = c.x; pz[1] = c.y; pz[2] = c.z; pz[3] = c.w;
            pw[0] = w.x; pw[1] = w.y; pw[2] = w.z; pw[3] = w.w;
        } else {
            cnt = (int)(P.n - base);
            for (int j = 0; j < 4; j++) {
                bool ok = j < cnt;
                px[j] = ok ? x[base + j] : 0.f;
                py[j] = ok ? y[base + j] : 0.f;
                pz[j] = ok ? z[base + j] : 0.f;
                pw[j] = ok ? rgbt[base + j] : 0u;
            }
        }

        Run cur;
        cur.key = 0;
        cur.leaf = 0; cur.qx = cur.qy = cur.qz = 0; cur.cr = cur.gb = 0; cur.tile = 0;
#pragma unroll 1
        for (int j = 0; j <= 4; j++) {
            unsigned long long key = 0, leafp = 0;
            long long qx = 0, qy = 0, qz = 0;
            uint32_t w = 0;
            if (j < cnt) {
                const float fx = px[j], fy = py[j], fz = pz[j];
                w = pw[j];
                if (isfinite(fx) && isfinite(fy) && isfinite(fz)) {
                    bmin[0] = fminf(bmin[0], fx); bmax[0] = fmaxf(bmax[0], fx);
                    bmin[1] = fminf(bmin[1], fy); bmax[1] = fmaxf(bmax[1], fy);
                    bmin[2] = fminf(bmin[2], fz); bmax[2] = fmaxf(bmax[2], fz);
                    // pcl::VoxelGrid: floor(p * inverse_leaf_size) with an fp32 product
                    const float gx = floorf(__fmul_rn(fx, P.inv_leaf)), gy = floorf(__fmul_rn(fy, P.inv_leaf)), gz = floorf(__fmul_rn(fz, P.inv_leaf));
                    const int vi = (int)gx, vj = (int)gy, vk = (int)gz;
                    const int di = vi - i0 + COORD_BIAS, dj = vj - j0 + COORD_BIAS, dk = vk - k0 + COORD_BIAS;
                    if ((unsigned)di >> COORD_BITS || (unsigned)dj >> COORD_BITS || (unsigned)dk >> COORD_BITS ||
                        fabsf(gx) > 1.0e9f || fabsf(gy) > 1.0e9f || fabsf(gz) > 1.0e9f) {
                        atomicOr(&T.ctrl[C_ERR], ERR_RANGE);
                    } else {
                        key = (1ull << 63) | ((unsigned long long)di) | ((unsigned long long)dj << COORD_BITS) | ((unsigned long long)dk << (2 * COORD_BITS));
                        if (P.leaf_split) {
                            // genOctreeKeyforPoint relative to the first box: floor((p - min) / resolution) in double
                            const int lx = (int)floor(((double)fx - mn0[0]) / P.res);
                            const int ly = (int)floor(((double)fy - mn0[1]) / P.res);
                            const int lz = (int)floor(((double)fz - mn0[2]) / P.res);
                            // a voxel touches at most two leaves per axis: the parity of the leaf index tells them apart
                            key |= ((unsigned long long)(lx & 1) << 60) | ((unsigned long long)(ly & 1) << 61) | ((unsigned long long)(lz & 1) << 62);
                            leafp = pack_leaf(lx, ly, lz);
                        }
                        // position inside the voxel in 2^-30 voxel units (exact integer sums => reproducible means)
                        qx = __double2ll_rn(((double)fx - (double)gx * P.leaf_d) * P.fix_scale);
                        qy = __double2ll_rn(((double)fy - (double)gy * P.leaf_d) * P.fix_scale);
                        qz = __double2ll_rn(((double)fz - (double)gz * P.leaf_d) * P.fix_scale);
                    }
                }
            }
            if (key != cur.key) {
                if (cur.key) lds_insert(L, T, P.table_mask, cur);
                cur.key = key; cur.leaf = leafp;
                cur.qx = cur.qy = cur.qz = 0; cur.cr = cur.gb = 0; cur.tile = 0;
            }
            if (key) {
                cur.qx += qx; cur.qy += qy; cur.qz += qz;
                cur.cr += (1ull << 32) | (unsigned long long)(w & 0xffu);
                cur.gb += ((unsigned long long)((w >> 8) & 0xffu) << 32) | (unsigned long long)((w >> 16) & 0xffu);
                cur.tile |= w >> 24;
            }
        }
    }

    // workgroup bounding box of this chunk (input of the octree replay)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; a++) {
        float lo = bmin[a], hi = bmax[a];
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_down(lo, off, 64));
            hi = fmaxf(hi, __shfl_down(hi, off, 64));
        }
        if (lane == 0) { red[a][wave] = lo; red[3 + a][wave] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < ACC_BLOCK / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        chunk_bbox[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }

    // flush the workgroup table
    for (int i = threadIdx.x; i < LDS_SLOTS; i += ACC_BLOCK) {
        if (L.keys[i]) {
            Run r;
            r.key = L.keys[i]; r.leaf = L.leaf[i];
            r.qx = (long long)L.sx[i]; r.qy = (long long)L.sy[i]; r.qz = (long long)L.sz[i];
            r.cr = L.cr[i]; r.gb = L.gb[i]; r.tile = L.tile[i];
            global_insert(T, P.table_mask, r);
        }
    }
}

// ---------------------------------------------------------------------------
// K2: octree bounding-box replay / global grid box
// ---------------------------------------------------------------------------
// Growth of pcl::octree::OctreePointCloud's box is sequential in input order,
// but a chunk whose box lies inside the current octree box cannot trigger a
// growth step, so only the few chunks that do are re-read point by point.
__global__ void __launch_bounds__(1024) octree_replay_kernel(VoxParams P, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, const float *__restrict__ chunk_bbox, uint32_t nchunks,
                                                            uint32_t *__restrict__ ctrl) {
    __shared__ double s_mn[3], s_mx[3];
    __shared__ int s_depth;
    __shared__ long long s_shift[3];
    __shared__ unsigned s_first;
    __shared__ int s_events;
    const int tid = threadIdx.x;

    if (!P.leaf_split) {
        // plain pcl::VoxelGrid: getMinMax3D, the 2^31-cell check, min_b / div_b
        __shared__ float s_red[6][16];
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t c = tid; c < nchunks; c += 1024) {
            for (int a = 0; a < 3; a++) {
                lo[a] = fminf(lo[a], chunk_bbox[(size_t)c * 6 + a]);
                hi[a] = fmaxf(hi[a], chunk_bbox[(size_t)c * 6 + 3 + a]);
            }
        }
        for (int a = 0; a < 3; a++) {
            for (int off = 32; off > 0; off >>= 1) {
                lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
                hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
            }
            if ((tid & 63) == 0) { s_red[a][tid >> 6] = lo[a]; s_red[3 + a][tid >> 6] = hi[a]; }
        }
        __syncthreads();
        if (tid == 0) {
            float mn[3], mx[3];
            for (int a = 0; a < 3; a++) {
                mn[a] = s_red[a][0]; mx[a] = s_red[3 + a][0];
                for (int w = 1; w < 16; w++) { mn[a] = fminf(mn[a], s_red[a][w]); mx[a] = fmaxf(mx[a], s_red[3 + a][w]); }
            }
            long long d[3];
            int minb[3], divb[3];
            for (int a = 0; a < 3; a++) {
                d[a] = (long long)(__fmul_rn(__fsub_rn(mx[a], mn[a]), P.inv_leaf)) + 1;
                minb[a] = (int)floorf(__fmul_rn(mn[a], P.inv_leaf));
                int maxb = (int)floorf(__fmul_rn(mx[a], P.inv_leaf));
                divb[a] = maxb - minb[a] + 1;
            }
            if (d[0] * d[1] * d[2] > (long long)INT32_MAX) atomicOr(&ctrl[C_ERR], ERR_GRID_OVERFLOW);
            for (int a = 0; a < 3; a++) { ctrl[C_MINB + a] = (uint32_t)minb[a]; ctrl[C_DIVB + a] = (uint32_t)divb[a]; }
        }
        return;
    }

    if (tid == 0) {
        double pp[3] = {(double)x[0], (double)y[0], (double)z[0]};
        double mn[3], mx[3];
        int d;
        first_box(pp, P.res, mn, mx, d);
        for (int a = 0; a < 3; a++) { s_mn[a] = mn[a]; s_mx[a] = mx[a]; s_shift[a] = 0; }
        s_depth = d;
        s_events = 0;
    }
    __syncthreads();

    const double eps = (double)FLT_EPSILON;
    uint32_t chunk = 0;
    while (chunk < nchunks) {
        // first chunk >= chunk whose box violates the current octree box
        if (tid == 0) s_first = 0xffffffffu;
        __syncthreads();
        {
            const double mn0 = s_mn[0], mn1 = s_mn[1], mn2 = s_mn[2], mx0 = s_mx[0], mx1 = s_mx[1], mx2 = s_mx[2];
            for (uint32_t c = chunk + tid; c < nchunks; c += 1024) {
                const float *b = chunk_bbox + (size_t)c * 6;
                bool viol = (double)b[0] < mn0 || (double)b[1] < mn1 || (double)b[2] < mn2 ||
                            (double)b[3] >= mx0 || (double)b[4] >= mx1 || (double)b[5] >= mx2;
                if (viol) { atomicMin(&s_first, c); break; }
            }
        }
        __syncthreads();
        const uint32_t hit = s_first;
        __syncthreads();
        if (hit == 0xffffffffu) break;

        // re-read that chunk (4 points per lane) and replay its violations in index order
        const size_t base = (size_t)hit * ACC_CHUNK + (size_t)tid * 4;
        float qx[4], qy[4], qz[4];
        for (int j = 0; j < 4; j++) {
            bool ok = base + j < P.n;
            qx[j] = ok ? x[base + j] : 0.f;
            qy[j] = ok ? y[base + j] : 0.f;
            qz[j] = ok ? z[base + j] : 0.f;
        }
        uint32_t from = 0;   // index inside the chunk from which violations are still unprocessed
        for (;;) {
            if (tid == 0) s_first = 0xffffffffu;
            __syncthreads();
            {
                const double mn0 = s_mn[0], mn1 = s_mn[1], mn2 = s_mn[2], mx0 = s_mx[0], mx1 = s_mx[1], mx2 = s_mx[2];
                for (int j = 0; j < 4; j++) {
                    uint32_t idx = (uint32_t)tid * 4 + j;
                    if (idx < from || base + j >= P.n) continue;
                    if (!(isfinite(qx[j]) && isfinite(qy[j]) && isfinite(qz[j]))) continue;
                    bool viol = (double)qx[j] < mn0 || (double)qy[j] < mn1 || (double)qz[j] < mn2 ||
                                (double)qx[j] >= mx0 || (double)qy[j] >= mx1 || (double)qz[j] >= mx2;
                    if (viol) { atomicMin(&s_first, idx); break; }
                }
            }
            __syncthreads();
            const uint32_t pidx = s_first;
            __syncthreads();
            if (pidx == 0xffffffffu) break;
            if (tid == (int)(pidx >> 2)) {
                // adoptBoundingBoxToPoint for this point: grow until it fits
                const double c[3] = {(double)qx[pidx & 3], (double)qy[pidx & 3], (double)qz[pidx & 3]};
                for (;;) {
                    bool up[3], any = false;
                    for (int a = 0; a < 3; a++) {
                        bool lo = c[a] < s_mn[a];
                        up[a] = c[a] >= s_mx[a];
                        any |= lo | up[a];
                    }
                    if (!any) break;
                    if (s_depth >= 31) { atomicOr(&ctrl[C_ERR], ERR_DEPTH); break; }
                    double side = (double)(1u << s_depth) * P.res;
                    for (int a = 0; a < 3; a++) {
                        if (!up[a]) {
                            s_mn[a] -= side;
                            s_shift[a] += (long long)1 << s_depth;   // existing keys move up on this axis
                        }
                    }
                    s_depth++;
                    side = (double)(1u << s_depth) * P.res - eps;
                    for (int a = 0; a < 3; a++) s_mx[a] = s_mn[a] + side;
                    s_events++;
                }
            }
            from = pidx + 1;
            __syncthreads();
        }
        chunk = hit + 1;
    }
    if (tid == 0) {
        ctrl[C_DEPTH] = (uint32_t)s_depth;
        ctrl[C_EVENTS] = (uint32_t)s_events;
        for (int a = 0; a < 3; a++) {
            ctrl[C_SHIFT + 2 * a] = (uint32_t)((unsigned long long)s_shift[a] & 0xffffffffu);
            ctrl[C_SHIFT + 2 * a + 1] = (uint32_t)((unsigned long long)s_shift[a] >> 32);
        }
    }
}

// ---------------------------------------------------------------------------
// K3: output-order keys
// ---------------------------------------------------------------------------
// positive cellsize: [ Morton code of the leaf's final octree key | k | j | i inside the leaf ]
// negative cellsize: pcl::VoxelGrid's idx = i + j*div_x + k*div_x*div_y
__global__ void __launch_bounds__(256) make_sort_keys_kernel(VoxParams P, VoxTable T, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t m, unsigned long long *__restrict__ sort_keys,
                                                            uint32_t *__restrict__ sort_vals) {
    uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t slot = T.occupied[r];
    const unsigned long long key = T.keys[slot];
    const int i0 = (int)floorf(x[0] * P.inv_leaf), j0 = (int)floorf(y[0] * P.inv_leaf), k0 = (int)floorf(z[0] * P.inv_leaf);
    const int vi = (int)(key & 0xfffff) - COORD_BIAS + i0;
    const int vj = (int)((key >> COORD_BITS) & 0xfffff) - COORD_BIAS + j0;
    const int vk = (int)((key >> (2 * COORD_BITS)) & 0xfffff) - COORD_BIAS + k0;
    unsigned long long sk;
    if (!P.leaf_split) {
        const long long di = vi - (int)T.ctrl[C_MINB], dj = vj - (int)T.ctrl[C_MINB + 1], dk = vk - (int)T.ctrl[C_MINB + 2];
        const long long dx = (int)T.ctrl[C_DIVB], dy = (int)T.ctrl[C_DIVB + 1];
        sk = (unsigned long long)(di + dj * dx + dk * dx * dy);
    } else {
        const int depth = (int)T.ctrl[C_DEPTH];
        const unsigned long long lp = T.leaf[slot];
        long long lk[3];
        int cell[3];
        const int v[3] = {vi, vj, vk};
        double pp[3] = {(double)x[0], (double)y[0], (double)z[0]}, mn0[3], mx0[3];
        int d0;
        first_box(pp, P.res, mn0, mx0, d0);
        bool bad = depth > 14;
        for (int a = 0; a < 3; a++) {
            const int lrel = unpack_leaf(lp, a);
            const long long shift = (long long)(((unsigned long long)T.ctrl[C_SHIFT + 2 * a + 1] << 32) | T.ctrl[C_SHIFT + 2 * a]);
            lk[a] = (long long)lrel + shift;
            if (lk[a] < 0 || lk[a] >= ((long long)1 << depth)) bad = true;
            // voxel index inside the leaf, relative to a base one cell below the leaf's lower face
            const int base = (int)floor(mn0[a] * (double)P.inv_leaf) - 1 + 64 * lrel;
            cell[a] = v[a] - base;
            if (cell[a] < 0 || cell[a] > 127) bad = true;
        }
        if (bad) {
            atomicOr(&T.ctrl[C_ERR], depth > 14 ? ERR_DEPTH : ERR_LEAF_RANGE);
            sk = ~0ull;
        } else {
            unsigned long long morton = 0;
            for (int b = depth - 1; b >= 0; b--) {
                morton = (morton << 3) | (((unsigned long long)(lk[0] >> b) & 1) << 2) | (((unsigned long long)(lk[1] >> b) & 1) << 1) |
                         ((unsigned long long)(lk[2] >> b) & 1);
            }
            sk = (morton << 21) | ((unsigned long long)cell[2] << 14) | ((unsigned long long)cell[1] << 7) | (unsigned long long)cell[0];
        }
    }
    sort_keys[r] = sk;
    sort_vals[r] = slot;
}

// ---------------------------------------------------------------------------
// K4: emit in output order and clean the table
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) emit_and_clean_kernel(VoxParams P, VoxTable T, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t m, const uint32_t *__restrict__ sorted_slots,
                                                            float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
                                                            uint32_t *__restrict__ ow, int emit) {
    uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t slot = emit ? sorted_slots[r] : T.occupied[r];
    if (emit) {
        const unsigned long long key = T.keys[slot];
        const int i0 = (int)floorf(x[0] * P.inv_leaf), j0 = (int)floorf(y[0] * P.inv_leaf), k0 = (int)floorf(z[0] * P.inv_leaf);
        const double vi = (double)((int)(key & 0xfffff) - COORD_BIAS + i0);
        const double vj = (double)((int)((key >> COORD_BITS) & 0xfffff) - COORD_BIAS + j0);
        const double vk = (double)((int)((key >> (2 * COORD_BITS)) & 0xfffff) - COORD_BIAS + k0);
        const unsigned long long cr = T.cr[slot], gb = T.gb[slot];
        const uint32_t cnt = (uint32_t)(cr >> 32);
        const double n = (double)cnt;
        // mean = voxel origin + mean offset; one rounding to fp32 at the end
        ox[r] = (float)(vi * P.leaf_d + ((double)(long long)T.sx[slot] / n) / P.fix_scale);
        oy[r] = (float)(vj * P.leaf_d + ((double)(long long)T.sy[slot] / n) / P.fix_scale);
        oz[r] = (float)(vk * P.leaf_d + ((double)(long long)T.sz[slot] / n) / P.fix_scale);
        // pcl AccumulatorRGBA: float sums (exact integers here) / n, truncated
        const float fn = (float)cnt;
        const uint32_t rr = (uint32_t)__fdiv_rn((float)(uint32_t)(cr & 0xffffffffu), fn);
        const uint32_t gg = (uint32_t)__fdiv_rn((float)(uint32_t)(gb >> 32), fn);
        const uint32_t bb = (uint32_t)__fdiv_rn((float)(uint32_t)(gb & 0xffffffffu), fn);
        ow[r] = (rr & 0xffu) | ((gg & 0xffu) << 8) | ((bb & 0xffu) << 16) | ((T.tile[slot] & 0xffu) << 24);
    }
    T.keys[slot] = 0; T.sx[slot] = 0; T.sy[slot] = 0; T.sz[slot] = 0; T.cr[slot] = 0; T.gb[slot] = 0; T.tile[slot] = 0; T.leaf[slot] = 0;
}

// ---------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------
struct Workspace {
    int device = -1;
    size_t slots = 0;      // table capacity (power of two)
    size_t list_cap = 0;   // occupied-list capacity
    size_t chunk_cap = 0;
    void *table_mem = nullptr;
    uint32_t *occupied = nullptr;
    float *chunk_bbox = nullptr;
    uint32_t *ctrl = nullptr;
    VoxTable view{};
    ~Workspace() {
        // process teardown: the runtime may be gone, ignore errors
        if (table_mem) (void)hipFree(table_mem);
        if (occupied) (void)hipFree(occupied);
        if (chunk_bbox) (void)hipFree(chunk_bbox);
        if (ctrl) (void)hipFree(ctrl);
    }
};

thread_local Workspace t_ws;

// Table bytes per slot: keys, leaf, sx, sy, sz, cr, gb (7 x 8) + tile (4)
constexpr size_t SLOT_BYTES = 7 * 8 + 4;

bool ensure_workspace(Workspace &ws, size_t n, hipStream_t s) {
    int dev = current_device();
    size_t want = 1 << 16;
    while (want < 2 * n) want <<= 1;
    if (ws.device != dev || ws.slots < want) {
        if (ws.table_mem) (void)hipFree(ws.table_mem);
        ws.table_mem = nullptr;
        ws.slots = 0;
        CW_HIP_TRY(hipMalloc(&ws.table_mem, want * SLOT_BYTES));
        CW_HIP_TRY(hipMemsetAsync(ws.table_mem, 0, want * SLOT_BYTES, s));
        ws.slots = want;
        char *p = (char *)ws.table_mem;
        ws.view.keys = (unsigned long long *)p; p += want * 8;
        ws.view.leaf = (unsigned long long *)p; p += want * 8;
        ws.view.sx = (unsigned long long *)p; p += want * 8;
        ws.view.sy = (unsigned long long *)p; p += want * 8;
        ws.view.sz = (unsigned long long *)p; p += want * 8;
        ws.view.cr = (unsigned long long *)p; p += want * 8;
        ws.view.gb = (unsigned long long *)p; p += want * 8;
        ws.view.tile = (uint32_t *)p;
    }
    if (ws.device != dev || ws.list_cap < n) {
        if (ws.occupied) (void)hipFree(ws.occupied);
        ws.occupied = nullptr;
        ws.list_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.occupied, n * sizeof(uint32_t)));
        ws.list_cap = n;
    }
    size_t nchunks = (n + ACC_CHUNK - 1) / ACC_CHUNK;
    if (ws.device != dev || ws.chunk_cap < nchunks) {
        if (ws.chunk_bbox) (void)hipFree(ws.chunk_bbox);
        ws.chunk_bbox = nullptr;
        ws.chunk_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.chunk_bbox, nchunks * 6 * sizeof(float)));
        ws.chunk_cap = nchunks;
    }
    if (ws.device != dev || !ws.ctrl) {
        if (ws.ctrl) (void)hipFree(ws.ctrl);
        ws.ctrl = nullptr;
        CW_HIP_TRY(hipMalloc((void **)&ws.ctrl, C_WORDS * sizeof(uint32_t)));
    }
    ws.device = dev;
    ws.view.occupied = ws.occupied;
    ws.view.ctrl = ws.ctrl;
    return true;
}

}  // namespace

std::shared_ptr<DeviceSoA> voxel_downsample(const DeviceSoA &src, float cellsize, bool leaf_split, int *error_code) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    const size_t n = src.npoints;
    Workspace &ws = t_ws;
    if (!ensure_workspace(ws, n, c.stream)) return nullptr;

    VoxParams P;
    P.n = n;
    P.leaf = cellsize;
    P.inv_leaf = 1.0f / cellsize;
    P.leaf_d = (double)cellsize;
    P.fix_scale = FIX_ONE / (double)cellsize;
    float octree_cellsize = (8 * 8) * cellsize;   // reference src/cwipc_filters.cpp:113-114
    P.res = (double)octree_cellsize;
    P.leaf_split = leaf_split ? 1 : 0;
    P.table_mask = (uint32_t)(ws.slots - 1);

    const uint32_t nchunks = (uint32_t)((n + ACC_CHUNK - 1) / ACC_CHUNK);
    bool ok = hipMemsetAsync(ws.ctrl, 0, C_WORDS * sizeof(uint32_t), c.stream) == hipSuccess;
    if (!ok) { hip_failed(hipGetLastError(), "hipMemsetAsync(ctrl)", __FILE__, __LINE__); return nullptr; }

    CW_LAUNCH("voxel_accumulate", voxel_accumulate_kernel, dim3(nchunks), dim3(ACC_BLOCK), 0, c.stream, P, src.x(), src.y(), src.z(), src.rgbt(),
              ws.view, ws.chunk_bbox);
    CW_LAUNCH("octree_replay", octree_replay_kernel, dim3(1), dim3(1024), 0, c.stream, P, src.x(), src.y(), src.z(), ws.chunk_bbox, nchunks,
              ws.ctrl);
    ok = hipMemcpyAsync(c.host_words, ws.ctrl, C_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    ok = c.sync() && ok;
    if (!ok) { hip_failed(hipGetLastError(), "voxel_accumulate", __FILE__, __LINE__); return nullptr; }

    uint32_t err = c.host_words[C_ERR];
    const uint32_t m = c.host_words[C_COUNT];
    std::shared_ptr<DeviceSoA> dst;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    uint32_t *vals_in = nullptr, *vals_out = nullptr;
    void *sort_tmp = nullptr;
    const unsigned mgrid = (m + 255) / 256;

    if (!err && m) {
        dst = soa_alloc(m);
        keys_in = (unsigned long long *)pool_alloc((size_t)m * 8 * 2);
        vals_in = (uint32_t *)pool_alloc((size_t)m * 4 * 2);
        if (!dst || !keys_in || !vals_in) {
            err |= 0x80000000u;
        } else {
            keys_out = keys_in + m;
            vals_out = vals_in + m;
            CW_LAUNCH("make_sort_keys", make_sort_keys_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, ws.view, src.x(), src.y(), src.z(), m,
                      keys_in, vals_in);
            size_t tmp_bytes = 0;
            hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0u, 64u, c.stream);
            if (e == hipSuccess) {
                sort_tmp = pool_alloc(tmp_bytes ? tmp_bytes : 256);
                if (!sort_tmp) e = hipErrorOutOfMemory;
            }
            if (e == hipSuccess) {
                if (profiling_enabled()) profile_begin("radix_sort_pairs", c.stream);
                e = rocprim::radix_sort_pairs(sort_tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0u, 64u, c.stream);
                if (profiling_enabled()) profile_end(c.stream);
            }
            if (e != hipSuccess) {
                hip_failed(e, "rocprim::radix_sort_pairs", __FILE__, __LINE__);
                err |= 0x80000000u;
            }
        }
    }
    if (m) {
        // emit (or, on error, only clean) -- the table must be left zeroed either way
        int emit = (!err && dst) ? 1 : 0;
        CW_LAUNCH("emit_and_clean", emit_and_clean_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, ws.view, src.x(), src.y(), src.z(), m,
                  vals_out, emit ? dst->x() : nullptr, emit ? dst->y() : nullptr, emit ? dst->z() : nullptr, emit ? dst->rgbt() : nullptr, emit);
        if (emit) {
            // sort-key construction may have raised an error flag
            ok = hipMemcpyAsync(c.host_words, ws.ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
        }
        ok = c.sync() && ok;
        if (emit && ok) err |= c.host_words[C_ERR];
    }
    pool_free(keys_in);
    pool_free(vals_in);
    pool_free(sort_tmp);
    if (error_code) *error_code = (int)err;
    if (!ok) { hip_failed(hipGetLastError(), "voxel emit", __FILE__, __LINE__); return nullptr; }
    if (err) {
        std::string why;
        if (err & ERR_GRID_OVERFLOW) why += " VoxelGrid: leaf size is too small for the input dataset, integer indices would overflow;";
        if (err & ERR_RANGE) why += " a voxel lies more than 2^19 cells from the first point;";
        if (err & ERR_TABLE_FULL) why += " voxel table full;";
        if (err & (ERR_DEPTH | ERR_LEAF_RANGE)) why += " octree deeper than 14 levels;";
        if (err & ERR_CELL_RANGE) why += " voxel outside its leaf grid;";
        if (err & 0x80000000u) why += " device allocation or sort failure;";
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed:" + why);
        return nullptr;
    }
    if (!m) {
        // only non-finite points: the reference's VoxelGrid path reports an empty result
        if (!leaf_split) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "VoxelGrid filter produced empty pointcloud");
            return nullptr;
        }
        return soa_alloc(0);
    }
    return dst;
}

}  // namespace cwipc_amd
