// kernels_voxel.hip -- voxel-grid downsample on gfx950 (MI355X).
//
// Reference: cwipc_downsample / cwipc_downsample_voxelgrid, src/cwipc_filters.cpp:30-172.
// The arithmetic the reference delegates to PCL is restated from the published
// upstream algorithms (scalar restatement: oracle/cwipc_oracle.c):
//   pcl::VoxelGrid            voxel (i,j,k) = floor(p * (1/leaf)) with an fp32 product; one
//                             output per occupied voxel = mean xyz, truncated mean rgb;
//                             outputs in ascending (k,j,i); grids above 2^31 cells refused.
//   pcl::octree::OctreePointCloud (positive cellsize only) leaves of side R = 64*leaf on a
//                             lattice anchored at the first point; every leaf is voxelised on
//                             its own, so a voxel cut by a leaf face yields one output per
//                             side; leaves are emitted in depth-first (Morton) order of their
//                             final octree keys, which depend on how the bounding box grew
//                             while the points were inserted in input order.
//   tile of an output         OR of the tiles of its contributors (src/cwipc_filters.cpp:64-74).
//
// Design for MI355X.  The job is HBM-bound integer work: 16 B per input point must be read
// once (160 MB at the 10 M-point configuration, ~29.5 us at the 5.4 TB/s a pure read of these
// planes reaches on this part), everything else has to hide behind that stream.  Measured
// constraints that shaped the kernel (scratch/ubench*.hip):
//   * scattered global integer atomics retire at ~23 G requests/s chip-wide whatever their
//     scope, but 8 lanes updating one 64-byte record cost ~1.3 requests (17 G records/s);
//   * an LDS atomic costs 13-20 cycles per WAVE INSTRUCTION almost independent of the
//     number of active lanes, so instructions have to be saved, not lanes.
// Hence:
//   K1 voxel_accumulate   persistent: one 1024-lane workgroup per CU, every wave streams a
//                         contiguous range of the planes (dwordx4 per plane per lane).  Lanes
//                         merge their own 4 points, DPP row-shifts merge runs across lanes,
//                         and only run ends touch the workgroup's LDS hash table (64-bit
//                         integer LDS atomics).  The table is flushed ONCE per workgroup into
//                         dense per-leaf grids of 64-byte records, 8 lanes per record, with
//                         returning 64-bit adds (the returned count tells the first toucher,
//                         which appends the cell to the occupied list).  All sums are
//                         fixed-point integers: results are bitwise reproducible.
//                         Each wave also emits the bounding box of its range.
//   K2 octree_replay      one workgroup replays the octree's bounding-box growth over the wave
//                         boxes, re-reading only the ranges that trigger a growth step
//                         (plain grid: reduces the boxes to the global one).
//   K3 make_sort_keys     one lane per occupied voxel: 64-bit output-order key.
//   (radix sort of the ~40 k keys)
//   K4 emit_and_clean     gathers the records in output order, writes the planes and zeroes
//                         what it read, so the workspace is clean for the next call.
#include "internal.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cfloat>
#include <cmath>

namespace cwipc_amd {

namespace {

// ---------------------------------------------------------------------------
// constants and shared structures
// ---------------------------------------------------------------------------
constexpr int K1_THREADS = 1024;
constexpr int K1_WAVES = K1_THREADS / 64;
constexpr int WAVE_STEP = 256;                 // points per wave per step (4 per lane)
constexpr int LTAB = 2048;                     // LDS table entries per workgroup
constexpr int LTAB_PROBES = 32;
constexpr int GRID_DIM = 68;                   // cells per axis of a leaf grid (64 + slack for fp rounding)
constexpr int CELLS = GRID_DIM * GRID_DIM * GRID_DIM;   // 314432 < 2^19
constexpr int CELL_BITS = 19;
constexpr uint32_t KEY_EMPTY = 0xffffffffu;
constexpr float FIX_ONE_F = 4194304.0f;        // 2^22 fixed-point units per voxel edge
constexpr int RECORD_WORDS = 8;                // 64-byte records: sx sy sz cr gb tlo thi tor

enum : uint32_t {
    ERR_RANGE = 1,           // voxel index outside +-2^26, or leaf index outside +-2^20
    ERR_LEAVES = 2,          // more octree leaves than the workspace has grids for (host regrows and retries)
    ERR_DEPTH = 4,           // octree deeper than the sort key can express
    ERR_GRID_OVERFLOW = 8,   // pcl::VoxelGrid: "Leaf size is too small ... indices would overflow"
    ERR_LEAF_RANGE = 16,
    ERR_FIRST_POINT = 32,    // the first point is not finite
    ERR_CELL_RANGE = 64,
    ERR_LIST_FULL = 128,
};

// control block, 32-bit words in device memory
enum { C_ERR = 0, C_COUNT = 1, C_DEPTH = 2, C_EVENTS = 3, C_SHIFT = 4 /* 3 x int64 */, C_MINB = 10, C_DIVB = 13, C_WORDS = 32 };

struct VoxParams {
    size_t n;
    size_t per_wave;        // points per wave range (multiple of WAVE_STEP)
    uint32_t nranges;       // number of wave ranges = waves in the K1 grid
    float inv_leaf;         // 1 / leaf in fp32, as pcl::VoxelGrid::setLeafSize
    float leaf;
    float fix_scale;        // 2^22 / leaf (fp32)
    double leaf_d;
    double res;             // octree resolution (double)(float)(64 * leaf)
    int leaf_split;
    uint32_t leaf_mask;     // leaf hash capacity - 1
    uint32_t list_cap;
};

struct VoxWork {
    unsigned long long *leaf_keys;   // [leaf hash] 0 = empty, else packed lattice coordinates | 1<<63 ; position = leaf id
    unsigned long long *records;     // [leaf hash][CELLS][8]
    uint32_t *occupied;              // list of (leaf id << 19 | cell) of touched records
    uint32_t *ctrl;
    float *bboxes;                   // [nranges][6]
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

// Leaf lattice anchor shared by K1, K3 and K4: mn0 = lower corner of the first octree box,
// ib = floor(mn0 / leaf).  Cell c of leaf l on an axis is voxel  c + ib + 64*l - 2.
struct Anchor {
    double mn0[3];
    int ib[3];
};

__device__ __forceinline__ Anchor make_anchor(const VoxParams &P, float p0x, float p0y, float p0z) {
    Anchor A;
    if (P.leaf_split) {
        double pp[3] = {(double)p0x, (double)p0y, (double)p0z}, mx0[3];
        int d0;
        first_box(pp, P.res, A.mn0, mx0, d0);
        for (int a = 0; a < 3; a++) A.ib[a] = (int)floor(A.mn0[a] / P.leaf_d);
    } else {
        // plain grid: "leaves" are bricks of 64^3 voxels aligned to the voxel lattice
        for (int a = 0; a < 3; a++) { A.mn0[a] = 0; A.ib[a] = 2; }
    }
    return A;
}

__device__ __forceinline__ unsigned long long pack_leaf(int lx, int ly, int lz) {
    return (1ull << 63) | ((unsigned long long)(uint32_t)(lx & 0x1fffff)) | ((unsigned long long)(uint32_t)(ly & 0x1fffff) << 21) |
           ((unsigned long long)(uint32_t)(lz & 0x1fffff) << 42);
}
__device__ __forceinline__ int unpack_leaf(unsigned long long v, int axis) {
    int t = (int)((v >> (21 * axis)) & 0x1fffff);
    return (t << 11) >> 11;   // sign-extend 21 bits
}

// One run of points of the same voxel, 32-bit in-wave form (at most 256 points).
struct Run32 {
    uint32_t key;       // leaf id << 19 | cell, KEY_EMPTY = none
    int qx, qy, qz;     // fixed-point offsets inside the voxel
    uint32_t cr;        // count << 16 | sum r
    uint32_t gb;        // sum g << 16 | sum b
    uint32_t tile;
};

// ---------------------------------------------------------------------------
// global side: leaf lookup, record updates
// ---------------------------------------------------------------------------
// Wave-uniform: returns the id (hash position) of leaf key k, inserting it if new.
__device__ __forceinline__ uint32_t leaf_lookup(const VoxWork &W, uint32_t mask, unsigned long long k) {
    uint32_t pos = (uint32_t)mix64(k) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {
        unsigned long long cur = __hip_atomic_load(&W.leaf_keys[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == k) return pos;
        if (cur == 0ull) {
            unsigned long long old = atomicCAS(&W.leaf_keys[pos], 0ull, k);
            if (old == 0ull || old == k) return pos;
        }
        pos = (pos + 1) & mask;
    }
    atomicOr(&W.ctrl[C_ERR], ERR_LEAVES);
    return 0xffffffffu;
}

__device__ __forceinline__ void note_first_touch(const VoxWork &W, const VoxParams &P, uint32_t key) {
    uint32_t idx = atomicAdd(&W.ctrl[C_COUNT], 1u);
    if (idx < P.list_cap) W.occupied[idx] = key;
    else atomicOr(&W.ctrl[C_ERR], ERR_LIST_FULL);
}

// Record of voxel key = leaf id << 19 | cell  (grids are CELLS records apart, not 2^19).
__device__ __forceinline__ unsigned long long *record_ptr(const VoxWork &W, uint32_t key) {
    return W.records + ((size_t)(key >> CELL_BITS) * CELLS + (key & ((1u << CELL_BITS) - 1))) * RECORD_WORDS;
}

// Slow path (workgroup table saturated): one lane updates a whole record.
__device__ __forceinline__ void global_insert_lane(const VoxWork &W, const VoxParams &P, uint32_t key, long long sx, long long sy, long long sz,
                                                   unsigned long long cr, unsigned long long gb, uint32_t tile) {
    unsigned long long *rec = record_ptr(W, key);
    atomicAdd(&rec[0], (unsigned long long)sx);
    atomicAdd(&rec[1], (unsigned long long)sy);
    atomicAdd(&rec[2], (unsigned long long)sz);
    unsigned long long old = atomicAdd(&rec[3], cr);
    atomicAdd(&rec[4], gb);
    atomicOr(&rec[7], (unsigned long long)tile);
    if ((old >> 32) == 0) note_first_touch(W, P, key);
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
struct LdsTable {
    uint32_t key[LTAB];
    uint32_t tile[LTAB];
    unsigned long long sx[LTAB], sy[LTAB], sz[LTAB], cr[LTAB], gb[LTAB];
    uint32_t fresh[LTAB];      // records this workgroup touched first (appended to the occupied list in one go)
    uint32_t nfresh, fresh_base;
};

__device__ __forceinline__ void lds_insert(LdsTable &L, const VoxWork &W, const VoxParams &P, const Run32 &r) {
    uint32_t slot = (r.key * 0x9E3779B1u) >> (32 - 11);   // LTAB = 2^11
    const long long sx = r.qx, sy = r.qy, sz = r.qz;
    const unsigned long long cr = ((unsigned long long)(r.cr >> 16) << 32) | (r.cr & 0xffffu);
    const unsigned long long gb = ((unsigned long long)(r.gb >> 16) << 32) | (r.gb & 0xffffu);
    for (int probe = 0; probe < LTAB_PROBES; probe++) {
        uint32_t old = atomicCAS(&L.key[slot], KEY_EMPTY, r.key);
        if (old == KEY_EMPTY || old == r.key) {
            atomicAdd(&L.sx[slot], (unsigned long long)sx);
            atomicAdd(&L.sy[slot], (unsigned long long)sy);
            atomicAdd(&L.sz[slot], (unsigned long long)sz);
            atomicAdd(&L.cr[slot], cr);
            atomicAdd(&L.gb[slot], gb);
            atomicOr(&L.tile[slot], r.tile);
            return;
        }
        slot = (slot + 1) & (LTAB - 1);
    }
    global_insert_lane(W, P, r.key, sx, sy, sz, cr, gb, r.tile);
}

// DPP row shift right by N lanes inside rows of 16 (lanes whose source is outside the row read 0).
template <int N>
__device__ __forceinline__ int dpp_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ int dpp_shl(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xf, 0xf, true);
}

template <int N>
__device__ __forceinline__ void scan_step(Run32 &v, int &flag) {
    const int pqx = dpp_shr<N>(v.qx), pqy = dpp_shr<N>(v.qy), pqz = dpp_shr<N>(v.qz);
    const int pcr = dpp_shr<N>((int)v.cr), pgb = dpp_shr<N>((int)v.gb), pt = dpp_shr<N>((int)v.tile);
    const int pf = dpp_shr<N>(flag);
    if (!flag) {
        v.qx += pqx; v.qy += pqy; v.qz += pqz;
        v.cr += (uint32_t)pcr; v.gb += (uint32_t)pgb; v.tile |= (uint32_t)pt;
    }
    flag |= pf;
}

__global__ void __launch_bounds__(K1_THREADS) voxel_accumulate_kernel(VoxParams P, const float *__restrict__ x, const float *__restrict__ y,
                                                                     const float *__restrict__ z, const uint32_t *__restrict__ rgbt, VoxWork W) {
    extern __shared__ __align__(16) unsigned char k1_smem[];
    LdsTable &L = *reinterpret_cast<LdsTable *>(k1_smem);

    for (int i = threadIdx.x; i < LTAB; i += K1_THREADS) {
        L.key[i] = KEY_EMPTY; L.tile[i] = 0;
        L.sx[i] = 0; L.sy[i] = 0; L.sz[i] = 0; L.cr[i] = 0; L.gb[i] = 0;
    }
    if (threadIdx.x == 0) L.nfresh = 0;

    const int lane = threadIdx.x & 63;
    const uint32_t range = blockIdx.x * K1_WAVES + (threadIdx.x >> 6);

    // Anchor: the first point (octree lattice phase).
    const float p0x = x[0], p0y = y[0], p0z = z[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(isfinite(p0x) && isfinite(p0y) && isfinite(p0z))) atomicOr(&W.ctrl[C_ERR], ERR_FIRST_POINT);
    const Anchor A = make_anchor(P, p0x, p0y, p0z);
    __syncthreads();

    float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const size_t lo = (size_t)range * P.per_wave;
    const size_t hi = lo + P.per_wave < P.n ? lo + P.per_wave : P.n;

    // wave-uniform leaf cache
    unsigned long long cache_key = 0;
    uint32_t cache_id = 0xffffffffu;

#pragma unroll 1
    for (size_t base = lo; base < hi; base += WAVE_STEP) {
        const size_t p = base + (size_t)lane * 4;
        float px[4], py[4], pz[4];
        uint32_t pw[4];
        int cnt = 0;
        if (p + 4 <= hi) {
            const float4 a = *(const float4 *)(x + p), b = *(const float4 *)(y + p), c = *(const float4 *)(z + p);
            const uint4 w = *(const uint4 *)(rgbt + p);
            px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w;
            py[0] = b.x; py[1] = b.y; py[2] = b.z; py[3] = b.w;
            pz[0] = c.x; pz[1] = c.y; pz[2] = c.z; pz[3] = c.w;
            pw[0] = w.x; pw[1] = w.y; pw[2] = w.z; pw[3] = w.w;
            cnt = 4;
        } else {
            cnt = p < hi ? (int)(hi - p) : 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool ok = j < cnt;
                px[j] = ok ? x[p + j] : 0.f;
                py[j] = ok ? y[p + j] : 0.f;
                pz[j] = ok ? z[p + j] : 0.f;
                pw[j] = ok ? rgbt[p + j] : 0u;
            }
        }

        // ---- per point: cell inside its leaf, leaf lattice coordinates, fixed-point offsets ----
        uint32_t key[4];
        unsigned long long lkey[4];
        int qx[4], qy[4], qz[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            key[j] = KEY_EMPTY;
            lkey[j] = 0;
            qx[j] = qy[j] = qz[j] = 0;
            const float f[3] = {px[j], py[j], pz[j]};
            if (j < cnt && isfinite(f[0]) && isfinite(f[1]) && isfinite(f[2])) {
                int cell[3], leaf[3], q[3];
                bool bad = false;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    bmin[a] = fminf(bmin[a], f[a]);
                    bmax[a] = fmaxf(bmax[a], f[a]);
                    // pcl::VoxelGrid: floor(p * inverse_leaf_size), fp32 product
                    const float g = floorf(__fmul_rn(f[a], P.inv_leaf));
                    bad |= !(fabsf(g) < 67108864.0f);
                    const int v = (int)g;
                    const int t = v - A.ib[a];
                    int l;
                    if (!P.leaf_split) {
                        l = t >> 6;
                    } else if (((t - 2) >> 6) == ((t + 2) >> 6)) {
                        l = t >> 6;   // the voxel is not near a leaf face: its leaf follows from its index
                    } else {
                        // genOctreeKeyforPoint: floor((p - min) / resolution) in double
                        l = (int)floor(((double)f[a] - A.mn0[a]) / P.res);
                    }
                    leaf[a] = l;
                    cell[a] = t - 64 * l + 2;
                    bad |= (unsigned)cell[a] >= (unsigned)GRID_DIM || l < -1048576 || l > 1048575;
                    // offset inside the voxel in 2^-22 voxel units; single rounding each (fma, product, convert)
                    q[a] = (int)rintf(__fmul_rn(fmaf(-g, P.leaf, f[a]), P.fix_scale));
                }
                if (bad) {
                    atomicOr(&W.ctrl[C_ERR], ERR_RANGE);
                } else {
                    key[j] = (uint32_t)((cell[2] * GRID_DIM + cell[1]) * GRID_DIM + cell[0]);
                    lkey[j] = pack_leaf(leaf[0], leaf[1], leaf[2]);
                    qx[j] = q[0]; qy[j] = q[1]; qz[j] = q[2];
                }
            }
        }

        // ---- leaf ids: wave-uniform lookups (a wave sees one or two leaves per step) ----
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned long long pending = lkey[j];
            for (;;) {
                const unsigned long long need = __ballot(pending != 0ull);
                if (!need) break;
                const int src = __ffsll((long long)need) - 1;
                const unsigned long long k = ((unsigned long long)(uint32_t)__shfl((int)(pending >> 32), src, 64) << 32) |
                                             (uint32_t)__shfl((int)(uint32_t)pending, src, 64);
                if (k != cache_key) {
                    uint32_t id = 0;
                    if (lane == src) id = leaf_lookup(W, P.leaf_mask, k);
                    cache_id = (uint32_t)__shfl((int)id, src, 64);
                    cache_key = k;
                }
                if (pending == k) {
                    pending = 0ull;
                    key[j] = cache_id == 0xffffffffu ? KEY_EMPTY : (key[j] | (cache_id << CELL_BITS));
                }
            }
        }

        // ---- lane-local runs; lanes whose 4 points share one voxel take part in the wave merge ----
        const bool uniform = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
        const bool single = uniform && key[0] != KEY_EMPTY;
        Run32 v;
        v.key = single ? key[0] : KEY_EMPTY;
        v.qx = v.qy = v.qz = 0;
        v.cr = v.gb = v.tile = 0;
        if (single) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v.qx += qx[j]; v.qy += qy[j]; v.qz += qz[j];
                v.cr += (1u << 16) | (pw[j] & 0xffu);
                v.gb += (((pw[j] >> 8) & 0xffu) << 16) | ((pw[j] >> 16) & 0xffu);
                v.tile |= pw[j] >> 24;
            }
        }
        // segmented inclusive scan over chains of consecutive single lanes with equal keys, inside rows of 16 lanes
        const int prev_key = dpp_shr<1>((int)v.key);
        int flag = (!single || (lane & 15) == 0 || (uint32_t)prev_key != v.key) ? 1 : 0;
        scan_step<1>(v, flag);
        scan_step<2>(v, flag);
        scan_step<4>(v, flag);
        scan_step<8>(v, flag);
        // a chain ends where the next lane does not continue it
        const int next_key = dpp_shl<1>((int)v.key);
        const bool chain_end = single && ((lane & 15) == 15 || (uint32_t)next_key != v.key);
        if (chain_end) lds_insert(L, W, P, v);

        // lanes with several voxels among their 4 points insert their runs directly
        if (__ballot(!uniform)) {
            if (!uniform) {
                Run32 r;
                r.key = KEY_EMPTY;
                r.qx = r.qy = r.qz = 0;
                r.cr = r.gb = r.tile = 0;
#pragma unroll 1
                for (int j = 0; j <= 4; j++) {
                    uint32_t kj = KEY_EMPTY, wj = 0;
                    int ax = 0, ay = 0, az = 0;
                    if (j < 4) {
                        kj = j == 0 ? key[0] : j == 1 ? key[1] : j == 2 ? key[2] : key[3];
                        wj = j == 0 ? pw[0] : j == 1 ? pw[1] : j == 2 ? pw[2] : pw[3];
                        ax = j == 0 ? qx[0] : j == 1 ? qx[1] : j == 2 ? qx[2] : qx[3];
                        ay = j == 0 ? qy[0] : j == 1 ? qy[1] : j == 2 ? qy[2] : qy[3];
                        az = j == 0 ? qz[0] : j == 1 ? qz[1] : j == 2 ? qz[2] : qz[3];
                    }
                    if (kj != r.key) {
                        if (r.key != KEY_EMPTY) lds_insert(L, W, P, r);
                        r.key = kj;
                        r.qx = r.qy = r.qz = 0;
                        r.cr = r.gb = r.tile = 0;
                    }
                    if (kj != KEY_EMPTY) {
                        r.qx += ax; r.qy += ay; r.qz += az;
                        r.cr += (1u << 16) | (wj & 0xffu);
                        r.gb += (((wj >> 8) & 0xffu) << 16) | ((wj >> 16) & 0xffu);
                        r.tile |= wj >> 24;
                    }
                }
            }
        }
    }

    // ---- bounding box of this wave's range (input of the octree replay) ----
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float vlo = bmin[a], vhi = bmax[a];
        for (int off = 32; off > 0; off >>= 1) {
            vlo = fminf(vlo, __shfl_down(vlo, off, 64));
            vhi = fmaxf(vhi, __shfl_down(vhi, off, 64));
        }
        if (lane == 0) {
            W.bboxes[(size_t)range * 6 + a] = vlo;
            W.bboxes[(size_t)range * 6 + 3 + a] = vhi;
        }
    }

    // ---- flush: 8 lanes per table entry update one 64-byte record with returning adds ----
    __syncthreads();
    const int sub = threadIdx.x & 7;
#pragma unroll 1
    for (int e = threadIdx.x >> 3; e < LTAB; e += K1_THREADS / 8) {
        const uint32_t k = L.key[e];
        if (k == KEY_EMPTY) continue;
        const uint32_t t = L.tile[e];
        unsigned long long val;
        switch (sub) {
        case 0: val = L.sx[e]; break;
        case 1: val = L.sy[e]; break;
        case 2: val = L.sz[e]; break;
        case 3: val = L.cr[e]; break;
        case 4: val = L.gb[e]; break;
        case 5:   // tile bits 0-3 as 16-bit contribution counters
            val = (unsigned long long)(t & 1u) | ((unsigned long long)((t >> 1) & 1u) << 16) | ((unsigned long long)((t >> 2) & 1u) << 32) |
                  ((unsigned long long)((t >> 3) & 1u) << 48);
            break;
        case 6:   // tile bits 4-7
            val = (unsigned long long)((t >> 4) & 1u) | ((unsigned long long)((t >> 5) & 1u) << 16) | ((unsigned long long)((t >> 6) & 1u) << 32) |
                  ((unsigned long long)((t >> 7) & 1u) << 48);
            break;
        default: val = 0; break;
        }
        unsigned long long *rec = record_ptr(W, k);
        const unsigned long long old = atomicAdd(&rec[sub], val);
        // One global counter bumped by every first toucher would serialise ~40 k same-address
        // atomics at the memory side; collect them per workgroup instead.
        if (sub == 3 && (old >> 32) == 0) L.fresh[atomicAdd(&L.nfresh, 1u)] = k;
    }
    __syncthreads();
    const uint32_t nfresh = L.nfresh;
    if (threadIdx.x == 0 && nfresh) L.fresh_base = atomicAdd(&W.ctrl[C_COUNT], nfresh);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nfresh; i += K1_THREADS) {
        const uint32_t idx = L.fresh_base + i;
        if (idx < P.list_cap) W.occupied[idx] = L.fresh[i];
        else atomicOr(&W.ctrl[C_ERR], ERR_LIST_FULL);
    }
}

// ---------------------------------------------------------------------------
// K2: octree bounding-box replay / global grid box
// ---------------------------------------------------------------------------
// Growth of pcl::octree::OctreePointCloud's box is sequential in input order, but a range
// whose box lies inside the current octree box cannot trigger a growth step, so only the few
// ranges that do are re-read point by point.
__global__ void __launch_bounds__(1024) octree_replay_kernel(VoxParams P, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, const float *__restrict__ bboxes,
                                                            uint32_t *__restrict__ ctrl) {
    __shared__ double s_mn[3], s_mx[3];
    __shared__ int s_depth;
    __shared__ long long s_shift[3];
    __shared__ unsigned long long s_first;
    __shared__ int s_events;
    const int tid = threadIdx.x;
    const uint32_t nranges = P.nranges;

    if (!P.leaf_split) {
        // plain pcl::VoxelGrid: getMinMax3D, the 2^31-cell check, min_b / div_b
        __shared__ float s_red[6][16];
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t c = tid; c < nranges; c += 1024) {
            for (int a = 0; a < 3; a++) {
                lo[a] = fminf(lo[a], bboxes[(size_t)c * 6 + a]);
                hi[a] = fmaxf(hi[a], bboxes[(size_t)c * 6 + 3 + a]);
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
    uint32_t range = 0;
    while (range < nranges) {
        // first range >= `range` whose box sticks out of the current octree box
        if (tid == 0) s_first = ~0ull;
        __syncthreads();
        {
            const double mn0 = s_mn[0], mn1 = s_mn[1], mn2 = s_mn[2], mx0 = s_mx[0], mx1 = s_mx[1], mx2 = s_mx[2];
            for (uint32_t c = range + tid; c < nranges; c += 1024) {
                const float *b = bboxes + (size_t)c * 6;
                const bool viol = (double)b[0] < mn0 || (double)b[1] < mn1 || (double)b[2] < mn2 ||
                                  (double)b[3] >= mx0 || (double)b[4] >= mx1 || (double)b[5] >= mx2;
                if (viol) { atomicMin(&s_first, (unsigned long long)c); break; }
            }
        }
        __syncthreads();
        const unsigned long long hit = s_first;
        __syncthreads();
        if (hit == ~0ull) break;

        // replay that range in index order, a tile of 4096 points at a time
        const size_t r_lo = (size_t)hit * P.per_wave;
        const size_t r_hi = r_lo + P.per_wave < P.n ? r_lo + P.per_wave : P.n;
        for (size_t tile = r_lo; tile < r_hi; tile += 4096) {
            const size_t base = tile + (size_t)tid * 4;
            float qx[4], qy[4], qz[4];
            for (int j = 0; j < 4; j++) {
                const bool ok = base + j < r_hi;
                qx[j] = ok ? x[base + j] : 0.f;
                qy[j] = ok ? y[base + j] : 0.f;
                qz[j] = ok ? z[base + j] : 0.f;
            }
            size_t from = tile;   // first index of this tile whose violation has not been handled yet
            for (;;) {
                if (tid == 0) s_first = ~0ull;
                __syncthreads();
                {
                    const double mn0 = s_mn[0], mn1 = s_mn[1], mn2 = s_mn[2], mx0 = s_mx[0], mx1 = s_mx[1], mx2 = s_mx[2];
                    for (int j = 0; j < 4; j++) {
                        const size_t idx = base + j;
                        if (idx < from || idx >= r_hi) continue;
                        if (!(isfinite(qx[j]) && isfinite(qy[j]) && isfinite(qz[j]))) continue;
                        const bool viol = (double)qx[j] < mn0 || (double)qy[j] < mn1 || (double)qz[j] < mn2 ||
                                          (double)qx[j] >= mx0 || (double)qy[j] >= mx1 || (double)qz[j] >= mx2;
                        if (viol) { atomicMin(&s_first, (unsigned long long)idx); break; }
                    }
                }
                __syncthreads();
                const unsigned long long pidx = s_first;
                __syncthreads();
                if (pidx == ~0ull) break;
                if ((size_t)pidx >= base && (size_t)pidx < base + 4) {
                    // adoptBoundingBoxToPoint for this point: grow until it fits
                    const int j = (int)((size_t)pidx - base);
                    const double c[3] = {(double)(j == 0 ? qx[0] : j == 1 ? qx[1] : j == 2 ? qx[2] : qx[3]),
                                         (double)(j == 0 ? qy[0] : j == 1 ? qy[1] : j == 2 ? qy[2] : qy[3]),
                                         (double)(j == 0 ? qz[0] : j == 1 ? qz[1] : j == 2 ? qz[2] : qz[3])};
                    for (;;) {
                        bool up[3], any = false;
                        for (int a = 0; a < 3; a++) {
                            const bool lo = c[a] < s_mn[a];
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
                from = (size_t)pidx + 1;
                __syncthreads();
            }
        }
        range = (uint32_t)hit + 1;
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
__global__ void __launch_bounds__(256) make_sort_keys_kernel(VoxParams P, VoxWork W, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t m, unsigned long long *__restrict__ sort_keys,
                                                            uint32_t *__restrict__ sort_vals) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t key = W.occupied[r];
    const uint32_t cell = key & ((1u << CELL_BITS) - 1), leaf_id = key >> CELL_BITS;
    const int c[3] = {(int)(cell % GRID_DIM), (int)((cell / GRID_DIM) % GRID_DIM), (int)(cell / (GRID_DIM * GRID_DIM))};
    const unsigned long long lp = W.leaf_keys[leaf_id];
    unsigned long long sk;
    if (!P.leaf_split) {
        const Anchor A = make_anchor(P, 0.f, 0.f, 0.f);
        long long d[3];
        for (int a = 0; a < 3; a++) d[a] = (long long)(c[a] + A.ib[a] + 64 * unpack_leaf(lp, a) - 2) - (long long)(int)W.ctrl[C_MINB + a];
        const long long dx = (int)W.ctrl[C_DIVB], dy = (int)W.ctrl[C_DIVB + 1];
        sk = (unsigned long long)(d[0] + d[1] * dx + d[2] * dx * dy);
    } else {
        const int depth = (int)W.ctrl[C_DEPTH];
        long long lk[3];
        bool bad = depth > 14;
        for (int a = 0; a < 3; a++) {
            const long long shift = (long long)(((unsigned long long)W.ctrl[C_SHIFT + 2 * a + 1] << 32) | W.ctrl[C_SHIFT + 2 * a]);
            lk[a] = (long long)unpack_leaf(lp, a) + shift;
            if (lk[a] < 0 || lk[a] >= ((long long)1 << depth)) bad = true;
        }
        if (bad) {
            atomicOr(&W.ctrl[C_ERR], depth > 14 ? ERR_DEPTH : ERR_LEAF_RANGE);
            sk = ~0ull;
        } else {
            unsigned long long morton = 0;
            for (int b = depth - 1; b >= 0; b--) {
                morton = (morton << 3) | (((unsigned long long)(lk[0] >> b) & 1) << 2) | (((unsigned long long)(lk[1] >> b) & 1) << 1) |
                         ((unsigned long long)(lk[2] >> b) & 1);
            }
            sk = (morton << 21) | ((unsigned long long)c[2] << 14) | ((unsigned long long)c[1] << 7) | (unsigned long long)c[0];
        }
    }
    sort_keys[r] = sk;
    sort_vals[r] = key;
}

// ---------------------------------------------------------------------------
// K4: emit in output order and clean the records
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) emit_and_clean_kernel(VoxParams P, VoxWork W, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, uint32_t m, const uint32_t *__restrict__ sorted_keys,
                                                            float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
                                                            uint32_t *__restrict__ ow, int emit) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t key = emit ? sorted_keys[r] : W.occupied[r];
    ulonglong2 *rec = reinterpret_cast<ulonglong2 *>(record_ptr(W, key));
    if (emit) {
        const ulonglong2 w01 = rec[0], w23 = rec[1], w45 = rec[2], w67 = rec[3];
        const uint32_t cell = key & ((1u << CELL_BITS) - 1), leaf_id = key >> CELL_BITS;
        const int c[3] = {(int)(cell % GRID_DIM), (int)((cell / GRID_DIM) % GRID_DIM), (int)(cell / (GRID_DIM * GRID_DIM))};
        const unsigned long long lp = W.leaf_keys[leaf_id];
        const Anchor A = make_anchor(P, x[0], y[0], z[0]);
        double vox[3];
        for (int a = 0; a < 3; a++) vox[a] = (double)(c[a] + A.ib[a] + 64 * unpack_leaf(lp, a) - 2);
        const unsigned long long cr = w23.y, gb = w45.x;
        const uint32_t cnt = (uint32_t)(cr >> 32);
        const double n = (double)cnt, unit = (double)P.fix_scale;
        // mean = voxel origin + mean offset; one rounding to fp32 at the end
        ox[r] = (float)(vox[0] * P.leaf_d + ((double)(long long)w01.x / n) / unit);
        oy[r] = (float)(vox[1] * P.leaf_d + ((double)(long long)w01.y / n) / unit);
        oz[r] = (float)(vox[2] * P.leaf_d + ((double)(long long)w23.x / n) / unit);
        // pcl AccumulatorRGBA: float sums (exact integers here) / n, truncated
        const float fn = (float)cnt;
        const uint32_t rr = (uint32_t)__fdiv_rn((float)(uint32_t)(cr & 0xffffffffu), fn);
        const uint32_t gg = (uint32_t)__fdiv_rn((float)(uint32_t)(gb >> 32), fn);
        const uint32_t bb = (uint32_t)__fdiv_rn((float)(uint32_t)(gb & 0xffffffffu), fn);
        // tile: bits 0-3 / 4-7 as contribution counters, plus the OR word of the slow path
        uint32_t tile = (uint32_t)w67.y & 0xffu;
        for (int b = 0; b < 4; b++) {
            if ((w45.y >> (16 * b)) & 0xffffull) tile |= 1u << b;
            if ((w67.x >> (16 * b)) & 0xffffull) tile |= 16u << b;
        }
        ow[r] = (rr & 0xffu) | ((gg & 0xffu) << 8) | ((bb & 0xffu) << 16) | (tile << 24);
    }
    const ulonglong2 zero = {0ull, 0ull};
    rec[0] = zero; rec[1] = zero; rec[2] = zero; rec[3] = zero;
}

// ---------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------
struct Workspace {
    int device = -1;
    uint32_t leaf_cap = 0;     // leaf hash capacity = number of grids (power of two)
    size_t list_cap = 0;
    size_t bbox_cap = 0;
    unsigned long long *leaf_keys = nullptr;
    unsigned long long *records = nullptr;
    uint32_t *occupied = nullptr;
    float *bboxes = nullptr;
    uint32_t *ctrl = nullptr;
    int num_cus = 0;
    void release() {
        // also runs at thread exit, when the runtime may be gone: errors ignored
        if (leaf_keys) (void)hipFree(leaf_keys);
        if (records) (void)hipFree(records);
        if (occupied) (void)hipFree(occupied);
        if (bboxes) (void)hipFree(bboxes);
        if (ctrl) (void)hipFree(ctrl);
        leaf_keys = nullptr; records = nullptr; occupied = nullptr; bboxes = nullptr; ctrl = nullptr;
        leaf_cap = 0; list_cap = 0; bbox_cap = 0;
    }
    ~Workspace() { release(); }
};

thread_local Workspace t_ws;

constexpr size_t GRID_BYTES = (size_t)CELLS * RECORD_WORDS * 8;   // 20.1 MB per leaf grid

bool ensure_workspace(Workspace &ws, size_t n, uint32_t leaf_cap, uint32_t nranges, hipStream_t s) {
    const int dev = current_device();
    if (ws.device != dev) {
        ws.release();
        hipDeviceProp_t prop;
        CW_HIP_TRY(hipGetDeviceProperties(&prop, dev));
        ws.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(LdsTable)));
        ws.device = dev;
    }
    if (ws.leaf_cap < leaf_cap) {
        if (ws.leaf_keys) (void)hipFree(ws.leaf_keys);
        if (ws.records) (void)hipFree(ws.records);
        ws.leaf_keys = nullptr; ws.records = nullptr; ws.leaf_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.leaf_keys, (size_t)leaf_cap * 8));
        CW_HIP_TRY(hipMalloc((void **)&ws.records, (size_t)leaf_cap * GRID_BYTES));
        CW_HIP_TRY(hipMemsetAsync(ws.records, 0, (size_t)leaf_cap * GRID_BYTES, s));   // once; K4 keeps it clean afterwards
        ws.leaf_cap = leaf_cap;
    }
    if (ws.list_cap < n) {
        if (ws.occupied) (void)hipFree(ws.occupied);
        ws.occupied = nullptr; ws.list_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.occupied, n * sizeof(uint32_t)));
        ws.list_cap = n;
    }
    if (ws.bbox_cap < nranges) {
        if (ws.bboxes) (void)hipFree(ws.bboxes);
        ws.bboxes = nullptr; ws.bbox_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.bboxes, (size_t)nranges * 6 * sizeof(float)));
        ws.bbox_cap = nranges;
    }
    if (!ws.ctrl) CW_HIP_TRY(hipMalloc((void **)&ws.ctrl, C_WORDS * sizeof(uint32_t)));
    return true;
}

}  // namespace

std::shared_ptr<DeviceSoA> voxel_downsample(const DeviceSoA &src, float cellsize, bool leaf_split, int *error_code) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    const size_t n = src.npoints;
    Workspace &ws = t_ws;

    uint32_t leaf_cap = ws.leaf_cap ? ws.leaf_cap : 64;   // 64 grids = 1.3 GB; grown x4 when a cloud has more leaves
    for (int attempt = 0; attempt < 5; attempt++) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || cus <= 0) cus = 256;
        // one persistent workgroup per CU; short clouds get fewer so that every wave has at least one step
        size_t nwaves = (size_t)cus * K1_WAVES;
        const size_t steps_total = (n + WAVE_STEP - 1) / WAVE_STEP;
        if (nwaves > steps_total) nwaves = ((steps_total + K1_WAVES - 1) / K1_WAVES) * K1_WAVES;
        const uint32_t nblocks = (uint32_t)(nwaves / K1_WAVES);
        if (!ensure_workspace(ws, n, leaf_cap, (uint32_t)nwaves, c.stream)) return nullptr;

        VoxParams P;
        P.n = n;
        P.per_wave = (((n + nwaves - 1) / nwaves + WAVE_STEP - 1) / WAVE_STEP) * WAVE_STEP;
        P.nranges = (uint32_t)nwaves;
        P.leaf = cellsize;
        P.inv_leaf = 1.0f / cellsize;
        P.fix_scale = FIX_ONE_F / cellsize;
        P.leaf_d = (double)cellsize;
        float octree_cellsize = (8 * 8) * cellsize;   // reference src/cwipc_filters.cpp:113-114
        P.res = (double)octree_cellsize;
        P.leaf_split = leaf_split ? 1 : 0;
        P.leaf_mask = ws.leaf_cap - 1;
        P.list_cap = (uint32_t)(ws.list_cap > 0xffffffffu ? 0xffffffffu : ws.list_cap);
        VoxWork W{ws.leaf_keys, ws.records, ws.occupied, ws.ctrl, ws.bboxes};

        bool ok = hipMemsetAsync(ws.ctrl, 0, C_WORDS * sizeof(uint32_t), c.stream) == hipSuccess &&
                  hipMemsetAsync(ws.leaf_keys, 0, (size_t)ws.leaf_cap * 8, c.stream) == hipSuccess;
        if (!ok) { hip_failed(hipGetLastError(), "hipMemsetAsync(voxel ctrl)", __FILE__, __LINE__); return nullptr; }

        CW_LAUNCH("voxel_accumulate", voxel_accumulate_kernel, dim3(nblocks), dim3(K1_THREADS), sizeof(LdsTable), c.stream, P, src.x(), src.y(),
                  src.z(), src.rgbt(), W);
        CW_LAUNCH("octree_replay", octree_replay_kernel, dim3(1), dim3(1024), 0, c.stream, P, src.x(), src.y(), src.z(), ws.bboxes, ws.ctrl);
        ok = hipMemcpyAsync(c.host_words, ws.ctrl, C_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
        ok = c.sync() && ok;
        if (!ok) { hip_failed(hipGetLastError(), "voxel_accumulate", __FILE__, __LINE__); return nullptr; }

        uint32_t err = c.host_words[C_ERR];
        const uint32_t m = c.host_words[C_COUNT] < P.list_cap ? c.host_words[C_COUNT] : P.list_cap;
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
                CW_LAUNCH("make_sort_keys", make_sort_keys_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, src.x(), src.y(), src.z(), m, keys_in,
                          vals_in);
                // only the bits that can be set take part in the sort
                unsigned end_bit = 64;
                if (leaf_split) end_bit = 21 + 3 * c.host_words[C_DEPTH];
                else end_bit = 32;
                if (end_bit > 64) end_bit = 64;
                size_t tmp_bytes = 0;
                hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0u, end_bit, c.stream);
                if (e == hipSuccess) {
                    sort_tmp = pool_alloc(tmp_bytes ? tmp_bytes : 256);
                    if (!sort_tmp) e = hipErrorOutOfMemory;
                }
                if (e == hipSuccess) {
                    if (profiling_enabled()) profile_begin("radix_sort_pairs", c.stream);
                    e = rocprim::radix_sort_pairs(sort_tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0u, end_bit, c.stream);
                    if (profiling_enabled()) profile_end(c.stream);
                }
                if (e != hipSuccess) {
                    hip_failed(e, "rocprim::radix_sort_pairs", __FILE__, __LINE__);
                    err |= 0x80000000u;
                }
            }
        }
        if (m) {
            // emit (or, on error, only clean): the records must be left zeroed either way
            const int emit = (!err && dst) ? 1 : 0;
            CW_LAUNCH("emit_and_clean", emit_and_clean_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, src.x(), src.y(), src.z(), m, vals_out,
                      emit ? dst->x() : nullptr, emit ? dst->y() : nullptr, emit ? dst->z() : nullptr, emit ? dst->rgbt() : nullptr, emit);
            if (emit) ok = hipMemcpyAsync(c.host_words, ws.ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
            ok = c.sync() && ok;
            if (emit && ok) err |= c.host_words[C_ERR];
        }
        pool_free(keys_in);
        pool_free(vals_in);
        pool_free(sort_tmp);
        if (error_code) *error_code = (int)err;
        if (!ok) { hip_failed(hipGetLastError(), "voxel emit", __FILE__, __LINE__); return nullptr; }

        if ((err & ERR_LEAVES) && !(err & ~(uint32_t)(ERR_LEAVES | ERR_LIST_FULL))) {
            // more leaves than grids: the touched records were cleaned above; grow and run again
            if ((size_t)leaf_cap * 4 * GRID_BYTES > ((size_t)200 << 30)) {
                cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: the cloud spans more octree leaves than fit in device memory");
                return nullptr;
            }
            leaf_cap *= 4;
            continue;
        }
        if (err) {
            std::string why;
            if (err & ERR_GRID_OVERFLOW) why += " VoxelGrid: leaf size is too small for the input dataset, integer indices would overflow;";
            if (err & ERR_RANGE) why += " voxel or leaf index out of range;";
            if (err & (ERR_DEPTH | ERR_LEAF_RANGE)) why += " octree deeper than 14 levels;";
            if (err & ERR_CELL_RANGE) why += " voxel outside its leaf grid;";
            if (err & ERR_FIRST_POINT) why += " the first point is not finite;";
            if (err & ERR_LIST_FULL) why += " occupied list full;";
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
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: could not size the workspace");
    return nullptr;
}

}  // namespace cwipc_amd
