// kernels_sor.hip -- statistical outlier removal on gfx950.
//
// Reference: cwipc_remove_outliers, src/cwipc_filters.cpp:181-278; the filter
// itself is pcl::StatisticalOutlierRemoval [PCL upstream], restated in
// oracle/cwipc_oracle.c (sor_filter):
//   d_i  = float( sum_{j=1..k} sqrtf(dist2_j) / k )   exact k+1 nearest (self = j 0),
//          dist2 in fp32 as FLANN's L2_Simple ((dx*dx + dy*dy) + dz*dz), ascending, sum in f64
//   mean / variance over all d_i in f64 (squares formed in fp32), threshold = mean + mul*stddev
//   keep point i iff !(d_i > threshold), input order preserved.
//
// The k nearest distances of a point are a well-defined multiset, so d_i does not
// depend on how the neighbour search is organised.  Here: points are bucketed
// into a uniform grid by a counting sort (one atomic per run of equal cells in a wave), each lane
// searches growing cubic shells of cells around its point -- its own row of cells first, rows that can
// no longer hold one of the k + 1 nearest skipped -- and keeps the k + 1 smallest distances as a sorted list
// in registers (insert = one v_med3 per slot; k <= 32; an LDS list up to k = 120); a shell
// radius r proves exactness once the (k+1)-th distance is <= r*h.
// Two layouts of the grid: dense (small and medium clouds; the grid itself -- box, cell size from an
// occupancy census -- is decided by two one-wave kernels on the device, no host round trip), and sparse
// (from 2^20 points: segments of 16 cells along x that exist only where points are, see SEG below).
#include "internal.hpp"
#include <vector>

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include <cfloat>
#include <cmath>
#include <mutex>

namespace cwipc_amd {

namespace {

constexpr int BLK = 256;
constexpr size_t MAX_CELLS = (size_t)1 << 27;   // dense grid cells at most (three 4-byte arrays of this length); a call uses 8 per point at most

struct Grid {
    float mn[3];
    int dim[3];
    double h;
    double inv_h;
    int nsegx;   // sparse layout: segments (16 cells along x) per row of cells
};

// Sparse layout of the grid for big clouds.  A surface occupies a percent or two of a fine 3-D grid: clearing and scanning a
// dense array of 10^8 cells costs more than the search saves.  Cells are grouped into SEGMENTS of 16 along x; only
// segments that hold points get cells, numbered in the order of the segments (x fastest), so the cells of a row of the
// grid are still one contiguous run of the sorted points, whatever segments are missing in between.
constexpr int SEG = 16, SEG_SHIFT = 4;

// The dense layout's grid is decided ON THE DEVICE (small and medium clouds: a tile of a frame is filtered in ~0.1 ms, and two
// host round trips -- for the bounding box, for the occupancy census -- were a third of that): the kernels read the grid
// from this block, which two one-wave kernels fill in.
struct GridMeta {
    Grid g;
    double ext[3], maxext;
    uint32_t occ;      // occupied cells of the first count (census)
    uint32_t refine;   // 1: the grid was coarsened after the census, the cells are counted again
};
__device__ __forceinline__ void grid_dims(Grid &g, const double ext[3], double h) {
    for (int a = 0; a < 3; a++) g.dim[a] = (int)floor(ext[a] / h) + 1;
    g.h = h;
    g.inv_h = 1.0 / h;
    g.nsegx = (g.dim[0] + SEG - 1) / SEG;
}
__device__ __forceinline__ size_t grid_cells(const Grid &g) { return (size_t)g.dim[0] * (size_t)g.dim[1] * (size_t)g.dim[2]; }

// one wave of the first workgroup: the cloud's box from the partial boxes, then the finest grid of at most cap_cells cells
// ... and the same launch clears the two per-cell arrays of the counting sort (its other workgroups: two memsets less)
__global__ void __launch_bounds__(BLK) grid_setup_zero_kernel(const float *__restrict__ partial, unsigned nb, size_t cap_cells, GridMeta *__restrict__ m,
                                                             uint32_t *__restrict__ counts, uint32_t *__restrict__ cursor) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    const size_t nvec = cap_cells / 4;
    uint4 *c4 = reinterpret_cast<uint4 *>(counts), *u4 = reinterpret_cast<uint4 *>(cursor);
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nvec; i += (size_t)gridDim.x * BLK) { c4[i] = zero; u4[i] = zero; }
    if (blockIdx.x == 0 && threadIdx.x < (cap_cells & 3)) { counts[nvec * 4 + threadIdx.x] = 0; cursor[nvec * 4 + threadIdx.x] = 0; }
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (unsigned b = threadIdx.x; b < nb; b += 64)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], partial[b * 6 + a]); hi[a] = fmaxf(hi[a], partial[b * 6 + 3 + a]); }
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64)); }
    if (threadIdx.x != 0) return;
    Grid g;
    double ext[3], maxext = 0;
    for (int a = 0; a < 3; a++) {
        g.mn[a] = lo[a] == FLT_MAX ? 0.f : lo[a];
        ext[a] = (double)hi[a] - (double)lo[a];
        if (!(ext[a] >= 0)) ext[a] = 0;   // no finite point
        if (ext[a] > maxext) maxext = ext[a];
    }
    if (!(maxext > 0)) maxext = 1.0;
    double h = maxext / 1024.0;
    grid_dims(g, ext, h);
    while (grid_cells(g) > cap_cells) { h *= 1.25; grid_dims(g, ext, h); }
    m->g = g;
    for (int a = 0; a < 3; a++) m->ext[a] = ext[a];
    m->maxext = maxext;
    m->occ = 0;
    m->refine = 0;
}

// The census says how many cells hold points (m->occ): coarsen the grid so that an occupied cell holds about `target` points
// (surface-like data: points per cell grow with h^2) and, if so, clear the counts for the second count -- one launch: every
// workgroup takes the decision from the same words (n, occ), the first one also writes the new grid (which nobody reads here).
__global__ void __launch_bounds__(BLK) grid_refine_zero_kernel(GridMeta *__restrict__ m, size_t n, double target, uint32_t *__restrict__ words, size_t nwords) {
    const uint32_t occ = m->occ;
    const double ppc = (double)n / (double)(occ ? occ : 1u);
    if (!(ppc < target)) return;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nwords; i += (size_t)gridDim.x * BLK) words[i] = 0;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double h = m->g.h * sqrt(target / ppc);
    if (h > m->maxext) h = m->maxext;
    Grid g = m->g;
    grid_dims(g, m->ext, h);
    m->g = g;
    m->refine = 1;
}

__device__ __forceinline__ int cell_coord(const Grid &g, float v, int a) {
    int c = (int)floor(((double)v - (double)g.mn[a]) * g.inv_h);
    c = c < 0 ? 0 : c;
    return c >= g.dim[a] ? g.dim[a] - 1 : c;
}

__device__ __forceinline__ uint32_t cell_of(const Grid &g, float x, float y, float z) {
    return (uint32_t)cell_coord(g, x, 0) + (uint32_t)g.dim[0] * ((uint32_t)cell_coord(g, y, 1) + (uint32_t)g.dim[1] * (uint32_t)cell_coord(g, z, 2));
}

// ---- bounding box ----
__global__ void __launch_bounds__(BLK) bbox_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                  float *__restrict__ partial /* [gridDim.x][6] */) {
    __shared__ float red[6][BLK / 64];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        float v[3] = {x[i], y[i], z[i]};
        if (!(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]))) continue;
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], v[a]); hi[a] = fmaxf(hi[a], v[a]); }
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < BLK / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// ---- occupancy probe / counting sort ----
// Consecutive points of a cloud in scan order mostly share a cell: one atomic per RUN of equal cells inside a wave, not per
// point (the scattered atomics of these two kernels were their whole cost: 10 M of them take ~0.4 ms).  `run` describes the
// run a lane belongs to: its first lane and its length.
struct WaveRun { int first; int length; bool leads; };
__device__ __forceinline__ WaveRun wave_run(uint32_t c, bool active) {
    const int lane = threadIdx.x & 63;
    const uint32_t prev = (uint32_t)__shfl_up((int)c, 1, 64);
    const bool leads = active && (lane == 0 || prev != c);
    const unsigned long long L = __ballot(leads), A = __ballot(active);
    WaveRun r;
    r.leads = leads;
    const unsigned long long upto = L & ((2ull << lane) - 1ull);                 // leaders at or below this lane
    r.first = upto ? 63 - __builtin_clzll(upto) : lane;
    const unsigned long long above = r.first < 63 ? (L >> (r.first + 1)) : 0ull;  // the next run's leader, if any
    const int end = above ? r.first + 1 + __builtin_ctzll(above) : (A ? 64 - __builtin_clzll(A) : 0);
    r.length = end - r.first;
    return r;
}

// census: also count the cells that get their first point here (the adds then return what was there), one atomic per workgroup
// on *census -- a separate pass over the whole cell array for it was a launch of its own
__global__ void __launch_bounds__(BLK) cell_count_kernel(Grid gv, const GridMeta *__restrict__ gm, int second_count, const float *__restrict__ x,
                                                        const float *__restrict__ y, const float *__restrict__ z, size_t n, uint32_t *__restrict__ counts,
                                                        uint32_t *__restrict__ cell_id, uint32_t *__restrict__ census) {
    if (gm && second_count && !gm->refine) return;   // the census's grid stands: its counts do too
    const Grid g = gm ? gm->g : gv;
    uint32_t fresh = 0;
    for (size_t base = (size_t)blockIdx.x * BLK; base < n; base += (size_t)gridDim.x * BLK) {
        const size_t i = base + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0xffffffffu;
        if (active) {
            c = cell_of(g, x[i], y[i], z[i]);
            if (cell_id) cell_id[i] = c;
        }
        const WaveRun r = wave_run(c, active);
        if (census) {
            if (r.leads && atomicAdd(&counts[c], (uint32_t)r.length) == 0u) fresh++;
        } else if (r.leads) {
            atomicAdd(&counts[c], (uint32_t)r.length);
        }
    }
    if (!census) return;
    __shared__ uint32_t wsum[BLK / 64];
    for (int off = 32; off > 0; off >>= 1) fresh += __shfl_down(fresh, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = fresh;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < BLK / 64; w++) t += wsum[w];
        if (t) atomicAdd(census, t);
    }
}

__global__ void __launch_bounds__(BLK) count_nonzero_kernel(const uint32_t *__restrict__ counts, size_t ncells, uint32_t *__restrict__ out) {
    // one atomic per workgroup: thousands of adds to one address serialise (about 6 ns each)
    __shared__ uint32_t wsum[BLK / 64];
    uint32_t c = 0;
    const size_t nvec = ncells / 4;
    const uint4 *v = reinterpret_cast<const uint4 *>(counts);
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nvec; i += (size_t)gridDim.x * BLK) {
        const uint4 q = v[i];
        c += (q.x != 0) + (q.y != 0) + (q.z != 0) + (q.w != 0);
    }
    if (blockIdx.x == 0 && threadIdx.x < (ncells & 3)) c += counts[nvec * 4 + threadIdx.x] != 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < BLK / 64; w++) t += wsum[w];
        if (t) atomicAdd(out, t);
    }
}

// starts = exclusive prefix sums of counts over the cells of the grid the device decided on (gm->g): ONE workgroup, for the
// small clouds of the dense layout (a camera tile of a frame: a few ten thousand cells).  rocprim's scan runs over the whole
// allocation (the host does not know the grid: 8 cells per point whatever the kernels made of them) in two launches, 9 + 3 us for
// such a tile; this one reads the cell count where the grid is and takes a pass to add and a pass to write.
constexpr int SCAN1_THREADS = 1024, SCAN1_PRE = 12;   // 12 groups of four cells per thread in registers: 48 k cells per round (16: spills)
__device__ __forceinline__ uint32_t scan1_wave_inclusive(uint32_t v) {   // four DPP row shifts inside rows of 16, two row broadcasts across them
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__global__ void __launch_bounds__(SCAN1_THREADS) small_scan_kernel(const GridMeta *__restrict__ gm, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ counts2,
                                                                  uint32_t *__restrict__ starts, size_t cap) {
    __shared__ uint32_t wsum[2][SCAN1_THREADS / 64];
    if (counts2 && gm[1].refine) { gm += 1; counts = counts2; }   // the small clouds' flow: the coarser grid's slot and counts
    size_t ncells = grid_cells(gm->g);
    if (ncells > cap) ncells = cap;
    // Whole 16-byte groups (the arrays are pool blocks, cap is a multiple of four, and what lies between the grid's last cell and the end
    // of its group is zeroes).  A round: up to twelve slices of 1024 groups, lane t of the workgroup taking group t of every slice -- every
    // load and store instruction of a wave covers 1 KB in one piece, and all of a round's loads are in flight before the first is used --
    // then slice by slice: the lanes' sums, a DPP scan over the wave, the waves' totals through LDS (one barrier per slice of 4096 cells).
    // (Versions before this one, all measured on a 35 k-cell grid, where an empty kernel of this shape takes 6 us by events: a contiguous
    // share per thread in two passes 15 us -- 144 bytes per lane apart, every wave instruction sixty-four separate requests on ONE compute
    // unit; tiles of 16 k cells with 64 contiguous bytes per lane 13 us; with that tile's loads under a condition, which the compiler turned
    // into sixteen one-word loads behind a branch each, 25 us.  rocprim's two launches over the whole allocation: 11.6.)
    const size_t nvec = (ncells + 3) / 4, vlast = cap / 4 - 1;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(counts);
    uint4 *s4 = reinterpret_cast<uint4 *>(starts);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint32_t carry = 0, flip = 0;
    for (size_t base = 0; base < nvec; base += (size_t)SCAN1_THREADS * SCAN1_PRE) {
        uint4 q[SCAN1_PRE];
#pragma unroll
        for (int j = 0; j < SCAN1_PRE; j++) {
            const size_t v = base + (size_t)j * SCAN1_THREADS + threadIdx.x;
            q[j] = c4[v < vlast ? v : vlast];   // (the address clamped into the allocation, the VALUE chosen afterwards: no load under a condition)
        }
#pragma unroll
        for (int j = 0; j < SCAN1_PRE; j++) {
            const size_t v = base + (size_t)j * SCAN1_THREADS + threadIdx.x;
            if (base + (size_t)j * SCAN1_THREADS >= nvec) continue;   // (the same for every lane: the barrier below is met by all or none)
            const uint4 w = v < nvec ? q[j] : zero;
            const uint32_t mine = w.x + w.y + w.z + w.w;
            const uint32_t incl = scan1_wave_inclusive(mine);
            uint32_t (&ws)[SCAN1_THREADS / 64] = wsum[flip & 1u];   // (two sets: the next slice's writers do not wait for this slice's readers)
            flip++;
            if (lane == 63) ws[wave] = incl;
            __syncthreads();
            uint32_t before = carry + incl - mine, total = 0;
#pragma unroll
            for (int k = 0; k < SCAN1_THREADS / 64; k++) {
                const uint32_t t = ws[k];
                if (k < wave) before += t;
                total += t;
            }
            carry += total;
            if (v < nvec) s4[v] = make_uint4(before, before + w.x, before + w.x + w.y, before + w.x + w.y + w.z);
        }
    }
}

// ---- small clouds (r4): the dense layout in ten launches instead of twelve ----
// A camera tile of a frame (a few ten thousand points) is filtered in ~0.1 ms, most of it the chain of small kernels in front of
// the search.  Here the box kernel also clears the per-cell arrays (their size is the host's: 8 cells per point), EVERY workgroup of
// the first count derives the grid from the partial boxes itself (a few KB from L2 and one lane's arithmetic: the same grid in every
// workgroup) and the first one writes it down; every workgroup of the second count takes the coarsening decision from the same
// two words and, if it stands, derives the coarser grid itself and counts into an array of its own (cleared with the others), the
// first one writing the grid into the block's second slot.  What follows reads slot 1 if its `refine` says so, slot 0 otherwise.
__device__ __forceinline__ void finest_grid(const float lo[3], const float hi[3], size_t cap_cells, GridMeta &m) {   // grid_setup_zero_kernel's arithmetic
    Grid g;
    double ext[3], maxext = 0;
    for (int a = 0; a < 3; a++) {
        g.mn[a] = lo[a] == FLT_MAX ? 0.f : lo[a];
        ext[a] = (double)hi[a] - (double)lo[a];
        if (!(ext[a] >= 0)) ext[a] = 0;   // no finite point
        if (ext[a] > maxext) maxext = ext[a];
    }
    if (!(maxext > 0)) maxext = 1.0;
    double h = maxext / 1024.0;
    grid_dims(g, ext, h);
    while (grid_cells(g) > cap_cells) { h *= 1.25; grid_dims(g, ext, h); }
    m.g = g;
    for (int a = 0; a < 3; a++) m.ext[a] = ext[a];
    m.maxext = maxext;
}

__global__ void __launch_bounds__(BLK) small_bbox_zero_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                             float *__restrict__ partial /* [gridDim.x][6] */, uint32_t *__restrict__ words, size_t nwords,
                                                             GridMeta *__restrict__ m) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 *w4 = reinterpret_cast<uint4 *>(words);
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nwords / 4; i += (size_t)gridDim.x * BLK) w4[i] = zero;   // (nwords: a multiple of four)
    if (blockIdx.x == 0 && threadIdx.x < 2) { m[threadIdx.x].occ = 0; m[threadIdx.x].refine = 0; }
    __shared__ float red[6][BLK / 64];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        float v[3] = {x[i], y[i], z[i]};
        if (!(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]))) continue;
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], v[a]); hi[a] = fmaxf(hi[a], v[a]); }
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < BLK / 64; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// PHASE 0: grid from the partial boxes, count + census into counts, slot 0.  PHASE 1: the coarser grid if the census asks for one,
// count into counts2, slot 1.
template <int PHASE>
__global__ void __launch_bounds__(BLK) small_count_kernel(const float *__restrict__ partial, unsigned nb, size_t cap_cells, double target, GridMeta *__restrict__ m,
                                                         const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                         uint32_t *__restrict__ counts, uint32_t *__restrict__ cell_id) {
    __shared__ GridMeta sm;
    __shared__ uint32_t wsum[BLK / 64];
    if (PHASE == 0) {
        if (threadIdx.x < 64) {
            float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (unsigned b = threadIdx.x; b < nb; b += 64)
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], partial[b * 6 + a]); hi[a] = fmaxf(hi[a], partial[b * 6 + 3 + a]); }
            for (int a = 0; a < 3; a++)
                for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64)); }
            if (threadIdx.x == 0) {
                finest_grid(lo, hi, cap_cells, sm);
                if (blockIdx.x == 0) { m[0].g = sm.g; for (int a = 0; a < 3; a++) m[0].ext[a] = sm.ext[a]; m[0].maxext = sm.maxext; }   // (occ: the census's adds, cleared by the box kernel)
            }
        }
    } else {
        const uint32_t occ = m[0].occ;
        const double ppc = (double)n / (double)(occ ? occ : 1u);
        if (!(ppc < target)) return;   // the census's grid stands, and its counts (the whole workgroup leaves: the same words for every thread)
        if (threadIdx.x == 0) {
            double h = m[0].g.h * sqrt(target / ppc);
            if (h > m[0].maxext) h = m[0].maxext;
            Grid g = m[0].g;
            double ext[3] = {m[0].ext[0], m[0].ext[1], m[0].ext[2]};
            grid_dims(g, ext, h);
            sm.g = g;
            if (blockIdx.x == 0) { m[1].g = g; m[1].refine = 1; }
        }
    }
    __syncthreads();
    const Grid g = sm.g;
    uint32_t fresh = 0;
    for (size_t base = (size_t)blockIdx.x * BLK; base < n; base += (size_t)gridDim.x * BLK) {
        const size_t i = base + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0xffffffffu;
        if (active) {
            c = cell_of(g, x[i], y[i], z[i]);
            cell_id[i] = c;
        }
        const WaveRun r = wave_run(c, active);
        if (PHASE == 0) {
            if (r.leads && atomicAdd(&counts[c], (uint32_t)r.length) == 0u) fresh++;
        } else if (r.leads) {
            atomicAdd(&counts[c], (uint32_t)r.length);
        }
    }
    if (PHASE != 0) return;
    for (int off = 32; off > 0; off >>= 1) fresh += __shfl_down(fresh, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = fresh;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < BLK / 64; w++) t += wsum[w];
        if (t) atomicAdd(&m[0].occ, t);
    }
}

// sorted[pos] = (x, y, z, original index)
__global__ void __launch_bounds__(BLK) cell_scatter_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                          const uint32_t *__restrict__ cell_id, const uint32_t *__restrict__ cell_start,
                                                          uint32_t *__restrict__ cell_fill, float4 *__restrict__ sorted) {
    for (size_t base = (size_t)blockIdx.x * BLK; base < n; base += (size_t)gridDim.x * BLK) {
        const size_t i = base + threadIdx.x;
        const bool active = i < n;
        const uint32_t c = active ? cell_id[i] : 0xffffffffu;
        const WaveRun r = wave_run(c, active);
        // the run's leader reserves room for the whole run; the order of points inside a cell does not matter to the search
        uint32_t at = 0;
        if (r.leads) at = cell_start[c] + atomicAdd(&cell_fill[c], (uint32_t)r.length);
        at = (uint32_t)__shfl((int)at, r.first, 64);
        if (active) sorted[at + (uint32_t)((threadIdx.x & 63) - r.first)] = make_float4(x[i], y[i], z[i], __uint_as_float((uint32_t)i));
    }
}

// ---- sparse layout: which segments exist, their cells ----
// id of a point's cell before the segments are numbered: segment << 4 | cell inside the segment
__device__ __forceinline__ uint32_t seg_cell_of(const Grid &g, float x, float y, float z) {
    const uint32_t cx = (uint32_t)cell_coord(g, x, 0);
    const uint32_t seg = (cx >> SEG_SHIFT) + (uint32_t)g.nsegx * ((uint32_t)cell_coord(g, y, 1) + (uint32_t)g.dim[1] * (uint32_t)cell_coord(g, z, 2));
    return (seg << SEG_SHIFT) | (cx & (SEG - 1));
}

// masks[segment] |= bit of the cell; cell_id[i] = seg_cell_of(point i)
__global__ void __launch_bounds__(BLK) seg_mark_kernel(Grid g, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                      uint32_t *__restrict__ masks, uint32_t *__restrict__ cell_id) {
    for (size_t base = (size_t)blockIdx.x * BLK; base < n; base += (size_t)gridDim.x * BLK) {
        const size_t i = base + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0xffffffffu;
        if (active) {
            c = seg_cell_of(g, x[i], y[i], z[i]);
            cell_id[i] = c;
        }
        const WaveRun r = wave_run(c, active);
        if (r.leads) atomicOr(&masks[c >> SEG_SHIFT], 1u << (c & (SEG - 1)));
    }
}

// flags[s] = segment s holds points; out[0] += occupied cells, out[1] += occupied segments (one atomic pair per workgroup)
__global__ void __launch_bounds__(BLK) seg_census_kernel(const uint32_t *__restrict__ masks, size_t nseg, uint32_t *__restrict__ flags, uint32_t *__restrict__ out) {
    __shared__ uint32_t wsum[2][BLK / 64];
    uint32_t cells = 0, segs = 0;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nseg; i += (size_t)gridDim.x * BLK) {
        const uint32_t m = masks[i];
        flags[i] = m != 0u;
        cells += (uint32_t)__popc(m);
        segs += m != 0u;
    }
    for (int off = 32; off > 0; off >>= 1) { cells += __shfl_down(cells, off, 64); segs += __shfl_down(segs, off, 64); }
    if ((threadIdx.x & 63) == 0) { wsum[0][threadIdx.x >> 6] = cells; wsum[1][threadIdx.x >> 6] = segs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int w = 0; w < BLK / 64; w++) { a += wsum[0][w]; b += wsum[1][w]; }
        if (a) atomicAdd(&out[0], a);
        if (b) atomicAdd(&out[1], b);
    }
}

// info[s] = (number of occupied segments before s) << 1 | (s is occupied): for an empty segment the first half names the
// next occupied one, which is what a range lookup wants from it
__global__ void __launch_bounds__(BLK) seg_pack_kernel(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ before, size_t nseg, uint32_t *__restrict__ info) {
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < nseg; i += (size_t)gridDim.x * BLK) info[i] = (before[i] << 1) | flags[i];
}

// cell_id[i]: seg_cell_of -> number of the cell among the cells that exist; counts[cell]++ (one atomic per run, as above)
__global__ void __launch_bounds__(BLK) seg_count_kernel(const uint32_t *__restrict__ info, size_t n, uint32_t *__restrict__ cell_id, uint32_t *__restrict__ counts) {
    for (size_t base = (size_t)blockIdx.x * BLK; base < n; base += (size_t)gridDim.x * BLK) {
        const size_t i = base + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0xffffffffu;
        if (active) {
            const uint32_t sc = cell_id[i];
            c = ((info[sc >> SEG_SHIFT] >> 1) << SEG_SHIFT) | (sc & (SEG - 1));
            cell_id[i] = c;
        }
        const WaveRun r = wave_run(c, active);
        if (r.leads) atomicAdd(&counts[c], (uint32_t)r.length);
    }
}

// ---- exact k-NN mean distance ----
// One lane per point (in cell order, so a wave's lanes search the same shells).
// best[] lives in LDS, one column per lane: best[j * QB + lane].
constexpr int QB = 128;

__global__ void __launch_bounds__(QB) knn_mean_dist_kernel(Grid gv, const GridMeta *__restrict__ gm, const float4 *__restrict__ sorted, size_t n,
                                                          const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ cell_count, int k,
                                                          float *__restrict__ dist_out, float *__restrict__ scratch) {
    const Grid g = gm ? gm->g : gv;
    extern __shared__ float best_all[];
    // the lists live in LDS while (k + 1) x 128 floats fit there (k <= 319), in a slab of device memory per workgroup beyond
    // (the reference takes any k: src/cwipc_filters.cpp:197-201); the workgroups walk over the query tiles
    float *best = scratch ? scratch + (size_t)blockIdx.x * (size_t)(k + 1) * QB + threadIdx.x : best_all + threadIdx.x;
    const int want = k + 1;
    for (size_t tile = blockIdx.x; tile * QB < n; tile += gridDim.x) {
    size_t qi = tile * QB + threadIdx.x;
    if (qi >= n) continue;
    const float4 q = sorted[qi];
    const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
    int have = 0;
    float worst = -1.0f;   // largest of the kept distances
    int worst_at = 0;
    const int maxring = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int ring = 0; ring <= maxring; ring++) {
        for (int dz = -ring; dz <= ring; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -ring; dy <= ring; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                const bool face = dz == -ring || dz == ring || dy == -ring || dy == ring;
                const int step = face ? 1 : (ring > 0 ? 2 * ring : 1);
                for (int dx = -ring; dx <= ring; dx += step) {
                    const int x = cx + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    const uint32_t c = (uint32_t)x + (uint32_t)g.dim[0] * ((uint32_t)y + (uint32_t)g.dim[1] * (uint32_t)z);
                    const uint32_t first = cell_start[c], cnt = cell_count[c];
                    for (uint32_t e = first; e < first + cnt; e++) {
                        const float4 p = sorted[e];
                        // FLANN L2_Simple<float>: separately rounded fp32 operations, x,y,z order
                        float d = __fsub_rn(q.x, p.x);
                        float d2 = __fmul_rn(d, d);
                        d = __fsub_rn(q.y, p.y);
                        d2 = __fadd_rn(d2, __fmul_rn(d, d));
                        d = __fsub_rn(q.z, p.z);
                        d2 = __fadd_rn(d2, __fmul_rn(d, d));
                        if (have < want) {
                            best[have * QB] = d2;
                            if (d2 > worst) { worst = d2; worst_at = have; }
                            have++;
                        } else if (d2 < worst) {
                            best[worst_at * QB] = d2;
                            worst = -1.0f;
                            for (int j = 0; j < want; j++) {
                                float v = best[j * QB];
                                if (v > worst) { worst = v; worst_at = j; }
                            }
                        }
                    }
                }
            }
        }
        if (have == want) {
            const double reach = (double)ring * g.h;
            if ((double)worst < reach * reach * (1.0 - 1e-6)) break;
        }
    }
    // ascending order, then the f64 sum of fp32 square roots, skipping the query itself
    for (int a = 1; a < have; a++) {
        float v = best[a * QB];
        int b = a;
        while (b > 0 && best[(b - 1) * QB] > v) { best[b * QB] = best[(b - 1) * QB]; b--; }
        best[b * QB] = v;
    }
    double sum = 0.0;
    for (int j = 1; j < have; j++) sum += (double)sqrtf(best[j * QB]);
    dist_out[__float_as_uint(q.w)] = (float)(sum / (double)k);
    }
}

// Launch of the list variant: LDS for the lists as long as they fit (64 KB by default, up to 160 KB once the limit has been
// raised), a slab of device memory per workgroup beyond.  `slab` receives the pool block to give back when the kernel is done.
bool launch_knn_list(const Grid &gv, const GridMeta *gm, const float4 *sorted, size_t n, const uint32_t *cell_start, const uint32_t *cell_count, int k,
                     float *dist_out, hipStream_t s, void **slab) {
    *slab = nullptr;
    const unsigned qgrid = (unsigned)((n + QB - 1) / QB);
    const size_t shmem = (size_t)(k + 1) * QB * sizeof(float);
    // (a device whose LDS limit cannot be raised that far -- the attribute call fails, or the device reports less -- takes the slab
    // path below for these k as well: any kNeighbors works, as in the reference)
    bool in_lds = shmem <= (size_t)160 * 1024 - 512;
    if (in_lds && shmem > (size_t)64 * 1024) {
        static std::mutex once;
        static int raised_on = -1, refused_on = -1;
        std::lock_guard<std::mutex> g(once);
        const int dev = current_device();
        if (refused_on == dev) {
            in_lds = false;
        } else if (raised_on != dev) {
            int lds_max = 0;
            if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); lds_max = 0; }
            if (lds_max >= 160 * 1024 - 512 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_mean_dist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess) {
                raised_on = dev;
            } else {
                (void)hipGetLastError();
                refused_on = dev;
                in_lds = false;
            }
        }
    }
    if (in_lds) {
        CW_LAUNCH("sor_knn_mean_dist", knn_mean_dist_kernel, dim3(qgrid), dim3(QB), shmem, s, gv, gm, sorted, n, cell_start, cell_count, k, dist_out, (float *)nullptr);
        return true;
    }
    unsigned blocks = std::min(qgrid, 1024u);
    while (blocks > 64 && (size_t)blocks * shmem > ((size_t)4 << 30)) blocks /= 2;
    if ((size_t)blocks * shmem > ((size_t)16 << 30)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_remove_outliers", "kNeighbors is too large for the device memory the candidate lists would take");
        return false;
    }
    float *scratch = (float *)pool_alloc((size_t)blocks * shmem);
    if (!scratch) return false;
    *slab = scratch;
    CW_LAUNCH("sor_knn_mean_dist", knn_mean_dist_kernel, dim3(blocks), dim3(QB), 0, s, gv, gm, sorted, n, cell_start, cell_count, k, dist_out, scratch);
    return true;
}

// (Round 3, measured: a camera tile's 36 k queries take this kernel 39 us whether they sit 64 or 16 to a wave (567 or 2266 waves
// on 1024 SIMDs): a query is ONE lane's chain of dependent loads -- row lookups, candidates four at a time -- and the kernel lasts
// as long as such a chain, however many run side by side.  Eight candidate loads in flight instead of four: 41 us, no change
// either (sixteen, through an array the compiler put into scratch memory: 149 us).  FOUR LANES PER QUERY (each scans every fourth
// candidate into a list of its own, the lists merged by two butterfly steps over the quad, rows and rings turned away on a bound
// the quad shares; d_i bit-identical, all tests green): 50 us.  The rows of a ring beyond the first looked up eight at a time
// instead of one after the other: 44 us.  So it is neither the candidates nor the row lookups one by one: per wave the counters
// say 4400 vector instructions (7 us), 182 loads and 52 % of 22 us waiting, the slowest waves twice that; what the waves wait
// for is not a cold L2 either (the arrays have just been written from all eight XCDs, but the same kernel launched a second
// time right behind the first takes 37.8 us against 42.4).  Not resolved in round 3.)
// The same search with the candidate list in registers (k + 1 <= KCAP): a sorted list kept by a
// compare-exchange chain, no LDS round trips per accepted candidate.  Unused leading slots hold -inf,
// so the largest kept distance is always the last register.
// SPARSE: cell_start is indexed by the cells that exist (one entry more than there are cells: the end), cell_count is the
// segment table (seg_pack_kernel).
// (r4, second session: six waves per SIMD for k <= 16 -- 80 registers, eight of them spilled, and still 6 % faster at 10 M points than five waves without
// a spill: the counters put 54 % of a wave's cycles there into waiting for its loads (SQ_WAIT_ANY; 249 loads, 5300 vector instructions per wave), and what
// hides a wait is another wave.  Eight waves (44 spilled) give it back; requesting the next four candidates before looking at these four changed nothing at
// five waves and cost 8 % at four.  profiles/r04_sor_knn_10m.txt)
template <int KCAP, bool SPARSE>
__global__ void __launch_bounds__(QB) __attribute__((amdgpu_waves_per_eu(KCAP <= 17 ? 6 : 4))) knn_mean_dist_reg_kernel(Grid gv, const GridMeta *__restrict__ gm, const float4 *__restrict__ sorted, size_t n,
                                                              const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ cell_count, int k,
                                                              float *__restrict__ dist_out, const uint32_t *__restrict__ cell_count2 = nullptr) {
    if (cell_count2 && gm[1].refine) { gm += 1; cell_count = cell_count2; }   // the small clouds' flow: the coarser grid's slot and counts
    const Grid g = gm ? gm->g : gv;
    const int want = k + 1, pad = KCAP - want;
    size_t qi = (size_t)blockIdx.x * QB + threadIdx.x;
    if (qi >= n) return;
    const float4 q = sorted[qi];
    const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
    float best[KCAP];
#pragma unroll
    for (int j = 0; j < KCAP; j++) best[j] = j < pad ? -INFINITY : INFINITY;
    int have = 0;
    // one candidate: FLANN L2_Simple<float> distance (separately rounded fp32 operations, x,y,z order), sorted insert
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 qxy = {q.x, q.y};
    auto candidate = [&](const float4 p) {
        // (x and y as one packed subtraction and one packed multiplication: the same separately rounded fp32 operations,
        // summed in the same order)
        const f32x2 dxy = qxy - f32x2{p.x, p.y};
        const f32x2 sq = dxy * dxy;
        const float dz = __fsub_rn(q.z, p.z);
        const float d2 = __fadd_rn(__fadd_rn(sq.x, sq.y), __fmul_rn(dz, dz));
        if (d2 < best[KCAP - 1]) {
            have++;
            // sorted insert, the largest value drops out: the new j-th smallest is the MEDIAN of the old (j-1)-th, the old
            // j-th and the newcomer -- one v_med3_f32 per slot, all from old values (top down, in place), no chain of
            // dependent min / max pairs (which was most of this kernel's instruction count)
#pragma unroll
            for (int j = KCAP - 1; j >= 1; j--) best[j] = __builtin_amdgcn_fmed3f(best[j - 1], best[j], d2);
            best[0] = fminf(best[0], d2);
        }
    };
    // a range of candidates, four loads in flight at a time (the loop is latency-bound otherwise: one dependent 16-byte
    // load per lane and iteration)
    auto scan = [&](uint32_t first, uint32_t last) {
        uint32_t e = first;
        for (; e + 4 <= last; e += 4) {
            const float4 p0 = sorted[e], p1 = sorted[e + 1], p2 = sorted[e + 2], p3 = sorted[e + 3];
            candidate(p0); candidate(p1); candidate(p2); candidate(p3);
        }
        for (; e < last; e++) candidate(sorted[e]);
    };
    // Cells that are neighbours along x are neighbours in `sorted` (the counting sort runs x fastest), so a
    // row of cells x0..x1 is ONE range of points: two index loads per row instead of two per cell.
    auto row_range = [&](int x0, int x1, int y, int z, uint32_t &first, uint32_t &last) {
        if (SPARSE) {
            // the cells of this row that exist, from the first at or after x0 to the last at or before x1: an empty segment's
            // entry names the next segment that exists, whose first cell is where everything before it ends
            const uint32_t rowseg = (uint32_t)g.nsegx * ((uint32_t)y + (uint32_t)g.dim[1] * (uint32_t)z);
            const uint32_t i0 = cell_count[rowseg + ((uint32_t)x0 >> SEG_SHIFT)], i1 = cell_count[rowseg + ((uint32_t)x1 >> SEG_SHIFT)];
            first = cell_start[((i0 >> 1) << SEG_SHIFT) + ((i0 & 1u) ? ((uint32_t)x0 & (SEG - 1)) : 0u)];
            last = cell_start[((i1 >> 1) << SEG_SHIFT) + ((i1 & 1u) ? ((uint32_t)x1 & (SEG - 1)) + 1u : 0u)];
            return;
        }
        const uint32_t base = (uint32_t)g.dim[0] * ((uint32_t)y + (uint32_t)g.dim[1] * (uint32_t)z);
        const uint32_t c1 = base + (uint32_t)x1;
        first = cell_start[base + (uint32_t)x0];
        last = cell_start[c1] + cell_count[c1];
    };
    // rings 0 and 1 together: 9 rows, their index loads issued before any of them is needed
    {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        uint32_t first[9], last[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int y = cy + (r % 3) - 1, z = cz + (r / 3) - 1;
            first[r] = last[r] = 0;
            if (y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) row_range(x0, x1, y, z, first[r], last[r]);
        }
        // the query's own row first; then a row only if it can still hold one of the k + 1 nearest: its nearest edge must be
        // closer than the worst distance kept so far (bound taken a little short: rounding never skips a row that matters)
        const float eps = (float)(g.h * 1e-5), hf = (float)g.h;
        const float ylo = (float)((double)g.mn[1] + (double)cy * g.h), zlo = (float)((double)g.mn[2] + (double)cz * g.h);
        auto gap = [&](float v, float lo_face, int o) {
            const float d = o == 0 ? 0.f : (o < 0 ? v - lo_face : lo_face + hf - v);
            const float t = fmaxf(d - eps, 0.f);
            return t * t;
        };
        scan(first[4], last[4]);
        // (rows that share a face with the query's row before the diagonal ones: the sooner the list holds near points,
        // the more rows and candidates the bound turns away)
        constexpr int order[8] = {1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int r = order[o];
            if (gap(q.y, ylo, r % 3 - 1) + gap(q.z, zlo, r / 3 - 1) >= best[KCAP - 1]) continue;
            scan(first[r], last[r]);
        }
    }
    const int maxring = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int ring = 1; ring <= maxring; ring++) {
        if (ring > 1) {
            // (r4) Shells beyond the first: the queries at a cloud's edge, a few lanes of every wave, and the whole wave waits for them.
            // A row (or an end cell of an inner row) is looked up only if it can still hold one of the k + 1 nearest: the squared distance
            // from the query to the row's cells, from the cells' faces in f64 (the cell of a point is floor((v - mn) / h) in f64 too),
            // against the worst distance kept, with the margin the ring's own exit test below takes.  Without the test a lane in ring 2
            // walked through 34 dependent pairs of loads (row index, candidates), most of them for cells on the far side.
            const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.dim[0] - 1);
            // (the margin goes in once, in f64, before the value is rounded to fp32: what is added up below in fp32 stays under the true
            // distance by more than the three roundings of the candidates' own fp32 distances)
            auto gap2 = [&](float v, int a, int cell, int o) -> float {   // squared distance from v to the cells `o` cells away from `cell` on axis a
                if (o == 0) return 0.f;
                const double face = (double)g.mn[a] + (double)(o < 0 ? cell + o + 1 : cell + o) * g.h;
                const double d = o < 0 ? (double)v - face : face - (double)v;
                return d > 0.0 ? (float)(d * d * (1.0 - 1e-6)) : 0.f;
            };
            const float gx_lo = gap2(q.x, 0, cx, -ring), gx_hi = gap2(q.x, 0, cx, ring);
            for (int dz = -ring; dz <= ring; dz++) {
                const int z = cz + dz;
                if (z < 0 || z >= g.dim[2]) continue;
                const float gz = gap2(q.z, 2, cz, dz);
                for (int dy = -ring; dy <= ring; dy++) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.dim[1]) continue;
                    const float gyz = gz + gap2(q.y, 1, cy, dy);
                    if (gyz >= best[KCAP - 1]) continue;
                    const bool face = dz == -ring || dz == ring || dy == -ring || dy == ring;
                    uint32_t first, last;
                    if (face) {   // the whole row belongs to the shell
                        row_range(x0, x1, y, z, first, last);
                        scan(first, last);
                    } else {      // only its two end cells do
                        if (cx - ring >= 0 && gyz + gx_lo < best[KCAP - 1]) {
                            row_range(cx - ring, cx - ring, y, z, first, last);
                            scan(first, last);
                        }
                        if (cx + ring < g.dim[0] && gyz + gx_hi < best[KCAP - 1]) {
                            row_range(cx + ring, cx + ring, y, z, first, last);
                            scan(first, last);
                        }
                    }
                }
            }
        }
        if (have >= want) {
            const double reach = (double)ring * g.h;
            if ((double)best[KCAP - 1] < reach * reach * (1.0 - 1e-6)) break;
        }
    }
    // the f64 sum of fp32 square roots in ascending order, skipping the query itself (the smallest)
    double sum = 0.0;
#pragma unroll
    for (int j = 1; j < KCAP; j++) {
        if (j > pad && best[j] < INFINITY) sum += (double)sqrtf(best[j]);
    }
    dist_out[__float_as_uint(q.w)] = (float)(sum / (double)k);
}

// ---- two lanes per query (r4, second session): small clouds, k = 16 ----
// A camera tile's 36 k queries are half a wave per SIMD: the kernel lasts as long as ONE query's chain -- ~76 candidates four at a time, a sorted insert
// for most of them -- however many queries run side by side.  Here two neighbouring lanes share a query: the same rows, the same ranges, but lane h takes
// every second candidate (first + h, + 2, ...) into a sorted list of its own, so the chain is half as long.  What made round 3's four-lane version slower
// (50 against 39 us) was merging the lists as sorted lists, 17 inserts per butterfly step; a query needs less:
//   * the k + 1 nearest of the pair's candidates are, as a SET, c[j] = min(a[j], b[16 - j]) over the two sorted lists (the first step of a bitonic merge):
//     seventeen DPP swaps and minima, and (k + 1)-th distance so far = max_j c[j] -- the bound that turns rows, rings and candidates away, refreshed
//     after the query's own row, after the face rows and at every ring's end;
//   * d_i = the f64 sum of the sixteen square roots in ascending order, and an f64 sum of seventeen fp32 values whose exponents lie within 2^23 of each
//     other is EXACT in any order (every partial sum is a multiple of the smallest term's unit below 2^53 of it): the set is summed as it stands, minus
//     the smallest term (the query itself); a query whose distances spread further (coincident points next to far ones) sorts its set first.
// Same d_i bit for bit (the outlier tests run through this kernel for k = 16 on small clouds).  Dense layout only, k + 1 = 17.
__device__ __forceinline__ float pair_swap(float v) {   // the other lane of the pair's value (quad_perm [1, 0, 3, 2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ int pair_swap_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }

__global__ void __launch_bounds__(QB) knn_pair_kernel(const GridMeta *__restrict__ gm, const float4 *__restrict__ sorted, size_t n, const uint32_t *__restrict__ cell_start,
                                                     const uint32_t *__restrict__ cell_count, const uint32_t *__restrict__ cell_count2, float *__restrict__ dist_out) {
    constexpr int KCAP = 17, k = 16;
    if (cell_count2 && gm[1].refine) { gm += 1; cell_count = cell_count2; }
    const Grid g = gm->g;
    const size_t qi0 = ((size_t)blockIdx.x * QB + threadIdx.x) >> 1;
    const uint32_t half = threadIdx.x & 1u;
    const bool real = qi0 < n;
    const size_t qi = real ? qi0 : n - 1;   // (lanes beyond the cloud run the last query along, so that every pair is whole: they write nothing)
    const float4 q = sorted[qi];
    const int cx = cell_coord(g, q.x, 0), cy = cell_coord(g, q.y, 1), cz = cell_coord(g, q.z, 2);
    float best[KCAP];
#pragma unroll
    for (int j = 0; j < KCAP; j++) best[j] = INFINITY;
    int have = 0;
    float thr = INFINITY;   // nothing at this distance or beyond can be among the pair's k + 1 nearest
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 qxy = {q.x, q.y};
    auto candidate = [&](const float4 p) {
        const f32x2 dxy = qxy - f32x2{p.x, p.y};
        const f32x2 sq = dxy * dxy;
        const float dz = __fsub_rn(q.z, p.z);
        const float d2 = __fadd_rn(__fadd_rn(sq.x, sq.y), __fmul_rn(dz, dz));
        if (d2 < thr) {
            have++;
#pragma unroll
            for (int j = KCAP - 1; j >= 1; j--) best[j] = __builtin_amdgcn_fmed3f(best[j - 1], best[j], d2);
            best[0] = fminf(best[0], d2);
            thr = fminf(thr, best[KCAP - 1]);
        }
    };
    // this lane's half of a range: first + half, every second one, four loads in flight
    auto scan = [&](uint32_t first, uint32_t last) {
        uint32_t e = first + half;
        for (; e + 6 < last; e += 8) {
            const float4 p0 = sorted[e], p1 = sorted[e + 2], p2 = sorted[e + 4], p3 = sorted[e + 6];
            candidate(p0); candidate(p1); candidate(p2); candidate(p3);
        }
        for (; e < last; e += 2) candidate(sorted[e]);
    };
    // the pair's (k + 1)-th distance so far, from the two lists
    auto refresh = [&]() {
        float mk = -INFINITY;
#pragma unroll
        for (int j = 0; j < KCAP; j++) mk = fmaxf(mk, fminf(best[j], pair_swap(best[KCAP - 1 - j])));
        thr = fminf(thr, mk);
    };
    auto row_range = [&](int x0, int x1, int y, int z, uint32_t &first, uint32_t &last) {
        const uint32_t base = (uint32_t)g.dim[0] * ((uint32_t)y + (uint32_t)g.dim[1] * (uint32_t)z);
        const uint32_t c1 = base + (uint32_t)x1;
        first = cell_start[base + (uint32_t)x0];
        last = cell_start[c1] + cell_count[c1];
    };
    {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        uint32_t first[9], last[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int y = cy + (r % 3) - 1, z = cz + (r / 3) - 1;
            first[r] = last[r] = 0;
            if (y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2]) row_range(x0, x1, y, z, first[r], last[r]);
        }
        const float eps = (float)(g.h * 1e-5), hf = (float)g.h;
        const float ylo = (float)((double)g.mn[1] + (double)cy * g.h), zlo = (float)((double)g.mn[2] + (double)cz * g.h);
        auto gap = [&](float v, float lo_face, int o) {
            const float d = o == 0 ? 0.f : (o < 0 ? v - lo_face : lo_face + hf - v);
            const float t = fmaxf(d - eps, 0.f);
            return t * t;
        };
        scan(first[4], last[4]);
        refresh();
        constexpr int order[8] = {1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int r = order[o];
            if (o == 4) refresh();
            if (gap(q.y, ylo, r % 3 - 1) + gap(q.z, zlo, r / 3 - 1) >= thr) continue;
            scan(first[r], last[r]);
        }
    }
    const int maxring = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int ring = 1; ring <= maxring; ring++) {
        if (ring > 1) {
            const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.dim[0] - 1);
            auto gap2 = [&](float v, int a, int cell, int o) -> float {   // (as in knn_mean_dist_reg_kernel)
                if (o == 0) return 0.f;
                const double face = (double)g.mn[a] + (double)(o < 0 ? cell + o + 1 : cell + o) * g.h;
                const double d = o < 0 ? (double)v - face : face - (double)v;
                return d > 0.0 ? (float)(d * d * (1.0 - 1e-6)) : 0.f;
            };
            const float gx_lo = gap2(q.x, 0, cx, -ring), gx_hi = gap2(q.x, 0, cx, ring);
            for (int dz = -ring; dz <= ring; dz++) {
                const int z = cz + dz;
                if (z < 0 || z >= g.dim[2]) continue;
                const float gz = gap2(q.z, 2, cz, dz);
                for (int dy = -ring; dy <= ring; dy++) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.dim[1]) continue;
                    const float gyz = gz + gap2(q.y, 1, cy, dy);
                    if (gyz >= thr) continue;
                    const bool face = dz == -ring || dz == ring || dy == -ring || dy == ring;
                    uint32_t first, last;
                    if (face) {
                        row_range(x0, x1, y, z, first, last);
                        scan(first, last);
                    } else {
                        if (cx - ring >= 0 && gyz + gx_lo < thr) {
                            row_range(cx - ring, cx - ring, y, z, first, last);
                            scan(first, last);
                        }
                        if (cx + ring < g.dim[0] && gyz + gx_hi < thr) {
                            row_range(cx + ring, cx + ring, y, z, first, last);
                            scan(first, last);
                        }
                    }
                }
            }
        }
        refresh();
        // (both lanes of a pair hold the same thr and the same sum of `have`: they leave together)
        if (have + pair_swap_i(have) >= KCAP) {
            const double reach = (double)ring * g.h;
            if ((double)thr < reach * reach * (1.0 - 1e-6)) break;
        }
    }
    // the pair's k + 1 nearest as a set
    float c[KCAP];
#pragma unroll
    for (int j = 0; j < KCAP; j++) c[j] = fminf(best[j], pair_swap(best[KCAP - 1 - j]));
    float smallest = INFINITY, largest = 0.f, least = INFINITY;   // least: the smallest distance that is not zero
#pragma unroll
    for (int j = 0; j < KCAP; j++) {
        smallest = fminf(smallest, c[j]);
        if (c[j] < INFINITY) largest = fmaxf(largest, c[j]);
        if (c[j] > 0.f) least = fminf(least, c[j]);
    }
    const bool exact = !(least < INFINITY) || sqrtf(largest) < sqrtf(least) * 8388608.f;
    double sum = 0.0;
    if (__builtin_expect(__ballot(!exact) != 0ull, 0)) {
        // distances spread over more than 2^23: the order of the sum may matter -- ascending, as the reference has it
        float srt[KCAP];
#pragma unroll
        for (int j = 0; j < KCAP; j++) srt[j] = INFINITY;
#pragma unroll 1
        for (int i = 0; i < KCAP; i++) {
            float v = c[0];
#pragma unroll
            for (int j = 1; j < KCAP; j++) v = i == j ? c[j] : v;
#pragma unroll
            for (int j = KCAP - 1; j >= 1; j--) srt[j] = __builtin_amdgcn_fmed3f(srt[j - 1], srt[j], v);
            srt[0] = fminf(srt[0], v);
        }
#pragma unroll
        for (int j = 1; j < KCAP; j++) if (srt[j] < INFINITY) sum += (double)sqrtf(srt[j]);
    } else {
#pragma unroll
        for (int j = 0; j < KCAP; j++) if (c[j] < INFINITY) sum += (double)sqrtf(c[j]);
        if (smallest < INFINITY) sum -= (double)sqrtf(smallest);
    }
    if (real && half == 0u) dist_out[__float_as_uint(q.w)] = (float)(sum / (double)k);
}

// (Round 4, built, measured and taken out again -- the commit before this comment's has the kernels, profiles/r04_sor_knn_staged.txt the figures:
// the neighbourhood of 64 consecutive sorted points staged in LDS by the whole wave -- the nine runs of cells [ca - 1, cb + 1] shifted by the row
// offsets, one round trip -- and every lane scanning its own three cells of each row there; before it, a wave per row of cells.  d_i bit-identical;
// 43.0 us against 42.3 for a 36 k-point camera tile, the same at every cell size from 4 to 50 points per cell (a wave per row: 61-67 us).  The
// kernel does not wait for its candidates: its time is the sorted insert -- 17 medians, run by the whole wave whenever ONE lane accepts, ~1400 of
// ~4400 vector instructions per wave at one wave per SIMD -- and the second ring of the queries at the cloud's edge, which every wave has.)

// ---- mean / variance, deterministic two-level sum ----
__global__ void __launch_bounds__(BLK) stats_partial_kernel(const float *__restrict__ d, size_t n, double *__restrict__ partial) {
    __shared__ double red[2][BLK / 64];
    double s = 0, q = 0;
    // contiguous slice per workgroup, strided by lane inside it: fixed order for a fixed launch shape
    size_t per = (n + gridDim.x - 1) / gridDim.x;
    size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += BLK) {
        float v = d[i];
        s += (double)v;
        q += (double)__fmul_rn(v, v);   // "distance * distance" is an fp32 product upstream
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        q += __shfl_down(q, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        for (int w = 0; w < BLK / 64; w++) { ts += red[0][w]; tq += red[1][w]; }
        partial[2 * blockIdx.x] = ts;
        partial[2 * blockIdx.x + 1] = tq;
    }
}

static inline unsigned grid_for(size_t n) {
    size_t g = (n + BLK - 1) / BLK;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (unsigned)g;
}

// The dense layout, driven from the device: box -> grid -> census -> (coarser grid, second count) -> counting sort -> k-NN,
// sixteen launches and no wait (the caller has one further down, behind the compaction).  Arrays are sized for the largest
// grid the rules allow (a few cells per point), whatever the kernels then decide.
bool sor_dense_on_device(const DeviceSoA &src, int k, float *dev_dist, float *partial, unsigned nb, ThreadCtx &c) {
    const size_t n = src.npoints;
    static const size_t cells_per_point = []() { const char *e = getenv("CWIPC_SOR_CELLS_PER_POINT"); return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)8; }();   // tuning knob
    const size_t cap = std::min<size_t>(MAX_CELLS, std::max<size_t>((size_t)1 << 16, cells_per_point * n));
    double target = (double)(k + 1) / 2.0;
    if (const char *t = getenv("CWIPC_SOR_CELL_TARGET")) target = (double)(k + 1) * atof(t);   // tuning knob: points per occupied cell / (k + 1)
    GridMeta *meta = (GridMeta *)pool_alloc(sizeof(GridMeta));
    uint32_t *counts = (uint32_t *)pool_alloc(cap * sizeof(uint32_t));
    uint32_t *starts = (uint32_t *)pool_alloc(cap * sizeof(uint32_t));
    uint32_t *cursor = (uint32_t *)pool_alloc(cap * sizeof(uint32_t));
    uint32_t *cell_id = (uint32_t *)pool_alloc(n * sizeof(uint32_t));
    float4 *sorted = (float4 *)pool_alloc(n * sizeof(float4));
    void *scan_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, counts, starts, 0u, cap, rocprim::plus<uint32_t>(), c.stream);
    if (e == hipSuccess) scan_tmp = pool_alloc(tmp_bytes ? tmp_bytes : 256);
    auto give_back = [&](bool later) {
        void *all[] = {partial, meta, counts, starts, cursor, cell_id, sorted, scan_tmp};
        for (void *b : all) { if (later) c.free_later(b); else pool_free(b); }
    };
    if (e != hipSuccess || !meta || !counts || !starts || !cursor || !cell_id || !sorted || !scan_tmp) {
        (void)c.sync();
        give_back(false);
        return hip_failed(e != hipSuccess ? e : hipErrorOutOfMemory, "sor workspace", __FILE__, __LINE__);
    }
    const Grid unused{};
    const unsigned cap_grid = std::min(1024u, grid_for(cap / 4 + 1));
    // (round 3: eleven launches instead of seventeen -- the two memsets ride with the grid's set-up, the census with the first
    // count, the coarsening with the clearing it asks for; a camera tile's kernels cost the device less than their launches
    // cost the host)
    CW_LAUNCH("sor_grid_setup", grid_setup_zero_kernel, dim3(cap_grid), dim3(BLK), 0, c.stream, partial, nb, cap, meta, counts, cursor);
    bool ok = true;
    if (ok) {
        CW_LAUNCH("sor_cell_count", cell_count_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, unused, meta, 0, src.x(), src.y(), src.z(), n, counts, cell_id, &meta->occ);
        CW_LAUNCH("sor_grid_refine", grid_refine_zero_kernel, dim3(cap_grid), dim3(BLK), 0, c.stream, meta, n, target, counts, cap);
        CW_LAUNCH("sor_cell_count", cell_count_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, unused, meta, 1, src.x(), src.y(), src.z(), n, counts, cell_id, (uint32_t *)nullptr);
        if (profiling_enabled()) profile_begin("sor_exclusive_scan", c.stream);
        e = rocprim::exclusive_scan(scan_tmp, tmp_bytes, counts, starts, 0u, cap, rocprim::plus<uint32_t>(), c.stream);
        if (profiling_enabled()) profile_end(c.stream);
        ok = e == hipSuccess;
    }
    if (ok) {
        CW_LAUNCH("sor_cell_scatter", cell_scatter_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, cell_id, starts, cursor,
                  sorted);
        const unsigned qgrid = (unsigned)((n + QB - 1) / QB);
        if (k + 1 <= 17) {
            CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<17, false>), dim3(qgrid), dim3(QB), 0, c.stream, unused, meta, sorted, n, starts, counts, k, dev_dist);
        } else if (k + 1 <= 33) {
            CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<33, false>), dim3(qgrid), dim3(QB), 0, c.stream, unused, meta, sorted, n, starts, counts, k, dev_dist);
        } else {
            void *slab = nullptr;
            ok = launch_knn_list(unused, meta, sorted, n, starts, counts, k, dev_dist, c.stream, &slab);
            c.free_later(slab);
        }
    }
    ok = hipGetLastError() == hipSuccess && ok;
    if (!ok) {
        hip_failed(e != hipSuccess ? e : hipGetLastError(), "sor k-NN", __FILE__, __LINE__);
        (void)c.sync();
        give_back(false);
        return false;
    }
    give_back(true);
    return true;
}

// The same for small clouds (cap <= 2^19 cells: up to 64 k points; k + 1 <= 33): ten launches with the compaction behind it, see small_bbox_zero_kernel.
bool sor_small_on_device(const DeviceSoA &src, int k, float *dev_dist, size_t cap, ThreadCtx &c) {
    const size_t n = src.npoints;
    double target = (double)(k + 1) / 2.0;
    if (const char *t = getenv("CWIPC_SOR_CELL_TARGET")) target = (double)(k + 1) * atof(t);   // tuning knob: points per occupied cell / (k + 1)
    const unsigned nb = std::min(256u, grid_for(n));
    float *partial = (float *)pool_alloc((size_t)nb * 6 * sizeof(float));
    GridMeta *meta = (GridMeta *)pool_alloc(2 * sizeof(GridMeta));
    uint32_t *words = (uint32_t *)pool_alloc(3 * cap * sizeof(uint32_t));   // counts | counts of the coarser grid | the scatter's cursor
    uint32_t *starts = (uint32_t *)pool_alloc(cap * sizeof(uint32_t));
    uint32_t *cell_id = (uint32_t *)pool_alloc(n * sizeof(uint32_t));
    float4 *sorted = (float4 *)pool_alloc(n * sizeof(float4));
    auto give_back = [&](bool later) {
        void *all[] = {partial, meta, words, starts, cell_id, sorted};
        for (void *b : all) { if (later) c.free_later(b); else pool_free(b); }
    };
    if (!partial || !meta || !words || !starts || !cell_id || !sorted) {
        (void)c.sync();
        give_back(false);
        return hip_failed(hipErrorOutOfMemory, "sor workspace", __FILE__, __LINE__);
    }
    uint32_t *counts = words, *counts2 = words + cap, *cursor = words + 2 * cap;
    const Grid unused{};
    const unsigned pgrid = grid_for(n);
    CW_LAUNCH("sor_bbox", small_bbox_zero_kernel, dim3(nb), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, partial, words, 3 * cap, meta);
    CW_LAUNCH("sor_cell_count", small_count_kernel<0>, dim3(pgrid), dim3(BLK), 0, c.stream, partial, nb, cap, target, meta, src.x(), src.y(), src.z(), n, counts, cell_id);
    CW_LAUNCH("sor_cell_count", small_count_kernel<1>, dim3(pgrid), dim3(BLK), 0, c.stream, partial, nb, cap, target, meta, src.x(), src.y(), src.z(), n, counts2, cell_id);
    CW_LAUNCH("sor_exclusive_scan", small_scan_kernel, dim3(1), dim3(SCAN1_THREADS), 0, c.stream, meta, counts, counts2, starts, cap);
    CW_LAUNCH("sor_cell_scatter", cell_scatter_kernel, dim3(pgrid), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, cell_id, starts, cursor, sorted);
    const unsigned qgrid = (unsigned)((n + QB - 1) / QB);
    static const bool pair_off = []() { const char *e = getenv("CWIPC_SOR_PAIR"); return e && atoi(e) == 0; }();   // test knob: a lane per query
    if (k == 16 && !pair_off) {
        CW_LAUNCH("sor_knn_mean_dist", knn_pair_kernel, dim3((unsigned)((2 * n + QB - 1) / QB)), dim3(QB), 0, c.stream, meta, sorted, n, starts, counts, counts2, dev_dist);
    } else if (k + 1 <= 17) {
        CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<17, false>), dim3(qgrid), dim3(QB), 0, c.stream, unused, meta, sorted, n, starts, counts, k, dev_dist, counts2);
    } else {
        CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<33, false>), dim3(qgrid), dim3(QB), 0, c.stream, unused, meta, sorted, n, starts, counts, k, dev_dist, counts2);
    }
    if (hipGetLastError() != hipSuccess) {
        hip_failed(hipGetLastError(), "sor k-NN", __FILE__, __LINE__);
        (void)c.sync();
        give_back(false);
        return false;
    }
    give_back(true);
    return true;
}

}  // namespace

bool sor_mean_distances(const DeviceSoA &src, int k, float *dev_dist) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return false;
    const size_t n = src.npoints;
    if (n == 0) return true;
    if (k < 1) {
        CW_HIP_TRY(hipMemsetAsync(dev_dist, 0, n * sizeof(float), c.stream));
        return c.sync();
    }
    // (any k, as the reference: lists in registers up to k = 32, in LDS up to k = 319, in device memory beyond: launch_knn_list)

    static const int sparse_knob = []() { const char *e = getenv("CWIPC_SOR_SPARSE"); return e ? atoi(e) : -1; }();   // test knob: 1 always, 0 never
    const bool sparse = (sparse_knob == 1 || (sparse_knob != 0 && n >= ((size_t)1 << 20))) && k + 1 <= 33;
    static const bool host_grid = []() { const char *e = getenv("CWIPC_SOR_HOST_GRID"); return e && atoi(e) != 0; }();   // test knob: the dense layout decided by the host
    if (!sparse && !host_grid && k + 1 <= 33) {
        // (r4) small clouds: two launches fewer and a one-workgroup scan over the cells the grid really has
        static const size_t cells_per_point = []() { const char *e = getenv("CWIPC_SOR_CELLS_PER_POINT"); return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)8; }();   // tuning knob
        static const size_t small_cells = []() { const char *e = getenv("CWIPC_SOR_SMALL_CELLS"); return e ? (size_t)atol(e) : (size_t)1 << 19; }();   // 0: never (test knob)
        const size_t cap = std::min<size_t>(MAX_CELLS, std::max<size_t>((size_t)1 << 16, cells_per_point * n));
        if (cap <= small_cells) return sor_small_on_device(src, k, dev_dist, cap, c);
    }
    // 1. bounding box
    const unsigned nb = grid_for(n);
    float *partial = (float *)pool_alloc((size_t)nb * 6 * sizeof(float));
    if (!partial) return false;
    CW_LAUNCH("sor_bbox", bbox_kernel, dim3(nb), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, partial);
    if (!sparse && !host_grid) return sor_dense_on_device(src, k, dev_dist, partial, nb, c);
    float *hpart = (float *)c.staging((size_t)nb * 6 * sizeof(float));
    bool ok = hpart && hipMemcpyAsync(hpart, partial, (size_t)nb * 6 * sizeof(float), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    ok = c.sync() && ok;
    pool_free(partial);
    if (!ok) return false;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (unsigned b = 0; b < nb; b++)
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], hpart[b * 6 + a]);
            mx[a] = fmaxf(mx[a], hpart[b * 6 + 3 + a]);
        }
    double ext[3], maxext = 0;
    for (int a = 0; a < 3; a++) {
        ext[a] = (double)mx[a] - (double)mn[a];
        if (!(ext[a] >= 0)) ext[a] = 0;   // no finite point
        if (ext[a] > maxext) maxext = ext[a];
    }
    if (!(maxext > 0)) maxext = 1.0;

    auto make_grid = [&](double h) {
        Grid g;
        for (int a = 0; a < 3; a++) {
            g.mn[a] = mn[a] == FLT_MAX ? 0.f : mn[a];
            g.dim[a] = (int)floor(ext[a] / h) + 1;
        }
        g.h = h;
        g.inv_h = 1.0 / h;
        g.nsegx = (g.dim[0] + SEG - 1) / SEG;
        return g;
    };
    auto cells_of = [](const Grid &g) { return (size_t)g.dim[0] * (size_t)g.dim[1] * (size_t)g.dim[2]; };
    // ---- big clouds: the sparse layout (segments of 16 cells, only those that hold points) ----
    if (sparse) {
        static const size_t sparse_cpp = []() { const char *e = getenv("CWIPC_SOR_SPARSE_CELLS_PER_POINT"); return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)16; }();
        // cells of the (virtual) fine grid: a few dozen per point, and segment numbers must fit 27 bits
        const size_t budget = std::min<size_t>((size_t)1 << 30, std::max<size_t>((size_t)1 << 16, sparse_cpp * n));
        auto segs_of = [](const Grid &gg) { return (size_t)gg.nsegx * (size_t)gg.dim[1] * (size_t)gg.dim[2]; };
        double hs = maxext / 2048.0;
        while (cells_of(make_grid(hs)) > budget || segs_of(make_grid(hs)) >= ((size_t)1 << 27)) hs *= 1.25;
        Grid g = make_grid(hs);
        size_t nseg = segs_of(g);
        uint32_t *cell_id = (uint32_t *)pool_alloc(n * sizeof(uint32_t));
        float4 *sorted = (float4 *)pool_alloc(n * sizeof(float4));
        uint32_t *masks = nullptr, *flags = nullptr, *before = nullptr, *info = nullptr, *counts = nullptr, *starts = nullptr, *cursor = nullptr;
        void *scan_tmp = nullptr;
        auto give_back = [&](bool later) {
            void *all[] = {cell_id, sorted, masks, flags, before, info, counts, starts, cursor, scan_tmp};
            for (void *b : all) { if (later) c.free_later(b); else pool_free(b); }
        };
        auto fail = [&]() { (void)c.sync(); give_back(false); return false; };
        if (!cell_id || !sorted) return fail();
        uint32_t occ_cells = 0, occ_segs = 0;
        auto census = [&]() -> bool {
            pool_free(masks); pool_free(flags);
            masks = (uint32_t *)pool_alloc(nseg * sizeof(uint32_t) + 256);
            flags = (uint32_t *)pool_alloc(nseg * sizeof(uint32_t));
            if (!masks || !flags) return false;
            uint32_t *out = masks + nseg;
            bool good = hipMemsetAsync(masks, 0, nseg * sizeof(uint32_t) + 8, c.stream) == hipSuccess;
            if (!good) return false;
            CW_LAUNCH("sor_seg_mark", seg_mark_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, g, src.x(), src.y(), src.z(), n, masks, cell_id);
            CW_LAUNCH("sor_seg_census", seg_census_kernel, dim3(std::min(1024u, grid_for(nseg))), dim3(BLK), 0, c.stream, masks, nseg, flags, out);
            good = hipMemcpyAsync(c.host_words, out, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
            good = c.sync() && good;
            occ_cells = c.host_words[0];
            occ_segs = c.host_words[1];
            return good;
        };
        if (!census()) return fail();
        {
            // coarsen so that an occupied cell holds about 0.3 (k + 1) points (surface-like data: points per cell grow with h^2).
            // (r4: 0.5 (k + 1) until the shells beyond the first got their bound per row; with it finer cells pay: 2 M points 0.71 -> 0.66 ms,
            // profiles/r04_sor_small_flow.txt.  10 M points are at the segment budget's cell size either way.)
            const double ppc = (double)n / (double)(occ_cells ? occ_cells : 1);
            double target = 0.3 * (double)(k + 1);
            if (const char *t = getenv("CWIPC_SOR_CELL_TARGET")) target = (double)(k + 1) * atof(t);
            if (ppc < target) {
                double h = hs * sqrt(target / ppc);
                if (h > maxext) h = maxext;
                g = make_grid(h);
                nseg = segs_of(g);
                if (!census()) return fail();
            }
        }
        const size_t ncomp = (size_t)occ_segs << SEG_SHIFT;
        before = (uint32_t *)pool_alloc(nseg * sizeof(uint32_t));
        info = (uint32_t *)pool_alloc(nseg * sizeof(uint32_t));
        counts = (uint32_t *)pool_alloc((ncomp + 1) * sizeof(uint32_t));
        starts = (uint32_t *)pool_alloc((ncomp + 1) * sizeof(uint32_t));
        cursor = (uint32_t *)pool_alloc((ncomp + 1) * sizeof(uint32_t));
        if (!before || !info || !counts || !starts || !cursor) return fail();
        size_t tmp_a = 0, tmp_b = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, tmp_a, flags, before, 0u, nseg, rocprim::plus<uint32_t>(), c.stream);
        if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, tmp_b, counts, starts, 0u, ncomp + 1, rocprim::plus<uint32_t>(), c.stream);
        const size_t tmp_bytes = std::max(tmp_a, tmp_b);
        if (e == hipSuccess) {
            scan_tmp = pool_alloc(tmp_bytes ? tmp_bytes : 256);
            if (!scan_tmp) e = hipErrorOutOfMemory;
        }
        if (e != hipSuccess) { hip_failed(e, "rocprim::exclusive_scan", __FILE__, __LINE__); return fail(); }
        if (profiling_enabled()) profile_begin("sor_exclusive_scan", c.stream);
        e = rocprim::exclusive_scan(scan_tmp, tmp_a, flags, before, 0u, nseg, rocprim::plus<uint32_t>(), c.stream);
        if (profiling_enabled()) profile_end(c.stream);
        bool ok = e == hipSuccess;
        if (ok) CW_LAUNCH("sor_seg_pack", seg_pack_kernel, dim3(std::min(2048u, grid_for(nseg))), dim3(BLK), 0, c.stream, flags, before, nseg, info);
        ok = ok && hipMemsetAsync(counts, 0, (ncomp + 1) * sizeof(uint32_t), c.stream) == hipSuccess &&
             hipMemsetAsync(cursor, 0, (ncomp + 1) * sizeof(uint32_t), c.stream) == hipSuccess;
        if (ok) {
            CW_LAUNCH("sor_cell_count", seg_count_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, info, n, cell_id, counts);
            if (profiling_enabled()) profile_begin("sor_exclusive_scan", c.stream);
            e = rocprim::exclusive_scan(scan_tmp, tmp_b, counts, starts, 0u, ncomp + 1, rocprim::plus<uint32_t>(), c.stream);
            if (profiling_enabled()) profile_end(c.stream);
            ok = e == hipSuccess;
        }
        if (ok) {
            CW_LAUNCH("sor_cell_scatter", cell_scatter_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, cell_id, starts,
                      cursor, sorted);
            const unsigned qgrid = (unsigned)((n + QB - 1) / QB);
            if (k + 1 <= 17) {
                CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<17, true>), dim3(qgrid), dim3(QB), 0, c.stream, g, (const GridMeta *)nullptr, sorted, n, starts, info, k, dev_dist);
            } else {
                CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<33, true>), dim3(qgrid), dim3(QB), 0, c.stream, g, (const GridMeta *)nullptr, sorted, n, starts, info, k, dev_dist);
            }
        }
        ok = hipGetLastError() == hipSuccess && ok;
        if (!ok) { hip_failed(e != hipSuccess ? e : hipGetLastError(), "sor k-NN (sparse grid)", __FILE__, __LINE__); return fail(); }
        give_back(true);   // (no wait here: every caller has one further down, and the temporaries go back to the pool there)
        return true;
    }

    // finest cell size whose dense grid stays within MAX_CELLS
    // (and, for small clouds, within a few cells per point: the probe is a pass over the grid)
    static const size_t cells_per_point = []() { const char *e = getenv("CWIPC_SOR_CELLS_PER_POINT"); return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)8; }();   // tuning knob
    const size_t probe_cells = std::min<size_t>(MAX_CELLS, std::max<size_t>((size_t)1 << 16, cells_per_point * n));
    double h_min = maxext / 1024.0;
    while (cells_of(make_grid(h_min)) > probe_cells) h_min *= 1.25;

    uint32_t *counts = (uint32_t *)pool_alloc(probe_cells * sizeof(uint32_t) + 256);
    uint32_t *fill = (uint32_t *)pool_alloc(probe_cells * sizeof(uint32_t));
    uint32_t *cell_id = (uint32_t *)pool_alloc(n * sizeof(uint32_t));
    float4 *sorted = (float4 *)pool_alloc(n * sizeof(float4));
    void *scan_tmp = nullptr;
    auto cleanup = [&]() { pool_free(counts); pool_free(fill); pool_free(cell_id); pool_free(sorted); pool_free(scan_tmp); };
    if (!counts || !fill || !cell_id || !sorted) { cleanup(); return false; }

    // 2. occupancy probe at h_min, then coarsen so that an occupied cell holds about (k+1)/3 points
    //    (surface-like data: points per cell grow with h^2)
    Grid g = make_grid(h_min);
    size_t ncells = cells_of(g);
    uint32_t *occ_dev = counts + probe_cells;
    ok = hipMemsetAsync(counts, 0, ncells * sizeof(uint32_t), c.stream) == hipSuccess &&
         hipMemsetAsync(occ_dev, 0, sizeof(uint32_t), c.stream) == hipSuccess;
    if (ok) {
        CW_LAUNCH("sor_cell_count", cell_count_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, g, (const GridMeta *)nullptr, 0, src.x(), src.y(), src.z(), n, counts, cell_id, (uint32_t *)nullptr);
        CW_LAUNCH("sor_count_nonzero", count_nonzero_kernel, dim3(std::min(1024u, grid_for(ncells / 4 + 1))), dim3(BLK), 0, c.stream, counts, ncells, occ_dev);
        ok = hipMemcpyAsync(c.host_words, occ_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    }
    ok = c.sync() && ok;
    if (!ok) { cleanup(); return false; }
    double ppc = (double)n / (double)(c.host_words[0] ? c.host_words[0] : 1);
    double target = (double)(k + 1) / 2.0;
    if (const char *t = getenv("CWIPC_SOR_CELL_TARGET")) target = (double)(k + 1) * atof(t);   // tuning knob: points per occupied cell / (k + 1)
    if (ppc < target) {
        double h = h_min * sqrt(target / ppc);
        if (h > maxext) h = maxext;
        g = make_grid(h);
        ncells = cells_of(g);
        ok = hipMemsetAsync(counts, 0, ncells * sizeof(uint32_t), c.stream) == hipSuccess;
        if (ok) CW_LAUNCH("sor_cell_count", cell_count_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, g, (const GridMeta *)nullptr, 0, src.x(), src.y(), src.z(), n, counts, cell_id, (uint32_t *)nullptr);
    }

    // 3. counting sort: exclusive scan of the counts, scatter
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, counts, fill, 0u, ncells, rocprim::plus<uint32_t>(), c.stream);
    if (e == hipSuccess) {
        scan_tmp = pool_alloc(tmp_bytes ? tmp_bytes : 256);
        if (!scan_tmp) e = hipErrorOutOfMemory;
    }
    if (e == hipSuccess) {
        if (profiling_enabled()) profile_begin("sor_exclusive_scan", c.stream);
        // fill <- starts ; counts stays ; a second buffer then serves as the fill cursor
        e = rocprim::exclusive_scan(scan_tmp, tmp_bytes, counts, fill, 0u, ncells, rocprim::plus<uint32_t>(), c.stream);
        if (profiling_enabled()) profile_end(c.stream);
    }
    if (e != hipSuccess || !ok) {
        hip_failed(e, "rocprim::exclusive_scan", __FILE__, __LINE__);
        cleanup();
        return false;
    }
    uint32_t *starts = fill;
    uint32_t *cursor = (uint32_t *)pool_alloc(ncells * sizeof(uint32_t));
    if (!cursor) { cleanup(); return false; }
    ok = hipMemsetAsync(cursor, 0, ncells * sizeof(uint32_t), c.stream) == hipSuccess;
    if (ok) {
        CW_LAUNCH("sor_cell_scatter", cell_scatter_kernel, dim3(grid_for(n)), dim3(BLK), 0, c.stream, src.x(), src.y(), src.z(), n, cell_id, starts,
                  cursor, sorted);
        // 4. the k-NN pass
        const unsigned qgrid = (unsigned)((n + QB - 1) / QB);
        if (k + 1 <= 17) {
            CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<17, false>), dim3(qgrid), dim3(QB), 0, c.stream, g, (const GridMeta *)nullptr, sorted, n, starts, counts, k, dev_dist);
        } else if (k + 1 <= 33) {
            CW_LAUNCH("sor_knn_mean_dist", (knn_mean_dist_reg_kernel<33, false>), dim3(qgrid), dim3(QB), 0, c.stream, g, (const GridMeta *)nullptr, sorted, n, starts, counts, k, dev_dist);
        } else {
            void *slab = nullptr;
            ok = launch_knn_list(g, nullptr, sorted, n, starts, counts, k, dev_dist, c.stream, &slab);
            c.free_later(slab);
        }
    }
    // no wait here: every caller has one further down (the compaction, a copy to the host), and the
    // temporaries go back to the pool there
    ok = hipGetLastError() == hipSuccess && ok;
    c.free_later(cursor); c.free_later(counts); c.free_later(fill); c.free_later(cell_id); c.free_later(sorted); c.free_later(scan_tmp);
    if (!ok) {
        hip_failed(hipGetLastError(), "sor k-NN", __FILE__, __LINE__);
        (void)c.sync();
    }
    return ok;
}

// mean, variance and threshold from the 1024 partial sums: a pairwise tree (s[i] = s[2i] + s[2i+1], ten
// levels), the order the host version (sor_threshold) follows too, then the same f64 expressions
__global__ void __launch_bounds__(1024) stats_final_kernel(const double *__restrict__ partial, size_t n, float stddev_mul, double *__restrict__ thr) {
    __shared__ double s[1024], q[1024];
    s[threadIdx.x] = partial[2 * threadIdx.x];
    q[threadIdx.x] = partial[2 * threadIdx.x + 1];
    __syncthreads();
    for (unsigned width = 512; width >= 1; width >>= 1) {
        double a = 0, b = 0;
        if (threadIdx.x < width) { a = s[2 * threadIdx.x] + s[2 * threadIdx.x + 1]; b = q[2 * threadIdx.x] + q[2 * threadIdx.x + 1]; }
        __syncthreads();
        if (threadIdx.x < width) { s[threadIdx.x] = a; q[threadIdx.x] = b; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double sum = s[0], sq_sum = q[0];
    const double valid = (double)n;
    const double mean = sum / valid;
    const double variance = (sq_sum - sum * sum / valid) / (valid - 1);
    const double stddev = sqrt(variance);
    *thr = mean + (double)stddev_mul * stddev;
}

// (Both in one launch -- the workgroup that takes the last ticket runs the tree -- was built and measured in round 3: 29 us with a
// ticket per partial sum, 16 us with sixteen partial sums per workgroup and ticket, against 3 + 3 us for the two kernels and the
// boundary between them.  A compaction's count and scan in one launch do pay: kernels_basic.hip.)
bool sor_threshold_device(const float *dev_dist, size_t n, float stddev_mul, double *thr_dev) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return false;
    const unsigned nb = 1024;
    double *partial = (double *)pool_alloc(nb * 2 * sizeof(double));
    if (!partial) return false;
    CW_LAUNCH("sor_stats", stats_partial_kernel, dim3(nb), dim3(BLK), 0, c.stream, dev_dist, n, partial);
    CW_LAUNCH("sor_stats_final", stats_final_kernel, dim3(1), dim3(1024), 0, c.stream, partial, n, stddev_mul, thr_dev);   // nb == 1024
    c.free_later(partial);
    return hipGetLastError() == hipSuccess;
}

bool sor_threshold(const float *dev_dist, size_t n, float stddev_mul, double *thr) {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return false;
    const unsigned nb = 1024;
    double *partial = (double *)pool_alloc(nb * 2 * sizeof(double));
    if (!partial) return false;
    CW_LAUNCH("sor_stats", stats_partial_kernel, dim3(nb), dim3(BLK), 0, c.stream, dev_dist, n, partial);
    double *h = (double *)c.staging(nb * 2 * sizeof(double));
    bool ok = h && hipMemcpyAsync(h, partial, nb * 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
    ok = c.sync() && ok;
    pool_free(partial);
    if (!ok) return false;
    // pairwise tree over the partial sums, as stats_final_kernel does it
    std::vector<double> s(nb), q(nb);
    for (unsigned b = 0; b < nb; b++) { s[b] = h[2 * b]; q[b] = h[2 * b + 1]; }
    for (unsigned width = nb / 2; width >= 1; width >>= 1)
        for (unsigned i = 0; i < width; i++) { s[i] = s[2 * i] + s[2 * i + 1]; q[i] = q[2 * i] + q[2 * i + 1]; }
    const double sum = s[0], sq_sum = q[0];
    // pcl::StatisticalOutlierRemoval: every point is "valid" here (finite input)
    double valid = (double)n;
    double mean = sum / valid;
    double variance = (sq_sum - sum * sum / valid) / (valid - 1);
    double stddev = sqrt(variance);
    *thr = mean + (double)stddev_mul * stddev;
    return true;
}

std::shared_ptr<DeviceSoA> sor_threshold_and_select(const DeviceSoA &src, const float *dev_dist, float stddev_mul, double *thr_dev) {
    const size_t n = src.npoints;
    static const bool fold_off = []() { const char *e = getenv("CWIPC_SOR_STATS_FOLD"); return e && atoi(e) == 0; }();   // test knob: the statistics' second kernel on its own
    if (n > k::compact_small_cloud_limit() || fold_off) {
        if (!sor_threshold_device(dev_dist, n, stddev_mul, thr_dev)) return nullptr;
        return sor_select(src, dev_dist, 0.0, thr_dev);
    }
    // small clouds: the partial sums only; the compaction's count kernel turns them into the threshold (kernels_basic.hip, threshold_from_partials)
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    const unsigned nb = 1024;
    double *partial = (double *)pool_alloc(nb * 2 * sizeof(double));
    if (!partial) return nullptr;
    CW_LAUNCH("sor_stats", stats_partial_kernel, dim3(nb), dim3(BLK), 0, c.stream, dev_dist, n, partial);
    k::Predicate p{};
    p.mode = 3;
    p.dist = dev_dist;
    p.thr_dev = thr_dev;
    p.stat_partial = partial; p.stat_n = n; p.stat_mul = stddev_mul; p.thr_out = thr_dev;
    auto out = compact(src, p);   // (waits for its kernels)
    c.free_later(partial);
    return out;
}

std::shared_ptr<DeviceSoA> sor_select(const DeviceSoA &src, const float *dev_dist, double thr, const double *thr_dev) {
    k::Predicate p{};
    p.mode = 3;
    p.dist = dev_dist;
    p.thr = thr;
    p.thr_dev = thr_dev;
    return compact(src, p);   // (waits for its kernels: the caller frees dev_dist afterwards)
}

}  // namespace cwipc_amd
