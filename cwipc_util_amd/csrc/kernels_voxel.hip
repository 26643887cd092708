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
//   * many lanes bumping ONE counter serialise at the memory side (40 k of them: ~250 us);
//   * an LDS atomic costs 13-20 cycles per WAVE INSTRUCTION almost independent of the
//     number of active lanes, so instructions have to be saved, not lanes;
//   * f64 and out-of-line calls in the streaming loop cost more than the stream itself.
// Hence:
//   host                  the octree lattice is anchored at the cloud's first point, which the
//                         host knows (or fetches once): it computes the anchor and, per axis, a
//                         table of leaf-face thresholds with the octree's own f64 arithmetic
//                         (the octree key floor((p - min)/res) is monotone in p, so "key >= m"
//                         is exactly "p >= T(m)" for one float T(m)).
//   K1 voxel_accumulate   persistent: one 1024-lane workgroup per CU, every wave streams a
//                         contiguous range of the planes (dwordx4 per plane per lane, next
//                         step prefetched in registers).  fp32/integer only.  Two variants with
//                         identical integer sums: the fast one (voxel_k1_fast.inc: linear voxel
//                         keys, leaf side bits from per-wave slabs, wave-wide prefix sums, one
//                         conflict-free LDS insert per run) takes clouds in scan order; the general
//                         one below (leaf by threshold compare, DPP segmented scan over chains of
//                         lanes, overflow path to the global records) takes everything else and is
//                         what the fast one hands a cloud back to (ERR_FAST_PATH).  Either way the
//                         workgroup's LDS table is flushed ONCE into dense per-leaf grids of
//                         64-byte records, 8 lanes per record; first touches (told by the returned
//                         count, in the fast variant by the leaf's occupancy bit) are counted per
//                         bitmap slice.  All sums are
//                         integers: results are bitwise reproducible.  Each wave also emits the
//                         bounding box of its range.
//   (partition)           clouds in no spatial order first go through voxel_partition.inc: bucket
//                         histogram + range boxes of the original order, scan, LDS-staged scatter;
//                         K1 (general) then runs on the moved copy.
//   K2 octree_replay      one workgroup replays the octree's bounding-box growth over the wave
//                         boxes, re-reading only the ranges that trigger a growth step
//                         (plain grid: reduces the boxes to the global one); publishes the pass's
//                         control words to pinned host memory.
//   rank_emit             octree path: output position = rank of the cell's bit in the occupancy
//                         bitmaps (leaves in Morton order of their final keys), centroid / colour /
//                         tile from the record, record and bit zeroed for the next call.  Launched
//                         right behind K2, before the host knows the count; a stream of frames gets
//                         its results back while these kernels run (PendingVoxel).
//   grid_mark .. unmark   plain grid: the same through a bitmap over the VoxelGrid index space.
//   make_sort_keys + rocprim radix sort + emit_and_clean: only for plain-grid index spaces beyond
//                         2^28 cells.
#include "internal.hpp"

#include <atomic>
#include <chrono>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cfloat>
#include <type_traits>
#include <cmath>

namespace cwipc_amd {

namespace {

// ---------------------------------------------------------------------------
// constants and shared structures
// ---------------------------------------------------------------------------
constexpr int K1_THREADS = 1024;
constexpr int K1_WAVES = K1_THREADS / 64;
constexpr int WAVE_STEP = 256;                 // points per wave per step (4 per lane)
constexpr size_t MAX_POINTS_PER_WAVE = 3840;   // 15 steps; 16 waves -> at most 61440 points per workgroup
constexpr int LTAB = 2048;                     // LDS table entries per workgroup
constexpr int LTAB_PROBES = 32;
constexpr int GRID_DIM = 68;                   // cells per axis of a leaf grid (64 + slack for fp rounding)
constexpr int CELLS = GRID_DIM * GRID_DIM * GRID_DIM;   // 314432 < 2^19
constexpr int CELL_BITS = 19;
constexpr int BITWORDS = CELLS / 32;            // 9826 occupancy words per leaf grid (CELLS is a multiple of 32)
constexpr uint32_t KEY_EMPTY = 0xffffffffu;
constexpr int HIST = 256;                      // slots of the per-workgroup histogram of first touches per bitmap slice
constexpr int LOCAL_LEAVES = 64;               // leaves a workgroup can name locally (keys carry the local slot, the flush translates)
// the finalize pass works on slices of a leaf's occupancy bitmap
constexpr int RANK_THREADS = 256;
constexpr int RANK_SEGS = 16;                  // a few leaves hold all the work: many slices per leaf for enough workgroups
constexpr int SEG_WORDS = (BITWORDS + RANK_SEGS - 1) / RANK_SEGS;                 // 615
constexpr int WORDS_PER_THREAD = (SEG_WORDS + RANK_THREADS - 1) / RANK_THREADS;   // 3
constexpr int RECORD_WORDS = 8;                // 64-byte records: sx sy sz cr gb tlo thi tor
constexpr int FACES = 128;                     // leaf faces per axis with a precomputed threshold
constexpr int FACE_BACK = 63;                  // the table starts 63 faces below the first point's leaf
// threshold table in 32-bit words: [3][FACES] thresholds T (float), [3][FACES] Tv (float: lower bound of the voxel above
// the one T lies in), [3][FACES] tf (int: index of the voxel T lies in); the general kernel reads the first part only
constexpr int FT_T = 0, FT_TV = 3 * FACES, FT_TF = 6 * FACES, FACE_TABLE_WORDS = 9 * FACES;

enum : uint32_t {
    ERR_RANGE = 1,           // voxel index outside +-2^26, or leaf index outside +-2^20
    ERR_LEAVES = 2,          // more octree leaves than the workspace has grids for (host regrows and retries)
    ERR_DEPTH = 4,           // octree deeper than the sort key can express
    ERR_GRID_OVERFLOW = 8,   // pcl::VoxelGrid: "Leaf size is too small ... indices would overflow"
    ERR_LEAF_RANGE = 16,
    ERR_FACE_TABLE = 32,     // a point lies beyond the threshold table (host reruns the exact variant)
    ERR_CELL_RANGE = 64,
    ERR_LIST_FULL = 128,
    ERR_LOCAL_LEAVES = 512,  // a workgroup met more leaves than its local leaf table holds: host reruns with global leaf ids in the hot loop
};

// control block, 32-bit words in device memory
enum {
    C_ERR = 0, C_COUNT = 1, C_DEPTH = 2, C_EVENTS = 3, C_SHIFT = 4 /* 3 x int64 */, C_MINB = 10, C_DIVB = 13,
    C_FALLBACK = 16,   // runs that found the workgroup table full and went to the global records one lane at a time
    C_MAXLOAD = 17,    // fullest workgroup table (entries)
    // 18: C_SCATTER (voxel_partition.inc)
    C_LEAVES = 20,     // octree leaves (or bricks of the plain grid) the pass has met = leaf grids in use
    C_FLUSHED = 19,    // table entries flushed by all workgroups = global record updates of the pass (the general variant counts them)
    C_SEQ = 31,        // number of published words (the host copy carries the pass's sequence number in the upper half of each 64-bit word)
    C_WORDS = 32
};

struct VoxParams {
    size_t n;
    size_t per_wave;        // points per wave range (multiple of WAVE_STEP)
    uint32_t range_base_q, range_inc_q;   // != 0: the ranges the replay kernel gets boxes of have growing lengths (range_first_step)
    uint32_t nranges;       // number of wave ranges = waves in the K1 grid
    float inv_leaf;         // 1 / leaf in fp32, as pcl::VoxelGrid::setLeafSize
    float leaf;
    double leaf_d;
    double vox_unit;        // 1 / inv_leaf: a voxel index times this is the voxel's lower corner
    double q_unit;          // 1 / (inv_leaf * 2^23): what one unit of the offset sums is worth
    double res;             // octree resolution (double)(float)(64 * leaf)
    // anchor (host): first octree box and the voxel index of its lower corner
    double mn0[3], mx0[3];
    int depth0;
    int ib[3];              // cell c of leaf l on axis a is voxel  c + ib[a] + 64*l - 2
    int face_base[3];       // faces[a][i] is the threshold of face face_base[a] + i
    int leaf_split;
    uint32_t leaf_mask;     // capacity of the leaf hash - 1 (the hash has four slots per leaf grid)
    uint32_t list_cap;
    uint32_t ablate;        // diagnostics only (CWIPC_VOXEL_ABLATE): skip parts of K1 to time the rest; results are wrong when non-zero
};

struct VoxWork {
    unsigned long long *leaf_keys;   // [leaf id] 0 = none yet, else packed lattice coordinates | 1<<63 (ids are handed out in order of arrival)
    unsigned long long *records;     // [leaf hash][CELLS][8]
    uint32_t *occupied;              // list of (leaf id << 19 | cell) of touched records
    uint32_t *ctrl;
    float *bboxes;                   // [nranges][6]
    const float *faces;              // [3][FACES] thresholds (positive cellsize only)
    uint32_t *bitmaps;               // [leaf hash][BITWORDS] occupancy of the leaf grids (bit = cell)
    uint32_t *seg_count;             // [leaf hash][RANK_SEGS] occupied cells per bitmap slice (accumulated by K1's flush)
    unsigned long long *hash_keys;   // [4 x leaf grids] leaf -> id: open addressing on the packed coordinates ...
    uint32_t *hash_ids;              //   ... and the id + 1 of the entry's leaf (0: not published yet, ~0: no grid left)
    uint32_t *dump_head;             // r4, fast accumulate kernel with FastParams::dump: per workgroup a DumpHead ...
    uint32_t *dump_ent;              //   ... and room for the entries of its table, DUMP_ENTRY_WORDS words each (voxel_k1_fast.inc)
};

// Ranges of growing length (r4, the fast accumulate kernel's workgroups: they then reach their flush one after the other instead of
// all at once): range b has base_q + b * inc_q 1024ths of a step (256 points); this is the first step of range b.  Shared by the
// accumulate kernel, the replay kernel (which reads a range again when its box does not settle the octree's growth) and the host.
inline __host__ __device__ uint32_t range_first_step(uint32_t b, uint32_t base_q, uint32_t inc_q) {
    const unsigned long long bb = b;
    return (uint32_t)((bb * base_q + (unsigned long long)inc_q * (bb * (bb > 0 ? bb - 1 : 0) / 2)) >> 10);
}

inline __host__ __device__ uint64_t mix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

__device__ __forceinline__ unsigned long long pack_leaf(int lx, int ly, int lz) {
    return (1ull << 63) | ((unsigned long long)(uint32_t)(lx & 0x1fffff)) | ((unsigned long long)(uint32_t)(ly & 0x1fffff) << 21) |
           ((unsigned long long)(uint32_t)(lz & 0x1fffff) << 42);
}
__device__ __forceinline__ int unpack_leaf(unsigned long long v, int axis) {
    int t = (int)((v >> (21 * axis)) & 0x1fffff);
    return (t << 11) >> 11;   // sign-extend 21 bits
}

// ---------------------------------------------------------------------------
// host: anchor and face thresholds  [PCL upstream octree_pointcloud.hpp]
// ---------------------------------------------------------------------------
// Octree box after the first point: adoptBoundingBoxToPoint's "octree is empty" branch followed
// by getKeyBitSize().
void first_box(const double p[3], double res, double mn[3], double mx[3], int &depth) {
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

// Smallest float for which a monotone predicate (false ... false true ... true over the ordered floats) holds, searched
// outwards from a guess: doubling steps until the answer is bracketed, then bisection.  The guess is a few float steps
// off as a rule (a dozen evaluations); it may be ~1e28 steps off near zero, where the double sum p - min absorbs
// them all (a face through a first point with a coordinate of -5e-17: a cloud rotated by 270 degrees) -- hence no
// fixed-width search, and hence not 32 bisection steps from the ends of the float line for every one of 768 table
// entries either (120 us per call for a cloud whose anchor is new, as every tile of a capture is).
template <class Pred>
float first_float_where(const Pred &passes, float guess) {
    const auto to_ord = [](float f) { int32_t b; memcpy(&b, &f, 4); return b >= 0 ? (int64_t)b : -(int64_t)(b & 0x7fffffff); };
    const auto from_ord = [](int64_t o) { int32_t b = o >= 0 ? (int32_t)o : (int32_t)(0x80000000u | (uint32_t)(-o)); float f; memcpy(&f, &b, 4); return f; };
    const int64_t lowest = to_ord(-FLT_MAX), highest = to_ord(FLT_MAX);
    if (!(guess >= -FLT_MAX && guess <= FLT_MAX)) guess = 0.f;
    int64_t lo, hi;   // invariant at the end: lo fails, hi passes
    const int64_t g = to_ord(guess);
    if (passes(from_ord(g))) {
        hi = g;
        int64_t step = 1;
        for (;;) {
            lo = hi - step;
            if (lo <= lowest) { lo = lowest; if (passes(from_ord(lo))) return -FLT_MAX; break; }
            if (!passes(from_ord(lo))) break;
            hi = lo;
            step *= 2;
        }
    } else {
        lo = g;
        int64_t step = 1;
        for (;;) {
            hi = lo + step;
            if (hi >= highest) { hi = highest; if (!passes(from_ord(hi))) return INFINITY; break; }
            if (passes(from_ord(hi))) break;
            lo = hi;
            step *= 2;
        }
    }
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (passes(from_ord(mid))) hi = mid; else lo = mid;
    }
    return from_ord(hi);
}

// The octree key of a coordinate, floor((p - min) / resolution) in double (genOctreeKeyforPoint), is monotone in p:
// "key >= m" is a threshold test p >= T(m).  Smallest float that passes.
float leaf_threshold(double mn0, double res, int m) {
    const double md = (double)m;
    return first_float_where([&](float p) { return floor(((double)p - mn0) / res) >= md; }, (float)(mn0 + md * res));
}

// Smallest float whose voxel index floor(fl(p * inv_leaf)) exceeds `voxel` (fp32 product, as the kernels compute it).
float voxel_upper_bound(float inv_leaf, int voxel) {
    return first_float_where([&](float p) { return floorf(p * inv_leaf) > (float)voxel; }, (float)(((double)voxel + 1.0) / (double)inv_leaf));
}

// ---------------------------------------------------------------------------
// global side: leaf lookup, record updates
// ---------------------------------------------------------------------------
// One lane: returns the id (hash position) of leaf key k, inserting it if new.
// The id (= grid) of leaf k, giving it the next free one if the pass has not met it yet; ~0 when the grids have run out
// (ERR_LEAVES is set: the host regrows and reruns).  The hash has four slots per grid: with one slot per grid (round 1: the
// slot WAS the id) a person-sized cloud's 12-16 leaves filled a 16-slot table and every lookup walked it, one global round
// trip per probe -- 5-10 us in the flush of every workgroup (time stamps of the debug-knob build).  Whoever claims a slot
// publishes the id right behind the claim; a lane that finds the key but not yet the id looks again in the SAME loop (no
// inner wait: lanes of one wave may be on either side).
__device__ __forceinline__ uint32_t leaf_lookup(const VoxWork &W, uint32_t mask, unsigned long long k) {
    const uint32_t cap = (mask + 1u) >> 2;
    uint32_t pos = (uint32_t)mix64(k) & mask;
    uint32_t probes = 0;
    for (uint32_t guard = 0; guard < (1u << 22); guard++) {
        unsigned long long cur = __hip_atomic_load(&W.hash_keys[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            cur = atomicCAS(&W.hash_keys[pos], 0ull, k);
            if (cur == 0ull) {
                const uint32_t id = atomicAdd(&W.ctrl[C_LEAVES], 1u);
                if (id < cap) {
                    W.leaf_keys[id] = k;
                    __hip_atomic_store(&W.hash_ids[pos], id + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return id;
                }
                atomicOr(&W.ctrl[C_ERR], ERR_LEAVES);
                __hip_atomic_store(&W.hash_ids[pos], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0xffffffffu;
            }
        }
        if (cur == k) {
            const uint32_t v = __hip_atomic_load(&W.hash_ids[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0u) return v == 0xffffffffu ? 0xffffffffu : v - 1u;
            continue;   // claimed a moment ago, id on its way
        }
        pos = (pos + 1) & mask;
        if (++probes > mask) break;
    }
    atomicOr(&W.ctrl[C_ERR], ERR_LEAVES);
    return 0xffffffffu;
}

// The same question asked through the caches first (r4, second session; the fast accumulate kernel's flush).  An entry of the leaf table never
// changes once its id is published, and this XCD's L2 holds nothing older than the kernel's start: a plain load that shows the key WITH its id
// shows the truth, and one that does not (an empty slot, a key without its id, a line that went into L2 before the leaf was claimed) sends the lane
// to leaf_lookup's loads at device scope.  Those go past the L2 to the memory side, where the lookups of ALL workgroups -- a cloud has a dozen
// leaves, a 300 k-point cloud 234 workgroups that flush at the same moment -- queue at a dozen addresses: 1.2 to 13 us per lookup by the time stamps.
__device__ __forceinline__ uint32_t leaf_lookup_cached(const VoxWork &W, uint32_t mask, unsigned long long k) {
    uint32_t pos = (uint32_t)mix64(k) & mask;
    for (uint32_t probes = 0; probes < 8u; probes++) {
        const unsigned long long cur = __hip_atomic_load(&W.hash_keys[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (cur == k) {
            const uint32_t v = __hip_atomic_load(&W.hash_ids[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (v != 0u && v != 0xffffffffu) return v - 1u;
            break;
        }
        if (cur == 0ull) break;
        pos = (pos + 1) & mask;
    }
    return leaf_lookup(W, mask, k);
}

// Record of voxel key = leaf id << 19 | cell  (grids are CELLS records apart, not 2^19).
__device__ __forceinline__ unsigned long long *record_ptr(const VoxWork &W, uint32_t key) {
    return W.records + ((size_t)(key >> CELL_BITS) * CELLS + (key & ((1u << CELL_BITS) - 1))) * RECORD_WORDS;
}

// index into seg_count of the bitmap slice that holds a record's bit
__device__ __forceinline__ uint32_t slice_of(uint32_t key) {
    const uint32_t cell = key & ((1u << CELL_BITS) - 1);
    return (key >> CELL_BITS) * RANK_SEGS + (cell >> 5) / SEG_WORDS;
}

__device__ __forceinline__ void mark_occupied(const VoxWork &W, uint32_t key) {
    const uint32_t cell = key & ((1u << CELL_BITS) - 1);
    atomicOr(&W.bitmaps[(size_t)(key >> CELL_BITS) * BITWORDS + (cell >> 5)], 1u << (cell & 31u));
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
// K1 is bound by instruction issue, not by HBM (rocprofv3: ~230 VALU instructions per point in the
// first version, VALU busy 60 %, waves parked 58 % with 4 waves per SIMD), so the hot loop below is
// written for instruction count: wave-uniform values in SGPRs, selects instead of branches, 32-bit
// arithmetic, byte permutes for the colour sums, and slow paths behind wave-uniform ballots.

// One run of points of the same voxel, 32-bit in-wave form (at most 256 points).
struct Run32 {
    uint32_t key;       // leaf id << 19 | cell, KEY_EMPTY = none
    uint32_t qx, qy, qz;   // sums of biased fixed-point offsets inside the voxel
    uint32_t cr;        // count << 16 | sum r
    uint32_t gb;        // sum g << 16 | sum b
    uint32_t tile;
};

// Workgroup table entry, 4 packed 64-bit sums.  A workgroup sees fewer than 65536 points, so
// count < 2^16, colour sums < 2^24, and a biased offset sum < 2^40.
struct LdsTable {
    uint32_t key[LTAB];
    uint32_t tile[LTAB];
    unsigned long long a[LTAB];   // sum (qx + bias)
    unsigned long long b[LTAB];   // sum (qy + bias)
    unsigned long long c[LTAB];   // sum (qz + bias) | sum b << 40
    unsigned long long d[LTAB];   // count | sum r << 16 | sum g << 40
    uint32_t fresh[LTAB];         // records this workgroup touched first
    float faces[3 * FACES];
    uint32_t htag[HIST], hcnt[HIST];  // first touches per bitmap slice of this workgroup (slice + 1, count); linear probing
    uint32_t nfresh, fresh_base;
    uint32_t nfallback, nused;    // table-full fallbacks of this workgroup; entries in use (counted by the flush)
    unsigned long long leaf_tab[LOCAL_LEAVES];   // packed leaf coordinates, 0 = free; position = local leaf slot
    uint32_t leaf_gid[LOCAL_LEAVES];             // global leaf id of each slot (filled before the flush)
    uint32_t nn_leaf[K1_WAVES][64];              // per wave: leaf (slot or id) of each leaf position relative to the cached faces, ~0 = not looked up yet
};

// The slim parameter block of K1 (kernel arguments live in SGPRs; K1 is short of them).
struct K1Params {
    uint32_t n, per_wave;
    float inv_leaf;
    int ib0, ib1, ib2;
    int fb0, fb1, fb2;
    uint32_t leaf_mask, list_cap, ablate;
    uint32_t local_leaves;   // 1: keys carry workgroup-local leaf slots (no global memory access in the hot loop)
    uint32_t want_list;      // 1: the touched records are listed in W.occupied (plain grid: the sort needs them); 0: only counted
    double mn0[3];      // MODE 2 only
    double res;
};

__device__ __forceinline__ unsigned long long u64_of(uint32_t lo, uint32_t hi) { return ((unsigned long long)hi << 32) | lo; }

// Add one run to the workgroup table.  All in-wave sums are 32-bit (<= 256 points), so the four
// packed 64-bit addends are assembled from 32-bit halves.
// One lane: slot of leaf k in the workgroup's local leaf table, inserting it if new; 0xffffffff when the table is full.
__device__ __forceinline__ uint32_t local_leaf_slot(LdsTable &L, unsigned long long k) {
    uint32_t pos = (((uint32_t)k ^ (uint32_t)(k >> 21) ^ (uint32_t)(k >> 42)) * 0x9E3779B1u) >> 26;   // LOCAL_LEAVES = 2^6
    for (int probe = 0; probe < LOCAL_LEAVES; probe++) {
        const unsigned long long cur = L.leaf_tab[pos];
        if (cur == k) return pos;
        if (cur == 0ull) {
            const unsigned long long old = atomicCAS(&L.leaf_tab[pos], 0ull, k);
            if (old == 0ull || old == k) return pos;
        }
        pos = (pos + 1) & (LOCAL_LEAVES - 1);
    }
    return 0xffffffffu;
}

// One lane: leaf k as the hot loop names it (a local slot, or the global id).  Whoever creates a local
// slot also fetches its global id right away, while the other waves keep streaming: the flush at the end
// of the kernel is a serial tail and should not start with a round trip to the global leaf table.
__device__ __forceinline__ uint32_t leaf_name(LdsTable &L, const VoxWork &W, const K1Params &P, unsigned long long k) {
    if (!P.local_leaves) return leaf_lookup(W, P.leaf_mask, k);
    const uint32_t slot = local_leaf_slot(L, k);
    if (slot != 0xffffffffu && atomicCAS(&L.leaf_gid[slot], 0xffffffffu, 0xfffffffeu) == 0xffffffffu) {
        L.leaf_gid[slot] = leaf_lookup(W, P.leaf_mask, k);   // ~0 if the global table is full (ERR_LEAVES is set then)
    }
    return slot;
}

__device__ __forceinline__ void lds_insert(LdsTable &L, const VoxWork &W, const K1Params &P, const Run32 &r, bool active) {
    // Called by the whole wave (active = this lane has something to insert).  The slot search is a loop
    // of its own, so that the adds are issued once per call however many probes the unluckiest lane needs:
    // LDS atomics are the scarcest resource of this kernel.
    uint32_t slot = (r.key * 0x9E3779B1u) >> (32 - 11);   // LTAB = 2^11
    bool pending = active;
#pragma unroll 1
    for (int probe = 0; probe < LTAB_PROBES; probe++) {
        if (pending) {
            const uint32_t old = atomicCAS(&L.key[slot], KEY_EMPTY, r.key);
            if (old == KEY_EMPTY || old == r.key) pending = false;
            else slot = (slot + 1) & (LTAB - 1);
        }
        if (__ballot(pending) == 0ull) break;
    }
    if (active && !pending) {
        // every offset carries its bias already, so the in-wave sums are plain unsigned 32-bit numbers
        atomicAdd(&L.a[slot], u64_of(r.qx, 0u));
        atomicAdd(&L.b[slot], u64_of(r.qy, 0u));
        atomicAdd(&L.c[slot], u64_of(r.qz, (r.gb & 0xffffu) << 8));                                            // | sum b << 40
        atomicAdd(&L.d[slot], u64_of(__builtin_amdgcn_alignbit(r.cr, r.cr, 16), (r.gb >> 16) << 8));         // count | sum r << 16 | sum g << 40
    }
    // the tile bits of a voxel are almost always there already: a plain read is much cheaper than an atomic
    const bool need_or = active && !pending && (L.tile[slot] & r.tile) != r.tile;
    if (__ballot(need_or) != 0ull) {
        if (need_or) atomicOr(&L.tile[slot], r.tile);
    }
    // ---- table saturated (sparse or incoherent input): the runs go straight to the global records ----
    const unsigned long long failed = __ballot(active && pending);
    if (__builtin_expect(failed != 0ull, 0)) {
        const bool mine = active && pending;
        const int lane = threadIdx.x & 63;
        // the record of this lane's run (global leaf id * CELLS + cell), ~0 if it has none
        uint32_t rec = 0xffffffffu;
        if (mine) {
            atomicAdd(&L.nfallback, 1u);
            uint32_t gid = r.key >> CELL_BITS;
            if (P.local_leaves) {
                gid = L.leaf_gid[r.key >> CELL_BITS];
                if (gid >= 0xfffffffeu) gid = leaf_lookup(W, P.leaf_mask, L.leaf_tab[r.key >> CELL_BITS]);   // ~0: ERR_LEAVES is set, the pass is discarded
            }
            if (gid != 0xffffffffu) rec = gid * (uint32_t)CELLS + (r.key & ((1u << CELL_BITS) - 1));
        }
        // Eight lanes per run update its 64-byte record with one instruction (one cache-line operation
        // in L2 instead of six: incoherent clouds are bound by exactly that), eight runs per instruction.
#pragma unroll 1
        for (int b = 0; b < 8; b++) {
            if (((failed >> (8 * b)) & 0xffull) == 0ull) continue;
            const int src = 8 * b + (lane >> 3), sub = lane & 7;
            const uint32_t s_rec = (uint32_t)__shfl((int)rec, src, 64);
            const uint32_t s_qx = (uint32_t)__shfl((int)r.qx, src, 64), s_qy = (uint32_t)__shfl((int)r.qy, src, 64), s_qz = (uint32_t)__shfl((int)r.qz, src, 64);
            const uint32_t s_cr = (uint32_t)__shfl((int)r.cr, src, 64), s_gb = (uint32_t)__shfl((int)r.gb, src, 64), s_tile = (uint32_t)__shfl((int)r.tile, src, 64);
            bool first = false;
            uint32_t s_key = 0;
            if (s_rec != 0xffffffffu && sub < 7) {
                const uint32_t cnt = s_cr >> 16;
                unsigned long long val;
                switch (sub) {
                case 0: val = s_qx; break;
                case 1: val = s_qy; break;
                case 2: val = s_qz; break;
                case 3: val = u64_of(s_cr & 0xffffu, cnt); break;                 // count << 32 | sum r
                case 4: val = u64_of(s_gb & 0xffffu, s_gb >> 16); break;         // sum g << 32 | sum b
                case 5:   // tile bits 0-3 as 16-bit contribution counters
                    val = (unsigned long long)(s_tile & 1u) | ((unsigned long long)((s_tile >> 1) & 1u) << 16) |
                          ((unsigned long long)((s_tile >> 2) & 1u) << 32) | ((unsigned long long)((s_tile >> 3) & 1u) << 48);
                    break;
                default:  // tile bits 4-7
                    val = (unsigned long long)((s_tile >> 4) & 1u) | ((unsigned long long)((s_tile >> 5) & 1u) << 16) |
                          ((unsigned long long)((s_tile >> 6) & 1u) << 32) | ((unsigned long long)((s_tile >> 7) & 1u) << 48);
                    break;
                }
                const unsigned long long old = atomicAdd(&W.records[(size_t)s_rec * RECORD_WORDS + sub], val);
                if (sub == 3 && (old >> 32) == 0) {
                    first = true;
                    s_key = ((s_rec / (uint32_t)CELLS) << CELL_BITS) | (s_rec % (uint32_t)CELLS);
                    mark_occupied(W, s_key);
                    atomicAdd(&W.seg_count[slice_of(s_key)], 1u);
                }
            }
            // records touched for the first time: counted (and listed) with one atomic per instruction
            const unsigned long long news = __ballot(first);
            if (news != 0ull) {
                const uint32_t nnew = (uint32_t)__popcll(news);
                uint32_t base = 0;
                if (lane == __ffsll((long long)news) - 1) base = atomicAdd(&W.ctrl[C_COUNT], nnew);
                if (P.want_list) {
                    base = (uint32_t)__shfl((int)base, __ffsll((long long)news) - 1, 64);
                    if (first) {
                        const uint32_t idx = base + (uint32_t)__popcll(news & ((1ull << lane) - 1ull));
                        if (idx < P.list_cap) W.occupied[idx] = s_key;
                        else atomicOr(&W.ctrl[C_ERR], ERR_LIST_FULL);
                    }
                }
            }
        }
    }
}

// DPP row shifts inside rows of 16 lanes (lanes whose source is outside the row read 0).
template <int N>
__device__ __forceinline__ int dpp_shr(int v) {   // lane l reads lane l - N
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ int dpp_shl(int v) {   // lane l reads lane l + N
    return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xf, 0xf, true);
}

// one step of the segmented inclusive scan: lanes that start a segment keep their value
template <int N>
__device__ __forceinline__ void scan_step(Run32 &v, int &flag) {
    const bool keep = flag != 0;
    const uint32_t qx = v.qx + (uint32_t)dpp_shr<N>((int)v.qx), qy = v.qy + (uint32_t)dpp_shr<N>((int)v.qy), qz = v.qz + (uint32_t)dpp_shr<N>((int)v.qz);
    const uint32_t cr = v.cr + (uint32_t)dpp_shr<N>((int)v.cr), gb = v.gb + (uint32_t)dpp_shr<N>((int)v.gb);
    const uint32_t tile = v.tile | (uint32_t)dpp_shr<N>((int)v.tile);
    v.qx = keep ? v.qx : qx; v.qy = keep ? v.qy : qy; v.qz = keep ? v.qz : qz;
    v.cr = keep ? v.cr : cr; v.gb = keep ? v.gb : gb; v.tile = keep ? v.tile : tile;
    flag |= dpp_shr<N>(flag);
}

// Two consecutive leaf faces of one axis, wave-uniform: faces mc and mc + 1 with their thresholds.
struct FaceCache {
    int mc;
    float tlo, thi;
    int cb;   // voxel index (of floor(p * inv_leaf)) that is cell 0 of leaf mc - 1:  ib + 64 * (mc - 1) - 2
};

struct PointOut {
    uint32_t key;   // cell inside the leaf grid, KEY_EMPTY if the point is skipped
    uint32_t nn;    // MODE 1: leaf relative to the cached faces, n0 | n1 << 2 | n2 << 4 with leaf_a = mc_a - 1 + n_a
    // One set of registers for two things that are never alive together: leaf lattice coordinates (MODE 0 / 2,
    // and MODE 1 once a step has gone the slow way), or the voxel index relative to cb (MODE 1, window test).
    union { int l0; int u0; };
    union { int l1; int u1; };
    union { int l2; int u2; };
    uint32_t q0, q1, q2; // biased fixed-point offsets inside the voxel (>= 0)
    bool seen;      // the point exists and is finite
};

// Position inside the voxel as an integer: prod = fl(p * inv_leaf) is the number pcl::VoxelGrid floors, so
// prod - floor(prod) in [0, 1) is where the point sits in its voxel, in voxel units.  Adding 1.0 rounds that to a
// multiple of 2^-23 (ties to even, unbiased) and leaves it in the mantissa: q in [0, 2^23].  One v_fract and one add;
// the centroid is rebuilt as (voxel + sum q / (n 2^23)) / inv_leaf in f64 by the emit kernels (VoxParams::vox_unit, q_unit).
// Both accumulate kernels (the fast one and the general one) use this very function: their integer sums are identical.
constexpr uint32_t Q_ONE_BITS = 0x3f800000u;   // bits of 1.0f
__device__ __forceinline__ float voxel_fract(float prod) { return __builtin_amdgcn_fractf(prod); }
__device__ __forceinline__ uint32_t voxel_offset(float prod) { return __float_as_uint(__fadd_rn(voxel_fract(prod), 1.0f)) - Q_ONE_BITS; }

// One coordinate: cell c inside the leaf grid, leaf (n or l), offset q inside the voxel.
// MODE 0: plain grid (bricks on the voxel lattice); 1: octree leaves by face thresholds; 2: octree leaves by f64 division.
template <int MODE>
__device__ __forceinline__ void axis_cell(const K1Params &P, int ib, const FaceCache &fc, int axis, float f, int &u, int &n, int &l, int &c,
                                          uint32_t &q) {
    const float prod = __fmul_rn(f, P.inv_leaf);
    const float g = floorf(prod);                        // pcl::VoxelGrid: floor(p * inverse_leaf_size), fp32 product
    const int ti = (int)g;                               // v_cvt_i32_f32 (saturating; non-finite points are masked by the caller)
    if (MODE == 1) {
        // leaf = (face mc - 1) + [f >= T(mc)] + [f >= T(mc + 1)], valid while the voxel lies between faces mc - 1/2 and mc + 3/2
        u = ti - fc.cb;
        n = (f >= fc.tlo ? -1 : 0) + (f >= fc.thi ? -1 : 0);   // minus the count: c is then one shift-and-add
        c = (n << 6) + u;
    } else {
        const int t = ti - ib;
        if (MODE == 0) l = t >> 6;
        else l = (int)floor(((double)f - P.mn0[axis]) / P.res);   // genOctreeKeyforPoint
        c = t - 64 * l + 2;
    }
    q = voxel_offset(prod);
}

template <int MODE>
__device__ __forceinline__ PointOut point_key(const K1Params &P, const FaceCache &f0, const FaceCache &f1, const FaceCache &f2, float fx, float fy,
                                              float fz, bool present) {
    PointOut o;
    int c0, c1, c2, n0 = 0, n1 = 0, n2 = 0;
    axis_cell<MODE>(P, P.ib0, f0, 0, fx, o.u0, n0, o.l0, c0, o.q0);
    axis_cell<MODE>(P, P.ib1, f1, 1, fy, o.u1, n1, o.l1, c1, o.q1);
    axis_cell<MODE>(P, P.ib2, f2, 2, fz, o.u2, n2, o.l2, c2, o.q2);
    o.nn = MODE == 1 ? (uint32_t)(-(n0 + (n1 << 2) + (n2 << 4))) : 0u;   // MODE 1: n_a = -(leaf position) here
    // Non-finite points are skipped as the octree does (addPointsFromInputCloud: isFinite).  One test for
    // the three coordinates: the sum is NaN or Inf iff one of them is (or they are beyond any sane range).
    o.seen = present && __builtin_isfinite(fx + fy + fz);
    // memory safety: a cell outside the leaf grid must never become a record address
    const uint32_t cm = max(max((uint32_t)c0, (uint32_t)c1), (uint32_t)c2);
    o.key = (o.seen && cm < (uint32_t)GRID_DIM) ? (uint32_t)__umul24(__umul24((uint32_t)c2, GRID_DIM) + (uint32_t)c1, GRID_DIM) + (uint32_t)c0 : KEY_EMPTY;
    return o;
}

// The same for a step whose voxels do not fit between one pair of cached faces per axis (a scan line
// wrapping around, a sparse cloud): every point looks up the two faces next to its own voxel in the
// threshold table in LDS.  Same arithmetic, same results, six LDS reads per point more.
__device__ __forceinline__ PointOut point_key_lookup(const K1Params &P, const float *faces, float fx, float fy, float fz, bool present, bool &off_table) {
    PointOut o;
    o.nn = 0;
    int c[3], l[3];
    uint32_t q[3];
    const float f[3] = {fx, fy, fz};
    const int ib[3] = {P.ib0, P.ib1, P.ib2}, fb[3] = {P.fb0, P.fb1, P.fb2};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float prod = __fmul_rn(f[a], P.inv_leaf);
        const float g = floorf(prod);
        const int t = (int)g - ib[a];
        const int m = (t + 32) >> 6;                         // the face nearest to this voxel
        const unsigned i = (unsigned)(m - fb[a]);
        const bool in_table = i + 1u < (unsigned)FACES;
        ok &= in_table;
        const unsigned ii = in_table ? i : 0u;
        const float tlo = faces[a * FACES + ii], thi = faces[a * FACES + ii + 1];
        l[a] = m - 1 + (f[a] >= tlo ? 1 : 0) + (f[a] >= thi ? 1 : 0);
        c[a] = t - 64 * l[a] + 2;
        q[a] = voxel_offset(prod);
    }
    o.l0 = l[0]; o.l1 = l[1]; o.l2 = l[2];
    o.q0 = q[0]; o.q1 = q[1]; o.q2 = q[2];
    o.seen = present && __builtin_isfinite(fx + fy + fz);
    off_table |= o.seen && !ok;
    const uint32_t cm = max(max((uint32_t)c[0], (uint32_t)c[1]), (uint32_t)c[2]);
    o.key = (o.seen && ok && cm < (uint32_t)GRID_DIM) ? (uint32_t)__umul24(__umul24((uint32_t)c[2], GRID_DIM) + (uint32_t)c[1], GRID_DIM) + (uint32_t)c[0] : KEY_EMPTY;
    return o;
}

// r,g,b,tile bytes of one point as addends of the run sums: byte permutes instead of shifts and masks
struct PointAdd {
    uint32_t cr;     // count << 16 | r
    uint32_t gb;     // g << 16 | b
    uint32_t tile;
};
__device__ __forceinline__ PointAdd point_add(uint32_t w) {
    PointAdd a;
    a.cr = (w & 0xffu) | 0x10000u;
    a.gb = __builtin_amdgcn_perm(0u, w, 0x0c010c02u);   // bytes [b, 0, g, 0]
    a.tile = w >> 24;
    return a;
}
__device__ __forceinline__ void add_point(Run32 &r, const PointOut &o, const PointAdd &a) {
    r.qx += o.q0; r.qy += o.q1; r.qz += o.q2;
    r.cr += a.cr; r.gb += a.gb; r.tile |= a.tile;
}

// v_min3_f32 / v_max3_f32 on operands known to be numbers
__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(K1_THREADS) voxel_accumulate_kernel(K1Params P, const float *__restrict__ x, const float *__restrict__ y,
                                                                     const float *__restrict__ z, const uint32_t *__restrict__ rgbt, VoxWork W) {
    extern __shared__ __align__(16) unsigned char k1_smem[];
    LdsTable &L = *reinterpret_cast<LdsTable *>(k1_smem);
    // stage switches for timing experiments exist in -DCWIPC_DEBUG_KNOBS builds only (results are wrong when set)
#ifdef CWIPC_DEBUG_KNOBS
    const uint32_t ablate = P.ablate;
#else
    constexpr uint32_t ablate = 0u;
#endif

    const int lane = threadIdx.x & 63;
    // everything that is the same for the whole wave lives in SGPRs
    const uint32_t range = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * K1_WAVES + (threadIdx.x >> 6)));
    float bn0 = FLT_MAX, bn1 = FLT_MAX, bn2 = FLT_MAX, bx0 = -FLT_MAX, bx1 = -FLT_MAX, bx2 = -FLT_MAX;
    uint32_t err = 0;
    // this wave's range [lo, hi); planes are padded to a multiple of 256 points, so whole steps can be loaded
    const uint32_t lo = range * P.per_wave;
    const uint32_t hi = min(lo + P.per_wave, P.n);
    const int npts = lo < hi ? (int)(hi - lo) : 0;
    const float4 *vx = reinterpret_cast<const float4 *>(x + lo) + lane;
    const float4 *vy = reinterpret_cast<const float4 *>(y + lo) + lane;
    const float4 *vz = reinterpret_cast<const float4 *>(z + lo) + lane;
    const uint4 *vw = reinterpret_cast<const uint4 *>(rgbt + lo) + lane;
    // the first step's loads go out before the table is initialised: their latency hides behind it
    float4 cx = make_float4(0, 0, 0, 0), cy = cx, cz = cx;
    uint4 cw = make_uint4(0, 0, 0, 0);
    if (npts > 0) { cx = vx[0]; cy = vy[0]; cz = vz[0]; cw = vw[0]; }

    if (!(ablate & 128u))
    for (int i = threadIdx.x; i < LTAB; i += K1_THREADS) {
        L.key[i] = KEY_EMPTY; L.tile[i] = 0;
        L.a[i] = 0; L.b[i] = 0; L.c[i] = 0; L.d[i] = 0;
    }
    if (MODE == 1) {
        for (int i = threadIdx.x; i < 3 * FACES; i += K1_THREADS) L.faces[i] = W.faces[i];
    }
    if (threadIdx.x < HIST) { L.htag[threadIdx.x] = 0; L.hcnt[threadIdx.x] = 0; }
    if (threadIdx.x < LOCAL_LEAVES) { L.leaf_tab[threadIdx.x] = 0ull; L.leaf_gid[threadIdx.x] = 0xffffffffu; }
    L.nn_leaf[threadIdx.x >> 6][threadIdx.x & 63] = 0xffffffffu;
    if (threadIdx.x == 0) { L.nfresh = 0; L.nfallback = 0; L.nused = 0; }
    __syncthreads();

    // wave-uniform caches: two leaf faces per axis, the last leaf and its id
    FaceCache fc0, fc1, fc2;
    fc0.mc = fc1.mc = fc2.mc = -(1 << 24);   // covers nothing yet
    fc0.tlo = fc0.thi = fc1.tlo = fc1.thi = fc2.tlo = fc2.thi = 0.f;
    fc0.cb = fc1.cb = fc2.cb = 1 << 30;
    int cl0 = 0, cl1 = 0, cl2 = 0;
    uint32_t cache_id = 0xffffffffu;
    bool cache_valid = false;

#pragma unroll 1
    for (int off = 0; off < npts; off += WAVE_STEP) {
        // keep the next step's 64 bytes per lane in flight while this step is processed
        float4 nx = cx, ny = cy, nz = cz;
        uint4 nw = cw;
        if (off + WAVE_STEP < npts) {
            const int v = (off + WAVE_STEP) >> 2;
            nx = vx[v]; ny = vy[v]; nz = vz[v]; nw = vw[v];
        }
        const int left = npts - off - lane * 4;   // points of this lane that exist: min(left, 4)

        if (ablate & 1u) {   // diagnostics: loads only
            bn0 = fminf(bn0, cx.x + cx.y + cx.z + cx.w + cy.x + cy.y + cy.z + cy.w + cz.x + cz.y + cz.z + cz.w + __uint_as_float(cw.x ^ cw.y ^ cw.z ^ cw.w));
            cx = nx; cy = ny; cz = nz; cw = nw;
            continue;
        }

        // ---- per point: cell, leaf, fixed-point offsets (branch-free) ----
        PointOut o0 = point_key<MODE>(P, fc0, fc1, fc2, cx.x, cy.x, cz.x, left > 0);
        PointOut o1 = point_key<MODE>(P, fc0, fc1, fc2, cx.y, cy.y, cz.y, left > 1);
        PointOut o2 = point_key<MODE>(P, fc0, fc1, fc2, cx.z, cy.z, cz.z, left > 2);
        PointOut o3 = point_key<MODE>(P, fc0, fc1, fc2, cx.w, cy.w, cz.w, left > 3);

        // ---- box of the wave's range (input of the octree replay): skipped points stay out of it ----
        if (__ballot(!(o0.seen && o1.seen && o2.seen && o3.seen)) == 0ull) {
            // (all twelve coordinates are finite here; written as instructions because fminf / fmaxf make the
            // compiler quiet every operand first, which costs more than the minimum itself)
            bn0 = min3f(min3f(bn0, cx.x, cx.y), cx.z, cx.w); bx0 = max3f(max3f(bx0, cx.x, cx.y), cx.z, cx.w);
            bn1 = min3f(min3f(bn1, cy.x, cy.y), cy.z, cy.w); bx1 = max3f(max3f(bx1, cy.x, cy.y), cy.z, cy.w);
            bn2 = min3f(min3f(bn2, cz.x, cz.y), cz.z, cz.w); bx2 = max3f(max3f(bx2, cz.x, cz.y), cz.z, cz.w);
        } else {
            // a ragged last step or non-finite points: NaN is the neutral element of v_min / v_max
            const float nan = __uint_as_float(0x7fc00000u);
            auto box = [&](bool seen, float fx, float fy, float fz) {
                const float sx = seen ? fx : nan, sy = seen ? fy : nan, sz = seen ? fz : nan;
                bn0 = fminf(bn0, sx); bx0 = fmaxf(bx0, sx);
                bn1 = fminf(bn1, sy); bx1 = fmaxf(bx1, sy);
                bn2 = fminf(bn2, sz); bx2 = fmaxf(bx2, sz);
            };
            box(o0.seen, cx.x, cy.x, cz.x); box(o1.seen, cx.y, cy.y, cz.y); box(o2.seen, cx.z, cy.z, cz.z); box(o3.seen, cx.w, cy.w, cz.w);
        }

        bool slow_step = false;   // MODE 1: this step's points carry leaf coordinates instead of positions around the cached faces
        if (MODE == 1) {
            // Are all voxels of this step between the cached faces?  u - 34 = voxel - (64 mc - 32) must lie in [0, 128).
            // Cheap test on the lane's extremes first; points that do not count (absent, non-finite) can only
            // raise a false alarm, which the exact test below sorts out.
            auto spread = [](int a, int b, int c, int d) {
                const int lo = min(min(a, b), min(c, d)), hi = max(max(a, b), max(c, d));
                return (uint32_t)(lo - 34) | (uint32_t)(hi - 34);
            };
            const uint32_t out = spread(o0.u0, o1.u0, o2.u0, o3.u0) | spread(o0.u1, o1.u1, o2.u1, o3.u1) | spread(o0.u2, o1.u2, o2.u2, o3.u2);
            if (__builtin_expect(__ballot(out >= 128u) != 0ull, 0)) {
                auto outside = [](const PointOut &o) {
                    return o.seen && (((uint32_t)(o.u0 - 34) | (uint32_t)(o.u1 - 34) | (uint32_t)(o.u2 - 34)) >= 128u);
                };
                if (__ballot(outside(o0) || outside(o1) || outside(o2) || outside(o3)) != 0ull) {
                    // Where do the voxels of this step lie?  (t = u + cb - ib, lowest and highest per axis)
                    const int big = 1 << 30;
                    int w0 = big, w1 = big, w2 = big, v0 = -big, v1 = -big, v2 = -big;
                    auto span = [&](const PointOut &o) {
                        if (o.seen) {
                            w0 = min(w0, o.u0); w1 = min(w1, o.u1); w2 = min(w2, o.u2);
                            v0 = max(v0, o.u0); v1 = max(v1, o.u1); v2 = max(v2, o.u2);
                        }
                    };
                    span(o0); span(o1); span(o2); span(o3);
                    for (int sft = 32; sft > 0; sft >>= 1) {
                        w0 = min(w0, __shfl_xor(w0, sft, 64)); w1 = min(w1, __shfl_xor(w1, sft, 64)); w2 = min(w2, __shfl_xor(w2, sft, 64));
                        v0 = max(v0, __shfl_xor(v0, sft, 64)); v1 = max(v1, __shfl_xor(v1, sft, 64)); v2 = max(v2, __shfl_xor(v2, sft, 64));
                    }
                    // A pair of faces covers 128 voxels.  If the step fits into that on every axis, move the
                    // caches to the lowest face it needs (thresholds come from the table the host computed)
                    // and redo it; if not, its points look their faces up one by one.
                    bool fits = true, off_table = false;
                    auto plan = [&](const FaceCache &fc, int umin, int umax, int ib, int fb, int &m) {
                        m = fc.mc;
                        if (umin == big) return;   // no point on this step at all
                        const int tmin = umin + fc.cb - ib, tmax = umax + fc.cb - ib;
                        m = (tmin + 32) >> 6;      // nearest face of the lowest voxel
                        fits &= tmax - (64 * m - 32) < 128;
                        fits &= (unsigned)(m - fb) + 1u < (unsigned)FACES;
                    };
                    int m0, m1, m2;
                    plan(fc0, __builtin_amdgcn_readfirstlane(w0), __builtin_amdgcn_readfirstlane(v0), P.ib0, P.fb0, m0);
                    plan(fc1, __builtin_amdgcn_readfirstlane(w1), __builtin_amdgcn_readfirstlane(v1), P.ib1, P.fb1, m1);
                    plan(fc2, __builtin_amdgcn_readfirstlane(w2), __builtin_amdgcn_readfirstlane(v2), P.ib2, P.fb2, m2);
                    if (fits) {
                        auto refill = [&](FaceCache &fc, int m, int ib, int fb, int axis) {
                            if (m == fc.mc) return;
                            const unsigned i = (unsigned)(m - fb);
                            fc.mc = m;
                            fc.tlo = L.faces[axis * FACES + i];
                            fc.thi = L.faces[axis * FACES + i + 1];
                            fc.cb = ib + 64 * (m - 1) - 2;
                        };
                        refill(fc0, m0, P.ib0, P.fb0, 0);
                        refill(fc1, m1, P.ib1, P.fb1, 1);
                        refill(fc2, m2, P.ib2, P.fb2, 2);
                        L.nn_leaf[threadIdx.x >> 6][lane] = 0xffffffffu;   // relative leaf positions mean other leaves now
                        o0 = point_key<MODE>(P, fc0, fc1, fc2, cx.x, cy.x, cz.x, left > 0);
                        o1 = point_key<MODE>(P, fc0, fc1, fc2, cx.y, cy.y, cz.y, left > 1);
                        o2 = point_key<MODE>(P, fc0, fc1, fc2, cx.z, cy.z, cz.z, left > 2);
                        o3 = point_key<MODE>(P, fc0, fc1, fc2, cx.w, cy.w, cz.w, left > 3);
                    } else {
                        slow_step = true;
                        o0 = point_key_lookup(P, L.faces, cx.x, cy.x, cz.x, left > 0, off_table);
                        o1 = point_key_lookup(P, L.faces, cx.y, cy.y, cz.y, left > 1, off_table);
                        o2 = point_key_lookup(P, L.faces, cx.z, cy.z, cz.z, left > 2, off_table);
                        o3 = point_key_lookup(P, L.faces, cx.w, cy.w, cz.w, left > 3, off_table);
                        // beyond the table (more than 60 leaves from the first point): the host reruns the f64 variant
                        if (__ballot(off_table) != 0ull) err |= ERR_FACE_TABLE;
                    }
                }
            }
        }

        if (ablate & 2u) {   // diagnostics: loads + per-point arithmetic only
            bx0 = fmaxf(bx0, __uint_as_float((o0.key ^ o1.key ^ o2.key ^ o3.key) + (o0.q0 + o1.q1 + o2.q2 + o3.q0 + o0.nn + o1.nn + o2.nn + o3.nn) +
                                              (uint32_t)(o0.l0 + o1.l1 + o2.l2)));
            cx = nx; cy = ny; cz = nz; cw = nw;
            continue;
        }

        // ---- leaf ids ----
        if (MODE == 1 && !slow_step && !(ablate & 64u)) {
            // The leaf of a point is one of the 27 positions around the cached faces (nn); this wave's table
            // in LDS says which leaf that is.  A plain LDS read per point; the lookup behind it runs once
            // per position (and again after the face caches moved).
            uint32_t *tab = L.nn_leaf[threadIdx.x >> 6];
            uint32_t s0 = tab[o0.nn], s1 = tab[o1.nn], s2 = tab[o2.nn], s3 = tab[o3.nn];
            const auto unknown = [](const PointOut &o, uint32_t sl) { return o.key != KEY_EMPTY && sl == 0xffffffffu; };
            // (first a test that may raise a false alarm for points that do not count: one maximum instead of
            // four two-part conditions; the loop behind it looks closely)
            if (__builtin_expect(__ballot(max(max(s0, s1), max(s2, s3)) == 0xffffffffu) != 0ull, 0)) {
                for (;;) {
                    const uint32_t want = unknown(o0, s0) ? o0.nn : unknown(o1, s1) ? o1.nn : unknown(o2, s2) ? o2.nn : unknown(o3, s3) ? o3.nn : 0xffu;
                    const unsigned long long need = __ballot(want != 0xffu);
                    if (!need) break;
                    const int src = __ffsll((long long)need) - 1;
                    const uint32_t nnv = (uint32_t)__builtin_amdgcn_readlane((int)want, src);
                    const int q0 = fc0.mc - 1 + (int)(nnv & 3u), q1 = fc1.mc - 1 + (int)((nnv >> 2) & 3u), q2 = fc2.mc - 1 + (int)(nnv >> 4);
                    uint32_t found = 0;
                    if (lane == src) {
                        found = leaf_name(L, W, P, pack_leaf(q0, q1, q2));
                        if (found != 0xffffffffu) tab[nnv] = found;
                    }
                    found = (uint32_t)__builtin_amdgcn_readlane((int)found, src);
                    if (found == 0xffffffffu) {
                        // no room for this leaf: its points are dropped from this pass, the host runs another one
                        if (P.local_leaves) err |= ERR_LOCAL_LEAVES;
                        if (o0.nn == nnv) o0.key = KEY_EMPTY;
                        if (o1.nn == nnv) o1.key = KEY_EMPTY;
                        if (o2.nn == nnv) o2.key = KEY_EMPTY;
                        if (o3.nn == nnv) o3.key = KEY_EMPTY;
                    } else {
                        if (o0.nn == nnv) s0 = found;
                        if (o1.nn == nnv) s1 = found;
                        if (o2.nn == nnv) s2 = found;
                        if (o3.nn == nnv) s3 = found;
                    }
                }
            }
            // KEY_EMPTY stays all ones
            o0.key |= s0 << CELL_BITS; o1.key |= s1 << CELL_BITS; o2.key |= s2 << CELL_BITS; o3.key |= s3 << CELL_BITS;
        }
        if ((MODE != 1 || __builtin_expect(slow_step, 0)) && !(ablate & 64u)) {
            // the points carry leaf lattice coordinates here; one leaf and its name are cached in scalar registers
            int mm = 0;
            mm |= o0.key != KEY_EMPTY ? (o0.l0 ^ cl0) | (o0.l1 ^ cl1) | (o0.l2 ^ cl2) : 0;
            mm |= o1.key != KEY_EMPTY ? (o1.l0 ^ cl0) | (o1.l1 ^ cl1) | (o1.l2 ^ cl2) : 0;
            mm |= o2.key != KEY_EMPTY ? (o2.l0 ^ cl0) | (o2.l1 ^ cl1) | (o2.l2 ^ cl2) : 0;
            mm |= o3.key != KEY_EMPTY ? (o3.l0 ^ cl0) | (o3.l1 ^ cl1) | (o3.l2 ^ cl2) : 0;
            const bool mism = mm != 0;
            if (cache_valid && __ballot(mism) == 0ull) {
                // the whole step lies in the cached leaf (the common case); KEY_EMPTY stays all ones
                const uint32_t hi_bits = cache_id << CELL_BITS;
                o0.key |= hi_bits; o1.key |= hi_bits; o2.key |= hi_bits; o3.key |= hi_bits;
            } else {
                // general case: resolve the distinct leaves of this step one at a time
                unsigned pend = (o0.key != KEY_EMPTY ? 1u : 0u) | (o1.key != KEY_EMPTY ? 2u : 0u) | (o2.key != KEY_EMPTY ? 4u : 0u) |
                                (o3.key != KEY_EMPTY ? 8u : 0u);
                for (;;) {
                    const unsigned long long need = __ballot(pend != 0u);
                    if (!need) break;
                    const int src = __ffsll((long long)need) - 1;
                    const int slot = __ffs((int)pend) - 1;   // meaningful in lane src
                    const int m0 = slot == 0 ? o0.l0 : slot == 1 ? o1.l0 : slot == 2 ? o2.l0 : o3.l0;
                    const int m1 = slot == 0 ? o0.l1 : slot == 1 ? o1.l1 : slot == 2 ? o2.l1 : o3.l1;
                    const int m2 = slot == 0 ? o0.l2 : slot == 1 ? o1.l2 : slot == 2 ? o2.l2 : o3.l2;
                    const int s0 = __builtin_amdgcn_readlane(m0, src), s1 = __builtin_amdgcn_readlane(m1, src), s2 = __builtin_amdgcn_readlane(m2, src);
                    if (!(cache_valid && s0 == cl0 && s1 == cl1 && s2 == cl2)) {
                        uint32_t found = 0;
                        if (lane == src) found = leaf_name(L, W, P, pack_leaf(s0, s1, s2));
                        cache_id = (uint32_t)__builtin_amdgcn_readlane((int)found, src);
                        if (P.local_leaves && cache_id == 0xffffffffu) err |= ERR_LOCAL_LEAVES;
                        cl0 = s0; cl1 = s1; cl2 = s2;
                        cache_valid = true;
                    }
                    const uint32_t hi_bits = cache_id << CELL_BITS;
                    const bool lost = cache_id == 0xffffffffu;
                    if ((pend & 1u) && o0.l0 == s0 && o0.l1 == s1 && o0.l2 == s2) { pend &= ~1u; o0.key = lost ? KEY_EMPTY : (o0.key | hi_bits); }
                    if ((pend & 2u) && o1.l0 == s0 && o1.l1 == s1 && o1.l2 == s2) { pend &= ~2u; o1.key = lost ? KEY_EMPTY : (o1.key | hi_bits); }
                    if ((pend & 4u) && o2.l0 == s0 && o2.l1 == s1 && o2.l2 == s2) { pend &= ~4u; o2.key = lost ? KEY_EMPTY : (o2.key | hi_bits); }
                    if ((pend & 8u) && o3.l0 == s0 && o3.l1 == s1 && o3.l2 == s2) { pend &= ~8u; o3.key = lost ? KEY_EMPTY : (o3.key | hi_bits); }
                }
            }
        }

        // ---- runs inside the lane ----
        // A lane's 4 consecutive points form 1 run (the usual case) or a head run, a tail run and up to two
        // runs in between.  The tail (the whole lane if it is one run) takes part in a segmented scan over
        // lanes; the head is handed to the previous lane, whose chain it ends; a run in between is complete
        // as it is.  So a run that spans several lanes is inserted into the workgroup table once, by the lane
        // where its chain ends, and most steps need one table insert per lane at most (LDS atomics are the
        // scarcest resource of this kernel).
        const uint32_t k0 = o0.key, k1 = o1.key, k2 = o2.key, k3 = o3.key;
        const bool e1 = k1 == k0, e2 = k2 == k1, e3 = k3 == k2;
        const int nb = (e1 ? 0 : 1) + (e2 ? 0 : 1) + (e3 ? 0 : 1);   // boundaries inside the lane
        const bool single = nb == 0, multi = nb != 0;
        const PointAdd a0 = point_add(cw.x), a1 = point_add(cw.y), a2 = point_add(cw.z), a3 = point_add(cw.w);
        Run32 all;   // the four points together
        all.key = k3;
        all.qx = (o0.q0 + o1.q0) + (o2.q0 + o3.q0);
        all.qy = (o0.q1 + o1.q1) + (o2.q1 + o3.q1);
        all.qz = (o0.q2 + o1.q2) + (o2.q2 + o3.q2);
        all.cr = (a0.cr + a1.cr) + (a2.cr + a3.cr);
        all.gb = (a0.gb + a1.gb) + (a2.gb + a3.gb);
        all.tile = (a0.tile | a1.tile) | (a2.tile | a3.tile);
        Run32 H;     // head: the points before the first boundary
        {
            const uint32_t m1 = e1 ? ~0u : 0u, m2 = (e1 && e2) ? ~0u : 0u;
            H.key = k0;
            H.qx = o0.q0 + (o1.q0 & m1) + (o2.q0 & m2);
            H.qy = o0.q1 + (o1.q1 & m1) + (o2.q1 & m2);
            H.qz = o0.q2 + (o1.q2 & m1) + (o2.q2 & m2);
            H.cr = a0.cr + (a1.cr & m1) + (a2.cr & m2);
            H.gb = a0.gb + (a1.gb & m1) + (a2.gb & m2);
            H.tile = a0.tile | (a1.tile & m1) | (a2.tile & m2);
        }
        Run32 T;     // tail: the points after the last boundary
        {
            const uint32_t n2 = e3 ? ~0u : 0u, n1 = (e3 && e2) ? ~0u : 0u;
            T.key = k3;
            T.qx = o3.q0 + (o2.q0 & n2) + (o1.q0 & n1);
            T.qy = o3.q1 + (o2.q1 & n2) + (o1.q1 & n1);
            T.qz = o3.q2 + (o2.q2 & n2) + (o1.q2 & n1);
            T.cr = a3.cr + (a2.cr & n2) + (a1.cr & n1);
            T.gb = a3.gb + (a2.gb & n2) + (a1.gb & n1);
            T.tile = a3.tile | (a2.tile & n2) | (a1.tile & n1);
        }
        Run32 X;     // what the lane contributes to the scan
        X.key = k3;
        X.qx = single ? all.qx : T.qx; X.qy = single ? all.qy : T.qy; X.qz = single ? all.qz : T.qz;
        X.cr = single ? all.cr : T.cr; X.gb = single ? all.gb : T.gb; X.tile = single ? all.tile : T.tile;
        // ---- segmented inclusive scan over chains of lanes; chains are cut every 8 lanes so that three
        // DPP steps (1, 2, 4) cover them completely
        const uint32_t prev_xkey = (uint32_t)dpp_shr<1>((int)k3);   // 0 in the first lane of a row of 16
        const int flag0 = (single && (lane & 7) != 0 && prev_xkey == k0) ? 0 : 1;   // 1: the lane starts a chain
        int flag = flag0;
        if (!(ablate & 16u)) {
            scan_step<1>(X, flag);
            scan_step<2>(X, flag);
            scan_step<4>(X, flag);
        }
        // ---- where chains end; the head of the next lane, if it continues this chain ----
        // (cross-lane reads first, into plain variables: inside a short-circuit they would run with part
        // of the wave switched off and read zeros from those lanes)
        const bool row_first = (lane & 15) == 0, row_last = (lane & 15) == 15;
        const int next_flag0 = dpp_shl<1>(flag0), next_multi = dpp_shl<1>(multi ? 1 : 0);
        const uint32_t next_k0 = (uint32_t)dpp_shl<1>((int)k0);
        const bool tail_final = row_last | (next_flag0 != 0);
        const bool take = !row_last & (next_multi != 0) & (next_k0 == k3);
        if (!(ablate & 32u)) {
            const int tm = take ? -1 : 0;
            X.qx += (uint32_t)(dpp_shl<1>((int)H.qx) & tm);
            X.qy += (uint32_t)(dpp_shl<1>((int)H.qy) & tm);
            X.qz += (uint32_t)(dpp_shl<1>((int)H.qz) & tm);
            X.cr += (uint32_t)(dpp_shl<1>((int)H.cr) & tm);
            X.gb += (uint32_t)(dpp_shl<1>((int)H.gb) & tm);
            X.tile |= (uint32_t)(dpp_shl<1>((int)H.tile) & tm);
        }
        // this lane's head is taken by the previous lane under exactly the condition `take` has there
        const bool head_taken = multi & !row_first & (prev_xkey == k0);

        // ---- table inserts ----
        // what a lane has to insert, in this order: its chain (if it ends here), its head (if nobody took
        // it), the run(s) between head and tail.  Round 0 takes the first of them, which is all there is in
        // most steps of a scan-ordered cloud.
        if (!(ablate & 4u)) {
            const bool have_t = tail_final & (k3 != KEY_EMPTY);
            const bool have_h = multi & !head_taken & (k0 != KEY_EMPTY);
            Run32 M1, M2;   // nb == 2: one run in between (everything but head and tail); nb == 3: points 1 and 2
            M1.key = M2.key = KEY_EMPTY;
            M1.qx = M1.qy = M1.qz = M1.cr = M1.gb = M1.tile = 0;
            M2 = M1;
            if (__ballot(nb >= 2) != 0ull) {
                const bool three = nb == 3;
                const uint32_t sm = three ? ~0u : 0u;   // nb == 3: point 2 is a run of its own
                M1.key = nb >= 2 ? (e1 ? k2 : k1) : KEY_EMPTY;
                M1.qx = all.qx - H.qx - T.qx - (o2.q0 & sm);
                M1.qy = all.qy - H.qy - T.qy - (o2.q1 & sm);
                M1.qz = all.qz - H.qz - T.qz - (o2.q2 & sm);
                M1.cr = all.cr - H.cr - T.cr - (a2.cr & sm);
                M1.gb = all.gb - H.gb - T.gb - (a2.gb & sm);
                M1.tile = three ? a1.tile : (e1 ? a2.tile : (a1.tile | (e2 ? a2.tile : 0u)));
                M2.key = three ? k2 : KEY_EMPTY;
                M2.qx = o2.q0; M2.qy = o2.q1; M2.qz = o2.q2; M2.cr = a2.cr; M2.gb = a2.gb; M2.tile = a2.tile;
            }
            const bool have_m1 = M1.key != KEY_EMPTY, have_m2 = M2.key != KEY_EMPTY;
            const int i_h = have_t ? 1 : 0, i_m1 = i_h + (have_h ? 1 : 0), i_m2 = i_m1 + (have_m1 ? 1 : 0);
            const int n_items = i_m2 + (have_m2 ? 1 : 0);
#pragma unroll 1
            for (int round = 0; round < 4; round++) {
                if (round > 0 && __ballot(n_items > round) == 0ull) break;
                const bool s_t = have_t & (round == 0), s_h = have_h & (round == i_h), s_m1 = have_m1 & (round == i_m1), s_m2 = have_m2 & (round == i_m2);
                Run32 r;
                r.key = s_t ? k3 : s_h ? k0 : s_m1 ? M1.key : s_m2 ? M2.key : KEY_EMPTY;
                r.qx = s_t ? X.qx : s_h ? H.qx : s_m1 ? M1.qx : M2.qx;
                r.qy = s_t ? X.qy : s_h ? H.qy : s_m1 ? M1.qy : M2.qy;
                r.qz = s_t ? X.qz : s_h ? H.qz : s_m1 ? M1.qz : M2.qz;
                r.cr = s_t ? X.cr : s_h ? H.cr : s_m1 ? M1.cr : M2.cr;
                r.gb = s_t ? X.gb : s_h ? H.gb : s_m1 ? M1.gb : M2.gb;
                r.tile = s_t ? X.tile : s_h ? H.tile : s_m1 ? M1.tile : M2.tile;
                lds_insert(L, W, P, r, r.key != KEY_EMPTY);
            }
        }
        cx = nx; cy = ny; cz = nz; cw = nw;
    }

    if (ablate & 128u) { if (bn0 == 1.2345f) W.bboxes[0] = bn0 + bx0; return; }   // diagnostics: no epilogue at all
    // ---- errors of this wave, bounding box of its range (input of the octree replay) ----
    {
        for (int s = 32; s > 0; s >>= 1) err |= (uint32_t)__shfl_xor((int)err, s, 64);
        if (lane == 0 && err) atomicOr(&W.ctrl[C_ERR], err);
        const float lo3[3] = {bn0, bn1, bn2}, hi3[3] = {bx0, bx1, bx2};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float vlo = lo3[a], vhi = hi3[a];
            for (int s = 32; s > 0; s >>= 1) {
                vlo = fminf(vlo, __shfl_down(vlo, s, 64));
                vhi = fmaxf(vhi, __shfl_down(vhi, s, 64));
            }
            if (lane == 0) {
                W.bboxes[(size_t)range * 6 + a] = vlo;
                W.bboxes[(size_t)range * 6 + 3 + a] = vhi;
            }
        }
    }

    // ---- flush: 8 lanes per table entry update one 64-byte record with returning adds ----
    // All adds of a lane are issued before the first result is looked at, so that their round
    // trips overlap (this is the serial tail of the kernel: nothing else is in flight any more).
    __syncthreads();
    if (P.local_leaves) {
        // local leaf slots -> global leaf ids (grids), one lookup per leaf and workgroup
        if (threadIdx.x < LOCAL_LEAVES) {
            // normally all there already (leaf_name); only a lookup that failed is tried again
            const unsigned long long lk = L.leaf_tab[threadIdx.x];
            if (lk != 0ull && L.leaf_gid[threadIdx.x] >= 0xfffffffeu) L.leaf_gid[threadIdx.x] = leaf_lookup(W, P.leaf_mask, lk);
        }
        __syncthreads();
    }
    const int sub = threadIdx.x & 7;
    constexpr int FLUSH_ITERS = LTAB / (K1_THREADS / 8);
    uint32_t fkey[FLUSH_ITERS];
    unsigned long long fold[FLUSH_ITERS];
    uint32_t used = 0;   // entries in use (lanes with sub == 0 count them)
#pragma unroll
    for (int it = 0; it < FLUSH_ITERS; it++) {
        const int e = (threadIdx.x >> 3) + it * (K1_THREADS / 8);
        uint32_t k = L.key[e];
        if (ablate & 8u) k = KEY_EMPTY;
        if (P.local_leaves && k != KEY_EMPTY) {
            const uint32_t gid = L.leaf_gid[k >> CELL_BITS];
            k = gid == 0xffffffffu ? KEY_EMPTY : ((gid << CELL_BITS) | (k & ((1u << CELL_BITS) - 1)));
        }
        fkey[it] = k;
        fold[it] = ~0ull;
        if (k == KEY_EMPTY) continue;
        used += sub == 0 ? 1u : 0u;
        const uint32_t t = L.tile[e];
        const unsigned long long ea = L.a[e], eb = L.b[e], ec = L.c[e], ed = L.d[e];
        const unsigned long long cnt = ed & 0xffffull;
        unsigned long long val;
        switch (sub) {
        case 0: val = ea; break;                                             // sum qx
        case 1: val = eb; break;
        case 2: val = ec & ((1ull << 40) - 1); break;
        case 3: val = (cnt << 32) | ((ed >> 16) & 0xffffffull); break;      // count << 32 | sum r
        case 4: val = ((ed >> 40) << 32) | (ec >> 40); break;               // sum g << 32 | sum b
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
        fold[it] = atomicAdd(&record_ptr(W, k)[sub], val);
    }
#pragma unroll
    for (int it = 0; it < FLUSH_ITERS; it++) {
        const uint32_t k = fkey[it];
        if (sub == 3 && k != KEY_EMPTY && (fold[it] >> 32) == 0) {
            // first touch of this record in this call: list it, set its bit, count it in its bitmap slice
            const uint32_t at = atomicAdd(&L.nfresh, 1u);
            if (P.want_list) L.fresh[at] = k;
            mark_occupied(W, k);
            const uint32_t sl = slice_of(k);
            uint32_t hs = (sl * 0x9E3779B1u) >> 24;   // HIST = 2^8
            bool counted = false;
            for (int probe = 0; probe < 8 && !counted; probe++) {
                const uint32_t tag = atomicCAS(&L.htag[hs], 0u, sl + 1u);
                if (tag == 0u || tag == sl + 1u) {
                    atomicAdd(&L.hcnt[hs], 1u);
                    counted = true;
                }
                hs = (hs + 1) & (HIST - 1);
            }
            if (!counted) atomicAdd(&W.seg_count[sl], 1u);
        }
    }
    for (int off = 32; off > 0; off >>= 1) used += (uint32_t)__shfl_xor((int)used, off, 64);
    if ((threadIdx.x & 63) == 0 && used) atomicAdd(&L.nused, used);
    __syncthreads();
    if (threadIdx.x < HIST && L.htag[threadIdx.x]) atomicAdd(&W.seg_count[L.htag[threadIdx.x] - 1u], L.hcnt[threadIdx.x]);
    const uint32_t nfresh = L.nfresh;
    if (threadIdx.x == 0) {
        // how the table fared: the host sizes the workgroups of the next call by it
        if (L.nfallback) atomicAdd(&W.ctrl[C_FALLBACK], L.nfallback);
        atomicMax(&W.ctrl[C_MAXLOAD], L.nused);
        if (L.nused) atomicAdd(&W.ctrl[C_FLUSHED], L.nused);
    }
    if (!P.want_list) {
        // octree path: the finalize pass finds the records through the occupancy bitmaps, so the count is
        // all that is needed here, and nobody waits for this add
        if (threadIdx.x == 0 && nfresh) atomicAdd(&W.ctrl[C_COUNT], nfresh);
        return;
    }
    if (threadIdx.x == 0 && nfresh) L.fresh_base = atomicAdd(&W.ctrl[C_COUNT], nfresh);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nfresh; i += K1_THREADS) {
        const uint32_t idx = L.fresh_base + i;
        if (idx < P.list_cap) W.occupied[idx] = L.fresh[i];
        else atomicOr(&W.ctrl[C_ERR], ERR_LIST_FULL);
    }
}

#include "voxel_k1_fast.inc"
#include "voxel_partition.inc"

// ---------------------------------------------------------------------------
// K2: octree bounding-box replay / global grid box
// ---------------------------------------------------------------------------
// Results straight into the host's pinned words: the host polls them instead of waiting for the stream
// (a blocking stream wait wakes up several microseconds late).
// Every control word travels as one 64-bit store with the sequence number in its upper half, so the host
// can tell word by word what has arrived: no release fence (a system-scope release writes back the whole
// L2, which K1 has just filled with dirty records) and no second store behind it.
__device__ __forceinline__ void publish(const uint32_t *ctrl, uint32_t *host_out, uint32_t seq) {
    const int tid = threadIdx.x;
    if (tid < C_SEQ) {
        const uint32_t v = __hip_atomic_load(&ctrl[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_out) + tid, ((unsigned long long)seq << 32) | v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Growth of pcl::octree::OctreePointCloud's box is sequential in input order, but a range
// whose box lies inside the current octree box cannot trigger a growth step, so only the few
// ranges that do are re-read point by point.
__global__ void __launch_bounds__(1024) octree_replay_kernel(VoxParams P, const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ z, const float *__restrict__ bboxes,
                                                            uint32_t *__restrict__ ctrl, const unsigned long long *__restrict__ leaf_keys,
                                                            uint32_t leaf_cap, uint32_t *__restrict__ next_head, uint32_t next_head_words,
                                                            uint32_t *__restrict__ host_out, uint32_t seq) {
    // housekeeping this single workgroup has threads to spare for: it zeroes the control block the NEXT call
    // on this workspace will use (the two blocks alternate, so no memset sits in front of that call's first
    // kernel) -- at the very end, after the results have gone out to the host, which is waiting for them
    const auto zero_next_head = [&]() {
        for (uint32_t i = threadIdx.x; i < next_head_words; i += 1024) next_head[i] = 0u;
    };
    __shared__ double s_mn[3], s_mx[3];
    __shared__ int s_resolved;
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
        __syncthreads();
        // results straight into the host's pinned words: the host only waits for the stream
        publish(ctrl, host_out, seq);
        zero_next_head();
        return;
    }

    if (tid == 0) {
        for (int a = 0; a < 3; a++) { s_mn[a] = P.mn0[a]; s_mx[a] = P.mx0[a]; s_shift[a] = 0; }
        s_depth = P.depth0;
        s_events = 0;
    }
    // (the wave boxes are read from global memory: K1 has just written them, they sit in L2; staging them
    // in 96 KB of LDS was measured to cost more than it saved)
    __syncthreads();

    const double eps = (double)FLT_EPSILON;
    // one step of adoptBoundingBoxToPoint's loop: double the box, keeping the corner on the axes in `up`
    auto grow = [&](const bool up[3]) {
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
    };
    uint32_t range = 0;
    while (range < nranges) {
        // first range >= `range` whose box sticks out of the current octree box
        if (tid == 0) s_first = ~0ull;
        __syncthreads();
        {
            const double mn0 = s_mn[0], mn1 = s_mn[1], mn2 = s_mn[2], mx0 = s_mx[0], mx1 = s_mx[1], mx2 = s_mx[2];
            uint32_t mine = 0xffffffffu;
            for (uint32_t c = range + tid; c < nranges; c += 1024) {
                float b[6];
                for (int i = 0; i < 6; i++) b[i] = bboxes[(size_t)c * 6 + i];
                const bool viol = (double)b[0] < mn0 || (double)b[1] < mn1 || (double)b[2] < mn2 ||
                                  (double)b[3] >= mx0 || (double)b[4] >= mx1 || (double)b[5] >= mx2;
                if (viol) { mine = c; break; }
            }
            // one LDS atomic per wave, not per lane: after a growth step most ranges still stick out
            for (int off = 32; off > 0; off >>= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, off, 64));
            if ((tid & 63) == 0 && mine != 0xffffffffu) atomicMin(&s_first, (unsigned long long)mine);
        }
        __syncthreads();
        const unsigned long long hit = s_first;
        __syncthreads();
        if (hit == ~0ull) break;

        // Shortcut on the range's box: a point triggers growth when it violates the octree box, and the
        // step it takes depends only on the axes where it lies above the box.  If the range sticks out
        // below only, or above on exactly one axis and nowhere below, every triggering point of the range
        // has the same pattern, so the steps follow from the box of the range without reading its points.
        if (tid == 0) {
            float b[6];
            for (int i = 0; i < 6; i++) b[i] = bboxes[(size_t)hit * 6 + i];
            int resolved = 0;
            for (;;) {
                bool up[3], any_low = false;
                int n_up = 0;
                for (int a = 0; a < 3; a++) {
                    up[a] = (double)b[3 + a] >= s_mx[a];
                    n_up += up[a] ? 1 : 0;
                    any_low |= (double)b[a] < s_mn[a];
                }
                if (n_up == 0 && !any_low) { resolved = 1; break; }
                if (!(n_up == 0 || (n_up == 1 && !any_low))) break;   // mixed patterns: replay point by point
                if (s_depth >= 31) { atomicOr(&ctrl[C_ERR], ERR_DEPTH); resolved = 1; break; }
                grow(up);
            }
            s_resolved = resolved;
        }
        __syncthreads();
        if (s_resolved) {
            range = (uint32_t)hit + 1;
            continue;
        }

        // replay that range in index order, a tile of 4096 points at a time
        size_t r_lo = (size_t)hit * P.per_wave;
        size_t r_hi = r_lo + P.per_wave < P.n ? r_lo + P.per_wave : P.n;
        if (P.range_base_q != 0u) {
            r_lo = (size_t)range_first_step((uint32_t)hit, P.range_base_q, P.range_inc_q) * WAVE_STEP;
            r_hi = (size_t)range_first_step((uint32_t)hit + 1u, P.range_base_q, P.range_inc_q) * WAVE_STEP;
            r_lo = r_lo < P.n ? r_lo : P.n;
            r_hi = r_hi < P.n ? r_hi : P.n;
        }
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
                    unsigned long long mine = ~0ull;
                    for (int j = 0; j < 4; j++) {
                        const size_t idx = base + j;
                        if (idx < from || idx >= r_hi) continue;
                        if (!(isfinite(qx[j]) && isfinite(qy[j]) && isfinite(qz[j]))) continue;
                        const bool viol = (double)qx[j] < mn0 || (double)qy[j] < mn1 || (double)qz[j] < mn2 ||
                                          (double)qx[j] >= mx0 || (double)qy[j] >= mx1 || (double)qz[j] >= mx2;
                        if (viol) { mine = (unsigned long long)idx; break; }
                    }
                    // lanes hold ascending indices: the lowest lane with a violation has the wave's minimum
                    const unsigned long long vote = __ballot(mine != ~0ull);
                    if (vote != 0ull && (tid & 63) == __ffsll((long long)vote) - 1) atomicMin(&s_first, mine);
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
                        grow(up);
                    }
                }
                from = (size_t)pidx + 1;
                __syncthreads();
            }
        }
        range = (uint32_t)hit + 1;
    }
    // the finalize pass orders leaves by the Morton code of their final keys: check here that it can
    // (depth, key range), so that it has no error of its own to report
    __syncthreads();
    {
        const int depth = s_depth;
        if (depth > 14) {
            if (tid == 0) atomicOr(&ctrl[C_ERR], ERR_DEPTH);
        } else {
            bool bad = false;
            for (uint32_t q = tid; q < leaf_cap; q += 1024) {
                const unsigned long long lp = leaf_keys[q];
                if (lp == 0ull) continue;
                for (int a = 0; a < 3; a++) {
                    const long long lk = (long long)unpack_leaf(lp, a) + s_shift[a];
                    bad |= lk < 0 || lk >= ((long long)1 << depth);
                }
            }
            if (bad) atomicOr(&ctrl[C_ERR], ERR_LEAF_RANGE);
        }
    }
    if (tid == 0) {
        ctrl[C_DEPTH] = (uint32_t)s_depth;
        ctrl[C_EVENTS] = (uint32_t)s_events;
        for (int a = 0; a < 3; a++) {
            ctrl[C_SHIFT + 2 * a] = (uint32_t)((unsigned long long)s_shift[a] & 0xffffffffu);
            ctrl[C_SHIFT + 2 * a + 1] = (uint32_t)((unsigned long long)s_shift[a] >> 32);
        }
    }
    __syncthreads();
    publish(ctrl, host_out, seq);
    zero_next_head();
}

// ---------------------------------------------------------------------------
// K3: output-order keys of the plain grid's sort path (index spaces beyond 2^28 cells)
// ---------------------------------------------------------------------------
// pcl::VoxelGrid's idx = i + j*div_x + k*div_x*div_y
__global__ void __launch_bounds__(256) make_sort_keys_kernel(VoxParams P, VoxWork W, uint32_t m, unsigned long long *__restrict__ sort_keys,
                                                            uint32_t *__restrict__ sort_vals) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t key = W.occupied[r];
    const uint32_t cell = key & ((1u << CELL_BITS) - 1), leaf_id = key >> CELL_BITS;
    const int c[3] = {(int)(cell % GRID_DIM), (int)((cell / GRID_DIM) % GRID_DIM), (int)(cell / (GRID_DIM * GRID_DIM))};
    const unsigned long long lp = W.leaf_keys[leaf_id];
    long long d[3];
    for (int a = 0; a < 3; a++) d[a] = (long long)(c[a] + P.ib[a] + 64 * unpack_leaf(lp, a) - 2) - (long long)(int)W.ctrl[C_MINB + a];
    const long long dx = (int)W.ctrl[C_DIVB], dy = (int)W.ctrl[C_DIVB + 1];
    sort_keys[r] = (unsigned long long)(d[0] + d[1] * dx + d[2] * dx * dy);
    sort_vals[r] = key;
}

// ---------------------------------------------------------------------------
// K4: emit in output order and clean the records
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) emit_and_clean_kernel(VoxParams P, VoxWork W, uint32_t m, const uint32_t *__restrict__ sorted_keys,
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
        double vox[3];
        for (int a = 0; a < 3; a++) vox[a] = (double)(c[a] + P.ib[a] + 64 * unpack_leaf(lp, a) - 2);
        const unsigned long long cr = w23.y, gb = w45.x;
        const uint32_t cnt = (uint32_t)(cr >> 32);
        const double scale = P.q_unit / (double)cnt;
        // mean = (voxel + mean position inside the voxel) / inv_leaf; one rounding to fp32 at the end
        ox[r] = (float)(vox[0] * P.vox_unit + (double)(long long)w01.x * scale);
        oy[r] = (float)(vox[1] * P.vox_unit + (double)(long long)w01.y * scale);
        oz[r] = (float)(vox[2] * P.vox_unit + (double)(long long)w23.x * scale);
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
    const uint32_t bit_cell = key & ((1u << CELL_BITS) - 1);
    atomicAnd(&W.bitmaps[(size_t)(key >> CELL_BITS) * BITWORDS + (bit_cell >> 5)], ~(1u << (bit_cell & 31u)));
}

// ---------------------------------------------------------------------------
// Sort-free output order for the octree path: leaves in Morton order of their final keys,
// cells in ascending index inside a leaf = rank of a bit in the occupancy bitmaps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void emit_record(const VoxParams &P, const VoxWork &W, unsigned long long lp, uint32_t key, uint32_t r,
                                            float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz, uint32_t *__restrict__ ow) {
    ulonglong2 *rec = reinterpret_cast<ulonglong2 *>(record_ptr(W, key));
    const ulonglong2 w01 = rec[0], w23 = rec[1], w45 = rec[2], w67 = rec[3];
    const uint32_t cell = key & ((1u << CELL_BITS) - 1);
    const int c[3] = {(int)(cell % GRID_DIM), (int)((cell / GRID_DIM) % GRID_DIM), (int)(cell / (GRID_DIM * GRID_DIM))};
    double vox[3];
    for (int a = 0; a < 3; a++) vox[a] = (double)(c[a] + P.ib[a] + 64 * unpack_leaf(lp, a) - 2);
    const unsigned long long cr = w23.y, gb = w45.x;
    const uint32_t cnt = (uint32_t)(cr >> 32);
    // mean = voxel origin + mean offset (f64, one division), one rounding to fp32 at the end
    const double scale = P.q_unit / (double)cnt;
    ox[r] = (float)(vox[0] * P.vox_unit + (double)(long long)w01.x * scale);
    oy[r] = (float)(vox[1] * P.vox_unit + (double)(long long)w01.y * scale);
    oz[r] = (float)(vox[2] * P.vox_unit + (double)(long long)w23.x * scale);
    const float fn = (float)cnt;
    const uint32_t rr = (uint32_t)__fdiv_rn((float)(uint32_t)(cr & 0xffffffffu), fn);
    const uint32_t gg = (uint32_t)__fdiv_rn((float)(uint32_t)(gb >> 32), fn);
    const uint32_t bb = (uint32_t)__fdiv_rn((float)(uint32_t)(gb & 0xffffffffu), fn);
    uint32_t tile = (uint32_t)w67.y & 0xffu;
    for (int b = 0; b < 4; b++) {
        if ((w45.y >> (16 * b)) & 0xffffull) tile |= 1u << b;
        if ((w67.x >> (16 * b)) & 0xffffull) tile |= 16u << b;
    }
    ow[r] = (rr & 0xffu) | ((gg & 0xffu) << 8) | ((bb & 0xffu) << 16) | (tile << 24);
    const ulonglong2 zero = {0ull, 0ull};
    rec[0] = zero; rec[1] = zero; rec[2] = zero; rec[3] = zero;
}

// RANK_SEGS workgroups per leaf, each owning a contiguous slice of the leaf's bitmap: output base
// of the leaf (cells of all leaves that precede it in Morton order) + occupied cells in the
// earlier slices, ranks of the slice's cells from a popcount scan, then gather, emit and clean.
// Replaces the key sort: no pass over the outputs other than the emit itself.

// speculative != 0: launched before the host knows the outcome of the pass, into a result buffer sized from
// the previous call: the kernel takes the count from the control block and does nothing at all when the pass
// reported an error or the count exceeds `m_or_cap` (the host then runs it again, with the facts).
__global__ void __launch_bounds__(RANK_THREADS) rank_emit_kernel(VoxParams P, VoxWork W, uint32_t leaf_cap, uint32_t m_or_cap, int speculative,
                                                                 uint32_t *order, float *__restrict__ ox, float *__restrict__ oy,
                                                                 float *__restrict__ oz, uint32_t *__restrict__ ow) {
    // A workgroup's life is a chain of dependent memory round trips (it handles some sixty cells), so
    // everything that does not depend on loaded data is requested at once, up front: control words, this
    // leaf, every leaf's key and slice counts, the slice's bitmap words.  The cells in rank order go
    // through LDS, not through global memory (unless a slice has more than RANK_LDS_CELLS of them).
    constexpr uint32_t RANK_LDS_CELLS = 2048;
    __shared__ uint32_t wave_tot[RANK_THREADS / 64];
    __shared__ uint32_t s_cells[RANK_LDS_CELLS];
    const uint32_t p = blockIdx.x / RANK_SEGS, seg = blockIdx.x % RANK_SEGS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *bm = W.bitmaps + (size_t)p * BITWORDS;
    const int w_lo = (int)seg * SEG_WORDS;
    const int w_hi = min(w_lo + SEG_WORDS, BITWORDS);
    // ---- loads ----
    const uint32_t c_count = W.ctrl[C_COUNT], c_err = W.ctrl[C_ERR], c_depth = W.ctrl[C_DEPTH];
    uint32_t c_shift[6];
#pragma unroll
    for (int i = 0; i < 6; i++) c_shift[i] = W.ctrl[C_SHIFT + i];
    const unsigned long long lp = W.leaf_keys[p];
    uint32_t words[WORDS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < WORDS_PER_THREAD; i++) {
        const int w = w_lo + threadIdx.x * WORDS_PER_THREAD + i;
        words[i] = bm[min(w, w_hi - 1)];
    }
    const uint32_t q0 = min((uint32_t)threadIdx.x, leaf_cap - 1u);   // this thread's leaf in the first round of the loop below
    const unsigned long long lq0 = W.leaf_keys[q0];
    uint4 sc0[RANK_SEGS / 4];
    {
        const uint4 *sc = reinterpret_cast<const uint4 *>(W.seg_count + (size_t)q0 * RANK_SEGS);
#pragma unroll
        for (int v = 0; v < RANK_SEGS / 4; v++) sc0[v] = sc[v];
    }
    const uint32_t own_earlier = threadIdx.x < seg ? W.seg_count[p * RANK_SEGS + threadIdx.x] : 0u;   // seg <= RANK_SEGS <= RANK_THREADS
    // ---- what they say ----
    uint32_t m = m_or_cap;
    if (speculative) {
        m = c_count;
        if (c_err != 0u || m > m_or_cap || m == 0u) return;
    }
    if (lp == 0ull) return;
    const int depth = (int)c_depth;
    long long shift[3];
    for (int a = 0; a < 3; a++) shift[a] = (long long)(((unsigned long long)c_shift[2 * a + 1] << 32) | c_shift[2 * a]);
    const auto morton = [&](unsigned long long leaf, unsigned long long &code) {
        long long lk[3];
        bool bad = false;
        for (int a = 0; a < 3; a++) {
            lk[a] = (long long)unpack_leaf(leaf, a) + shift[a];
            if (lk[a] < 0 || lk[a] >= ((long long)1 << depth)) bad = true;
        }
        code = 0;
        for (int b = depth - 1; b >= 0; b--) {
            code = (code << 3) | (((unsigned long long)(lk[0] >> b) & 1) << 2) | (((unsigned long long)(lk[1] >> b) & 1) << 1) |
                   ((unsigned long long)(lk[2] >> b) & 1);
        }
        return !bad;
    };
    unsigned long long mine;
    if (depth > 14 || !morton(lp, mine)) {
        if (threadIdx.x == 0) atomicOr(&W.ctrl[C_ERR], depth > 14 ? ERR_DEPTH : ERR_LEAF_RANGE);
        return;   // the host cleans up through the occupied list
    }
    // ---- base: cells of the leaves that come first, plus this leaf's cells in earlier slices ----
    uint32_t before = own_earlier;
    for (uint32_t q = threadIdx.x; q < leaf_cap; q += RANK_THREADS) {
        unsigned long long lq = lq0;
        uint4 scq[RANK_SEGS / 4];
#pragma unroll
        for (int v = 0; v < RANK_SEGS / 4; v++) scq[v] = sc0[v];
        if (q >= RANK_THREADS) {   // more leaves than threads (rare): the later rounds load as they go
            lq = W.leaf_keys[q];
            const uint4 *sc = reinterpret_cast<const uint4 *>(W.seg_count + (size_t)q * RANK_SEGS);
#pragma unroll
            for (int v = 0; v < RANK_SEGS / 4; v++) scq[v] = sc[v];
        }
        unsigned long long other;
        if (lq != 0ull && q != p && morton(lq, other) && other < mine) {
#pragma unroll
            for (int v = 0; v < RANK_SEGS / 4; v++) before += scq[v].x + scq[v].y + scq[v].z + scq[v].w;
        }
    }
    // (earlier slices may already have been cleaned by their own workgroups, so their cells are counted
    // from the per-slice totals K1 accumulated, not from the bitmaps)
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
    // ---- ranks inside the slice: each lane owns WORDS_PER_THREAD consecutive bitmap words ----
    uint32_t mycount = 0;
#pragma unroll
    for (int i = 0; i < WORDS_PER_THREAD; i++) {
        const int w = w_lo + threadIdx.x * WORDS_PER_THREAD + i;
        if (w >= w_hi) words[i] = 0u;
        mycount += __popc(words[i]);
    }
    uint32_t inc = mycount;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    __shared__ uint32_t wave_cells[RANK_THREADS / 64];
    if (lane == 0) wave_tot[wave] = before;
    if (lane == 63) wave_cells[wave] = inc;
    __syncthreads();
    uint32_t base = 0, wbase = 0, total = 0;
    for (int w = 0; w < RANK_THREADS / 64; w++) {
        base += wave_tot[w];
        if (w < wave) wbase += wave_cells[w];
        total += wave_cells[w];
    }
    // phase 1: the slice's cells in rank order, the bitmap words cleaned
    uint32_t local = wbase + inc - mycount;
    if (base + total > m) {
        if (threadIdx.x == 0) atomicOr(&W.ctrl[C_ERR], ERR_LIST_FULL);
    }
    const bool in_lds = total <= RANK_LDS_CELLS;
#pragma unroll
    for (int i = 0; i < WORDS_PER_THREAD; i++) {
        uint32_t bits = words[i];
        const int w = w_lo + threadIdx.x * WORDS_PER_THREAD + i;
        if (bits) bm[w] = 0u;   // clean
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const uint32_t cell = (p << CELL_BITS) | (uint32_t)(w * 32 + b);
            if (in_lds) s_cells[local] = cell;
            else if (base + local < m) order[base + local] = cell;
            local++;
        }
    }
    __syncthreads();   // (this workgroup's order[] stores are visible to its own lanes from here on)
    // phase 2: gather, emit and clean, one cell per lane
    for (uint32_t i = threadIdx.x; i < total; i += RANK_THREADS) {
        const uint32_t r = base + i;
        if (r < m) emit_record(P, W, lp, in_lds ? s_cells[i] : __hip_atomic_load(&order[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), r, ox, oy, oz, ow);
    }
}

// ---------------------------------------------------------------------------
// Sort-free output order for the plain grid: pcl::VoxelGrid emits voxels by ascending
// idx = i + j * div_x + k * div_x * div_y.  A bitmap over that index space (it has at most 2^31
// cells by VoxelGrid's own rule; the bitmap path takes up to 2^28) turns the order into popcount ranks.
// ---------------------------------------------------------------------------
constexpr uint32_t GRID_BITMAP_MAX_CELLS = 1u << 28;
constexpr int GB_WORDS_PER_BLOCK = 1024;

__device__ __forceinline__ uint32_t voxelgrid_index(const VoxParams &P, const VoxWork &W, uint32_t key) {
    const uint32_t cell = key & ((1u << CELL_BITS) - 1), leaf_id = key >> CELL_BITS;
    const int c[3] = {(int)(cell % GRID_DIM), (int)((cell / GRID_DIM) % GRID_DIM), (int)(cell / (GRID_DIM * GRID_DIM))};
    const unsigned long long lp = W.leaf_keys[leaf_id];
    long long d[3];
    for (int a = 0; a < 3; a++) d[a] = (long long)(c[a] + P.ib[a] + 64 * unpack_leaf(lp, a) - 2) - (long long)(int)W.ctrl[C_MINB + a];
    const long long dx = (int)W.ctrl[C_DIVB], dy = (int)W.ctrl[C_DIVB + 1];
    return (uint32_t)(d[0] + d[1] * dx + d[2] * dx * dy);
}

// The five passes below can be launched before the host knows the outcome of the pass (right behind the replay
// kernel, like rank_emit_kernel on the octree path): they then take count and grid size from the control block
// and do nothing at all unless the pass succeeded and everything fits what the host provided for.
struct GridSpec {
    int on;                        // 0: m and nwords are the host's
    uint32_t m_cap;                // room in the result
    uint32_t words_cap;            // room in the index bitmap
    unsigned long long cells_max;  // largest index space the bitmap path takes
};
struct GridGate { uint32_t m, nwords; bool go; };
__device__ __forceinline__ GridGate grid_gate(const VoxWork &W, uint32_t m_host, uint32_t nwords_host, const GridSpec &spec) {
    if (!spec.on) return GridGate{m_host, nwords_host, true};
    const uint32_t m = W.ctrl[C_COUNT], err = W.ctrl[C_ERR];
    const unsigned long long cells = (unsigned long long)W.ctrl[C_DIVB] * W.ctrl[C_DIVB + 1] * W.ctrl[C_DIVB + 2];
    const unsigned long long nwords = (cells + 31) / 32;
    const bool go = err == 0u && m != 0u && m <= spec.m_cap && cells <= spec.cells_max && nwords <= spec.words_cap;
    return GridGate{m, (uint32_t)nwords, go};
}

// one bit per touched record; its index is kept for the passes that follow
__global__ void __launch_bounds__(256) grid_mark_kernel(VoxParams P, VoxWork W, uint32_t m_host, GridSpec spec, uint32_t *__restrict__ gbits,
                                                       uint32_t *__restrict__ gidx) {
    const GridGate gate = grid_gate(W, m_host, 0u, spec);
    if (!gate.go) return;
    const uint32_t m = gate.m;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t idx = voxelgrid_index(P, W, W.occupied[r]);
    gidx[r] = idx;
    atomicOr(&gbits[idx >> 5], 1u << (idx & 31u));
}

// per block of 1024 bitmap words: set bits before each word (inside the block), set bits of the block
__global__ void __launch_bounds__(256) grid_block_kernel(VoxWork W, GridSpec spec, const uint32_t *__restrict__ gbits, uint32_t nwords_host,
                                                        uint32_t *__restrict__ word_prefix, uint32_t *__restrict__ block_sum) {
    __shared__ uint32_t wave_tot[4];
    const GridGate gate = grid_gate(W, 0u, nwords_host, spec);
    if (!gate.go) return;
    const uint32_t nwords = gate.nwords;
    if (blockIdx.x * (uint32_t)GB_WORDS_PER_BLOCK >= nwords) return;   // (a speculative launch covers the whole bitmap buffer)
    const uint32_t w0 = blockIdx.x * GB_WORDS_PER_BLOCK + threadIdx.x * 4;
    uint32_t c[4];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c[i] = w0 + i < nwords ? __popc(gbits[w0 + i]) : 0u;
        mine += c[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = inc - mine, total = 0;
    for (int w = 0; w < 4; w++) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (w0 + i < nwords) word_prefix[w0 + i] = before;
        before += c[i];
    }
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

// exclusive scan of the block sums, one workgroup
__global__ void __launch_bounds__(1024) grid_blockscan_kernel(VoxWork W, GridSpec spec, uint32_t *__restrict__ block_sum, uint32_t nwords_host) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    const GridGate gate = grid_gate(W, 0u, nwords_host, spec);
    if (!gate.go) return;
    const uint32_t nblocks = (gate.nwords + GB_WORDS_PER_BLOCK - 1) / GB_WORDS_PER_BLOCK;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sum[i] : 0;
        uint32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; w++) wbase += wave_tot[w];
        const uint32_t c = carry;
        if (i < nblocks) block_sum[i] = c + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wbase + inc;
        __syncthreads();
    }
}

// rank of the record's bit = its output position; emit, clean the record and the leaf bitmap bit
__global__ void __launch_bounds__(256) grid_emit_kernel(VoxParams P, VoxWork W, uint32_t m_host, GridSpec spec, const uint32_t *__restrict__ gidx,
                                                       const uint32_t *__restrict__ gbits, const uint32_t *__restrict__ word_prefix,
                                                       const uint32_t *__restrict__ block_sum, float *__restrict__ ox, float *__restrict__ oy,
                                                       float *__restrict__ oz, uint32_t *__restrict__ ow) {
    const GridGate gate = grid_gate(W, m_host, 0u, spec);
    if (!gate.go) return;
    const uint32_t m = gate.m;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const uint32_t key = W.occupied[r], idx = gidx[r], w = idx >> 5;
    const uint32_t rank = block_sum[w / GB_WORDS_PER_BLOCK] + word_prefix[w] + __popc(gbits[w] & ((1u << (idx & 31u)) - 1u));
    if (rank < m) emit_record(P, W, W.leaf_keys[key >> CELL_BITS], key, rank, ox, oy, oz, ow);
    const uint32_t bit_cell = key & ((1u << CELL_BITS) - 1);
    atomicAnd(&W.bitmaps[(size_t)(key >> CELL_BITS) * BITWORDS + (bit_cell >> 5)], ~(1u << (bit_cell & 31u)));
}

// the index bitmap is left zeroed for the next call (after every rank has been read)
__global__ void __launch_bounds__(256) grid_unmark_kernel(VoxWork W, uint32_t m_host, GridSpec spec, const uint32_t *__restrict__ gidx,
                                                         uint32_t *__restrict__ gbits) {
    const GridGate gate = grid_gate(W, m_host, 0u, spec);
    if (!gate.go) return;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < gate.m) gbits[gidx[r] >> 5] = 0u;
}

// Error path of the octree variant (no list of touched records there): zero every record whose bit is
// set, and the bitmaps.  Same launch shape as rank_emit_kernel.
__global__ void __launch_bounds__(RANK_THREADS) clean_by_bitmap_kernel(VoxWork W) {
    const uint32_t p = blockIdx.x / RANK_SEGS, seg = blockIdx.x % RANK_SEGS;
    if (W.leaf_keys[p] == 0ull) return;
    uint32_t *bm = W.bitmaps + (size_t)p * BITWORDS;
    const int w_lo = (int)seg * SEG_WORDS, w_hi = min(w_lo + SEG_WORDS, BITWORDS);
    for (int w = w_lo + threadIdx.x; w < w_hi; w += RANK_THREADS) {
        uint32_t bits = bm[w];
        if (!bits) continue;
        bm[w] = 0u;
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            ulonglong2 *rec = reinterpret_cast<ulonglong2 *>(record_ptr(W, (p << CELL_BITS) | (uint32_t)(w * 32 + b)));
            const ulonglong2 zero = {0ull, 0ull};
            rec[0] = zero; rec[1] = zero; rec[2] = zero; rec[3] = zero;
        }
    }
}

// ---------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------
// An octree pass whose call has returned (DeferredResult): accumulate, replay and finalize kernels are in flight, the
// finalize kernel writes into `spec_dst`, sized from the passes before.  Whoever needs the outcome polls the workspace's
// pinned words for this pass's sequence number.  If the pass did not go through (an error word, more outputs than
// spec_dst holds) the result is computed again by an ordinary, waiting call, and the workspace's next user cleans up.
struct PendingVoxel : DeferredResult {
    std::recursive_mutex lock;   // (settle() may run the pass again on this thread, which first asks the workspace's pending pass -- this one -- for its outcome)
    std::shared_ptr<DeviceSoA> src, spec_dst, result;
    volatile unsigned long long *words = nullptr;
    hipStream_t stream = nullptr;
    uint32_t seq = 0, spec_cap = 0;
    float cellsize = 0;
    bool known = false, ok = false, settled = false;
    uint32_t err = 0, m = 0;
    bool leaf_split = true;                      // octree pass; false: the plain grid, whose finalize kernels ran only if the index space fits (gspec)
    GridSpec gspec{0, 0u, 0u, 0ull};
    bool partitioned = false;                    // the pass ran on a partitioned copy of the cloud ...
    uint32_t scatter = 0, steps_total = 0;       //   ... and this is how scattered the cloud was as it came (of how many wave steps)
    uint32_t leaves = 0;                         // leaf grids the pass used
    bool outcome_locked();                       // waits for the replay kernel's report; true: spec_dst holds the result
    bool outcome() { std::lock_guard<std::recursive_mutex> g(lock); return outcome_locked(); }
    std::shared_ptr<DeviceSoA> settle() override;
};

std::atomic<size_t> g_workspace_bytes{0};   // device memory held by voxel workspaces (cwipc_hip_workspace_bytes)

struct Workspace {
    int device = -1;
    size_t grid_bytes = 0;             // what this workspace has added to g_workspace_bytes
    int cus = 0, cus_device = -1;   // compute units of the device the workspace was last used on
    uint32_t leaf_cap = 0;     // leaf hash capacity = number of grids (power of two)
    size_t list_cap = 0;
    size_t bbox_cap = 0;
    void *head = nullptr;              // two blocks of ctrl | leaf_keys | seg_count, used by alternate passes
    size_t head_bytes = 0;             // bytes of one block
    int parity = 0;                    // block of the next pass
    uint32_t seq = 0;                  // sequence number of the last pass (never 0 once used)
    uint32_t last_m = 0;               // outputs of this thread's last octree pass (sizes the speculative result of the next)
    uint32_t last_m_grid = 0;          //   ... and of its last plain-grid pass
    int shrink = 0;                    // log2 of how much smaller than "one workgroup per CU" the workgroups are made (sparse clouds)
    int calm = 0;                      // calls in a row whose tables stayed less than a third full
    bool incoherent = false;           // smaller workgroups did not stop the overflows: stay with full-size ones
    bool no_fast = false;              // the fast accumulate kernel gave this kind of cloud back (ERR_FAST_PATH): use the general one
    int streak = 0;                    // octree passes of this kind in a row that went through without a retry
    int roomy = 0;                     // passes in a row that used at most a quarter of the leaf grids
    uint32_t shrink_to = 0;            // != 0: give the grids back and start again with this many (at the next call, when nothing is in flight)
    // A leaf grid is 20 MB: a thread that once met a cloud of many leaves must not sit on them while it filters camera tiles
    // of three or four.  Eight roomy passes in a row -> the next call reallocates for what these passes needed.
    void note_leaves(uint32_t leaves) {
        uint32_t need = 4;
        while (need < leaves) need <<= 1;
        if (need * 4 <= leaf_cap) {
            if (++roomy >= 8) shrink_to = need;
        } else {
            roomy = 0;
            shrink_to = 0;
        }
    }
    uint32_t *host_words = nullptr;    // page-locked: the replay kernel publishes the pass's control words here (64-bit, tagged with seq)
    std::shared_ptr<struct PendingVoxel> pending;   // the pass still in flight on this workspace, if the call that started it has returned
    size_t hint_n = 0;                 // the kind of call ws.shrink was learned on
    float hint_cell = 0.f;
    bool hint_split = true;
    bool head_clean[2] = {false, false};   // the block is known to be zero (the replay kernel of the pass before zeroed it)
    unsigned long long *leaf_keys = nullptr;
    unsigned long long *hash_keys = nullptr;
    uint32_t *hash_ids = nullptr;
    unsigned long long *records = nullptr;
    uint32_t *occupied = nullptr;
    uint32_t *order = nullptr;         // records in output order (finalize pass), list_cap entries
    uint32_t *gbits = nullptr;         // plain grid: bitmap over the VoxelGrid index space, its per-word and per-block prefixes
    uint32_t *gprefix = nullptr, *gblock = nullptr;
    size_t gwords_cap = 0;
    float *bboxes = nullptr;           // [2][bbox_cap][6]: the ranges' boxes the replay kernel reads; behind them room for boxes nobody reads
    uint32_t *dump_head = nullptr, *dump_ent = nullptr;   // what the fast accumulate kernel leaves for the merge kernel (r4) ...
    size_t dump_blocks = 0, dump_entries = 0;             //   ... for this many workgroups of this many table entries
    float *part = nullptr;             // partition pass: the cloud moved into spatial buckets, four planes of part_stride elements
    size_t part_stride = 0;
    uint32_t *part_hist = nullptr;     //   ... and its table: a row of PART_BUCKETS counts per range, the rows' sums by segments, the buckets' starts
    size_t part_rows = 0;              //   (room for this many rows)
    uint32_t *ctrl = nullptr;
    uint32_t *bitmaps = nullptr;
    uint32_t *seg_count = nullptr;
    float *faces = nullptr;            // device copy of the threshold table
    uint32_t faces_host[FACE_TABLE_WORDS];   // what the device copy holds
    double faces_mn0[3] = {0, 0, 0};   //   ... and what it was computed from
    double faces_res = 0;
    bool faces_valid = false;
    void drop_partition_buffers() {
        if (part) { (void)hipFree(part); g_workspace_bytes -= 16 * part_stride; }
        if (part_hist) (void)hipFree(part_hist);
        part = nullptr; part_hist = nullptr; part_stride = 0; part_rows = 0;
    }
    void drop_dump_buffers() {
        if (dump_head) (void)hipFree(dump_head);
        if (dump_ent) { (void)hipFree(dump_ent); g_workspace_bytes -= dump_blocks * dump_entries * DUMP_ENTRY_WORDS * 4; }
        dump_head = dump_ent = nullptr; dump_blocks = dump_entries = 0;
    }
    void release() {
        // also runs at thread exit, when the runtime may be gone: errors ignored
        drop_dump_buffers();
        if (head) (void)hipFree(head);
        if (records) (void)hipFree(records);
        if (occupied) (void)hipFree(occupied);
        if (order) (void)hipFree(order);
        if (gbits) (void)hipFree(gbits);
        if (gprefix) (void)hipFree(gprefix);
        if (gblock) (void)hipFree(gblock);
        gbits = gprefix = gblock = nullptr; gwords_cap = 0;
        if (bboxes) (void)hipFree(bboxes);
        drop_partition_buffers();
        if (faces) (void)hipFree(faces);
        if (host_words) (void)hipHostFree(host_words);
        host_words = nullptr;
        if (bitmaps) (void)hipFree(bitmaps);
        g_workspace_bytes -= grid_bytes;
        grid_bytes = 0;
        bitmaps = nullptr; seg_count = nullptr; head = nullptr; head_bytes = 0;
        leaf_keys = nullptr; records = nullptr; occupied = nullptr; order = nullptr; bboxes = nullptr; ctrl = nullptr; faces = nullptr;
        leaf_cap = 0; list_cap = 0; bbox_cap = 0;
        faces_valid = false;
    }
    ~Workspace() { release(); }
};

// Two workspaces per thread, used by alternate calls, each with its own stream (ThreadCtx::stream / stream_alt):
// a call's finalize kernel, still running when the call returns, works on grids the next call does not touch.
// The workspaces of threads that have ended wait here for the next thread (1.3 GB each: a program that starts a
// thread per frame must not allocate and free that much every time).
std::mutex g_ws_pool_mutex;
std::vector<Workspace *> *g_ws_pool = new std::vector<Workspace *>();   // never destroyed: threads may end after the statics

constexpr int MAX_WS = 2 + ThreadCtx::EXTRA_STREAMS;   // workspaces (and streams) a thread's calls rotate over, at most
struct WorkspaceLease {
    Workspace *ws[MAX_WS] = {nullptr, nullptr, nullptr, nullptr};
    Workspace &get(int which) {
        if (!ws[which]) {
            const int dev = current_device();
            std::lock_guard<std::mutex> lock(g_ws_pool_mutex);
            for (size_t i = 0; i < g_ws_pool->size(); i++) {
                if ((*g_ws_pool)[i]->device == dev) {   // one that already holds grids on this device
                    ws[which] = (*g_ws_pool)[i];
                    g_ws_pool->erase(g_ws_pool->begin() + (long)i);
                    break;
                }
            }
            if (!ws[which]) ws[which] = new Workspace();
        }
        return *ws[which];
    }
    ~WorkspaceLease() {
        // thread exit: a finalize kernel of this thread's last call may still be using a workspace
        bool any = false;
        for (int i = 0; i < MAX_WS; i++) any = any || ws[i];
        if (any) (void)hipDeviceSynchronize();
        // A pass of this thread that was handed out while it ran (ws.pending) reads its report from the workspace's pinned words
        // when somebody asks for the result -- possibly another thread, long after this one has gone and the workspace with
        // it.  The report is final now: take it (the outcome is cached in the pending pass, the words are not looked at again).
        // A workspace that goes to the pool keeps its `pending`: its next user learns from it whether the records were left clean.
        for (int i = 0; i < MAX_WS; i++)
            if (ws[i] && ws[i]->pending) (void)ws[i]->pending->outcome();
        // (r3: at most eight wait here -- a program with a thread per tile and frame finds one each; what threads held beyond
        // that, mostly second workspaces that a busy moment made them take, is given back to the device)
        std::vector<Workspace *> surplus;
        {
            std::lock_guard<std::mutex> lock(g_ws_pool_mutex);
            for (int i = 0; i < MAX_WS; i++) {
                if (!ws[i]) continue;
                if (g_ws_pool->size() < 8) g_ws_pool->push_back(ws[i]);
                else surplus.push_back(ws[i]);
            }
        }
        for (Workspace *w : surplus) delete w;
    }
};
thread_local WorkspaceLease t_ws;
thread_local int t_ws_next = 0;
thread_local uint32_t t_busy_sample = 0;    // bit i: the stream of this thread's workspace i had work in flight when cwipc_downsample was entered
thread_local bool t_busy_sampled = false;  // ... valid for the call that follows (voxel_sample_streams)
thread_local int t_ws_idle = 0;   // calls in a row that found both of the thread's streams idle while it holds a second workspace

// For the duration of a call: the thread's current stream is the one of the workspace in use.
struct StreamOfWorkspace {
    ThreadCtx &c;
    hipStream_t *other = nullptr;   // the stream that changed places with c.stream for the duration of the call
    StreamOfWorkspace(ThreadCtx &ctx, int which) : c(ctx) {
        if (which == 1 && c.stream_alt) other = &c.stream_alt;
        else if (which >= 2 && c.extra_stream(which - 2)) other = &c.stream_extra[which - 2];
        if (other) std::swap(c.stream, *other);
    }
    ~StreamOfWorkspace() {
        if (other) std::swap(c.stream, *other);
    }
};

constexpr size_t GRID_BYTES = (size_t)CELLS * RECORD_WORDS * 8;   // 20.1 MB per leaf grid
constexpr size_t HEAD_CTRL_BYTES = 256;                           // C_WORDS words, padded

bool ensure_workspace(Workspace &ws, size_t n, uint32_t leaf_cap, uint32_t nranges, hipStream_t s) {
    const int dev = current_device();
    if (ws.device != dev) {
        ws.release();
        const int lds = (int)sizeof(LdsTable);
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int lds_fast = (int)sizeof(FastTable);
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_fast_kernel<0, K1_THREADS, LTAB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_fast));
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_fast_kernel<1, K1_THREADS, LTAB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_fast));
        const int lds_pair = (int)sizeof(PairTable);
        static_assert(2 * sizeof(PairTable) <= 160 * 1024, "two workgroups of the paired accumulate kernel share a CU's LDS");
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_fast_kernel<0, PAIR_THREADS, PAIR_LTAB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_pair));
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_accumulate_fast_kernel<1, PAIR_THREADS, PAIR_LTAB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_pair));
        CW_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&partition_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PartLds)));
        ws.device = dev;
    }
    if (ws.leaf_cap < leaf_cap) {
        if (ws.head) (void)hipFree(ws.head);
        if (ws.records) (void)hipFree(ws.records);
        if (ws.bitmaps) (void)hipFree(ws.bitmaps);
        ws.head = nullptr; ws.ctrl = nullptr; ws.leaf_keys = nullptr; ws.seg_count = nullptr;
        ws.records = nullptr; ws.bitmaps = nullptr; ws.leaf_cap = 0;
        ws.head_bytes = HEAD_CTRL_BYTES + (size_t)leaf_cap * 8 + (size_t)leaf_cap * RANK_SEGS * sizeof(uint32_t) + (size_t)leaf_cap * 4 * (8 + 4);
        ws.head_bytes = (ws.head_bytes + 255) & ~(size_t)255;
        CW_HIP_TRY(hipMalloc(&ws.head, 2 * ws.head_bytes));
        ws.head_clean[0] = ws.head_clean[1] = false;
        ws.parity = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.bitmaps, (size_t)leaf_cap * BITWORDS * sizeof(uint32_t)));
        CW_HIP_TRY(hipMemsetAsync(ws.bitmaps, 0, (size_t)leaf_cap * BITWORDS * sizeof(uint32_t), s));
        CW_HIP_TRY(hipMalloc((void **)&ws.records, (size_t)leaf_cap * GRID_BYTES));
        CW_HIP_TRY(hipMemsetAsync(ws.records, 0, (size_t)leaf_cap * GRID_BYTES, s));   // once; K4 keeps it clean afterwards
        ws.leaf_cap = leaf_cap;
        g_workspace_bytes -= ws.grid_bytes;
        ws.grid_bytes = (size_t)leaf_cap * (GRID_BYTES + BITWORDS * sizeof(uint32_t)) + 2 * ws.head_bytes;
        g_workspace_bytes += ws.grid_bytes;
    }
    if (ws.list_cap < n) {
        if (ws.occupied) (void)hipFree(ws.occupied);
        if (ws.order) (void)hipFree(ws.order);
        ws.occupied = nullptr; ws.order = nullptr; ws.list_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.occupied, n * sizeof(uint32_t)));
        CW_HIP_TRY(hipMalloc((void **)&ws.order, n * sizeof(uint32_t)));
        ws.list_cap = n;
    }
    if (ws.bbox_cap < nranges) {
        if (ws.bboxes) (void)hipFree(ws.bboxes);
        ws.bboxes = nullptr; ws.bbox_cap = 0;
        CW_HIP_TRY(hipMalloc((void **)&ws.bboxes, 2 * (size_t)nranges * 6 * sizeof(float)));
        ws.bbox_cap = nranges;
    }
    if (!ws.faces) CW_HIP_TRY(hipMalloc((void **)&ws.faces, FACE_TABLE_WORDS * sizeof(uint32_t)));
    if (!ws.host_words) {
        CW_HIP_TRY(hipHostMalloc((void **)&ws.host_words, 2 * C_WORDS * sizeof(uint32_t), hipHostMallocDefault));
        memset(ws.host_words, 0, 2 * C_WORDS * sizeof(uint32_t));
    }
    return true;
}

// The first point of a cloud the host has not seen (a filter result), once per cloud: one lane writes
// it into the pinned words (one launch instead of three copy operations).
__global__ void first_point_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, float *__restrict__ host_out) {
    host_out[0] = x[0]; host_out[1] = y[0]; host_out[2] = z[0];
}

// The first FINITE point (the octree skips the others, so it is the anchor): one workgroup walks the
// cloud from the front, 1024 points at a time, until a chunk holds one.  out = x, y, z, found.
__global__ void __launch_bounds__(1024) first_finite_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z, size_t n,
                                                           float *__restrict__ host_out) {
    __shared__ unsigned long long s_first;
    for (size_t base = 0; base < n; base += 1024) {
        if (threadIdx.x == 0) s_first = ~0ull;
        __syncthreads();
        const size_t i = base + threadIdx.x;
        if (i < n && isfinite(x[i]) && isfinite(y[i]) && isfinite(z[i])) atomicMin(&s_first, (unsigned long long)i);
        __syncthreads();
        const unsigned long long f = s_first;
        __syncthreads();
        if (f != ~0ull) {
            if (threadIdx.x == 0) { host_out[0] = x[f]; host_out[1] = y[f]; host_out[2] = z[f]; host_out[3] = 1.0f; }
            return;
        }
    }
    if (threadIdx.x == 0) host_out[3] = 0.0f;
}

bool fetch_first_point(const DeviceSoA &src, ThreadCtx &c) {
    if (src.has_first) return true;
    float *h = (float *)c.host_words;
    hipLaunchKernelGGL(first_point_kernel, dim3(1), dim3(1), 0, c.stream, src.x(), src.y(), src.z(), h);
    bool ok = hipGetLastError() == hipSuccess;
    ok = c.sync() && ok;
    if (!ok) return hip_failed(hipGetLastError(), "fetch of the first point", __FILE__, __LINE__);
    src.first[0] = h[0]; src.first[1] = h[1]; src.first[2] = h[2];
    src.has_first = true;
    return true;
}

bool PendingVoxel::outcome_locked() {
    if (known) return ok;
    uint32_t hw[C_SEQ];
    const auto take = [&]() {   // true when every word carries this pass's tag
        for (int i = 0; i < C_SEQ; i++) {
            const unsigned long long w = words[i];
            if ((uint32_t)(w >> 32) != seq) return false;
            hw[i] = (uint32_t)w;
        }
        return true;
    };
    bool seen = false;
    const auto t_give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(poll_budget_us());
    for (int spin = 0;; spin++) {
        if ((uint32_t)(words[C_COUNT] >> 32) == seq && take()) { seen = true; break; }
        if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_give_up) break;
        __builtin_ia32_pause();
    }
    if (!seen) {
        if (hipStreamSynchronize(stream) != hipSuccess) (void)hipGetLastError();
        seen = take();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    known = true;
    err = seen ? hw[C_ERR] : 0x80000000u;
    m = seen ? hw[C_COUNT] : 0u;
    scatter = seen ? hw[C_SCATTER] : 0xffffffffu;
    leaves = seen ? hw[C_LEAVES] : 0u;
    ok = seen && err == 0u && m <= spec_cap;
    if (ok && !leaf_split) {
        // did the speculative passes run?  (the test they made on the device, on the same words)
        const unsigned long long cells = (unsigned long long)hw[C_DIVB] * hw[C_DIVB + 1] * hw[C_DIVB + 2];
        ok = cells <= gspec.cells_max && (cells + 31) / 32 <= gspec.words_cap;
    }
    return ok;
}

std::shared_ptr<DeviceSoA> PendingVoxel::settle() {
    std::lock_guard<std::recursive_mutex> g(lock);
    if (settled) return result;
    if (outcome_locked()) {
        if (m == 0 && !leaf_split) {
            cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "VoxelGrid filter produced empty pointcloud");   // reference src/cwipc_filters.cpp:58-62
            result = nullptr;
        } else if (m == 0) {
            result = soa_alloc(0);   // only points that do not count: no leaves, an empty cloud
        } else {
            spec_dst->npoints = m;   // the finalize kernel is filling (or has filled) the first m slots; the planes carry its `ready` event
            result = spec_dst;
        }
    } else {
        // not a pass that could be handed out early after all: once more, the waiting way (the dirty records of the
        // failed pass are cleaned by the next user of its workspace)
        int code = 0;
        result = voxel_downsample(src, cellsize, leaf_split, &code, nullptr);
    }
    spec_dst.reset();
    src.reset();
    settled = true;
    return result;
}

}  // namespace

// "Is the workspace whose turn it is still at work?" is asked of its stream -- and has to be asked BEFORE the call orders that stream
// behind the producer of its input: a cloud that came out of another filter with its last kernel still running (r4: colorize, as
// a join's result since round 2) puts a wait into the thread's first stream, which then reads as busy although the workspace has
// been idle since the frame before, and every thread of a per-tile chain took a second and a third workspace (0.3 GB each) for
// calls that could not overlap anyway.  cwipc_downsample samples the streams on entry; the call that follows uses the sample.
void voxel_sample_streams() {
    ThreadCtx &c = tctx();
    if (!c.ensure()) return;
    uint32_t bits = 0;
    for (int i = 0; i < MAX_WS && t_ws.ws[i]; i++) {
        hipStream_t s = i == 0 ? c.stream : i == 1 ? c.stream_alt : c.extra_stream(i - 2);
        if (s && hipStreamQuery(s) == hipErrorNotReady) bits |= 1u << i;
        (void)hipGetLastError();   // (hipErrorNotReady is an answer, not a failure)
    }
    t_busy_sample = bits;
    t_busy_sampled = true;
}

std::shared_ptr<DeviceSoA> voxel_downsample(const std::shared_ptr<DeviceSoA> &src_ptr, float cellsize, bool leaf_split, int *error_code,
                                            std::shared_ptr<DeferredResult> *deferred) {
    const DeviceSoA &src = *src_ptr;
    ThreadCtx &c = tctx();
    if (!c.ensure()) return nullptr;
    const size_t n = src.npoints;
    // Workspaces (and streams) per thread, taken in turn, so that a call queued right behind another does not wait for that one's
    // finalize kernel.  The second one comes into being only when it is needed: a thread whose downsample calls are separated by
    // other work (a per-tile filter chain) finds its first workspace idle every time and never pays the 80+ MB of grids for a second.
    // r4: and a third (CWIPC_WORKSPACES, 1 to 4, default 3) when the one whose turn it is still has kernels in flight: the kernel
    // trace of a stream of calls (gpurun_out/r4_trace_dump.log) shows a call's chain on its stream -- accumulate 54-65 us next to its
    // neighbour, replay 7, finalize 16-27, and the host's turn-around -- at ~110 us, i.e. two streams give a call every 55 us whatever
    // the accumulate kernel does; the chip had nothing to stream for 12 of every 110 us.
    static const int max_ws = []() { const char *e = getenv("CWIPC_WORKSPACES"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : v > MAX_WS ? MAX_WS : v; }();
    int have = 0;
    while (have < MAX_WS && t_ws.ws[have]) have++;
    const auto stream_of = [&](int i) -> hipStream_t { return i == 0 ? c.stream : i == 1 ? c.stream_alt : c.extra_stream(i - 2); };
    const bool sampled = t_busy_sampled;
    t_busy_sampled = false;
    const auto busy = [&](int i) {
        if (sampled) return ((t_busy_sample >> i) & 1u) != 0u;   // as the thread's streams were when the call came in (voxel_sample_streams)
        hipStream_t s = stream_of(i);
        const bool b = s && hipStreamQuery(s) == hipErrorNotReady;
        (void)hipGetLastError();   // (hipErrorNotReady is an answer, not a failure)
        return b;
    };
    int which = 0;
    if (have > 0) {
        which = t_ws_next % have;
        if (have < max_ws && busy(which)) {
            which = have;          // this thread's calls come faster than its workspaces turn around: one more
        } else if (have > 1 && !t_ws.ws[have - 1]->pending) {
            // ... and the last one goes again when it has not been needed for a while: sixteen calls in a row that found all streams idle (a
            // thread that has stopped calling back to back: 0.3 GB of leaf grids it no longer needs)
            bool idle = true;
            for (int i = 0; i < have && idle; i++) idle = !busy(i);
            t_ws_idle = idle ? t_ws_idle + 1 : 0;
            if (t_ws_idle >= 16) {
                delete t_ws.ws[have - 1];
                t_ws.ws[have - 1] = nullptr;
                have--;
                t_ws_idle = 0;
                which = which % have;
            }
        }
    }
    t_ws_next = which + 1;
    Workspace &ws = t_ws.get(which);
    StreamOfWorkspace on_its_stream(c, which);
    if (ws.pending) {
        // the pass before last of this thread was handed out while it ran: its report is in by now (its words are about to
        // be reused); if it did not go through, its records are still dirty
        const std::shared_ptr<PendingVoxel> p = ws.pending;
        ws.pending.reset();
        if (p->outcome()) {
            if (p->leaf_split) ws.last_m = p->m; else ws.last_m_grid = p->m;
            ws.note_leaves(p->leaves);
            // (a partitioned pass tells how scattered the cloud was as it came: in scan order again, no partition next time)
            if (p->partitioned && (size_t)p->scatter * 4 < p->steps_total) ws.incoherent = false;
        } else {
            if (p->leaf_split) ws.last_m = 0; else ws.last_m_grid = 0;
            ws.streak = 0;
            if (p->err & (ERR_FAST_PATH | ERR_CELL_RANGE)) ws.no_fast = true;
            VoxWork W{ws.leaf_keys, ws.records, ws.occupied, ws.ctrl, ws.bboxes, ws.faces, ws.bitmaps, ws.seg_count, ws.hash_keys, ws.hash_ids, ws.dump_head, ws.dump_ent};
            CW_LAUNCH("clean_by_bitmap", clean_by_bitmap_kernel, dim3(ws.leaf_cap * RANK_SEGS), dim3(RANK_THREADS), 0, c.stream, W);
            if (!c.sync()) { hip_failed(hipGetLastError(), "voxel workspace clean-up", __FILE__, __LINE__); return nullptr; }
        }
    }
    src.wait_on(c.stream);   // (the caller ordered the thread's first stream behind the input's producer; this may be the second)
    if (n >= ((size_t)1 << 31)) {
        cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: more than 2^31 points");
        return nullptr;
    }

    // (asked once per workspace and device: a runtime call on the way to the first launch is time the GPU waits)
    if (ws.cus_device != current_device() || ws.cus <= 0) {
        int cus_now = 0;
        if (hipDeviceGetAttribute(&cus_now, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || cus_now <= 0) cus_now = 256;
        ws.cus = cus_now;
        ws.cus_device = current_device();
    }
    // The persistent grid leaves one compute unit per XCD free (8 of 256 on MI355X).  In a stream of calls consecutive
    // accumulate kernels run on the thread's two streams: the next one's workgroups start on the free CUs and take over the
    // others as the previous kernel's workgroups finish, so one kernel's start-up and flush (about 10 of its 58 us, during
    // which a CU streams nothing) lie behind the other's stream: 61.3 -> 52.5 us per call at 10 M points.  It takes a free CU
    // on EVERY XCD (workgroups are dealt to the XCDs in turn: with 6 spare CUs nothing is gained, with 8 everything); a
    // single kernel is as fast on 248 CUs as on 256 (58.8 / 58.5 us).  CWIPC_SPARE_CUS=n overrides (a process that runs a
    // multi-GPU join next to its downsamples leaves a few more: the exchange's kernels need room too).
    static const int spare_knob = []() { const char *e = getenv("CWIPC_SPARE_CUS"); return e ? atoi(e) : -1; }();
    const int spare_cus = spare_knob >= 0 ? spare_knob : ws.cus / 32;
    const int cus = ws.cus - spare_cus > 8 ? ws.cus - spare_cus : ws.cus;
    // one persistent workgroup per CU; short clouds get fewer so that every wave has at least one step,
    // very large clouds get more (sequential) workgroups: the packed table needs < 65536 points per workgroup
    // Clouds with few points per voxel fill the workgroup table (2048 voxels): the previous calls of this
    // thread tell (ws.shrink) how much smaller the workgroups have to be for it to hold; the extra
    // workgroups run one after the other on the same CUs.
    if (ws.hint_cell != cellsize || ws.hint_split != leaf_split || n > 2 * ws.hint_n || 2 * n < ws.hint_n) {   // another kind of cloud or call: start over
        ws.shrink = 0;
        ws.calm = 0;
        ws.incoherent = false;
        ws.no_fast = false;
        ws.streak = 0;
    }
    ws.hint_cell = cellsize;
    ws.hint_split = leaf_split;
    ws.hint_n = n;
    size_t nwaves = ((size_t)cus * K1_WAVES) << ws.shrink;
    const size_t steps_total = (n + WAVE_STEP - 1) / WAVE_STEP;
    if (nwaves > steps_total) nwaves = ((steps_total + K1_WAVES - 1) / K1_WAVES) * K1_WAVES;
    const size_t min_waves = (n + MAX_POINTS_PER_WAVE - 1) / MAX_POINTS_PER_WAVE;
    if (nwaves < min_waves) nwaves = ((min_waves + K1_WAVES - 1) / K1_WAVES) * K1_WAVES;
    const uint32_t nblocks = (uint32_t)(nwaves / K1_WAVES);

    VoxParams P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.per_wave = (((n + nwaves - 1) / nwaves + WAVE_STEP - 1) / WAVE_STEP) * WAVE_STEP;
    P.nranges = (uint32_t)nwaves;
    P.leaf = cellsize;
    P.inv_leaf = 1.0f / cellsize;
    P.vox_unit = 1.0 / (double)P.inv_leaf;
    P.q_unit = P.vox_unit / 8388608.0;
    bool local_leaves = true;   // leaf ids resolved per workgroup at flush time; off after ERR_LOCAL_LEAVES
    P.leaf_d = (double)cellsize;
    const float octree_cellsize = (8 * 8) * cellsize;   // reference src/cwipc_filters.cpp:113-114
    P.res = (double)octree_cellsize;
    P.leaf_split = leaf_split ? 1 : 0;
#ifdef CWIPC_DEBUG_KNOBS
    static const uint32_t ablate_knob = []() { const char *e = getenv("CWIPC_VOXEL_ABLATE"); return e ? (uint32_t)atoi(e) : 0u; }();   // read once
    if (ablate_knob) cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_downsample", "CWIPC_VOXEL_ABLATE is set: results are WRONG (timing experiments only)");
    P.ablate = ablate_knob;
#endif

    // ---- anchor and face thresholds (host, f64) ----
    uint32_t faces_host[FACE_TABLE_WORDS];
    double faces_key_mn0[3] = {0, 0, 0};
    memset(faces_host, 0, sizeof(faces_host));
    if (leaf_split) {
        if (!fetch_first_point(src, c)) return nullptr;
        float anchor[3] = {src.first[0], src.first[1], src.first[2]};
        if (!(std::isfinite(anchor[0]) && std::isfinite(anchor[1]) && std::isfinite(anchor[2]))) {
            // the octree skips non-finite points (addPointsFromInputCloud): its first point is the first finite one
            float *h = (float *)c.host_words;
            hipLaunchKernelGGL(first_finite_kernel, dim3(1), dim3(1024), 0, c.stream, src.x(), src.y(), src.z(), n, h);
            bool ok = hipGetLastError() == hipSuccess;
            ok = c.sync() && ok;
            if (!ok) { hip_failed(hipGetLastError(), "search for the first finite point", __FILE__, __LINE__); return nullptr; }
            if (h[3] == 0.0f) return soa_alloc(0);   // no finite point at all: no leaves, an empty cloud
            anchor[0] = h[0]; anchor[1] = h[1]; anchor[2] = h[2];
        }
        const double pp[3] = {(double)anchor[0], (double)anchor[1], (double)anchor[2]};
        first_box(pp, P.res, P.mn0, P.mx0, P.depth0);
        for (int a = 0; a < 3; a++) {
            P.ib[a] = (int)floor(P.mn0[a] / P.leaf_d);
            // the first point sits in leaf 1 of its box: cover faces 1 - FACE_BACK .. 1 - FACE_BACK + FACES - 1
            P.face_base[a] = 1 - FACE_BACK;
        }
        // the thresholds depend on the anchor's box and the resolution only: a stream of frames of one scene
        // (and every repeat of a call) finds them in the workspace, host copy and device copy
        const bool cached = ws.faces_valid && ws.faces_res == P.res && ws.faces_mn0[0] == P.mn0[0] && ws.faces_mn0[1] == P.mn0[1] && ws.faces_mn0[2] == P.mn0[2];
        if (cached) {
            memcpy(faces_host, ws.faces_host, sizeof(faces_host));
        } else {
            for (int a = 0; a < 3; a++) {
                for (int i = 0; i < FACES; i++) {
                    const float T = leaf_threshold(P.mn0[a], P.res, P.face_base[a] + i);
                    // the voxel the face cuts (the fp32 product and floor of the kernels), and where the voxel above it begins
                    const float g = floorf(T * P.inv_leaf);
                    const bool sane = std::isfinite(T) && fabsf(g) < 1.0e9f;
                    const int tf = sane ? (int)g : (T > 0 ? INT32_MAX : INT32_MIN);
                    const float Tv = sane ? voxel_upper_bound(P.inv_leaf, tf) : T;
                    memcpy(&faces_host[FT_T + a * FACES + i], &T, 4);
                    memcpy(&faces_host[FT_TV + a * FACES + i], &Tv, 4);
                    memcpy(&faces_host[FT_TF + a * FACES + i], &tf, 4);
                }
            }
        }
        faces_key_mn0[0] = P.mn0[0]; faces_key_mn0[1] = P.mn0[1]; faces_key_mn0[2] = P.mn0[2];
    } else {
        for (int a = 0; a < 3; a++) P.ib[a] = 2;   // bricks of 64 voxels aligned to the voxel lattice
    }

    // Points in no spatial order (learned from the calls before: the workgroup tables overflowed whatever the workgroup size):
    // move them into coarse spatial buckets first and accumulate the moved copy (voxel_partition.inc).
    static const bool partition_off = []() { const char *e = getenv("CWIPC_VOXEL_PARTITION"); return e && atoi(e) == 0; }();   // test knob
    const bool partition = ws.incoherent && !partition_off && n >= 65536;
    const size_t part_stride = (n + 1023) & ~(size_t)1023;
    const size_t part_nseg = (nblocks + PART_SEG_ROWS - 1) / PART_SEG_ROWS;
    if (partition) {
        if (ws.part_stride < part_stride || ws.part_rows < nblocks) {
            ws.drop_partition_buffers();
            if (hipMalloc((void **)&ws.part, 16 * part_stride) != hipSuccess ||
                hipMalloc((void **)&ws.part_hist, ((size_t)nblocks + part_nseg + 1) * PART_BUCKETS * sizeof(uint32_t)) != hipSuccess) {
                (void)hipGetLastError();
                ws.drop_partition_buffers();
                hip_failed(hipErrorOutOfMemory, "voxel partition buffers", __FILE__, __LINE__);
                return nullptr;
            }
            ws.part_stride = part_stride;
            ws.part_rows = nblocks;
            g_workspace_bytes += 16 * part_stride;
        }
    } else if (ws.part) {
        ws.drop_partition_buffers();   // (nothing of this workspace's stream is in flight: its last pass has reported above or long ago)
    }

    if (ws.shrink_to && ws.shrink_to < ws.leaf_cap) {
        // (this thread's last pass on the workspace may have its finalize kernel in flight still)
        if (!c.sync()) return nullptr;
        if (ws.head) (void)hipFree(ws.head);
        if (ws.records) (void)hipFree(ws.records);
        if (ws.bitmaps) (void)hipFree(ws.bitmaps);
        ws.head = nullptr; ws.ctrl = nullptr; ws.leaf_keys = nullptr; ws.seg_count = nullptr; ws.records = nullptr; ws.bitmaps = nullptr;
        g_workspace_bytes -= ws.grid_bytes;
        ws.grid_bytes = 0;
        ws.leaf_cap = 0;
    }
    // 4 grids = 80 MB to begin with (a camera tile at 1 cm has 2 to 4 leaves, a person-sized cloud 12 to 16); x4 when a cloud has more
    uint32_t leaf_cap = ws.leaf_cap ? ws.leaf_cap : (ws.shrink_to ? ws.shrink_to : 4);
    if (ws.shrink_to) ws.roomy = 0;
    ws.shrink_to = 0;
    int mode = leaf_split ? 1 : 0;
    bool used_fast = false;
    for (int attempt = 0; attempt < 10; attempt++) {
        if (!ensure_workspace(ws, n, leaf_cap, (uint32_t)nwaves, c.stream)) return nullptr;
        P.leaf_mask = 4 * ws.leaf_cap - 1;
        P.list_cap = (uint32_t)(ws.list_cap > 0xffffffffu ? 0xffffffffu : ws.list_cap);
        const uint32_t seq = ++ws.seq ? ws.seq : ++ws.seq;
        for (int i = 0; i < C_SEQ; i++) ws.host_words[2 * i + 1] = 0;   // the tags of the words the replay kernel will publish
        // control words, leaf table, slice counts: this pass's block (zeroed by the previous pass's replay kernel)
        const int blk = ws.parity;
        char *head = (char *)ws.head + (size_t)blk * ws.head_bytes, *next_head = (char *)ws.head + (size_t)(1 - blk) * ws.head_bytes;
        ws.ctrl = (uint32_t *)head;
        ws.leaf_keys = (unsigned long long *)(head + HEAD_CTRL_BYTES);
        ws.seg_count = (uint32_t *)(head + HEAD_CTRL_BYTES + (size_t)ws.leaf_cap * 8);
        ws.hash_keys = (unsigned long long *)(head + HEAD_CTRL_BYTES + (size_t)ws.leaf_cap * 8 + (size_t)ws.leaf_cap * RANK_SEGS * sizeof(uint32_t));
        ws.hash_ids = (uint32_t *)((char *)ws.hash_keys + (size_t)ws.leaf_cap * 4 * 8);
        VoxWork W{ws.leaf_keys, ws.records, ws.occupied, ws.ctrl, ws.bboxes, ws.faces, ws.bitmaps, ws.seg_count, ws.hash_keys, ws.hash_ids, ws.dump_head, ws.dump_ent};
        bool ok = true;
        if (!ws.head_clean[blk]) ok = hipMemsetAsync(head, 0, ws.head_bytes, c.stream) == hipSuccess;
        ws.head_clean[blk] = false;
        if (ok && mode == 1 && !(ws.faces_valid && memcmp(ws.faces_host, faces_host, sizeof(faces_host)) == 0)) {
            uint32_t *stage = (uint32_t *)c.staging(sizeof(faces_host));
            ok = stage != nullptr;
            if (ok) {
                memcpy(stage, faces_host, sizeof(faces_host));
                ok = hipMemcpyAsync(ws.faces, stage, sizeof(faces_host), hipMemcpyHostToDevice, c.stream) == hipSuccess;
                memcpy(ws.faces_host, faces_host, sizeof(faces_host));
                ws.faces_valid = ok;
                ws.faces_res = P.res;
                ws.faces_mn0[0] = faces_key_mn0[0]; ws.faces_mn0[1] = faces_key_mn0[1]; ws.faces_mn0[2] = faces_key_mn0[2];
            }
        }
        if (!ok) { hip_failed(hipGetLastError(), "voxel workspace setup", __FILE__, __LINE__); return nullptr; }

        K1Params K;
        memset(&K, 0, sizeof(K));
        K.n = (uint32_t)n; K.per_wave = (uint32_t)P.per_wave;
        K.inv_leaf = P.inv_leaf;
        K.ib0 = P.ib[0]; K.ib1 = P.ib[1]; K.ib2 = P.ib[2];
        K.fb0 = P.face_base[0]; K.fb1 = P.face_base[1]; K.fb2 = P.face_base[2];
        K.leaf_mask = P.leaf_mask; K.list_cap = P.list_cap; K.ablate = P.ablate;
        K.local_leaves = local_leaves ? 1u : 0u;
        K.want_list = leaf_split ? 0u : 1u;
        K.mn0[0] = P.mn0[0]; K.mn0[1] = P.mn0[1]; K.mn0[2] = P.mn0[2];
        K.res = P.res;
        // The fast variant takes coherent clouds that fit its workgroup table and key; it says so (ERR_FAST_PATH) when a
        // cloud does not, and the pass is run again with the general variant (which is remembered for the clouds to come).
        static const bool fast_off = []() { const char *e = getenv("CWIPC_VOXEL_GENERAL"); return e && atoi(e) != 0; }();   // test knob: general variant only
        const bool fast = mode != 2 && !ws.no_fast && ws.shrink == 0 && !fast_off && !partition;
        // the planes the accumulate kernel reads, and where it leaves its boxes
        const float *kx = src.x(), *ky = src.y(), *kz = src.z();
        const uint32_t *kw = src.rgbt();
        VoxWork Wk = W;
        if (partition) {
            float *px = ws.part, *py = px + ws.part_stride, *pz = py + ws.part_stride;
            uint32_t *pw = (uint32_t *)(pz + ws.part_stride);
            const uint32_t padded = (uint32_t)((n + WAVE_STEP - 1) / WAVE_STEP * WAVE_STEP);
            // the table: rows [part_rows][buckets], then the segments' sums [nseg][buckets], then the buckets' starts (every word is
            // written by the kernels that follow: nothing to clear)
            uint32_t *rows = ws.part_hist, *seg = rows + ws.part_rows * PART_BUCKETS, *start = seg + part_nseg * PART_BUCKETS;
            const uint32_t per_wg = (uint32_t)(P.per_wave * K1_WAVES);
            CW_LAUNCH("partition_count", partition_count_kernel, dim3(nblocks), dim3(K1_THREADS), 0, c.stream, (uint32_t)n, per_wg, P.inv_leaf,
                      src.x(), src.y(), src.z(), ws.bboxes, rows, ws.ctrl);
            CW_LAUNCH("partition_scan", partition_segsum_kernel, dim3((unsigned)part_nseg, PART_BUCKETS / 256), dim3(256), 0, c.stream, rows, nblocks, seg);
            CW_LAUNCH("partition_scan", partition_starts_kernel, dim3(1), dim3(K1_THREADS), 0, c.stream, seg, (uint32_t)part_nseg, start);
            CW_LAUNCH("partition_scatter", partition_scatter_kernel, dim3(nblocks), dim3(K1_THREADS), sizeof(PartLds), c.stream,
                      (uint32_t)n, per_wg, padded, P.inv_leaf, src.x(), src.y(), src.z(), src.rgbt(), px, py, pz, pw, rows, seg, start);
            kx = px; ky = py; kz = pz; kw = pw;
            Wk.bboxes = ws.bboxes + (size_t)ws.bbox_cap * 6;   // the boxes of the moved points: nobody reads them
        }
        used_fast = fast;
        uint32_t fast_blocks = 0, fast_per_wg = 0, range_base_q = 0, range_inc_q = 0;
        if (fast) {
            FastParams F;
            memset(&F, 0, sizeof(F));
            // The workgroups' ranges: the cloud's steps dealt evenly over the CUs the grid may use (a workgroup's waves share its
            // range step by step, so a range need not be a multiple of sixteen steps): a 300 k-point cloud gets 235 workgroups
            // of 5 steps, five busy waves each, instead of 74 workgroups whose sixteen waves queue up on four SIMDs.
            // CWIPC_K1_PAIR=1 (experiments): two workgroups of 8 waves and half the range per CU instead of one of 16 (voxel_k1_fast.inc)
            static const int pair_knob = []() { const char *e = getenv("CWIPC_K1_PAIR"); return e ? atoi(e) : 0; }();
            // CWIPC_K1_PAIR=2: the 8-wave workgroups with FULL ranges, one per CU and kernel -- in a stream of calls a CU then holds a workgroup
            // of each of two consecutive kernels, and one streams while the other sets up or flushes
            const bool pair = pair_knob != 0;
            const size_t slots = (size_t)cus * (pair_knob == 1 ? 2 : 1);
            const size_t wg_steps = std::min<size_t>(std::max<size_t>((steps_total + slots - 1) / slots, 1), MAX_POINTS_PER_WAVE * K1_WAVES / WAVE_STEP);
            fast_blocks = (uint32_t)((steps_total + wg_steps - 1) / wg_steps);
            fast_per_wg = (uint32_t)(wg_steps * WAVE_STEP);
            // r4: the ranges' lengths grow linearly with the workgroup's number, from (1 - p %) to (1 + p %) of the mean, so that the
            // workgroups reach their flush one after the other: the memory side takes ~10 us for all the flushes' atomics (112 k
            // entries x seven), during which nothing streams when 248 workgroups arrive within two microseconds.  The longest range
            // sets the kernel's end now (a workgroup's streaming time goes with its length: the vector port, not the memory, bounds it),
            // so only part of those 10 us comes back: 53.9-54.4 -> 51.2-51.9 us alone at p = 20-25, 15 does nothing, 30-40 lose it again;
            // a call in a stream is what it was (47-48 us: there the next kernel's workgroups fill the gaps anyway), call-then-count
            // 66.0 -> 64.2 (profiles/r04_k1_stagger.txt).  CWIPC_K1_STAGGER=p overrides (0: equal ranges, rounds 1-3).  Only for
            // ranges of 24 steps or more (clouds from 1.5 M points): a short range is mostly set-up and flush.
            static const int stagger_knob = []() { const char *e = getenv("CWIPC_K1_STAGGER"); return e ? atoi(e) : 25; }();
            if (stagger_knob > 0 && stagger_knob < 60 && fast_blocks >= 64 && !pair && steps_total >= (size_t)24 * fast_blocks && n < ((size_t)1 << 31)) {
                const double mean_q = (double)steps_total * 1024.0 / (double)fast_blocks, s_frac = stagger_knob / 100.0;
                const size_t max_steps = MAX_POINTS_PER_WAVE * K1_WAVES / WAVE_STEP;
                uint32_t inc = (uint32_t)ceil(2.0 * s_frac * mean_q / (double)(fast_blocks - 1));
                uint32_t base = (uint32_t)ceil(mean_q * (1.0 - s_frac));
                while (range_first_step(fast_blocks, base, inc) < steps_total) base++;
                const size_t longest = (size_t)((base + (unsigned long long)(fast_blocks - 1) * inc + 2047) >> 10);
                if (base >= 1024 && longest <= max_steps) {
                    range_base_q = base; range_inc_q = inc;
                    fast_per_wg = (uint32_t)(longest * WAVE_STEP);
                }
            }
            F.n = K.n; F.per_wg = fast_per_wg; F.inv_leaf = K.inv_leaf;
            F.range_base_q = range_base_q; F.range_inc_q = range_inc_q;
            static const bool stagger_rev = []() { const char *e = getenv("CWIPC_K1_STAGGER_REV"); return e && atoi(e) != 0; }();
            F.range_reverse = range_base_q != 0u && stagger_rev ? 1u : 0u;
            F.ib0 = K.ib0; F.ib1 = K.ib1; F.ib2 = K.ib2;
            F.fb0 = K.fb0; F.fb1 = K.fb1; F.fb2 = K.fb2;
            F.leaf_mask = K.leaf_mask; F.list_cap = K.list_cap; F.want_list = K.want_list;
            // r4: the accumulate kernel may leave its tables' entries in the workspace, for a kernel of small workgroups right behind it to
            // take to the records (voxel_k1_fast.inc, fast_dump / voxel_merge_kernel): the accumulate kernel alone 53.4 -> 48.1 us, the merge
            // kernel 20 us alone -- a loss for a call that is waited for (64.6 -> 78.7 us with count()), possibly a gain in a stream of calls,
            // where the merge workgroups (one wave per SIMD, 64 registers, 19 KB of LDS) run beside the NEXT call's accumulate kernel.
            // CWIPC_K1_DUMP: 0 never, 1 always, 2 when this call is going to return with its kernels in flight (a stream); default 0
            // until the stream figure says otherwise (profiles/r04_k1_dump_merge.txt)
            static const int dump_mode = []() { const char *e = getenv("CWIPC_K1_DUMP"); return e ? atoi(e) : 0; }();
            static const bool defer_allowed = []() { const char *e = getenv("CWIPC_DEFER"); return !e || atoi(e) != 0; }();
            const bool will_defer = deferred && defer_allowed && attempt == 0 && ws.streak >= 2 && !profiling_enabled() &&
                                    (leaf_split ? ws.last_m > 0 : (ws.last_m_grid > 0 && ws.gwords_cap > 0));
            const bool dump_knob = dump_mode == 1 || (dump_mode == 2 && will_defer);
            const size_t table_entries = pair ? (size_t)PAIR_LTAB : (size_t)LTAB;
            bool dump = dump_knob;
            if (dump && (ws.dump_blocks < fast_blocks || ws.dump_entries != table_entries)) {
                ws.drop_dump_buffers();
                const size_t blocks_cap = std::max<size_t>(fast_blocks, 256);
                if (hipMalloc((void **)&ws.dump_head, blocks_cap * sizeof(DumpHead)) != hipSuccess ||
                    hipMalloc((void **)&ws.dump_ent, blocks_cap * table_entries * DUMP_ENTRY_WORDS * 4) != hipSuccess) {
                    (void)hipGetLastError();
                    if (ws.dump_head) (void)hipFree(ws.dump_head);
                    ws.dump_head = ws.dump_ent = nullptr;
                    dump = false;   // no room for the tables: this pass updates the records from the accumulate kernel
                } else {
                    ws.dump_blocks = blocks_cap; ws.dump_entries = table_entries;
                    g_workspace_bytes += blocks_cap * table_entries * DUMP_ENTRY_WORDS * 4;
                }
            }
            W.dump_head = ws.dump_head; W.dump_ent = ws.dump_ent;
            F.dump = dump ? 1u : 0u;
#ifdef CWIPC_DEBUG_KNOBS
            static const uint32_t fast_dbg = []() { const char *e = getenv("CWIPC_FAST_DBG"); return e ? (uint32_t)atoi(e) : 0u; }();
            if (fast_dbg) cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_downsample", "CWIPC_FAST_DBG is set: results are WRONG (timing experiments only)");
            F.dbg = fast_dbg;
#endif
            // (a workgroup that leaves its table needs the LDS up to the list of entries in use only: 105 instead of 151 KB)
            const size_t lds_pair = dump ? PairTable::DUMP_LDS_BYTES : sizeof(PairTable), lds_one = dump ? FastTable::DUMP_LDS_BYTES : sizeof(FastTable);
            if (pair && mode == 0) {
                CW_LAUNCH("voxel_accumulate", (voxel_accumulate_fast_kernel<0, PAIR_THREADS, PAIR_LTAB>), dim3(fast_blocks), dim3(PAIR_THREADS), lds_pair, c.stream, F,
                          src.x(), src.y(), src.z(), src.rgbt(), W);
            } else if (pair) {
                CW_LAUNCH("voxel_accumulate", (voxel_accumulate_fast_kernel<1, PAIR_THREADS, PAIR_LTAB>), dim3(fast_blocks), dim3(PAIR_THREADS), lds_pair, c.stream, F,
                          src.x(), src.y(), src.z(), src.rgbt(), W);
            } else if (mode == 0) {
                CW_LAUNCH("voxel_accumulate", (voxel_accumulate_fast_kernel<0, K1_THREADS, LTAB>), dim3(fast_blocks), dim3(K1_THREADS), lds_one, c.stream, F, src.x(),
                          src.y(), src.z(), src.rgbt(), W);
            } else {
                CW_LAUNCH("voxel_accumulate", (voxel_accumulate_fast_kernel<1, K1_THREADS, LTAB>), dim3(fast_blocks), dim3(K1_THREADS), lds_one, c.stream, F, src.x(),
                          src.y(), src.z(), src.rgbt(), W);
            }
            if (dump && mode == 0) {
                CW_LAUNCH("voxel_merge", voxel_merge_kernel<0>, dim3(fast_blocks), dim3(MERGE_THREADS), 0, c.stream, F, (uint32_t)table_entries, W);
            } else if (dump) {
                CW_LAUNCH("voxel_merge", voxel_merge_kernel<1>, dim3(fast_blocks), dim3(MERGE_THREADS), 0, c.stream, F, (uint32_t)table_entries, W);
            }
        } else if (mode == 0) {
            CW_LAUNCH("voxel_accumulate_general", voxel_accumulate_kernel<0>, dim3(nblocks), dim3(K1_THREADS), sizeof(LdsTable), c.stream, K, kx, ky, kz, kw, Wk);
        } else if (mode == 1) {
            CW_LAUNCH("voxel_accumulate_general", voxel_accumulate_kernel<1>, dim3(nblocks), dim3(K1_THREADS), sizeof(LdsTable), c.stream, K, kx, ky, kz, kw, Wk);
        } else {
            CW_LAUNCH("voxel_accumulate_exact", voxel_accumulate_kernel<2>, dim3(nblocks), dim3(K1_THREADS), sizeof(LdsTable), c.stream, K, kx, ky, kz, kw, Wk);
        }
        // (the fast variant and the partition pass leave one box per workgroup range, the general variant one per wave range)
        VoxParams Pr = P;
        if (fast) {
            Pr.per_wave = fast_per_wg;
            Pr.nranges = fast_blocks;
            Pr.range_base_q = range_base_q; Pr.range_inc_q = range_inc_q;
        } else if (partition) {   // the counting kernel's boxes: one per workgroup range of the cloud as it came
            Pr.per_wave = P.per_wave * K1_WAVES;
            Pr.nranges = nblocks;
        }
        CW_LAUNCH("octree_replay", octree_replay_kernel, dim3(1), dim3(1024), 0, c.stream, Pr, src.x(), src.y(), src.z(), ws.bboxes, ws.ctrl,
                  ws.leaf_keys, ws.leaf_cap, (uint32_t *)next_head, (uint32_t)(ws.head_bytes / 4), ws.host_words, seq);
        const hipError_t launch_err = hipGetLastError();
        ok = launch_err == hipSuccess;
        ws.head_clean[1 - blk] = ok;
        ws.parity = 1 - blk;
        // Octree variant: the finalize pass goes out right behind the replay kernel, before the host knows the
        // count, into a result sized from the previous call of this thread (+25 %); it checks count and error
        // word on the device and leaves everything untouched if they do not fit.
        std::shared_ptr<DeviceSoA> spec_dst;
        uint32_t spec_cap = 0;
        if (ok && leaf_split && ws.last_m > 0 && !profiling_enabled()) {
            spec_cap = ws.last_m + ws.last_m / 4 + 1024;
            spec_dst = soa_alloc(spec_cap);
            if (spec_dst) {
                hipLaunchKernelGGL(rank_emit_kernel, dim3(ws.leaf_cap * RANK_SEGS), dim3(RANK_THREADS), 0, c.stream, P, W, ws.leaf_cap, spec_cap, 1, ws.order,
                                   spec_dst->x(), spec_dst->y(), spec_dst->z(), spec_dst->rgbt());
                // its `ready` event now, while the kernels run, not after the wait below: what the host does
                // between the end of that wait and the next call's first launch is time the GPU stands still
                spec_dst->mark_pending(c.stream);
            }
        }
        // Plain grid: the same, five small kernels instead of one (mark, block counts, block scan, emit, unmark), with
        // room for last call's count (+25 %) and the index bitmap as it stands; each of them checks on the device
        // that the pass succeeded and fits, and does nothing otherwise.
        GridSpec gspec{0, 0u, 0u, 0ull};
        unsigned long long bitmap_max = GRID_BITMAP_MAX_CELLS;
        if (const char *e = getenv("CWIPC_GRID_BITMAP_MAX")) bitmap_max = strtoull(e, nullptr, 10);   // test knob: force the sort path
        if (ok && !leaf_split && ws.last_m_grid > 0 && ws.gwords_cap > 0 && !profiling_enabled()) {
            spec_cap = ws.last_m_grid + ws.last_m_grid / 4 + 1024;
            if (spec_cap > P.list_cap) spec_cap = P.list_cap;
            spec_dst = soa_alloc(spec_cap);
            if (spec_dst) {
                gspec = GridSpec{1, spec_cap, (uint32_t)std::min<size_t>(ws.gwords_cap, 0xffffffffu), bitmap_max};
                const unsigned sgrid = (spec_cap + 255) / 256, sblk = (unsigned)((gspec.words_cap + GB_WORDS_PER_BLOCK - 1) / GB_WORDS_PER_BLOCK);
                hipLaunchKernelGGL(grid_mark_kernel, dim3(sgrid), dim3(256), 0, c.stream, P, W, 0u, gspec, ws.gbits, ws.order);
                hipLaunchKernelGGL(grid_block_kernel, dim3(sblk), dim3(256), 0, c.stream, W, gspec, ws.gbits, 0u, ws.gprefix, ws.gblock);
                hipLaunchKernelGGL(grid_blockscan_kernel, dim3(1), dim3(1024), 0, c.stream, W, gspec, ws.gblock, 0u);
                hipLaunchKernelGGL(grid_emit_kernel, dim3(sgrid), dim3(256), 0, c.stream, P, W, 0u, gspec, ws.order, ws.gbits, ws.gprefix, ws.gblock,
                                   spec_dst->x(), spec_dst->y(), spec_dst->z(), spec_dst->rgbt());
                hipLaunchKernelGGL(grid_unmark_kernel, dim3(sgrid), dim3(256), 0, c.stream, W, 0u, gspec, ws.order, ws.gbits);
                spec_dst->mark_pending(c.stream);
            }
        }
        // A stream of frames (the two passes before went through at the first attempt): the call returns here, with its
        // kernels in flight (r3: the plain grid's seven too).  The host does not wait for the count any more, so the next call's accumulate kernel
        // is queued while this pass's replay and finalize kernels still run (on the thread's other stream), and the
        // 15 us that lay between two accumulate kernels (replay kernel + the host's return, next entry and launch) are gone.
        static const bool defer_on = []() { const char *e = getenv("CWIPC_DEFER"); return !e || atoi(e) != 0; }();
        if (deferred && defer_on && attempt == 0 && ok && spec_dst && (leaf_split || gspec.on) && ws.streak >= 2) {
            auto p = std::make_shared<PendingVoxel>();
            p->src = src_ptr;
            p->spec_dst = spec_dst;
            p->words = reinterpret_cast<volatile unsigned long long *>(ws.host_words);
            p->stream = c.stream;
            p->seq = seq;
            p->spec_cap = spec_cap;
            p->cellsize = cellsize;
            p->leaf_split = leaf_split;
            p->gspec = gspec;
            p->partitioned = partition;
            p->steps_total = (uint32_t)steps_total;
            src.note_reader(c.stream);   // the input's planes are not recycled before the accumulate kernel is done with them
            ws.pending = p;
            *deferred = p;
            return nullptr;
        }
        // wait for the replay kernel's sequence number in pinned memory (a few hundred microseconds of
        // polling at most, then the ordinary stream wait, which also reports launch failures)
        uint32_t hw[C_SEQ];   // the published control words
        {
            volatile unsigned long long *words = reinterpret_cast<volatile unsigned long long *>(ws.host_words);
            const auto take = [&]() {   // true when every word carries this pass's tag
                for (int i = 0; i < C_SEQ; i++) {
                    const unsigned long long w = words[i];
                    if ((uint32_t)(w >> 32) != seq) return false;
                    hw[i] = (uint32_t)w;
                }
                return true;
            };
            bool seen = false;
            if (ok && !profiling_enabled()) {
                const auto t_give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(poll_budget_us());
                for (int spin = 0;; spin++) {
                    if ((uint32_t)(words[C_COUNT] >> 32) == seq && take()) { seen = true; break; }
                    if ((spin & 255) == 255 && std::chrono::steady_clock::now() > t_give_up) break;
                    __builtin_ia32_pause();
                }
            }
            if (!seen) {
                ok = c.sync() && ok;
                if (ok && !take()) { ok = false; cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: the replay kernel did not report"); }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!ok) { hip_failed(launch_err != hipSuccess ? launch_err : hipGetLastError(), "voxel_accumulate", __FILE__, __LINE__); return nullptr; }

        uint32_t err = hw[C_ERR];
#ifdef CWIPC_DEBUG_KNOBS
        if (used_fast && getenv("CWIPC_FAST_STAMPS")) {
            unsigned long long st[2][16];
            (void)hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fast_stamps), sizeof(st)) == hipSuccess) {
                for (int w = 0; w < 2; w++) {
                    std::string line = "debug: workgroup " + std::string(w ? "mid" : "0") + " phases (us since its start): ";
                    const char *names[9] = {"start", "table ready", "steps done", "boxes out", "entries compacted", "keys decoded", "leaf ids", "records updated", "end"};
                    for (int i = 1; i < 9; i++) line += std::string(names[i]) + " " + std::to_string((double)(st[w][i] - st[w][0]) * 0.01).substr(0, 5) + "; ";
                    cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_downsample", line);
                }
                static unsigned long long wt[1024][4];
                if (hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_wg_times), sizeof(wt)) == hipSuccess && fast_blocks > 0 && fast_blocks <= 1024) {
                    unsigned long long t0 = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0, d_min = ~0ull, d_max = 0, a_max = 0;
                    for (uint32_t b = 0; b < fast_blocks; b++) {
                        t0 = std::min(t0, wt[b][0]); s_max = std::max(s_max, wt[b][0]);
                        e_min = std::min(e_min, wt[b][3]); e_max = std::max(e_max, wt[b][3]);
                        d_min = std::min(d_min, wt[b][3] - wt[b][0]); d_max = std::max(d_max, wt[b][3] - wt[b][0]);
                        a_max = std::max(a_max, wt[b][2]);
                    }
                    char buf[256];
                    snprintf(buf, sizeof(buf), "debug: %u workgroups: last start %.2f us after the first; records updated between %.2f and %.2f us; all waves done by %.2f; a workgroup lives %.2f to %.2f us",
                             fast_blocks, (double)(s_max - t0) * 0.01, (double)(e_min - t0) * 0.01, (double)(e_max - t0) * 0.01, (double)(a_max - t0) * 0.01, (double)d_min * 0.01, (double)d_max * 0.01);
                    cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_downsample", buf);
                    if (const char *path = getenv("CWIPC_FAST_STAMPS_FILE")) {   // every workgroup's row, for a look at the spread
                        if (FILE *f = fopen(path, "w")) {
                            fprintf(f, "# workgroup start stream_done(thread 0's wave) all_waves_done end   (us since the first start)\n");
                            for (uint32_t b = 0; b < fast_blocks; b++)
                                fprintf(f, "%u %.2f %.2f %.2f %.2f\n", b, (double)(wt[b][0] - t0) * 0.01, (double)(wt[b][1] - t0) * 0.01, (double)(wt[b][2] - t0) * 0.01, (double)(wt[b][3] - t0) * 0.01);
                            fclose(f);
                        }
                    }
                }
                unsigned long long wd[K1_WAVES];
                if (hipMemcpyFromSymbol(wd, HIP_SYMBOL(g_fast_wave_done), sizeof(wd)) == hipSuccess) {
                    std::string line = "debug: workgroup 0, waves done with their steps at (us):";
                    for (int w = 0; w < K1_WAVES; w++) line += " " + std::to_string((double)(wd[w] - st[0][0]) * 0.01).substr(0, 5);
                    cwipc_log(CWIPC_LOG_LEVEL_WARNING, "cwipc_downsample", line);
                }
            }
        }
#endif
        {
            // adapt the workgroup size for the next call
            const uint32_t fallbacks = hw[C_FALLBACK], maxload = hw[C_MAXLOAD];
            if (partition) {
                // how scattered the cloud was as it came: in scan order again -> no partition next time
                if ((size_t)hw[C_SCATTER] * 4 < steps_total) ws.incoherent = false;
            } else if (!used_fast && !err && n >= 65536 && 2 * ((size_t)hw[C_FLUSHED] + fallbacks) > n && n > 4 * (size_t)hw[C_COUNT]) {
                // Voxels hold several points each, and yet most points went to the global records on their own (a cloud in scan
                // order: one update per ~90 points at the 10 M configuration): the workgroups' ranges are all over the place.
                // The partition pass takes such clouds from the next call on.
                ws.incoherent = true;
                ws.shrink = 0;
                ws.calm = 0;
            } else if (ws.incoherent) {
                // nothing to adapt: this kind of cloud defeats the table whatever its size
            } else if (ws.shrink >= 2 && (size_t)fallbacks * 2 > n) {
                ws.incoherent = true;   // smaller workgroups did not help: points in no order at all
                ws.shrink = 0;
                ws.calm = 0;
            } else if ((size_t)fallbacks * 64 > n && ws.shrink < 6) {
                ws.shrink++;
                ws.calm = 0;
            } else if (ws.shrink > 0 && fallbacks == 0 && maxload * 3 < (uint32_t)LTAB) {
                if (++ws.calm >= 4) { ws.shrink--; ws.calm = 0; }
            } else {
                ws.calm = 0;
            }
        }
        const uint32_t m = hw[C_COUNT] < P.list_cap ? hw[C_COUNT] : P.list_cap;
        std::shared_ptr<DeviceSoA> dst;
        unsigned long long *keys_in = nullptr, *keys_out = nullptr;
        uint32_t *vals_in = nullptr, *vals_out = nullptr;
        void *sort_tmp = nullptr;
        const unsigned mgrid = (m + 255) / 256;

        bool ranked = false;
        bool grid_ranked = false;
        if (leaf_split) ws.last_m = err ? 0 : m;
        else ws.last_m_grid = err ? 0 : m;
        if (!err && m && !leaf_split && spec_dst && gspec.on) {
            // did the speculative passes run?  (the test they made on the device, on the same words)
            const unsigned long long cells = (unsigned long long)hw[C_DIVB] * hw[C_DIVB + 1] * hw[C_DIVB + 2];
            if (hw[C_COUNT] <= gspec.m_cap && cells <= gspec.cells_max && (cells + 31) / 32 <= gspec.words_cap) {
                dst = spec_dst;
                dst->npoints = m;
                grid_ranked = true;
            }
        }
        if (!err && m && leaf_split && spec_dst && m <= spec_cap) {
            // the speculative finalize pass is doing the work: the result uses the first m slots of its planes
            dst = spec_dst;
            dst->npoints = m;
            ranked = true;
        }
        spec_dst.reset();
        if (!ranked && !err && m && leaf_split) {
            // octree path: rank the occupied cells through the bitmaps, emit, clean -- no sort.  The
            // replay kernel has already checked everything this pass could trip over, so the call
            // returns with it in flight: the result carries a `ready` event, later work of this thread
            // (including the next call's use of the workspace) is ordered behind it on the stream.
            dst = soa_alloc(m);
            if (!dst) {
                err |= 0x80000000u;
            } else {
                CW_LAUNCH("rank_emit", rank_emit_kernel, dim3(ws.leaf_cap * RANK_SEGS), dim3(RANK_THREADS), 0, c.stream, P, W, ws.leaf_cap, m, 0, ws.order,
                          dst->x(), dst->y(), dst->z(), dst->rgbt());
                dst->mark_pending(c.stream);
                ranked = true;
            }
        }
        if (!err && m && !leaf_split && !grid_ranked) {
            // plain grid, the usual case: output order from a bitmap over the VoxelGrid index space, no sort;
            // like the octree variant the call returns with these passes in flight
            const unsigned long long cells = (unsigned long long)hw[C_DIVB] * hw[C_DIVB + 1] * hw[C_DIVB + 2];
            if (cells <= bitmap_max) {
                const uint32_t nwords = (uint32_t)((cells + 31) / 32), nblk = (nwords + GB_WORDS_PER_BLOCK - 1) / GB_WORDS_PER_BLOCK;
                bool ready = true;
                if (ws.gwords_cap < nwords) {
                    if (ws.gbits) (void)hipFree(ws.gbits);
                    if (ws.gprefix) (void)hipFree(ws.gprefix);
                    if (ws.gblock) (void)hipFree(ws.gblock);
                    ws.gbits = ws.gprefix = ws.gblock = nullptr; ws.gwords_cap = 0;
                    const size_t cap = std::max<size_t>((size_t)nwords * 2, (size_t)1 << 16);
                    ready = hipMalloc((void **)&ws.gbits, cap * 4) == hipSuccess && hipMalloc((void **)&ws.gprefix, cap * 4) == hipSuccess &&
                            hipMalloc((void **)&ws.gblock, (cap / GB_WORDS_PER_BLOCK + 2) * 4) == hipSuccess &&
                            hipMemsetAsync(ws.gbits, 0, cap * 4, c.stream) == hipSuccess;
                    if (ready) ws.gwords_cap = cap; else (void)hipGetLastError();
                }
                dst = ready ? soa_alloc(m) : nullptr;
                if (dst) {
                    const GridSpec known{0, 0u, 0u, 0ull};
                    CW_LAUNCH("grid_mark", grid_mark_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, m, known, ws.gbits, ws.order);
                    CW_LAUNCH("grid_block", grid_block_kernel, dim3(nblk), dim3(256), 0, c.stream, W, known, ws.gbits, nwords, ws.gprefix, ws.gblock);
                    CW_LAUNCH("grid_blockscan", grid_blockscan_kernel, dim3(1), dim3(1024), 0, c.stream, W, known, ws.gblock, nwords);
                    CW_LAUNCH("grid_emit", grid_emit_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, m, known, ws.order, ws.gbits, ws.gprefix, ws.gblock,
                              dst->x(), dst->y(), dst->z(), dst->rgbt());
                    CW_LAUNCH("grid_unmark", grid_unmark_kernel, dim3(mgrid), dim3(256), 0, c.stream, W, m, known, ws.order, ws.gbits);
                    dst->mark_pending(c.stream);
                    grid_ranked = true;
                }
            }
        }
        if (!err && m && !leaf_split && !grid_ranked) {
            // index spaces beyond 2^28 cells: sort the touched records by index
            dst = soa_alloc(m);
            keys_in = (unsigned long long *)pool_alloc((size_t)m * 8 * 2);
            vals_in = (uint32_t *)pool_alloc((size_t)m * 4 * 2);
            if (!dst || !keys_in || !vals_in) {
                err |= 0x80000000u;
            } else {
                keys_out = keys_in + m;
                vals_out = vals_in + m;
                CW_LAUNCH("make_sort_keys", make_sort_keys_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, m, keys_in, vals_in);
                // only the bits that can be set take part in the sort
                const unsigned end_bit = 32;   // idx < 2^31
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
        if (leaf_split && !ranked) {
            // octree variant, error: the records must be left zeroed
            CW_LAUNCH("clean_by_bitmap", clean_by_bitmap_kernel, dim3(ws.leaf_cap * RANK_SEGS), dim3(RANK_THREADS), 0, c.stream, W);
            ok = c.sync() && ok;
        }
        if (!leaf_split && m && !grid_ranked) {
            // emit (or, on error, only clean): the records must be left zeroed either way
            const int emit = (!err && dst) ? 1 : 0;
            CW_LAUNCH("emit_and_clean", emit_and_clean_kernel, dim3(mgrid), dim3(256), 0, c.stream, P, W, m, vals_out, emit ? dst->x() : nullptr,
                      emit ? dst->y() : nullptr, emit ? dst->z() : nullptr, emit ? dst->rgbt() : nullptr, emit);
            if (emit) ok = hipMemcpyAsync(c.host_words, ws.ctrl, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
            ok = c.sync() && ok;
            if (emit && ok) err |= c.host_words[C_ERR];
        }
        pool_free(keys_in);
        pool_free(vals_in);
        pool_free(sort_tmp);
        if (error_code) *error_code = (int)err;
        if (!ok) { hip_failed(hipGetLastError(), "voxel emit", __FILE__, __LINE__); return nullptr; }

        const uint32_t retryable = ERR_LEAVES | ERR_FACE_TABLE | ERR_LOCAL_LEAVES | ERR_LIST_FULL | ERR_FAST_PATH | (used_fast ? ERR_CELL_RANGE : 0u);
        if (used_fast && (err & (ERR_FAST_PATH | ERR_CELL_RANGE)) && !(err & ~retryable)) {
            // not a cloud for the fast variant (its table, its key or its slabs): the touched records were cleaned above
            ws.no_fast = true;
            continue;
        }
        if ((err & (ERR_LEAVES | ERR_FACE_TABLE | ERR_LOCAL_LEAVES)) && !(err & ~retryable)) {
            if (err & ERR_LOCAL_LEAVES) local_leaves = false;   // a workgroup spans more than 64 leaves: global ids in the hot loop
            // the touched records were cleaned above; change what was too small and run again
            if (err & ERR_FACE_TABLE) mode = 2;   // points beyond the threshold table: per-point f64 variant
            if (err & ERR_LEAVES) {
                if ((size_t)leaf_cap * 4 * GRID_BYTES > ((size_t)200 << 30)) {
                    cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: the cloud spans more octree leaves than fit in device memory");
                    return nullptr;
                }
                leaf_cap *= 4;
            }
            continue;
        }
        if (err) {
            std::string why;
            if (err & ERR_GRID_OVERFLOW) why += " VoxelGrid: leaf size is too small for the input dataset, integer indices would overflow;";
            if (err & ERR_RANGE) why += " voxel or leaf index out of range;";
            if (err & (ERR_DEPTH | ERR_LEAF_RANGE)) why += " octree deeper than 14 levels;";
            if (err & ERR_CELL_RANGE) why += " voxel outside its leaf grid;";
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
        ws.streak = attempt == 0 ? ws.streak + 1 : 0;
        ws.note_leaves(hw[C_LEAVES]);
        return dst;
    }
    cwipc_log(CWIPC_LOG_LEVEL_ERROR, "cwipc_downsample", "voxel grid failed: could not size the workspace");
    return nullptr;
}

size_t voxel_workspace_bytes() { return g_workspace_bytes.load(); }

// The workspaces that threads which have ended left for the next threads (at most eight) go back to the device.  For a
// deployment that wants the memory back after a burst of threads, and for tests that measure a footprint.
size_t voxel_release_pooled_workspaces() {
    std::vector<Workspace *> gone;
    {
        std::lock_guard<std::mutex> lock(g_ws_pool_mutex);
        gone.swap(*g_ws_pool);
    }
    if (!gone.empty()) (void)hipDeviceSynchronize();
    for (Workspace *w : gone) {
        if (w->pending) (void)w->pending->outcome();   // (a result somebody still holds reads its report from this workspace's words)
        delete w;
    }
    return gone.size();
}

}  // namespace cwipc_amd

extern "C" _CWIPC_UTIL_EXPORT size_t cwipc_hip_workspace_bytes(void) { return cwipc_amd::voxel_workspace_bytes(); }
extern "C" _CWIPC_UTIL_EXPORT size_t cwipc_hip_workspace_trim(void) { return cwipc_amd::voxel_release_pooled_workspaces(); }
