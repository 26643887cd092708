// kernels_basic.hip -- gfx950 kernels for the exact per-point filters and the
// AoS <-> SoA copy path.  All of them are HBM-bound byte/integer work: the
// design rules are coalesced 4/16-byte accesses on the SoA planes, wave64
// ballot/prefix for stable compaction, and enough workgroups (>> 256) to fill
// the 8 XCDs.  No MFMA: nothing here is a contraction.
//
// Reference semantics (bit-exact, including output order):
//   tilefilter  src/cwipc_filters.cpp:295-299      crop      :347-354
//   tilemap     src/cwipc_filters.cpp:322-325      colormap  :376-380
//   join        src/cwipc_filters.cpp:403-409      colorize  python/cwipc/filters/colorize.py:100-119
#include "internal.hpp"

namespace cwipc_amd {
namespace k {

static constexpr int BLOCK = 256;
static constexpr int WAVES = BLOCK / 64;

// ---------------------------------------------------------------------------
// AoS <-> SoA
// ---------------------------------------------------------------------------
// One 16-byte record per lane: a wave reads/writes 1 KiB contiguous on the AoS
// side and four 256-byte runs on the plane side.
__global__ void __launch_bounds__(BLOCK) aos_to_soa_kernel(const uint4 *__restrict__ aos, float *__restrict__ x, float *__restrict__ y,
                                                          float *__restrict__ z, uint32_t *__restrict__ rgbt, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t stride = (size_t)gridDim.x * BLOCK;
    for (; i < n; i += stride) {
        uint4 p = aos[i];
        x[i] = __uint_as_float(p.x);
        y[i] = __uint_as_float(p.y);
        z[i] = __uint_as_float(p.z);
        rgbt[i] = p.w;
    }
}

__global__ void __launch_bounds__(BLOCK) soa_to_aos_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                          const uint32_t *__restrict__ rgbt, uint4 *__restrict__ aos, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t stride = (size_t)gridDim.x * BLOCK;
    for (; i < n; i += stride) {
        uint4 p;
        p.x = __float_as_uint(x[i]);
        p.y = __float_as_uint(y[i]);
        p.z = __float_as_uint(z[i]);
        p.w = rgbt[i];
        aos[i] = p;
    }
}

// Gathered slots (one per rank, each with room for the largest cloud) -> planes, in slot order.
struct SlotTable {
    uint32_t first[MAX_SLOTS + 1];   // output index of slot s's first record; first[nslots] = total
};
__global__ void __launch_bounds__(BLOCK) slots_to_soa_kernel(const uint4 *__restrict__ slots, SlotTable t, int nslots, size_t slot_rows, size_t header_rows,
                                                            float *__restrict__ x, float *__restrict__ y, float *__restrict__ z,
                                                            uint32_t *__restrict__ rgbt) {
    const size_t n = t.first[nslots];
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (; i < n; i += stride) {
        int s = 0;
        while (s + 1 < nslots && i >= t.first[s + 1]) s++;   // (a handful of slots: a scan beats a search)
        const uint4 p = slots[(size_t)s * slot_rows + header_rows + (i - t.first[s])];
        x[i] = __uint_as_float(p.x);
        y[i] = __uint_as_float(p.y);
        z[i] = __uint_as_float(p.z);
        rgbt[i] = p.w;
    }
}

static inline unsigned grid_for(size_t n, size_t per_block) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;   // grid-stride beyond 32 blocks per CU
    return (unsigned)g;
}

void slots_to_soa(const void *slots, int nslots, size_t slot_rows, size_t header_rows, const uint32_t *counts, const DeviceSoA &dst, hipStream_t s) {
    SlotTable t;
    uint32_t at = 0;
    for (int i = 0; i < nslots; i++) { t.first[i] = at; at += counts[i]; }
    for (int i = nslots; i <= MAX_SLOTS; i++) t.first[i] = at;
    if (!at) return;
    CW_LAUNCH("slots_to_soa", slots_to_soa_kernel, dim3(grid_for(at, BLOCK)), dim3(BLOCK), 0, s, (const uint4 *)slots, t, nslots, slot_rows, header_rows,
              dst.x(), dst.y(), dst.z(), dst.rgbt());
}

void aos_to_soa(const cwipc_point *aos, const DeviceSoA &dst, size_t n, hipStream_t s) {
    if (!n) return;
    CW_LAUNCH("aos_to_soa", aos_to_soa_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, (const uint4 *)aos, dst.x(), dst.y(), dst.z(), dst.rgbt(), n);
}

void soa_to_aos(const DeviceSoA &src, cwipc_point *aos, size_t n, hipStream_t s) {
    if (!n) return;
    CW_LAUNCH("soa_to_aos", soa_to_aos_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, src.x(), src.y(), src.z(), src.rgbt(), (uint4 *)aos, n);
}

// ---------------------------------------------------------------------------
// Stable compaction (tilefilter / crop / masked tilefilter)
// ---------------------------------------------------------------------------
// A workgroup owns TILE consecutive points.  Pass 1 counts the kept points per
// workgroup, a one-workgroup scan turns counts into offsets, pass 2 re-evaluates
// the predicate and writes every kept point at offset + rank, where rank comes
// from a wave64 ballot/popcount prefix -- so output order == input order.
static constexpr int ITEMS = 4;                         // points per lane and step (one dwordx4 per plane)
static constexpr int STEPS = 4;                         // steps per workgroup
static constexpr int TILE = BLOCK * ITEMS * STEPS;      // 4096 points per workgroup ...
// ... and 1024 for clouds of up to 256 k points (round 3): a camera tile of 36 k points is nine workgroups of 4096, on nine of 256
// compute units, each walking through its four steps (8.5 us for the scatter kernel alone)
static constexpr size_t SMALL_CLOUD = 262144;
static constexpr int SMALL_STEPS = 1;
static constexpr int SMALL_TILE = BLOCK * ITEMS * SMALL_STEPS;

size_t compact_small_cloud_limit() { return SMALL_CLOUD; }
size_t compact_blocks(size_t n) { return n <= SMALL_CLOUD ? (n + SMALL_TILE - 1) / SMALL_TILE : (n + TILE - 1) / TILE; }

struct PredArgs {
    int mode;
    int tile;
    float b0, b1, b2, b3, b4, b5;
    const float *dist;
    double thr;
    const double *thr_dev;   // mode 3: the threshold, if a kernel computed it (else thr)
    const double *stat_partial;   // mode 3, compact_count_scan_kernel: the statistics' partial sums instead of a threshold (Predicate)
    size_t stat_n;
    float stat_mul;
    double *thr_out;
};

__device__ __forceinline__ bool keep_point(const PredArgs &p, float x, float y, float z, uint32_t w, size_t idx) {
    int t = (int)(w >> 24);
    if (p.mode == 0) return p.tile == 0 || p.tile == t;
    if (p.mode == 2) return (t & p.tile) != 0;
    if (p.mode == 3) return !((double)p.dist[idx] > (p.thr_dev ? *p.thr_dev : p.thr));   // fp32 d_i widened for the f64 compare
    return p.b0 <= x && x < p.b1 && p.b2 <= y && y < p.b3 && p.b4 <= z && z < p.b5;
}

// Loads ITEMS consecutive points of one lane; out-of-range items report keep = false.
template <bool NEED_XYZ, bool NEED_W = true>
__device__ __forceinline__ unsigned lane_mask(const PredArgs &p, const float *__restrict__ x, const float *__restrict__ y,
                                              const float *__restrict__ z, const uint32_t *__restrict__ rgbt, size_t base, size_t n,
                                              float4 &vx, float4 &vy, float4 &vz, uint4 &vw) {
    unsigned m = 0;
    if (base + ITEMS <= n) {
        if (NEED_W) vw = *(const uint4 *)(rgbt + base);
        if (NEED_XYZ) {
            vx = *(const float4 *)(x + base);
            vy = *(const float4 *)(y + base);
            vz = *(const float4 *)(z + base);
        }
        m |= keep_point(p, vx.x, vy.x, vz.x, vw.x, base) ? 1u : 0u;
        m |= keep_point(p, vx.y, vy.y, vz.y, vw.y, base + 1) ? 2u : 0u;
        m |= keep_point(p, vx.z, vy.z, vz.z, vw.z, base + 2) ? 4u : 0u;
        m |= keep_point(p, vx.w, vy.w, vz.w, vw.w, base + 3) ? 8u : 0u;
    } else if (base < n) {
        float ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
        uint32_t aw[4] = {0, 0, 0, 0};
        for (int j = 0; j < ITEMS; j++) {
            if (base + j < n) {
                if (NEED_W) aw[j] = rgbt[base + j];
                if (NEED_XYZ) { ax[j] = x[base + j]; ay[j] = y[base + j]; az[j] = z[base + j]; }
                if (keep_point(p, ax[j], ay[j], az[j], aw[j], base + j)) m |= 1u << j;
            }
        }
        vx = make_float4(ax[0], ax[1], ax[2], ax[3]);
        vy = make_float4(ay[0], ay[1], ay[2], ay[3]);
        vz = make_float4(az[0], az[1], az[2], az[3]);
        vw = make_uint4(aw[0], aw[1], aw[2], aw[3]);
    }
    return m;
}

// The scatter kernel's loads (r4): the planes the predicate looks at first, the others only by lanes that keep a point -- a tile
// filter that keeps half of a 10 M-point cloud does not read the coordinates of the other half (60 of 240 MB).
__device__ __forceinline__ unsigned lane_mask_lazy(const PredArgs &p, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                   const uint32_t *__restrict__ rgbt, size_t base, size_t n, float4 &vx, float4 &vy, float4 &vz, uint4 &vw) {
    if (base + ITEMS > n) return lane_mask<true>(p, x, y, z, rgbt, base, n, vx, vy, vz, vw);   // (the ragged last lane, and lanes beyond the end)
    unsigned m = 0;
    if (p.mode == 1) {           // crop: the coordinates decide
        vx = *(const float4 *)(x + base); vy = *(const float4 *)(y + base); vz = *(const float4 *)(z + base);
        m |= keep_point(p, vx.x, vy.x, vz.x, 0u, base) ? 1u : 0u;
        m |= keep_point(p, vx.y, vy.y, vz.y, 0u, base + 1) ? 2u : 0u;
        m |= keep_point(p, vx.z, vy.z, vz.z, 0u, base + 2) ? 4u : 0u;
        m |= keep_point(p, vx.w, vy.w, vz.w, 0u, base + 3) ? 8u : 0u;
        if (m) vw = *(const uint4 *)(rgbt + base);
    } else {                     // tile value, tile mask: the colour / tile word decides; outlier removal: d_i (no plane at all)
        if (p.mode != 3) vw = *(const uint4 *)(rgbt + base);
        m |= keep_point(p, 0.f, 0.f, 0.f, vw.x, base) ? 1u : 0u;
        m |= keep_point(p, 0.f, 0.f, 0.f, vw.y, base + 1) ? 2u : 0u;
        m |= keep_point(p, 0.f, 0.f, 0.f, vw.z, base + 2) ? 4u : 0u;
        m |= keep_point(p, 0.f, 0.f, 0.f, vw.w, base + 3) ? 8u : 0u;
        if (m) {
            if (p.mode == 3) vw = *(const uint4 *)(rgbt + base);
            vx = *(const float4 *)(x + base); vy = *(const float4 *)(y + base); vz = *(const float4 *)(z + base);
        }
    }
    return m;
}

// kept points of one workgroup's tile (S steps of BLOCK * ITEMS points); the predicate's planes only
template <int S>
__device__ __forceinline__ uint32_t count_tile(const PredArgs &p, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                               const uint32_t *__restrict__ rgbt, size_t n, uint32_t *wave_sum) {
    size_t tile0 = (size_t)blockIdx.x * (BLOCK * ITEMS * S);
    uint32_t cnt = 0;
    float4 vx = make_float4(0, 0, 0, 0), vy = vx, vz = vx;
    uint4 vw = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < S; s++) {
        size_t base = tile0 + (size_t)s * BLOCK * ITEMS + (size_t)threadIdx.x * ITEMS;
        // (crop looks at the coordinates only, outlier removal at d_i only, the tile predicates at the colour / tile word only)
        unsigned m = (p.mode == 1) ? lane_mask<true, false>(p, x, y, z, rgbt, base, n, vx, vy, vz, vw)
                   : (p.mode == 3) ? lane_mask<false, false>(p, x, y, z, rgbt, base, n, vx, vy, vz, vw)
                                   : lane_mask<false, true>(p, x, y, z, rgbt, base, n, vx, vy, vz, vw);
        cnt += __popc(m);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    uint32_t t = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < WAVES; w++) t += wave_sum[w];
    return t;   // (thread 0's value is the tile's count)
}

// The outlier filter's threshold from its 1024 pairs of partial sums (kernels_sor.hip, stats_partial_kernel): the pairwise tree and the f64
// expressions of stats_final_kernel there, level by level in the same order, run by the BLOCK lanes of a workgroup -- every workgroup of the
// count kernel runs it and gets the same bits; that kernel and its boundary (~5 us of a 36 k-point tile's ~90) go.
__device__ __forceinline__ double threshold_from_partials(const double *__restrict__ partial, size_t n, float stddev_mul) {
    constexpr int NP = 1024, PER = NP / BLOCK;
    __shared__ double s[NP], q[NP];
    for (int i = threadIdx.x; i < NP; i += BLOCK) { s[i] = partial[2 * i]; q[i] = partial[2 * i + 1]; }
    __syncthreads();
    for (unsigned width = NP / 2; width >= 1; width >>= 1) {
        double a[PER], b[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const unsigned i = threadIdx.x + (unsigned)r * BLOCK;
            a[r] = b[r] = 0;
            if (i < width) { a[r] = s[2 * i] + s[2 * i + 1]; b[r] = q[2 * i] + q[2 * i + 1]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const unsigned i = threadIdx.x + (unsigned)r * BLOCK;
            if (i < width) { s[i] = a[r]; q[i] = b[r]; }
        }
        __syncthreads();
    }
    const double sum = s[0], sq_sum = q[0];
    const double valid = (double)n;
    const double mean = sum / valid;
    const double variance = (sq_sum - sum * sum / valid) / (valid - 1);
    const double stddev = sqrt(variance);
    return mean + (double)stddev_mul * stddev;
}

template <int S>
__global__ void __launch_bounds__(BLOCK) compact_count_kernel(PredArgs p, const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ z, const uint32_t *__restrict__ rgbt, size_t n,
                                                             uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t wave_sum[WAVES];
    if (p.mode == 3 && p.stat_partial) {   // (as in compact_count_scan_kernel, should a small cloud ever come this way)
        const double thr = threshold_from_partials(p.stat_partial, p.stat_n, p.stat_mul);
        if (blockIdx.x == 0 && threadIdx.x == 0) *p.thr_out = thr;
        p.thr = thr;
        p.thr_dev = nullptr;
    }
    const uint32_t t = count_tile<S>(p, x, y, z, rgbt, n, wave_sum);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = t;
}

// The exclusive scan of the block counts by the workgroup that runs it (NT lanes): counts[i] <- sum of those before, counts[nblocks]
// <- the total, which also goes to a pinned host word with `tag` in its upper half (one 64-bit store, no fence: the host polls for
// the tag instead of waiting for the stream).
template <int NT>
__device__ __forceinline__ void scan_block_counts(uint32_t *__restrict__ counts, size_t nblocks, unsigned long long *__restrict__ total, uint32_t tag) {
    __shared__ uint32_t wave_tot[NT / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t base = 0; base < nblocks; base += NT) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? counts[i] : 0;
        uint32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t wave_base = 0;
        for (int w = 0; w < wave; w++) wave_base += wave_tot[w];
        uint32_t c = carry;
        if (i < nblocks) counts[i] = c + wave_base + inc - v;
        __syncthreads();
        if (threadIdx.x == NT - 1) carry = c + wave_base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[nblocks] = carry;   // for the scatter kernel: a compaction that keeps every point moves nothing
        __hip_atomic_store(total, ((unsigned long long)tag << 32) | carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Count and scan in one launch: every workgroup leaves its count and takes a ticket; the one that takes the last ticket has
// every count in front of it (release / acquire at device scope around the ticket) and scans them.  For small clouds, whose
// kernels cost the device less than their launches cost the host (a camera tile through the outlier filter: sixteen launches).
template <int S>
__global__ void __launch_bounds__(BLOCK) compact_count_scan_kernel(PredArgs p, const float *__restrict__ x, const float *__restrict__ y,
                                                                  const float *__restrict__ z, const uint32_t *__restrict__ rgbt, size_t n,
                                                                  uint32_t *__restrict__ block_counts, uint32_t *__restrict__ ticket,
                                                                  unsigned long long *__restrict__ total, uint32_t tag) {
    __shared__ uint32_t wave_sum[WAVES];
    __shared__ uint32_t is_last;
    if (p.mode == 3 && p.stat_partial) {
        const double thr = threshold_from_partials(p.stat_partial, p.stat_n, p.stat_mul);
        if (blockIdx.x == 0 && threadIdx.x == 0) *p.thr_out = thr;   // (the scatter kernel reads it there)
        p.thr = thr;
        p.thr_dev = nullptr;
    }
    const uint32_t t = count_tile<S>(p, x, y, z, rgbt, n, wave_sum);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&block_counts[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t before = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        is_last = before == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next kernel that uses it
    scan_block_counts<BLOCK>(block_counts, gridDim.x, total, tag);
}

// Exclusive scan of up to millions of block counts by ONE workgroup of 1024 lanes
// (the count array is tiny: N/4096 entries).
// The total goes to a pinned host word with `tag` in its upper half (one 64-bit store, no fence): the
// host polls for the tag instead of waiting for the stream.
__global__ void __launch_bounds__(1024) compact_scan_kernel(uint32_t *__restrict__ counts, size_t nblocks, unsigned long long *__restrict__ total,
                                                            uint32_t tag) {
    scan_block_counts<1024>(counts, nblocks, total, tag);
}

// SUMS: the workgroup adds the counts of the workgroups in front of it up itself (big clouds, r4: the one-workgroup scan kernel between the
// count and the scatter kernel and its two kernel boundaries go; a workgroup reads 1200 counts on average, next to a tile of 64 KB), and
// workgroup 0 leaves the total for the host.  Otherwise block_offsets are offsets already (small clouds: scanned by the count kernel's last
// workgroup), the total behind them.
template <int S, bool SUMS>
__global__ void __launch_bounds__(BLOCK) compact_scatter_kernel(PredArgs p, const float *__restrict__ x, const float *__restrict__ y,
                                                               const float *__restrict__ z, const uint32_t *__restrict__ rgbt, size_t n,
                                                               const uint32_t *__restrict__ block_offsets, float *__restrict__ ox,
                                                               float *__restrict__ oy, float *__restrict__ oz, uint32_t *__restrict__ ow,
                                                               unsigned long long *__restrict__ total_host, uint32_t tag) {
    __shared__ uint32_t wave_sum[S][WAVES];
    __shared__ uint32_t prefix_sum[2][WAVES];
    __shared__ uint32_t stage[WAVES][4][64 * ITEMS];   // per wave: the kept points of one step in output order, plane by plane
    uint32_t before_me = 0, everything = 0;
    if (SUMS) {
        uint32_t lo = 0, all = 0;
        for (uint32_t i = threadIdx.x; i < gridDim.x; i += BLOCK) {
            const uint32_t v = block_offsets[i];
            all += v;
            if (i < blockIdx.x) lo += v;
        }
        for (int off = 32; off > 0; off >>= 1) { lo += __shfl_down(lo, off, 64); all += __shfl_down(all, off, 64); }
        if ((threadIdx.x & 63) == 0) { prefix_sum[0][threadIdx.x >> 6] = lo; prefix_sum[1][threadIdx.x >> 6] = all; }
        __syncthreads();
        for (int w = 0; w < WAVES; w++) { before_me += prefix_sum[0][w]; everything += prefix_sum[1][w]; }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(total_host, ((unsigned long long)tag << 32) | everything, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        before_me = block_offsets[blockIdx.x];
        everything = block_offsets[gridDim.x];
    }
    // every point kept (a tile filter on a cloud of that one tile, a crop box around everything): the host hands the input's
    // planes on as the result, nothing is copied
    if (everything == n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile0 = (size_t)blockIdx.x * (BLOCK * ITEMS * S);
    const size_t out0 = before_me;
    // the whole tile is loaded before anything else happens (16 loads of 16 B per lane in flight);
    // the scatter needs all four planes of the kept points
    float4 vx[S], vy[S], vz[S];
    uint4 vw[S];
    unsigned m[S];
    uint32_t inc[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const size_t base = tile0 + (size_t)s * BLOCK * ITEMS + (size_t)threadIdx.x * ITEMS;
        vx[s] = make_float4(0, 0, 0, 0); vy[s] = vx[s]; vz[s] = vx[s];
        vw[s] = make_uint4(0, 0, 0, 0);
        m[s] = lane_mask_lazy(p, x, y, z, rgbt, base, n, vx[s], vy[s], vz[s], vw[s]);
        inc[s] = __popc(m[s]);
    }
    // ranks inside the tile (points run step-major, then lane-major): wave prefixes, one barrier
#pragma unroll
    for (int s = 0; s < S; s++) {
        const uint32_t c = inc[s];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc[s], off, 64);
            if (lane >= off) inc[s] += t;
        }
        if (lane == 63) wave_sum[s][wave] = inc[s];
        inc[s] -= c;   // exclusive inside the wave
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int s = 0; s < S; s++) {
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const uint32_t t = wave_sum[s][w];
            if (w < wave) before += t;
            tot += t;
        }
        // r4: the wave's kept points of this step -- a run of the output, wave_total long from wave_pos0 on -- go through LDS in output
        // order and out in 16-byte stores (1 KiB per wave instruction and plane) with a scalar head and tail around the aligned part.
        // (Rounds 1-3 had every lane store its own points, 4 bytes at a time at a lane stride of up to 16: the scatter kernel of a
        // tile filter that keeps half of 10 M points ran at 3.3 TB/s, a plain copy of the same planes at 5.)
        const uint32_t wave_total = wave_sum[s][wave];
        const size_t wave_pos0 = out0 + run + before;
        run += tot;
        if (wave_total == 0) continue;   // (wave-uniform)
        {
            const float ax[4] = {vx[s].x, vx[s].y, vx[s].z, vx[s].w}, ay[4] = {vy[s].x, vy[s].y, vy[s].z, vy[s].w};
            const float az[4] = {vz[s].x, vz[s].y, vz[s].z, vz[s].w};
            const uint32_t aw[4] = {vw[s].x, vw[s].y, vw[s].z, vw[s].w};
            uint32_t at = inc[s];
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                if (m[s] & (1u << j)) {
                    stage[wave][0][at] = __float_as_uint(ax[j]);
                    stage[wave][1][at] = __float_as_uint(ay[j]);
                    stage[wave][2][at] = __float_as_uint(az[j]);
                    stage[wave][3][at] = aw[j];
                    at++;
                }
            }
        }
        // (LDS operations of one wave are carried out in the order they were issued: no barrier, but the compiler must keep the order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t head = min(wave_total, (uint32_t)((4u - (uint32_t)(wave_pos0 & 3u)) & 3u));
        const uint32_t quads = (wave_total - head) >> 2, tail0 = head + 4u * quads;
        float *const outs[4] = {ox, oy, oz, reinterpret_cast<float *>(ow)};
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
            float *const o = outs[pl] + wave_pos0;
            const uint32_t *const st = stage[wave][pl];
            if ((uint32_t)lane < head) o[lane] = __uint_as_float(st[lane]);
            if ((uint32_t)lane < quads) {
                const uint32_t i = head + 4u * (uint32_t)lane;
                *reinterpret_cast<float4 *>(o + i) = make_float4(__uint_as_float(st[i]), __uint_as_float(st[i + 1]), __uint_as_float(st[i + 2]), __uint_as_float(st[i + 3]));
            }
            if (tail0 + (uint32_t)lane < wave_total) o[tail0 + lane] = __uint_as_float(st[tail0 + lane]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

static PredArgs to_args(const Predicate &p) {
    PredArgs a;
    a.mode = p.mode;
    a.tile = p.tile;
    a.b0 = p.bbox[0]; a.b1 = p.bbox[1]; a.b2 = p.bbox[2];
    a.b3 = p.bbox[3]; a.b4 = p.bbox[4]; a.b5 = p.bbox[5];
    a.dist = p.dist;
    a.thr = p.thr;
    a.thr_dev = p.thr_dev;
    a.stat_partial = p.stat_partial; a.stat_n = p.stat_n; a.stat_mul = p.stat_mul; a.thr_out = p.thr_out;
    return a;
}

void compact_count(const DeviceSoA &src, const Predicate &p, uint32_t *block_counts, hipStream_t s) {
    size_t nb = compact_blocks(src.npoints);
    if (!nb) return;
    if (src.npoints <= SMALL_CLOUD) {
        CW_LAUNCH("compact_count", compact_count_kernel<SMALL_STEPS>, dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(), src.rgbt(),
                  src.npoints, block_counts);
    } else {
        CW_LAUNCH("compact_count", compact_count_kernel<STEPS>, dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(), src.rgbt(),
                  src.npoints, block_counts);
    }
}

bool compact_count_scan(const DeviceSoA &src, const Predicate &p, uint32_t *block_counts, uint32_t *ticket, unsigned long long *total_host, uint32_t tag,
                        hipStream_t s) {
    size_t nb = compact_blocks(src.npoints);
    // Small clouds only.  (r4: tried at every size -- 2441 workgroups for 10 M points, each with its release / acquire around the one
    // ticket word: tilefilter 81 -> 129 us, crop 70 -> 147.  An agent-scope release writes the XCD's L2 back, and all tickets
    // are one address at the memory side.)
    if (!nb || src.npoints > SMALL_CLOUD || !ticket) return false;
    CW_LAUNCH("compact_count", compact_count_scan_kernel<SMALL_STEPS>, dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(), src.rgbt(),
              src.npoints, block_counts, ticket, total_host, tag);
    return true;
}

void compact_scan(uint32_t *block_counts, size_t nblocks, unsigned long long *total_host, uint32_t tag, hipStream_t s) {
    CW_LAUNCH("compact_scan", compact_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nblocks, total_host, tag);
}

void compact_scatter(const DeviceSoA &src, const Predicate &p, const uint32_t *block_offsets, const DeviceSoA &dst, hipStream_t s, unsigned long long *total_host,
                     uint32_t tag) {
    size_t nb = compact_blocks(src.npoints);
    if (!nb) return;
    // total_host != nullptr: block_offsets hold the count kernel's counts as they are (see the kernel)
    if (src.npoints <= SMALL_CLOUD) {
        if (total_host) {
            CW_LAUNCH("compact_scatter", (compact_scatter_kernel<SMALL_STEPS, true>), dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(),
                      src.rgbt(), src.npoints, block_offsets, dst.x(), dst.y(), dst.z(), dst.rgbt(), total_host, tag);
        } else {
            CW_LAUNCH("compact_scatter", (compact_scatter_kernel<SMALL_STEPS, false>), dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(),
                      src.rgbt(), src.npoints, block_offsets, dst.x(), dst.y(), dst.z(), dst.rgbt(), total_host, tag);
        }
    } else if (total_host) {
        CW_LAUNCH("compact_scatter", (compact_scatter_kernel<STEPS, true>), dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(),
                  src.rgbt(), src.npoints, block_offsets, dst.x(), dst.y(), dst.z(), dst.rgbt(), total_host, tag);
    } else {
        CW_LAUNCH("compact_scatter", (compact_scatter_kernel<STEPS, false>), dim3((unsigned)nb), dim3(BLOCK), 0, s, to_args(p), src.x(), src.y(), src.z(),
                  src.rgbt(), src.npoints, block_offsets, dst.x(), dst.y(), dst.z(), dst.rgbt(), total_host, tag);
    }
}

// ---------------------------------------------------------------------------
// Per-point maps on the packed colour/tile word
// ---------------------------------------------------------------------------
enum { MAP_TILE = 0, MAP_BITS = 1, MAP_COLORIZE = 2 };

struct MapArgs {
    uint32_t clear_mask;   // MAP_BITS: word &= ~clear ; word |= set   (already in rgbt byte order)
    uint32_t set_mask;
};

// Layout of the colorize table in device memory (doubles):
//   [0, 768)    cw[t][c]  = colour[t][c] * weight        (host product, Python float semantics)
//   [768, 1024) oldf[v]   = v / 255.0
//   [1024]      omw       = 1 - weight
//   [1025,1281) valid[t]  (non-zero = the colour map has an entry for tile t)
static constexpr int COLORIZE_DOUBLES = 1025 + 256;

// COPY_XYZ = false: the result shares the input's coordinate planes (soa_with_new_rgbt), only the words are touched.
template <int KIND, bool COPY_XYZ>
__global__ void __launch_bounds__(BLOCK) map_kernel(MapArgs a, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                   const uint32_t *__restrict__ rgbt, float *__restrict__ ox, float *__restrict__ oy,
                                                   float *__restrict__ oz, uint32_t *__restrict__ ow, size_t n, const void *__restrict__ table) {
    __shared__ double lut[KIND == MAP_COLORIZE ? COLORIZE_DOUBLES : 1];
    __shared__ uint8_t tmap[KIND == MAP_TILE ? 256 : 1];
    if (KIND == MAP_TILE) {
        tmap[threadIdx.x] = ((const uint8_t *)table)[threadIdx.x];   // BLOCK == 256
        __syncthreads();
    }
    if (KIND == MAP_COLORIZE) {
        for (int i = threadIdx.x; i < COLORIZE_DOUBLES; i += BLOCK) lut[i] = ((const double *)table)[i];
        __syncthreads();
    }
    const size_t nvec = n / 4;
    size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t v = (size_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec + 1; v += stride) {
        uint32_t w[4];
        int cnt = 4;
        if (v < nvec) {
            uint4 t = ((const uint4 *)rgbt)[v];
            w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
            if (COPY_XYZ) {
                ((float4 *)ox)[v] = ((const float4 *)x)[v];
                ((float4 *)oy)[v] = ((const float4 *)y)[v];
                ((float4 *)oz)[v] = ((const float4 *)z)[v];
            }
        } else {
            cnt = (int)(n - nvec * 4);   // ragged tail, handled by one lane
            for (int j = 0; j < cnt; j++) {
                size_t i = nvec * 4 + j;
                w[j] = rgbt[i];
                if (COPY_XYZ) { ox[i] = x[i]; oy[i] = y[i]; oz[i] = z[i]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= cnt) break;
            uint32_t word = w[j];
            if (KIND == MAP_TILE) {
                word = (word & 0x00ffffffu) | ((uint32_t)tmap[word >> 24] << 24);
            } else if (KIND == MAP_BITS) {
                word = (word & ~a.clear_mask) | a.set_mask;
            } else {
                uint32_t t = word >> 24;
                if (lut[1025 + t] != 0.0) {
                    const double omw = lut[1024];
                    uint32_t outw = word & 0xff000000u;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        uint32_t old = (word >> (8 * c)) & 0xffu;
                        // new = colour*w + (old/255.0)*(1-w), then int(new*255): separately rounded f64 ops
                        double blended = __dadd_rn(lut[t * 3 + c], __dmul_rn(lut[768 + old], omw));
                        long long q = (long long)__dmul_rn(blended, 255.0);
                        outw |= ((uint32_t)q & 0xffu) << (8 * c);
                    }
                    word = outw;
                }
            }
            w[j] = word;
        }
        if (v < nvec) {
            ((uint4 *)ow)[v] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int j = 0; j < cnt; j++) ow[nvec * 4 + j] = w[j];
        }
    }
}

template <int KIND>
static void launch_map(const char *name, const DeviceSoA &src, const DeviceSoA &dst, MapArgs a, const void *table, hipStream_t s) {
    size_t n = src.npoints;
    if (!n) return;
    if (dst.x() == src.x()) {   // shared coordinate planes
        CW_LAUNCH(name, (map_kernel<KIND, false>), dim3(grid_for(n / 4 + 1, BLOCK)), dim3(BLOCK), 0, s, a, src.x(), src.y(), src.z(), src.rgbt(),
                  dst.x(), dst.y(), dst.z(), dst.rgbt(), n, table);
    } else {
        CW_LAUNCH(name, (map_kernel<KIND, true>), dim3(grid_for(n / 4 + 1, BLOCK)), dim3(BLOCK), 0, s, a, src.x(), src.y(), src.z(), src.rgbt(),
                  dst.x(), dst.y(), dst.z(), dst.rgbt(), n, table);
    }
}

void map_tile(const DeviceSoA &src, const DeviceSoA &dst, const uint8_t *dev_map256, hipStream_t s) {
    launch_map<MAP_TILE>("map_tile", src, dst, MapArgs{0, 0}, dev_map256, s);
}

// The reference masks PCL's rgba word (a<<24 | r<<16 | g<<8 | b, reference
// include/cwipc_util/api_pcl.h:20-70); the planes hold r | g<<8 | b<<16 | tile<<24,
// so the r and b bytes of both masks are swapped once on the host.
static inline uint32_t swap_rb(uint32_t m) { return (m & 0xff00ff00u) | ((m & 0xffu) << 16) | ((m >> 16) & 0xffu); }

void map_color_bits(const DeviceSoA &src, const DeviceSoA &dst, uint32_t clearBits, uint32_t setBits, hipStream_t s) {
    launch_map<MAP_BITS>("map_color_bits", src, dst, MapArgs{swap_rb(clearBits), swap_rb(setBits)}, nullptr, s);
}

void map_colorize(const DeviceSoA &src, const DeviceSoA &dst, const double *dev_table, hipStream_t s) {
    launch_map<MAP_COLORIZE>("map_colorize", src, dst, MapArgs{0, 0}, dev_table, s);
}

// ---------------------------------------------------------------------------
// geometry maps: affine transform, offset + scale (both in f64, one rounding to fp32)
// ---------------------------------------------------------------------------
struct AffineArgs {
    double m[12];   // rows of the 3x4 matrix [R | t]
    int mode;       // 0: R p + t  (reference python/cwipc/registration/util.py:295-309); 1: (p + t) * m[0]  (filters/transform.py:45-48)
};

__global__ void __launch_bounds__(BLOCK) affine_kernel(AffineArgs a, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                      const uint32_t *__restrict__ rgbt, float *__restrict__ ox, float *__restrict__ oy,
                                                      float *__restrict__ oz, uint32_t *__restrict__ ow, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const double px = (double)x[i], py = (double)y[i], pz = (double)z[i];
        double rx, ry, rz;
        if (a.mode == 0) {
            // numpy's `rotmat @ xyz.T` is a dgemm whose inner loop is a chain of fused multiply-adds in index order: the first
            // product rounded, the other two fused into the running sum; then the translation, a separate addition.  (Round 4:
            // found with the outputs of the reference function itself, tests/golden/helper_vectors.npz -- three separately
            // rounded products, the form of rounds 1-3, differ from it in the last bit for 3 of 3000 values of one matrix.)
            rx = __dadd_rn(__fma_rn(a.m[2], pz, __fma_rn(a.m[1], py, __dmul_rn(a.m[0], px))), a.m[3]);
            ry = __dadd_rn(__fma_rn(a.m[6], pz, __fma_rn(a.m[5], py, __dmul_rn(a.m[4], px))), a.m[7]);
            rz = __dadd_rn(__fma_rn(a.m[10], pz, __fma_rn(a.m[9], py, __dmul_rn(a.m[8], px))), a.m[11]);
        } else {
            rx = __dmul_rn(__dadd_rn(px, a.m[3]), a.m[0]);
            ry = __dmul_rn(__dadd_rn(py, a.m[7]), a.m[0]);
            rz = __dmul_rn(__dadd_rn(pz, a.m[11]), a.m[0]);
        }
        ox[i] = (float)rx; oy[i] = (float)ry; oz[i] = (float)rz;
        if (ow != rgbt) ow[i] = rgbt[i];   // (a result that shares the input's colour / tile words has nothing to copy)
    }
}

void map_affine(const DeviceSoA &src, const DeviceSoA &dst, const double m[12], int mode, hipStream_t s) {
    const size_t n = src.npoints;
    if (!n) return;
    AffineArgs a;
    for (int i = 0; i < 12; i++) a.m[i] = m[i];
    a.mode = mode;
    CW_LAUNCH("map_affine", affine_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, a, src.x(), src.y(), src.z(), src.rgbt(), dst.x(), dst.y(), dst.z(),
              dst.rgbt(), n);
}

// The synthetic source's points, generated where they are going to be used (reference src/cwipc_synthetic.cpp:182-222).
// Per point the reference takes one product per coordinate of a per-row value (the radius) and a per-column value (sin /
// cos of the angle): those two tables come from the host, made with the host's libm exactly as the reference makes
// them, so the coordinates and tiles are the reference's bit for bit.  The colours take a double-precision sine of a sum
// that differs per point; that one is computed here.  An error of a unit in the last place of that sine reaches the
// 8-bit colour only across two rounding boundaries (double -> float, then the truncation of 255 r): the test suite
// compares whole clouds against the host generator and has not seen a differing byte.
struct SyntheticArgs {
    int hsteps, asteps;
    float m_angle;
    int eyes_white;   // fmod(m_angle, pi / 2) > 0.08 (reference :206-210), evaluated on the host
    const float *radius, *height, *angle;   // [hsteps], [hsteps], [asteps]
    const double *sin_a, *cos_a;            // [asteps]
};

__global__ void __launch_bounds__(BLOCK) synthetic_kernel(SyntheticArgs a, float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
                                                         uint32_t *__restrict__ ow, size_t n) {
    const float pi = 3.14159265358979f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const int hi = (int)(i / (size_t)a.asteps), ai = (int)(i % (size_t)a.asteps);
        const float height = a.height[hi], angle = a.angle[ai], radius = a.radius[hi];
        const float x = (float)__dmul_rn((double)radius, a.sin_a[ai]);
        const float y = (float)__dmul_rn((double)radius, a.cos_a[ai]);
        // float r = (1 + sin((double)(2 * pi * height + m_angle + angle))) / 2;  -- the argument is a float expression
        const float t2 = __fadd_rn(__fadd_rn(__fmul_rn(__fmul_rn(2.0f, pi), height), a.m_angle), angle);
        const float t3 = __fadd_rn(__fadd_rn(__fmul_rn(__fmul_rn(3.0f, pi), height), a.m_angle), angle);
        const float t4 = __fadd_rn(__fadd_rn(__fmul_rn(__fmul_rn(4.0f, pi), height), a.m_angle), angle);
        const float r = (float)__ddiv_rn(__dadd_rn(1.0, sin((double)t2)), 2.0);
        const float g = (float)__ddiv_rn(__dadd_rn(1.0, sin((double)t3)), 2.0);
        const float b = (float)__ddiv_rn(__dadd_rn(1.0, sin((double)t4)), 2.0);
        int rr = (int)__dmul_rn((double)r, 255.0), gg = (int)__dmul_rn((double)g, 255.0), bb = (int)__dmul_rn((double)b, 255.0);
        if (height > 1.7 && height < 1.8 &&
            (((double)angle > (double)pi * 0.083 && (double)angle < (double)pi * 0.1667) ||
             ((double)angle > (double)pi * 1.833 && (double)angle < (double)pi * 1.917))) {
            if (a.eyes_white) rr = gg = bb = 255;
        }
        ox[i] = -x;
        oy[i] = height;
        oz[i] = y;
        ow[i] = (uint32_t)(rr & 0xff) | ((uint32_t)(gg & 0xff) << 8) | ((uint32_t)(bb & 0xff) << 16) | ((y < 0 ? 1u : 2u) << 24);
    }
}

void synthetic_fill(const DeviceSoA &dst, int hsteps, int asteps, float m_angle, bool eyes_white, const float *radius, const float *height,
                    const float *angle, const double *sin_a, const double *cos_a, hipStream_t s) {
    const size_t n = (size_t)hsteps * asteps;
    if (!n) return;
    SyntheticArgs a{hsteps, asteps, m_angle, eyes_white ? 1 : 0, radius, height, angle, sin_a, cos_a};
    CW_LAUNCH("synthetic_fill", synthetic_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, a, dst.x(), dst.y(), dst.z(), dst.rgbt(), n);
}

// simulated cameras, hard assignment (reference python/cwipc/filters/simulatecams.py:44-70): every point gets the tile mask
// 1 << c of the camera direction c (on a circle in the x-z plane) its centred, flattened position has the largest dot
// product with.  The arithmetic is numpy's: position minus centroid in float32, the dot product in float64 accumulated
// in index order the way numpy.dot (cblas_ddot on x86 with FMA) does for three terms -- fl(x cx) first, then one fused
// multiply-add for z cz, the y terms being zero -- and of equal dot products the camera with the HIGHER index wins
// (numpy.argsort's insertion sort is stable, and the reference takes the last of the ascending order).
struct CameraArgs {
    int ncam;
    float cen_x, cen_z;
    double dir[2 * 32];   // cos, sin per camera
};

__global__ void __launch_bounds__(BLOCK) camera_assign_kernel(CameraArgs a, const float *__restrict__ x, const float *__restrict__ z,
                                                             const uint32_t *__restrict__ rgbt, uint32_t *__restrict__ ow, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const double vx = (double)__fsub_rn(x[i], a.cen_x), vz = (double)__fsub_rn(z[i], a.cen_z);
        int best = 0;
        double best_d = 0;
        for (int c = 0; c < a.ncam; c++) {
            const double d = fma(vz, a.dir[2 * c + 1], __dmul_rn(vx, a.dir[2 * c]));
            if (c == 0 || d >= best_d) { best = c; best_d = d; }
        }
        // (the reference stores 1 << c into a float matrix column and casts it to uint8)
        ow[i] = (rgbt[i] & 0x00ffffffu) | (((1u << best) & 0xffu) << 24);
    }
}

void map_cameras(const DeviceSoA &src, const DeviceSoA &dst, int ncam, float cen_x, float cen_z, const double *dirs, hipStream_t s) {
    const size_t n = src.npoints;
    if (!n) return;
    CameraArgs a;
    a.ncam = ncam;
    a.cen_x = cen_x;
    a.cen_z = cen_z;
    for (int i = 0; i < 2 * ncam; i++) a.dir[i] = dirs[i];
    CW_LAUNCH("map_cameras", camera_assign_kernel, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, a, src.x(), src.z(), src.rgbt(), dst.rgbt(), n);
}

// which tile values occur: 256-bit set (8 words)
__global__ void __launch_bounds__(BLOCK) tiles_used_kernel(const uint32_t *__restrict__ rgbt, size_t n, uint32_t *__restrict__ bits) {
    __shared__ uint32_t local[8];
    if (threadIdx.x < 8) local[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const uint32_t t = rgbt[i] >> 24;
#pragma unroll
        for (int w = 0; w < 8; w++) mine[w] |= (t >> 5) == (uint32_t)w ? 1u << (t & 31u) : 0u;
    }
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t v = mine[w];
        for (int off = 32; off > 0; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicOr(&local[w], v);
    }
    __syncthreads();
    if (threadIdx.x < 8 && local[threadIdx.x]) atomicOr(&bits[threadIdx.x], local[threadIdx.x]);
}

void tiles_used(const DeviceSoA &src, uint32_t *dev_bits8, hipStream_t s) {
    const size_t n = src.npoints;
    if (!n) return;
    CW_LAUNCH("tiles_used", tiles_used_kernel, dim3(grid_for(n, BLOCK) > 1024 ? 1024 : grid_for(n, BLOCK)), dim3(BLOCK), 0, s, src.rgbt(), n, dev_bits8);
}

// first[t] = index of the first point with tile value t (0xffffffff: none): what the per-tile outlier filter needs to
// visit the tiles in first-appearance order (reference src/cwipc_filters.cpp:238-250) without the tile plane leaving the device.
// The plain read in front of the LDS atomic filters all but the first few touches of a tile per workgroup.
__global__ void __launch_bounds__(BLOCK) tile_first_kernel(const uint32_t *__restrict__ rgbt, size_t n, uint32_t *__restrict__ first) {
    __shared__ uint32_t local[256];
    for (int t = threadIdx.x; t < 256; t += BLOCK) local[t] = 0xffffffffu;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const uint32_t t = rgbt[i] >> 24;
        if (((volatile uint32_t *)local)[t] > (uint32_t)i) atomicMin(&local[t], (uint32_t)i);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 256; t += BLOCK)
        if (local[t] != 0xffffffffu) atomicMin(&first[t], local[t]);
}

void tile_first_index(const DeviceSoA &src, uint32_t *dev_first256, hipStream_t s) {
    const size_t n = src.npoints;
    (void)hipMemsetAsync(dev_first256, 0xff, 256 * sizeof(uint32_t), s);
    if (!n) return;
    CW_LAUNCH("tile_first_index", tile_first_kernel, dim3(grid_for(n, BLOCK) > 1024 ? 1024 : grid_for(n, BLOCK)), dim3(BLOCK), 0, s, src.rgbt(), n, dev_first256);
}

// ---------------------------------------------------------------------------
// join: plane-wise concatenation
// ---------------------------------------------------------------------------
// blockIdx.y selects the plane.  Planes start on 1 KiB boundaries, so a part whose destination offset is a multiple
// of four points (the usual case: clouds of any size first, and every count that is a multiple of 4 after) can
// move 16 bytes per lane; the general form moves 4.
template <bool VEC>
__global__ void __launch_bounds__(BLOCK) join_copy_kernel(JoinPart part, float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz,
                                                         uint32_t *__restrict__ ow) {
    const uint32_t *src;
    uint32_t *dst;
    switch (blockIdx.y) {
    case 0: src = (const uint32_t *)part.x; dst = (uint32_t *)ox; break;
    case 1: src = (const uint32_t *)part.y; dst = (uint32_t *)oy; break;
    case 2: src = (const uint32_t *)part.z; dst = (uint32_t *)oz; break;
    default: src = part.rgbt; dst = ow; break;
    }
    dst += part.dst_offset;
    const size_t stride = (size_t)gridDim.x * BLOCK;
    if (VEC) {
        const size_t n4 = part.n / 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) d4[i] = s4[i];
        const size_t tail = n4 * 4 + threadIdx.x;
        if (blockIdx.x == 0 && tail < part.n) dst[tail] = src[tail];   // up to three points
    } else {
        for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < part.n; i += stride) dst[i] = src[i];
    }
}

void join_copy(const JoinPart &part, const DeviceSoA &dst, hipStream_t s) {
    if (!part.n) return;
    if (part.dst_offset % 4 == 0) {   // source planes always start aligned; the destination does when the points before it come in fours
        const unsigned gx = grid_for(part.n / 4 + 1, BLOCK * 2);
        CW_LAUNCH("join_copy", join_copy_kernel<true>, dim3(gx, 4), dim3(BLOCK), 0, s, part, dst.x(), dst.y(), dst.z(), dst.rgbt());
    } else {
        const unsigned gx = grid_for(part.n, BLOCK * 4);
        CW_LAUNCH("join_copy", join_copy_kernel<false>, dim3(gx, 4), dim3(BLOCK), 0, s, part, dst.x(), dst.y(), dst.z(), dst.rgbt());
    }
}

}  // namespace k
}  // namespace cwipc_amd
