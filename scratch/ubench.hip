// Microbenchmarks that decide the structure of the voxel accumulate kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) stream_read4(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ z, const uint4* __restrict__ w, size_t nvec, float* out) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float4 a = x[i], b = y[i], c = z[i]; uint4 d = w[i];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + __uint_as_float(d.x ^ d.y ^ d.z ^ d.w);
    }
    if (acc == 1.2345f) out[0] = acc;
}
// chunked: block b reads a contiguous chunk (like the accumulate kernel), UNROLL loads in flight
template <int STEPS>
__global__ void __launch_bounds__(256) stream_chunk(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ z, const uint4* __restrict__ w, size_t nvec, float* out) {
    float acc = 0;
    size_t base = (size_t)blockIdx.x * 256 * STEPS + threadIdx.x;
    float4 a[STEPS], b[STEPS], c[STEPS]; uint4 d[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) { size_t i = base + s * 256; if (i < nvec) { a[s] = x[i]; b[s] = y[i]; c[s] = z[i]; d[s] = w[i]; } else { a[s] = b[s] = c[s] = make_float4(0,0,0,0); d[s] = make_uint4(0,0,0,0);} }
#pragma unroll
    for (int s = 0; s < STEPS; s++) acc += a[s].x + a[s].y + a[s].z + a[s].w + b[s].x + b[s].y + b[s].z + b[s].w + c[s].x + c[s].y + c[s].z + c[s].w + __uint_as_float(d[s].x ^ d[s].y ^ d[s].z ^ d[s].w);
    if (acc == 1.2345f) out[0] = acc;
}

__device__ __forceinline__ uint32_t xorshift(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// mode 0: device-scope atomicAdd u64 no return; 1: workgroup-scope (L2-local) ; 2: device-scope u32; 3: CAS returning u64; 4: wg-scope u32
template <int MODE>
__global__ void __launch_bounds__(256) atomic_bench(unsigned long long* table, uint32_t mask, int per_thread, int group) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (int i = 0; i < per_thread; i++) {
        s = xorshift(s);
        // `group` consecutive lanes hit consecutive slots (record-like access)
        uint32_t slot = ((s & ~(uint32_t)(group - 1)) + (threadIdx.x & (group - 1))) & mask;
        if (group > 1) { uint32_t lead = __shfl(s, (threadIdx.x & 63) & ~(group - 1), 64); slot = ((lead * (uint32_t)group) + (threadIdx.x & (group - 1))) & mask; }
        if (MODE == 0) atomicAdd(&table[slot], 1ull);
        if (MODE == 1) __hip_atomic_fetch_add(&table[slot], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) atomicAdd((uint32_t*)&table[slot], 1u);
        if (MODE == 3) acc += atomicCAS(&table[slot], 0ull, (unsigned long long)s);
        if (MODE == 4) __hip_atomic_fetch_add((uint32_t*)&table[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc == 0x1234567) table[0] = acc;
}

// LDS atomics: each lane does `iters` ds atomics into a 1024-slot table; conflict = lanes per identical slot
template <int WIDE>
__global__ void __launch_bounds__(256) lds_atomic_bench(int iters, int conflict, unsigned long long* out) {
    __shared__ unsigned long long tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 777u;
    for (int i = 0; i < iters; i++) {
        s = xorshift(s);
        uint32_t lead = __shfl(s, (threadIdx.x & 63) / conflict * conflict, 64);
        uint32_t slot = lead & 1023;
        if (WIDE) atomicAdd(&tab[slot], 1ull); else atomicAdd((uint32_t*)&tab[slot], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[5] == 0x7fffffffffull) out[0] = tab[5];
}

int main() {
    const size_t n = 9998244; const size_t stride = (n + 63) / 64 * 64; const size_t nvec = stride / 4;
    const int NC = 4;
    float* bufs[NC];
    for (int c = 0; c < NC; c++) { CK(hipMalloc(&bufs[c], stride * 16)); CK(hipMemset(bufs[c], 1, stride * 16)); }
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, int reps, double bytes_or_ops, const char* unit) {
        for (int i = 0; i < 3; i++) launch(i);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < reps; i++) { CK(hipEventRecord(e0)); launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
        printf("%-44s avg %8.2f us  best %8.2f us   %8.2f %s (avg)\n", name, tot / reps * 1e3, best * 1e3, bytes_or_ops / (tot / reps * 1e-3) / 1e9, unit);
    };
    double bytes = (double)stride * 16;
    auto planes = [&](int i, const float4*& x, const float4*& y, const float4*& z, const uint4*& w) { float* b = bufs[i % NC]; x = (const float4*)b; y = (const float4*)(b + stride); z = (const float4*)(b + 2 * stride); w = (const uint4*)(b + 3 * stride); };
    for (int grid : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, 64, "stream_read4 grid-stride grid=%d", grid);
        timeit(nm, [&](int i) { const float4 *x, *y, *z; const uint4* w; planes(i, x, y, z, w); hipLaunchKernelGGL(stream_read4, dim3(grid), dim3(256), 0, 0, x, y, z, w, nvec, out); }, 20, bytes, "GB/s");
    }
    timeit("stream_chunk<1> (1024 pts/block)", [&](int i) { const float4 *x, *y, *z; const uint4* w; planes(i, x, y, z, w); hipLaunchKernelGGL(stream_chunk<1>, dim3((nvec + 255) / 256), dim3(256), 0, 0, x, y, z, w, nvec, out); }, 20, bytes, "GB/s");
    timeit("stream_chunk<2> (2048 pts/block)", [&](int i) { const float4 *x, *y, *z; const uint4* w; planes(i, x, y, z, w); hipLaunchKernelGGL(stream_chunk<2>, dim3((nvec + 511) / 512), dim3(256), 0, 0, x, y, z, w, nvec, out); }, 20, bytes, "GB/s");
    timeit("stream_chunk<4> (4096 pts/block)", [&](int i) { const float4 *x, *y, *z; const uint4* w; planes(i, x, y, z, w); hipLaunchKernelGGL(stream_chunk<4>, dim3((nvec + 1023) / 1024), dim3(256), 0, 0, x, y, z, w, nvec, out); }, 20, bytes, "GB/s");

    // global atomics
    unsigned long long* table; size_t tslots = (size_t)1 << 25; CK(hipMalloc(&table, tslots * 8)); CK(hipMemset(table, 0, tslots * 8));
    const int blocks = 2048, per_thread = 8; double ops = (double)blocks * 256 * per_thread;
    for (uint32_t lg : {16u, 20u, 25u}) {
        uint32_t mask = (1u << lg) - 1;
        for (int group : {1, 4}) {
            char nm[96];
            snprintf(nm, 96, "atomic u64 device   slots=2^%u group=%d", lg, group);
            timeit(nm, [&](int) { hipLaunchKernelGGL(atomic_bench<0>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, group); }, 10, ops, "Gatom/s");
            snprintf(nm, 96, "atomic u64 wg-scope slots=2^%u group=%d", lg, group);
            timeit(nm, [&](int) { hipLaunchKernelGGL(atomic_bench<1>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, group); }, 10, ops, "Gatom/s");
        }
        char nm[96];
        snprintf(nm, 96, "atomic u32 device   slots=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL(atomic_bench<2>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, 1); }, 10, ops, "Gatom/s");
        snprintf(nm, 96, "atomic u32 wg-scope slots=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL(atomic_bench<4>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, 1); }, 10, ops, "Gatom/s");
        snprintf(nm, 96, "atomicCAS u64 device (returning) slots=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL(atomic_bench<3>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, 1); }, 10, ops, "Gatom/s");
    }
    // LDS atomics
    unsigned long long* o2; CK(hipMalloc(&o2, 64));
    const int lblocks = 2048, iters = 64; double lops = (double)lblocks * 256 * iters;
    for (int conflict : {1, 4, 16, 64}) {
        char nm[96];
        snprintf(nm, 96, "LDS atomic u32 conflict=%d", conflict);
        timeit(nm, [&](int) { hipLaunchKernelGGL(lds_atomic_bench<0>, dim3(lblocks), dim3(256), 0, 0, iters, conflict, o2); }, 10, lops, "Gatom/s");
        snprintf(nm, 96, "LDS atomic u64 conflict=%d", conflict);
        timeit(nm, [&](int) { hipLaunchKernelGGL(lds_atomic_bench<1>, dim3(lblocks), dim3(256), 0, 0, iters, conflict, o2); }, 10, lops, "Gatom/s");
    }
    return 0;
}
