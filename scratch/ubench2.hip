#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t xorshift(uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// persistent streaming: grid = nblocks, block = T threads; each WAVE owns a contiguous range
template <int T>
__global__ void __launch_bounds__(T) stream_persist(const float* __restrict__ base, size_t stride, size_t n, float* out) {
    const int nwaves = gridDim.x * (T / 64);
    const int wave = blockIdx.x * (T / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    size_t per = ((n + nwaves - 1) / nwaves + 255) / 256 * 256;
    size_t lo = (size_t)wave * per, hi = lo + per < n ? lo + per : n;
    const float4* x = (const float4*)base; const float4* y = (const float4*)(base + stride); const float4* z = (const float4*)(base + 2 * stride); const uint4* w = (const uint4*)(base + 3 * stride);
    float acc = 0;
    for (size_t p = lo + lane * 4; p < hi; p += 512) {
        size_t i = p >> 2, i2 = (p + 256) >> 2;
        float4 a = x[i], b = y[i], c = z[i]; uint4 d = w[i];
        float4 a2 = a, b2 = b, c2 = c; uint4 d2 = d;
        if (p + 256 < hi) { a2 = x[i2]; b2 = y[i2]; c2 = z[i2]; d2 = w[i2]; }
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + __uint_as_float(d.x ^ d.y ^ d.z ^ d.w);
        acc += a2.x + a2.y + a2.z + a2.w + b2.x + b2.y + b2.z + b2.w + c2.x + c2.y + c2.z + c2.w + __uint_as_float(d2.x ^ d2.y ^ d2.z ^ d2.w);
    }
    if (acc == 1.2345f) out[0] = acc;
}

// record flush: G lanes cooperate on one 64-byte record (G u64 words), returning or not
template <int G, int RET>
__global__ void __launch_bounds__(256) record_flush(unsigned long long* table, uint32_t rec_mask, int per_thread) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (int i = 0; i < per_thread; i++) {
        s = xorshift(s);
        uint32_t lead = __shfl(s, (threadIdx.x & 63) & ~(G - 1), 64);
        size_t word = (size_t)(lead & rec_mask) * 8 + (threadIdx.x & (G - 1));
        if (RET) acc += atomicAdd(&table[word], 1ull); else atomicAdd(&table[word], 1ull);
    }
    if (acc == 0x1234567) table[0] = acc;
}

// LDS atomics with a sparse exec mask: only lanes with (lane % sparsity == 0) act
template <int WIDE>
__global__ void __launch_bounds__(256) lds_sparse(int iters, int sparsity, unsigned long long* out) {
    __shared__ unsigned long long tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 777u;
    bool active = (threadIdx.x % sparsity) == 0;
    for (int i = 0; i < iters; i++) {
        s = xorshift(s);
        if (active) { if (WIDE) atomicAdd(&tab[s & 1023], 1ull); else atomicAdd((uint32_t*)&tab[s & 1023], 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[5] == 0x7fffffffffull) out[0] = tab[5];
}
// plain LDS read-modify-write (no atomic) for comparison
__global__ void __launch_bounds__(256) lds_rmw(int iters, int sparsity, unsigned long long* out) {
    __shared__ unsigned long long tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 777u;
    bool active = (threadIdx.x % sparsity) == 0;
    for (int i = 0; i < iters; i++) {
        s = xorshift(s);
        if (active) { volatile unsigned long long* p = &tab[s & 1023]; *p = *p + 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[5] == 0x7fffffffffull) out[0] = tab[5];
}

int main() {
    const size_t n = 9998244; const size_t stride = (n + 63) / 64 * 64;
    const int NC = 4; float* bufs[NC];
    for (int c = 0; c < NC; c++) { CK(hipMalloc(&bufs[c], stride * 16)); CK(hipMemset(bufs[c], 1, stride * 16)); }
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, int reps, double q, const char* unit) {
        for (int i = 0; i < 3; i++) launch(i);
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int i = 0; i < reps; i++) { CK(hipEventRecord(e0)); launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
        printf("%-52s avg %8.2f us  best %8.2f us   %9.2f %s\n", name, tot / reps * 1e3, best * 1e3, q / (tot / reps * 1e-3) / 1e9, unit);
    };
    double bytes = (double)n * 16;
    for (int grid : {256, 512, 1024}) {
        char nm[80];
        snprintf(nm, 80, "stream_persist<1024> grid=%d", grid);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(stream_persist<1024>, dim3(grid), dim3(1024), 0, 0, bufs[i % NC], stride, n, out); }, 20, bytes, "GB/s");
        snprintf(nm, 80, "stream_persist<512> grid=%d", grid);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(stream_persist<512>, dim3(grid), dim3(512), 0, 0, bufs[i % NC], stride, n, out); }, 20, bytes, "GB/s");
        snprintf(nm, 80, "stream_persist<256> grid=%d", grid * 4);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(stream_persist<256>, dim3(grid * 4), dim3(256), 0, 0, bufs[i % NC], stride, n, out); }, 20, bytes, "GB/s");
    }
    unsigned long long* table; size_t recs = (size_t)1 << 22; CK(hipMalloc(&table, recs * 64)); CK(hipMemset(table, 0, recs * 64));
    const int blocks = 1024, per_thread = 8; 
    for (uint32_t lg : {14u, 22u}) {
        uint32_t mask = (1u << lg) - 1; char nm[96];
        double recs_done;
        recs_done = (double)blocks * 256 * per_thread / 8; snprintf(nm, 96, "record flush 8 lanes/rec  no-ret  recs=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL((record_flush<8, 0>), dim3(blocks), dim3(256), 0, 0, table, mask, per_thread); }, 10, recs_done, "Grec/s");
        snprintf(nm, 96, "record flush 8 lanes/rec  return  recs=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL((record_flush<8, 1>), dim3(blocks), dim3(256), 0, 0, table, mask, per_thread); }, 10, recs_done, "Grec/s");
        recs_done = (double)blocks * 256 * per_thread / 4; snprintf(nm, 96, "record flush 4 lanes/rec  no-ret  recs=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL((record_flush<4, 0>), dim3(blocks), dim3(256), 0, 0, table, mask, per_thread); }, 10, recs_done, "Grec/s");
        recs_done = (double)blocks * 256 * per_thread / 1; snprintf(nm, 96, "record flush 1 lane/rec   no-ret  recs=2^%u", lg);
        timeit(nm, [&](int) { hipLaunchKernelGGL((record_flush<1, 0>), dim3(blocks), dim3(256), 0, 0, table, mask, per_thread); }, 10, recs_done, "Grec/s");
    }
    unsigned long long* o2; CK(hipMalloc(&o2, 64));
    const int lblocks = 2048, iters = 64;
    for (int sp : {1, 2, 4, 8, 16}) {
        char nm[96]; double instrs = (double)lblocks * 4 * iters;   // wave-instructions
        snprintf(nm, 96, "LDS atomic u64, 1 of %d lanes active", sp);
        timeit(nm, [&](int) { hipLaunchKernelGGL(lds_sparse<1>, dim3(lblocks), dim3(256), 0, 0, iters, sp, o2); }, 10, instrs, "Gwave-instr/s");
        snprintf(nm, 96, "LDS atomic u32, 1 of %d lanes active", sp);
        timeit(nm, [&](int) { hipLaunchKernelGGL(lds_sparse<0>, dim3(lblocks), dim3(256), 0, 0, iters, sp, o2); }, 10, instrs, "Gwave-instr/s");
        snprintf(nm, 96, "LDS plain rmw u64, 1 of %d lanes active", sp);
        timeit(nm, [&](int) { hipLaunchKernelGGL(lds_rmw, dim3(lblocks), dim3(256), 0, 0, iters, sp, o2); }, 10, instrs, "Gwave-instr/s");
    }
    return 0;
}
