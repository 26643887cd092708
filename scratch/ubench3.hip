// Load-pattern microbenchmark: persistent grid (CUs x 1024 threads), each wave a contiguous range of four
// 4-byte planes.  A: one 16-byte load per lane and plane per 256 points (lane l <- points 4l..4l+3).
// B: two 16-byte loads per lane and plane per 512 points (lane l <- points 8l..8l+7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) stream(const float* __restrict__ base, size_t stride, size_t n, float* out) {
    const int nwaves = gridDim.x * 16;
    const int wave = blockIdx.x * 16 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t per = ((n + nwaves - 1) / nwaves + 511) / 512 * 512;
    const size_t lo = (size_t)wave * per, hi = lo + per < n ? lo + per : n;
    const float4* x = (const float4*)(base + lo); const float4* y = (const float4*)(base + stride + lo);
    const float4* z = (const float4*)(base + 2 * stride + lo); const uint4* w = (const uint4*)(base + 3 * stride + lo);
    const int npts = lo < hi ? (int)(hi - lo) : 0;
    float acc = 0;
    if (MODE == 0) {
        float4 a = x[lane], b = y[lane], c = z[lane]; uint4 d = w[lane];
        for (int off = 0; off < npts; off += 256) {
            float4 na = a, nb = b, nc = c; uint4 nd = d;
            if (off + 256 < npts) { const int v = ((off + 256) >> 2) + lane; na = x[v]; nb = y[v]; nc = z[v]; nd = w[v]; }
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + __uint_as_float(d.x ^ d.y ^ d.z ^ d.w);
            a = na; b = nb; c = nc; d = nd;
        }
    } else {
        float4 a0 = x[2 * lane], a1 = x[2 * lane + 1], b0 = y[2 * lane], b1 = y[2 * lane + 1], c0 = z[2 * lane], c1 = z[2 * lane + 1];
        uint4 d0 = w[2 * lane], d1 = w[2 * lane + 1];
        for (int off = 0; off < npts; off += 512) {
            float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1, nc0 = c0, nc1 = c1; uint4 nd0 = d0, nd1 = d1;
            if (off + 512 < npts) {
                const int v = ((off + 512) >> 2) + 2 * lane;
                na0 = x[v]; na1 = x[v + 1]; nb0 = y[v]; nb1 = y[v + 1]; nc0 = z[v]; nc1 = z[v + 1]; nd0 = w[v]; nd1 = w[v + 1];
            }
            acc += a0.x + a0.y + a0.z + a0.w + b0.x + b0.y + b0.z + b0.w + c0.x + c0.y + c0.z + c0.w + __uint_as_float(d0.x ^ d0.y ^ d0.z ^ d0.w);
            acc += a1.x + a1.y + a1.z + a1.w + b1.x + b1.y + b1.z + b1.w + c1.x + c1.y + c1.z + c1.w + __uint_as_float(d1.x ^ d1.y ^ d1.z ^ d1.w);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; c0 = nc0; c1 = nc1; d0 = nd0; d1 = nd1;
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    const size_t n = 9998244, stride = (n + 511) / 512 * 512 + 512;
    const int copies = 4;
    float* buf[copies]; float* out;
    for (int i = 0; i < copies; i++) { CK(hipMalloc(&buf[i], stride * 16)); CK(hipMemset(buf[i], 1, stride * 16)); }
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; i++) {
                if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(256), dim3(1024), 0, 0, buf[i % copies], stride, n, out);
                else hipLaunchKernelGGL(stream<1>, dim3(256), dim3(1024), 0, 0, buf[i % copies], stride, n, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d: %.1f us per pass, %.2f TB/s\n", mode, ms / 20 * 1000, n * 16.0 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
