// How fast is rocprim's single-pass select on this path's data (four planes, keep points of one tile)?  Compared with the
// library's count / scan / scatter compaction (79 us at 10 M points).  hipcc --offload-arch=gfx950 -O3 scratch/ubench_select.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/zip_iterator.hpp>
#include <cstdio>
#include <vector>
#include <cstdint>

struct KeepTile {
    uint32_t tile;
    __device__ bool operator()(const rocprim::tuple<float, float, float, uint32_t> &p) const { return (rocprim::get<3>(p) >> 24) == tile; }
};

int main() {
    const size_t n = 9998244;
    float *x, *y, *z, *ox, *oy, *oz;
    uint32_t *w, *ow;
    unsigned int *count;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&z, n * 4); hipMalloc(&w, n * 4);
    hipMalloc(&ox, n * 4); hipMalloc(&oy, n * 4); hipMalloc(&oz, n * 4); hipMalloc(&ow, n * 4);
    hipMalloc(&count, 4);
    std::vector<uint32_t> hw(n);
    for (size_t i = 0; i < n; i++) hw[i] = ((i / 3162) & 1 ? 1u : 2u) << 24 | (uint32_t)(i * 2654435761u & 0xffffff);
    hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4); hipMemset(z, 0, n * 4);
    auto in = rocprim::make_zip_iterator(rocprim::make_tuple(x, y, z, w));
    auto out = rocprim::make_zip_iterator(rocprim::make_tuple(ox, oy, oz, ow));
    size_t tmp_bytes = 0;
    rocprim::select(nullptr, tmp_bytes, in, out, count, n, KeepTile{1});
    void *tmp;
    hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) rocprim::select(tmp, tmp_bytes, in, out, count, n, KeepTile{1});
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; i++) rocprim::select(tmp, tmp_bytes, in, out, count, n, KeepTile{1});
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned int kept;
    hipMemcpy(&kept, count, 4, hipMemcpyDeviceToHost);
    printf("rocprim::select, %zu points, %u kept: %.1f us per call (%zu temp bytes)\n", n, kept, ms / reps * 1e3, tmp_bytes);
    return 0;
}
