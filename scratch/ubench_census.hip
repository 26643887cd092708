// Where do the workgroups of a 2-per-CU grid land (round 4)?  496 workgroups x 512 threads x 66 KB of LDS: every workgroup
// records its XCD, its HW_ID (shader engine / CU), its LDS allocation register and when it started, then stays for ~30 us so
// that all of them are resident together.  Printed: per CU the workgroups it held, whether exactly two, and whether the LDS
// base tells them apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(512, 4) census(uint32_t* out, int spin_us) {
    extern __shared__ uint32_t lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
        const uint32_t la = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);       // HW_REG_LDS_ALLOC
        out[blockIdx.x * 8 + 0] = xcc; out[blockIdx.x * 8 + 1] = hw; out[blockIdx.x * 8 + 2] = la;
        out[blockIdx.x * 8 + 3] = (uint32_t)t0; out[blockIdx.x * 8 + 4] = (uint32_t)(t0 >> 32);
    }
    lds[threadIdx.x] = threadIdx.x;
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 511] == 0xdeadbeefu) out[0] = 1;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 496, lds_bytes = argc > 2 ? atoi(argv[2]) : 66 * 1024;
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    uint32_t* d; CK(hipMalloc(&d, blocks * 32));
    std::vector<uint32_t> h(blocks * 8);
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(d, 0, blocks * 32));
        hipLaunchKernelGGL(census, dim3(blocks), dim3(512), lds_bytes, 0, d, 30);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, blocks * 32, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::vector<int>> by_cu;
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < blocks; b++) tmin = std::min(tmin, ((unsigned long long)h[b * 8 + 4] << 32) | h[b * 8 + 3]);
        for (int b = 0; b < blocks; b++) by_cu[(h[b * 8 + 0] << 16) | ((h[b * 8 + 1] >> 8) & 0xffu)].push_back(b);
        int hist[8] = {0}, told_apart = 0, pairs = 0;
        for (auto& kv : by_cu) {
            hist[std::min<size_t>(kv.second.size(), 7)]++;
            if (kv.second.size() == 2) { pairs++; if ((h[kv.second[0] * 8 + 2] & 0xfffu) != (h[kv.second[1] * 8 + 2] & 0xfffu)) told_apart++; }
        }
        printf("rep %d: %zu CUs used; CUs holding 1/2/3/4 workgroups: %d %d %d %d; pairs whose LDS base differs: %d of %d\n", rep, by_cu.size(), hist[1], hist[2], hist[3], hist[4], told_apart, pairs);
        if (rep == 1) {
            int shown = 0;
            for (auto& kv : by_cu) {
                if (shown++ >= 40) break;
                printf("  xcc %u se/sh/cu %02x:", kv.first >> 16, kv.first & 0xff);
                for (int b : kv.second) printf("  wg %3d (lds_alloc %08x, hw_id %08x, start +%.2f us)", b, h[b * 8 + 2], h[b * 8 + 1], (double)((((unsigned long long)h[b * 8 + 4] << 32) | h[b * 8 + 3]) - tmin) * 0.01);
                printf("\n");
            }
        }
    }
    return 0;
}
