// What a vector instruction of the accumulate kernel's step costs the SIMD (round 4): one 1024-thread workgroup per CU (four waves
// per SIMD, as in the kernel), every wave issues REPS x 64 instructions of one kind on eight independent registers, the kernel is
// timed with the 100 MHz counter and with clock64() (shader clock): cycles of the SIMD per wave-instruction =
// shader cycles / (instructions per wave x waves per SIMD).  The counters of the accumulate kernel (profiles/r04_k1_pmc.txt) say
// 506 VALU per wave step and a step of ~8160 cycles with four waves per SIMD: 4 cycles each would be all of it, 2 half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int REPS = 512;

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY64(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP)

#define KERNEL(NAME, OP)                                                                                              \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, unsigned long long *times, uint32_t seed) {          \
        uint32_t r[8];                                                                                                \
        float f[8];                                                                                                   \
        for (int i = 0; i < 8; i++) { r[i] = seed * (threadIdx.x + 1u) + i; f[i] = (float)(r[i] & 1023u) * 0.37f; } \
        uint32_t s = seed | 1u; float g = (float)seed * 0.001f + 1.0f;                                                \
        (void)s; (void)g;                                                                                             \
        const unsigned long long w0 = wall_clock64(), c0 = clock64();                                                 \
        for (int rep = 0; rep < REPS; rep++) { BODY64(OP) }                                                           \
        const unsigned long long c1 = clock64(), w1 = wall_clock64();                                                 \
        uint32_t acc = 0;                                                                                             \
        for (int i = 0; i < 8; i++) acc ^= r[i] ^ __float_as_uint(f[i]);                                              \
        if (acc == 0x12345u) out[0] = acc;                                                                            \
        if (threadIdx.x == 0) { times[blockIdx.x * 2] = c1 - c0; times[blockIdx.x * 2 + 1] = w1 - w0; }              \
    }

#define OP_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define OP_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define OP_MUL_F32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : );
#define OP_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define OP_ADD_DPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define OP_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define OP_CVT_FLR(i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(r[i]) : "v"(f[i]));
#define OP_FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(f[i]));
#define OP_LSHL_ADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(s));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define OP_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(s) : "vcc");
#define OP_CMP_F(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(g) : "vcc");
#define OP_MUL_U24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r[i]));
#define OP_SUB_F32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(s));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_LSHLREV(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
#define OP_MAX_F32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
#define OP_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define OP_CVT_F32_U32(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(r[i]));
#define OP_SNOP(i) asm volatile("s_nop 0");
#define OP_SADD(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");   // (writes SCC: without the clobber the loop never ends)

#define OP_MIX_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));   // 4 instructions
#define OP_MIX_ADD(i) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));        // 4 instructions
#define OP_CND_SGPR(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "s"(m64));
#define OP_CMP_CND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : "vcc");   // 2 instructions
#define OP_CMP_SGPR(i) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m64) : "v"(r[i]), "v"(s));
#define OP_MAX_U32(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[i]) : "v"(s));
#define OP_AND_OR(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define OP_MAD_U24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define OP_BPERMUTE(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(s));
#define OP_BPERMUTE8(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(r[i]) : "v"(s));
#define OP_READLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(r[i]));
#define OP_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(s));
#define OP_ADD_CO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(s) : "vcc");
#define OP_PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f2v[i]) : "v"(g2));
#define OP_LSHL_ADD_U64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q8));
#define OP_SWIZZLE(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]));
#define OP_LDS_ATOMIC(i) asm volatile("ds_add_u64 %0, %1" : : "v"(ldsaddr), "v"(q[i]) : "memory");
#define OP_LDS_ATOMIC32(i) asm volatile("ds_add_u32 %0, %1" : : "v"(ldsaddr), "v"(r[i]) : "memory");
#define OP_LDS_CAS(i) asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(ldsaddr), "v"(s) : "memory");

KERNEL(k_add_u32, OP_ADD_U32)
KERNEL(k_add_f32, OP_ADD_F32)
KERNEL(k_fma_f32, OP_FMA_F32)
KERNEL(k_mul_f32, OP_MUL_F32)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_mov_dpp, OP_MOV_DPP)
KERNEL(k_add_dpp, OP_ADD_DPP)
KERNEL(k_min3, OP_MIN3)
KERNEL(k_cvt_flr, OP_CVT_FLR)
KERNEL(k_fract, OP_FRACT)
KERNEL(k_lshl_add, OP_LSHL_ADD)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_perm, OP_PERM)
KERNEL(k_cmp_u32, OP_CMP)
KERNEL(k_cmp_f32, OP_CMP_F)
KERNEL(k_mul_u24, OP_MUL_U24)
KERNEL(k_and, OP_AND)
KERNEL(k_bfe, OP_BFE)
KERNEL(k_sub_f32, OP_SUB_F32)
KERNEL(k_mov, OP_MOV)
KERNEL(k_xor, OP_XOR)
KERNEL(k_lshlrev, OP_LSHLREV)
KERNEL(k_max_f32, OP_MAX_F32)
KERNEL(k_med3, OP_MED3)
KERNEL(k_cvt_f32_u32, OP_CVT_F32_U32)
KERNEL(k_s_nop, OP_SNOP)
KERNEL(k_s_add, OP_SADD)

#define KERNEL2(NAME, OP)                                                                                             \
    __global__ void __launch_bounds__(1024) NAME(uint32_t *out, unsigned long long *times, uint32_t seed) {          \
        __shared__ unsigned long long lds[2048];                                                                      \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                        \
        uint32_t r[8]; unsigned long long q[8]; f2 f2v[8];                                                            \
        for (int i = 0; i < 8; i++) { r[i] = seed * (threadIdx.x + 1u) + i; q[i] = r[i]; f2v[i] = f2{(float)r[i], 1.f}; } \
        for (int i = threadIdx.x; i < 2048; i += 1024) lds[i] = 0;                                                    \
        __syncthreads();                                                                                              \
        uint32_t s = (threadIdx.x * 4u) & 255u; unsigned long long m64 = seed * 0x9E3779B97F4A7C15ull, q8 = seed;     \
        f2 g2 = f2{1.0001f, 0.9999f};                                                                                 \
        const uint32_t ldsaddr = (uint32_t)(((threadIdx.x * 2654435761u) >> 21) * 8u);                                \
        (void)s; (void)m64; (void)q8; (void)g2; (void)ldsaddr;                                                        \
        const unsigned long long w0 = wall_clock64(), c0 = clock64();                                                 \
        for (int rep = 0; rep < REPS; rep++) { BODY64(OP) }                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                         \
        const unsigned long long c1 = clock64(), w1 = wall_clock64();                                                 \
        uint32_t acc = s ^ (uint32_t)m64;                                                                             \
        for (int i = 0; i < 8; i++) acc ^= r[i] ^ (uint32_t)q[i] ^ __float_as_uint(f2v[i].x + f2v[i].y);             \
        __syncthreads();                                                                                              \
        if (acc == 0x12345u) out[0] = acc + (uint32_t)lds[threadIdx.x];                                               \
        if (threadIdx.x == 0) { times[blockIdx.x * 2] = c1 - c0; times[blockIdx.x * 2 + 1] = w1 - w0; }              \
    }
KERNEL2(k_mix_cnd, OP_MIX_CND)
KERNEL2(k_mix_add, OP_MIX_ADD)
KERNEL2(k_cnd_sgpr, OP_CND_SGPR)
KERNEL2(k_cmp_cnd, OP_CMP_CND)
KERNEL2(k_cmp_sgpr, OP_CMP_SGPR)
KERNEL2(k_max_u32, OP_MAX_U32)
KERNEL2(k_bfi, OP_BFI)
KERNEL2(k_and_or, OP_AND_OR)
KERNEL2(k_mad_u24, OP_MAD_U24)
KERNEL2(k_bpermute, OP_BPERMUTE)
KERNEL2(k_bpermute8, OP_BPERMUTE8)
KERNEL2(k_readlane, OP_READLANE)
KERNEL2(k_mbcnt, OP_MBCNT)
KERNEL2(k_add_co, OP_ADD_CO)
KERNEL2(k_pk_mul, OP_PK_MUL)
KERNEL2(k_lshl_add_u64, OP_LSHL_ADD_U64)
KERNEL2(k_swizzle, OP_SWIZZLE)
KERNEL2(k_lds_add64, OP_LDS_ATOMIC)
KERNEL2(k_lds_add32, OP_LDS_ATOMIC32)
KERNEL2(k_lds_cas, OP_LDS_CAS)

// v_pk_add_f32 works on register pairs
__global__ void __launch_bounds__(1024) k_pk_add(uint32_t *out, unsigned long long *times, uint32_t seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 f[8];
    for (int i = 0; i < 8; i++) f[i] = f2{(float)((seed + i) & 1023u) * 0.37f, (float)threadIdx.x};
    f2 g = f2{(float)seed * 0.001f + 1.0f, 0.5f};
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
#define OP_PK(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(g));
    for (int rep = 0; rep < REPS; rep++) { BODY64(OP_PK) }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += f[i].x + f[i].y;
    if (acc == 12345.f) out[0] = 1;
    if (threadIdx.x == 0) { times[blockIdx.x * 2] = c1 - c0; times[blockIdx.x * 2 + 1] = w1 - w0; }
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 1024;
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount * (argc > 2 ? atoi(argv[2]) : 1);
    uint32_t *out; unsigned long long *times;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&times, blocks * 16));
    std::vector<unsigned long long> h(blocks * 2);
    Entry list[] = {{"v_add_u32", k_add_u32}, {"v_add_f32", k_add_f32}, {"v_sub_f32", k_sub_f32}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_pk_add_f32", k_pk_add},
                    {"v_max_f32", k_max_f32}, {"v_min3_f32", k_min3}, {"v_med3_f32", k_med3}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov}, {"v_mov_b32 dpp row_shr", k_mov_dpp},
                    {"v_add_u32 dpp row_shr", k_add_dpp}, {"v_cvt_flr_i32_f32", k_cvt_flr}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_fract_f32", k_fract},
                    {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_f32", k_cmp_f32},
                    {"v_mul_u32_u24", k_mul_u24}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_bfe_u32", k_bfe}, {"v_lshlrev_b32", k_lshlrev}, {"s_nop", k_s_nop}, {"s_add_u32", k_s_add},
                    {"4x v_add_u32 (x4)", k_mix_add}, {"cndmask + 3 add (x4)", k_mix_cnd}, {"v_cndmask_b32 sgpr mask", k_cnd_sgpr}, {"v_cmp + v_cndmask (x2)", k_cmp_cnd},
                    {"v_cmp_lt_u32 -> sgpr pair", k_cmp_sgpr}, {"v_max_u32", k_max_u32}, {"v_bfi_b32", k_bfi}, {"v_and_or_b32", k_and_or}, {"v_mad_u32_u24", k_mad_u24},
                    {"v_mbcnt_lo", k_mbcnt}, {"v_add_co_u32", k_add_co}, {"v_pk_mul_f32", k_pk_mul}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_readlane_b32", k_readlane},
                    {"ds_bpermute + wait", k_bpermute}, {"ds_bpermute (8 in flight)", k_bpermute8}, {"ds_swizzle + wait", k_swizzle},
                    {"ds_add_u64 (hashed addr)", k_lds_add64}, {"ds_add_u32 (hashed addr)", k_lds_add32}, {"ds_cmpst_rtn + wait", k_lds_cas}};
    const int waves_per_simd = threads / 64 / 4;
    printf("# %d workgroups (%d per CU) of %d threads (%d waves per SIMD and workgroup), %d x 64 instructions per wave; clock rate %d kHz\n", blocks, argc > 2 ? atoi(argv[2]) : 1, threads, waves_per_simd > 0 ? waves_per_simd : 1, REPS, prop.clockRate);
    printf("# %-24s %12s %12s %14s %14s\n", "instruction", "clk64 cyc", "us (100MHz)", "cyc/instr/wave", "cyc/instr/SIMD");
    for (auto &e : list) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, times, 12345u + rep);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), times, blocks * 16, hipMemcpyDeviceToHost));
        double c = 0, w = 0;
        for (int b = 0; b < blocks; b++) { c += (double)h[b * 2]; w += (double)h[b * 2 + 1]; }
        c /= blocks; w /= blocks;
        const double instrs = (double)REPS * 64.0;
        // shader cycles from the wall clock at the nominal clock rate
        const double cyc_wall = w * 0.01 * (double)prop.clockRate * 1e-3;
        const int wps = threads >= 256 ? threads / 256 : 1;
        printf("  %-24s %12.0f %12.2f %14.2f %14.2f\n", e.name, c, w * 0.01, cyc_wall / instrs, cyc_wall / (instrs * wps));
        fflush(stdout);
    }
    return 0;
}
