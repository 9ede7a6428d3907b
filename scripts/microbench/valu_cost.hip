// valu_cost.hip -- issue cost of the VALU instructions the stage-parallel kernel is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o valu_cost valu_cost.hip && ./valu_cost
// One wave per workgroup; every wave runs REPS x 64 back-to-back instructions of one kind on 8 independent
// register chains (no dependency stalls) between two s_memtime reads.  Launched with 1, 2 and 4 waves per SIMD:
// cycles / instruction of ONE wave alone, and the SIMD's cycles per instruction when 4 waves share it
// (= the issue cost that binds af_flow_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REPS 256

#define BODY8(INS)                                                                                     \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                 : "v"(b), "v"(c)                                                                      \
                 : "vcc", "s4", "s5")

#define KERNEL32(NAME, INS)                                                                             \
    __global__ void __launch_bounds__(64) NAME(unsigned long long* out, uint32_t seed) {              \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = seed * 3u + 0x12345u, c = threadIdx.x | 1u;                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
        for (int r = 0; r < REPS; ++r) {                                                                \
            BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); \
        }                                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x7fffffffu) out[blockIdx.x] = 0;                 \
    }

#define KERNEL64(NAME, INS)                                                                             \
    __global__ void __launch_bounds__(64) NAME(unsigned long long* out, uint32_t seed) {              \
        double a0 = threadIdx.x + seed + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        double b = 1.0000001 + seed * 1e-9, c = 0.9999999;                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
        for (int r = 0; r < REPS; ++r) {                                                                \
            BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); \
        }                                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.123) out[blockIdx.x] = 0;                        \
    }

// 32-bit chains
#define I_ADD_U32(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_MUL_U24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define I_MAD_U24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define I_MUL_LO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MUL_HI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CNDMASK_E64(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[4:5]\n"
#define I_ADD_CNDMASK(k) "v_add_u32 %" #k ", %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_CMP_CNDMASK(k) "v_cmp_lt_u32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_CMP_CNDMASK_S(k) "v_cmp_lt_u32 s[4:5], %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, s[4:5]\n"
#define I_CMP_ADDC(k) "v_cmp_lt_u32 vcc, %" #k ", %8\nv_addc_co_u32 %" #k ", vcc, %" #k ", %9, vcc\n"
#define I_CMP_U32(k) "v_cmp_lt_u32 vcc, %" #k ", %8\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHL_ADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 3, %8\n"
#define I_DPP(k) "v_add_u32_dpp %" #k ", %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_READLANE(k) "v_readlane_b32 s4, %" #k ", 3\n"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %8, %" #k "\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 9\n"
#define I_CVT_F32(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define I_ADD_F32(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_FMA_F32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
// 64-bit chains
#define I_ADD_F64(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define I_MUL_F64(k) "v_mul_f64 %" #k ", %" #k ", %9\n"
#define I_FMA_F64(k) "v_fma_f64 %" #k ", %" #k ", %9, %8\n"
#define I_MIN_F64(k) "v_min_f64 %" #k ", %" #k ", %8\n"
#define I_CMP_F64(k) "v_cmp_lt_f64 vcc, %" #k ", %8\n"
#define I_MOV_B64(k) "v_mov_b64 %" #k ", %8\n"
#define I_LSHL_ADD_U64(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %8\n"
#define I_CVT_U32_F64(k) "v_cvt_u32_f64 %" #k ", %" #k "\n"
#define I_RCP_F64(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define I_LDEXP_F64(k) "v_ldexp_f64 %" #k ", %" #k ", 1\n"
#define I_CMP_U64(k) "v_cmp_lt_u64 vcc, %" #k ", %8\n"
#define I_PK_ADD_F32(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n"
#define I_PK_MUL_F32(k) "v_pk_mul_f32 %" #k ", %" #k ", %9\n"

KERNEL32(k_add_u32, I_ADD_U32)
KERNEL32(k_xor, I_XOR)
KERNEL32(k_mov, I_MOV)
KERNEL32(k_mul_u24, I_MUL_U24)
KERNEL32(k_mad_u24, I_MAD_U24)
KERNEL32(k_mul_lo, I_MUL_LO)
KERNEL32(k_mul_hi, I_MUL_HI)
KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_cndmask_e64, I_CNDMASK_E64)
KERNEL32(k_add_cndmask, I_ADD_CNDMASK)
KERNEL32(k_cmp_cndmask, I_CMP_CNDMASK)
KERNEL32(k_cmp_cndmask_s, I_CMP_CNDMASK_S)
KERNEL32(k_cmp_addc, I_CMP_ADDC)
KERNEL32(k_cmp_u32, I_CMP_U32)
KERNEL32(k_add3, I_ADD3)
KERNEL32(k_lshl_add, I_LSHL_ADD)
KERNEL32(k_dpp, I_DPP)
KERNEL32(k_readlane, I_READLANE)
KERNEL32(k_mbcnt, I_MBCNT)
KERNEL32(k_bfe, I_BFE)
KERNEL32(k_cvt_f32, I_CVT_F32)
KERNEL32(k_add_f32, I_ADD_F32)
KERNEL32(k_fma_f32, I_FMA_F32)
KERNEL64(k_add_f64, I_ADD_F64)
KERNEL64(k_mul_f64, I_MUL_F64)
KERNEL64(k_fma_f64, I_FMA_F64)
KERNEL64(k_min_f64, I_MIN_F64)
KERNEL64(k_cmp_f64, I_CMP_F64)
KERNEL64(k_mov_b64, I_MOV_B64)
KERNEL64(k_lshl_add_u64, I_LSHL_ADD_U64)
KERNEL64(k_rcp_f64, I_RCP_F64)
KERNEL64(k_ldexp_f64, I_LDEXP_F64)
// v_mad_u64_u32 (the Philox round's 32 x 32 -> 64 multiply): written in C, the compiler emits the instruction
__global__ void __launch_bounds__(64) k_mad_u64(unsigned long long* out, uint32_t seed) {
    unsigned long long a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + seed + k;
    const uint32_t m = 0xD2511F53u + seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = (unsigned long long)(uint32_t)a[k] * m + a[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]) == 0x7fffffffull) out[blockIdx.x] = 0;
}
KERNEL64(k_cmp_u64, I_CMP_U64)
KERNEL64(k_pk_add_f32, I_PK_ADD_F32)
KERNEL64(k_pk_mul_f32, I_PK_MUL_F32)

// LDS: broadcast read (all lanes one address) and per-lane read, 8 in flight
__global__ void __launch_bounds__(64) k_ds_read_b64(unsigned long long* out, uint32_t seed) {
    __shared__ double buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = i + seed;
    __syncthreads();
    double acc = 0.0;
    const uint32_t base = (threadIdx.x * 8u) & 1023u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS * 8; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += buf[(base + u * 64 + r) & 1023u];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0.123) out[blockIdx.x] = 0;
}

struct K { const char* name; void (*fn)(unsigned long long*, uint32_t); double per_iter; };

int main() {
    std::vector<K> ks = {
        {"v_add_u32", k_add_u32, 64}, {"v_xor_b32", k_xor, 64}, {"v_mov_b32", k_mov, 64}, {"v_mul_u32_u24", k_mul_u24, 64},
        {"v_mad_u32_u24", k_mad_u24, 64}, {"v_mul_lo_u32", k_mul_lo, 64}, {"v_mul_hi_u32", k_mul_hi, 64}, {"v_cndmask_b32 (vcc)", k_cndmask, 64}, {"v_cndmask_b32 (sgpr pair)", k_cndmask_e64, 64},
        {"PAIR v_add_u32 + v_cndmask(vcc)", k_add_cndmask, 64}, {"PAIR v_cmp->vcc + v_cndmask(vcc)", k_cmp_cndmask, 64},
        {"PAIR v_cmp->s[4:5] + v_cndmask(s[4:5])", k_cmp_cndmask_s, 64}, {"PAIR v_cmp->vcc + v_addc(vcc)", k_cmp_addc, 64},
        {"v_cmp_lt_u32", k_cmp_u32, 64}, {"v_add3_u32", k_add3, 64}, {"v_lshl_add_u32", k_lshl_add, 64}, {"v_add_u32_dpp", k_dpp, 64},
        {"v_readlane_b32", k_readlane, 64}, {"v_mbcnt_lo", k_mbcnt, 64}, {"v_bfe_u32", k_bfe, 64}, {"v_cvt_f32_u32", k_cvt_f32, 64},
        {"v_add_f32", k_add_f32, 64}, {"v_fma_f32", k_fma_f32, 64},
        {"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_min_f64", k_min_f64, 64},
        {"v_cmp_lt_f64", k_cmp_f64, 64}, {"v_mov_b64", k_mov_b64, 64}, {"v_lshl_add_u64", k_lshl_add_u64, 64}, {"v_rcp_f64", k_rcp_f64, 64},
        {"v_ldexp_f64", k_ldexp_f64, 64}, {"v_mad_u64_u32", k_mad_u64, 64}, {"v_cmp_lt_u64", k_cmp_u64, 64},
        {"v_pk_add_f32", k_pk_add_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64},
        {"ds_read_b64+v_add_f64 (x8 per iteration x8)", k_ds_read_b64, 64},
    };
    unsigned long long* d = nullptr;
    const int max_waves = 256 * 4 * 4;
    hipMalloc(&d, max_waves * 8);
    std::vector<unsigned long long> h(max_waves);
    std::printf("%-46s %10s %10s %10s   (shader cycles per wave-instruction: one wave alone on its SIMD / SIMD cycles per instruction with 2 and 4 waves sharing it)\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (const K& k : ks) {
        double res[3];
        int idx = 0;
        for (int wps : {1, 2, 4}) {
            const int waves = 256 * 4 * wps;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.fn, dim3(waves), dim3(64), 0, 0, d, 7u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < waves; ++i) sum += (double)h[i];
            const double per_wave = sum / waves / (REPS * k.per_iter);     // cycles per instruction seen by a wave
            res[idx++] = per_wave / wps;                                   // SIMD cycles per instruction
        }
        std::printf("%-46s %10.2f %10.2f %10.2f\n", k.name, res[0], res[1], res[2]);
    }
    hipFree(d);
    return 0;
}
