// valu_calibration.hip -- what is ONE count of SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES on gfx950?
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_calibration valu_calibration.hip
//   ./valu_calibration                                   (prints its own hipEvent timings and the instruction counts it KNOWS)
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
//             --kernel-trace --output-format csv -d out -o cal -- ./valu_calibration
//
// Kernels of KNOWN VALU wave-instruction count (inline asm, 8 independent register chains, REPS x 64 instructions per
// wave and nothing else in the loop but one s_sub / s_cbranch), launched at 1, 2, 4 and 8 waves per SIMD over all 1 024
// SIMDs, in two instruction mixes whose issue costs differ (v_fma_f64: 16 lanes per clock; v_add_u32: 32 or 64):
//   * SQ_INSTS_VALU per launch must equal waves x REPS x 64 (+ the handful outside the loop): the counter counts
//     wave-instructions, one per issue, whatever the lanes do;
//   * with >= 4 waves per SIMD of independent chains the VALU of every SIMD is busy for the whole launch BY
//     CONSTRUCTION, so SQ_ACTIVE_INST_VALU x k / (1 024 SIMDs x launch cycles) = 1 fixes k -- for both mixes if the
//     counter measures busy TIME (and not instructions);
//   * at 1 wave per SIMD the same kernel leaves the VALU idle between dependent issues: the fraction must fall.
// scripts/make_binding_json.py reads the resulting table (profiles/rNN/valu_calibration.json) and states the stage-parallel
// kernel's VALU busy fraction in units of a kernel that is VALU-bound by construction (VERDICT r4 "What's weak" 5).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define REPS 4096

#define BODY8(INS)                                                                                     \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                 : "v"(b), "v"(c))

#define I_FMA_F64(k) "v_fma_f64 %" #k ", %" #k ", %9, %8\n"
#define I_ADD_U32(k) "v_add_u32 %" #k ", %" #k ", %8\n"

__global__ void __launch_bounds__(64) cal_fma_f64(double* out, uint32_t seed) {
    double a0 = threadIdx.x + seed + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1e-9 * seed, c = 0.9999999;
    for (int r = 0; r < REPS; ++r) {
        BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.123) out[blockIdx.x] = 0;
}

__global__ void __launch_bounds__(64) cal_add_u32(double* out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed * 3u + 0x12345u, c = threadIdx.x | 1u;
    for (int r = 0; r < REPS; ++r) {
        BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32); BODY8(I_ADD_U32);
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x7fffffffu) out[blockIdx.x] = 0;
}

// half the lanes masked off: does the counter see lanes (it must not), does SQ_THREAD_CYCLES_VALU (it must)
__global__ void __launch_bounds__(64) cal_fma_f64_half_lanes(double* out, uint32_t seed) {
    double a0 = threadIdx.x + seed + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1e-9 * seed, c = 0.9999999;
    if (threadIdx.x & 1u) {
        for (int r = 0; r < REPS; ++r) {
            BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64); BODY8(I_FMA_F64);
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0.123) out[blockIdx.x] = 0;
}

struct K {
    const char* name;
    void (*fn)(double*, uint32_t);
};

int main() {
    const K ks[] = {{"cal_fma_f64", cal_fma_f64}, {"cal_add_u32", cal_add_u32}, {"cal_fma_f64_half_lanes", cal_fma_f64_half_lanes}};
    double* d = nullptr;
    const int n_simd = 256 * 4;
    hipMalloc(&d, n_simd * 8 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::printf("{\"reps\": %d, \"valu_wave_insts_per_wave_in_loop\": %d, \"launches\": [\n", REPS, REPS * 64);
    bool first = true;
    for (const K& k : ks) {
        for (int wps : {1, 2, 4, 8}) {
            const int waves = n_simd * wps;
            hipLaunchKernelGGL(k.fn, dim3(waves), dim3(64), 0, 0, d, 7u);   // warm (code object load, clocks)
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(waves), dim3(64), 0, 0, d, 7u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            std::printf("%s {\"kernel\": \"%s\", \"waves_per_simd\": %d, \"waves\": %d, \"ms\": %.4f, \"known_valu_wave_insts\": %.0f}", first ? "" : ",\n",
                        k.name, wps, waves, ms, (double)waves * REPS * 64.0);
            first = false;
        }
    }
    std::printf("\n]}\n");
    hipFree(d);
    return 0;
}
