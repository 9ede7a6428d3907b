// af_pregen.hpp -- the windowed arrival sampler with ONE LANE PER SCENARIO for the order-dependent part (round 3,
// af_arrival_groups in this file, launched by engine.hip).
//
// What it replaces: samplers/poisson_poisson.py:51-82 / gaussian_poisson.py:63-94 (the sampler's virtual clock, the
// once-a-window user draw, the exponential gaps) and rqs_generator.py:97-119 (arrival k = arrival k-1 + gap k on the
// simulation clock); af::gen_next_gap (af_core.hpp) is the sequential statement the oracle and the host builds follow.
//
// Why the work is split.  A gap is -log(1 - u_i) / lambda: the expensive part (Philox block, logarithm: ~110 instructions)
// is a pure function of the draw index i and can be worked out by any lane of any wave.  What is left is order-dependent --
// which window a draw falls in (lambda), the two running f64 sums -- and costs five f64 operations per draw: a lane walks
// ITS scenario's draws in order, eight per step, and does exactly the sequential sampler's additions, 64 scenarios per
// instruction.  The row formulation (af_pregen_arrivals_rows: 16 lanes per scenario, lane l re-adds gaps 0..l) has 4
// scenarios per instruction and spends ~100 instructions per 64 draws on the sums alone, this one 16.
//
// The division by lambda is the compiler's own IEEE sequence (v_div_scale x2, v_rcp_f64, 4 x v_fma refining the
// reciprocal, v_mul, v_fma, v_div_fmas, v_div_fixup) with the part that depends only on the divisor hoisted out of the
// loop: lambda changes once per window.  For operands that v_div_scale leaves alone (no scaling: `div_in_range`, and a
// numerator in [1e-15, 37]) and that v_div_fixup passes through (normal numbers) the three instructions left per draw
// compute the same bits; windows whose lambda is outside that range take the plain `/`.
//
// Device code under hipcc; plain C++ under g++ for tests/hostcheck (TEST-ONLY: the per-lane functions below against
// af::gen_next_gap, bit for bit).
#pragma once

#include <stdint.h>

#include "af_core.hpp"

namespace afp {

constexpr uint32_t kBatch = 8u;   // draws per lane and step
constexpr uint32_t kRound = 8u;   // steps per round: variates and arrival times change hands a round's worth at a time
constexpr uint32_t kChunk = kBatch * kRound;   // draws per lane and round

// -log(1 - u) of draw `idx` of the generator's stream: the gap's numerator (poisson_poisson.py:69-70, u clamped at 1e-15)
AF_HD double unit_variate(uint64_t seed, uint32_t idx) {
    const af::U4 r = af::draw_block(seed, af::STREAM_GENERATOR, idx, 0u);
    double u = af::u53(r.x, r.y);
    if (u < 1e-15) u = 1e-15;
    return -af::af_log_unit(1.0 - u);
}

// the divisor's half of the IEEE division sequence: two Newton steps on the hardware's reciprocal estimate `r0`
AF_HD double refine_rcp(double d, double r0) {
    const double e0 = __builtin_fma(-d, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-d, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
// ... and the numerator's: quotient estimate, remainder, correction (v_div_fmas without scaling is this fma)
AF_HD double div_by(double n, double d, double r2) {
    const double q = n * r2;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r2, q);
}
AF_HD bool div_in_range(double d) { return d >= 0x1p-64 && d <= 0x1p64; }
AF_HD double rcp_estimate(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);
#else
    return (double)(1.0f / (float)d);   // host builds (tests): an estimate good to 2^-24 like the hardware's is to ~2^-26
#endif
}

enum : uint32_t { LANE_WAIT = 0u, LANE_RUN = 1u, LANE_DONE = 2u };
constexpr uint32_t kFlagDrawOverflow = 1u << 6;   // AF_FLAG_DRAW_OVERFLOW (include/asyncflow_hip.h; engine.hip checks the value)

// One scenario's sampler.  WAIT: the next thing to do is a window start (or to find the horizon reached); RUN: inside a
// window, drawing; DONE: horizon reached or the arrival array full.
struct Lane {
    double g_now, g_wend, lam, rinv, t;   // sampler clock, window end, rate, refined 1 / rate, simulation clock
    uint32_t draws, k, flags, state;      // next draw index, arrivals so far
    bool fast_div;
};
AF_HD void lane_init(Lane& L) {
    L.g_now = L.g_wend = L.lam = L.rinv = L.t = 0.0;
    L.draws = L.k = L.flags = 0u;
    L.state = LANE_WAIT;
    L.fast_div = false;
}
// `while now < simulation_time: if now >= window_end: ...; if lam <= 0: now = window_end; continue`
// (poisson_poisson.py:55-66); `users(idx)` = the number of active users drawn at index idx
template <class UsersDraw>
AF_HD void window_start(Lane& L, double T, double window_s, double rps_per_user, UsersDraw&& users) {
    if (!(L.g_now < T)) {
        L.state = LANE_DONE;
        return;
    }
    L.g_wend = L.g_now + window_s;
    L.lam = users(L.draws++) * rps_per_user;
    if (L.lam <= 0.0) {   // nobody active: on to the next window (the lane stays in WAIT)
        L.g_now = L.g_wend;
        return;
    }
    L.fast_div = div_in_range(L.lam);
    L.rinv = L.fast_div ? refine_rcp(L.lam, rcp_estimate(L.lam)) : 0.0;
    L.state = LANE_RUN;
}
// One step of a running lane: e[j] = unit variate of draw index L.draws + j.  The sums are the sequential sampler's own
// (now += delta_t, env.now + gap): G[j] / S[j] = the two clocks after draw j.  Gaps are positive, so the clocks never
// decrease: when the LAST draw of the step neither crosses the window end nor the horizon, none does -- the usual step,
// eight arrivals and nothing else.  Otherwise the first draw that stops is looked for: beyond the horizon -> exhausted;
// across the window end -> discarded, the clock jumps to the window end (poisson_poisson.py:72-79).
template <bool FAST_DIV>
AF_HD void lane_sums(const Lane& L, const double (&e)[kBatch], double (&G)[kBatch], double (&S)[kBatch]) {
    double g = L.g_now, s = L.t;
#pragma unroll
    for (uint32_t j = 0u; j < kBatch; ++j) {
        const double dt = af::test_quant(FAST_DIV ? div_by(e[j], L.lam, L.rinv) : e[j] / L.lam);
        g += dt;
        s = s + dt;
        G[j] = g;
        S[j] = s;
    }
}
AF_HD bool lane_usual(const Lane& L, const double (&G)[kBatch], double T, uint32_t n_draw) {
    return (G[kBatch - 1u] <= T) & (G[kBatch - 1u] < L.g_wend) & (L.k + kBatch <= n_draw);
}
AF_HD void lane_commit(Lane& L, const double (&G)[kBatch], const double (&S)[kBatch]) {
    L.k += kBatch;
    L.draws += kBatch;
    L.g_now = G[kBatch - 1u];
    L.t = S[kBatch - 1u];
}
// the step that stops somewhere (or whose arrivals do not all fit); `stored`: S[0..7] is at out[k..k+8) already -- the
// kernel stores every step's eight sums where they fit; what lies behind the accepted ones is written again by a later
// step or by the +inf fill
AF_HD void lane_slow(Lane& L, const double (&G)[kBatch], const double (&S)[kBatch], double T, uint32_t n_draw, double* out,
                     bool stored) {
    uint32_t first = kBatch;
    bool over = false;
#pragma unroll
    for (uint32_t j = 0u; j < kBatch; ++j)
        if (first == kBatch && (G[j] > T || G[j] >= L.g_wend)) {
            first = j;
            over = G[j] > T;
        }
    uint32_t n_acc = first;
    bool full = false;
    if (L.k + n_acc > n_draw) {   // more arrivals than the array holds
        n_acc = n_draw - L.k;
        full = true;
    }
    if (!stored) {
        double* o = out + L.k;
#pragma unroll
        for (uint32_t j = 0u; j < kBatch; ++j)
            if (j < n_acc) o[j] = S[j];
    }
    L.k += n_acc;
    if (full) {
        L.flags = kFlagDrawOverflow;
        L.state = LANE_DONE;
    } else if (first == kBatch) {
        L.g_now = G[kBatch - 1u];
        L.t = S[kBatch - 1u];
        L.draws += kBatch;
    } else {
#pragma unroll
        for (uint32_t j = 0u; j + 1u < kBatch; ++j)
            if (j + 1u == first) L.t = S[j];
        L.draws += first + 1u;
        if (over) {
            L.state = LANE_DONE;
        } else {
            L.g_now = L.g_wend;
            L.state = LANE_WAIT;
        }
    }
}
// (host builds, and the statement of what a kernel step does)
template <bool FAST_DIV>
AF_HD void lane_step(Lane& L, const double (&e)[kBatch], double T, uint32_t n_draw, double* out) {
    double G[kBatch], S[kBatch];
    lane_sums<FAST_DIV>(L, e, G, S);
    if (lane_usual(L, G, T, n_draw)) {
        double* o = out + L.k;
#pragma unroll
        for (uint32_t j = 0u; j < kBatch; ++j) o[j] = S[j];
        lane_commit(L, G, S);
    } else {
        lane_slow(L, G, S, T, n_draw, out, false);
    }
}

// the number of active users of a window (poisson_poisson.py:60, gaussian_poisson.py:72-76 + common_helpers.py:32-33)
AF_HD double users_draw(uint32_t dist, double mean, double sigma, uint64_t seed, uint32_t idx) {
    if (dist == af::DIST_NORMAL) {
        const double v = mean + sigma * af::af_norminv(af::uniform_j(seed, af::STREAM_GENERATOR, idx, 0u));
        return v > 0.0 ? v : 0.0;
    }
    return (double)af::af_poisson(mean, seed, af::STREAM_GENERATOR, idx, 0u);
}

// what the kernel is launched with (engine.hip fills it from the engine's launch arguments)
struct ArrivalArgs {
    double total_time, users_mean, users_sigma, rpm, window_s;   // plan values; per-scenario columns override them
    uint32_t users_dist, n_scen, n_draw;
    uint32_t stride;   // doubles per scenario in `out`
    uint32_t group;    // scenarios per workgroup (<= 64)
    const uint64_t* seeds;
    const uint32_t* scen_map;   // slot j holds scenario scen_map[j] (null: j)
    uint32_t n_ovr;
    const uint32_t* ovr_param;
    const uint32_t* ovr_index;
    const double* ovr_values;
    uint32_t ovr_stride;
    double* out;           // [n_scen][stride]: arrival times, +inf behind the last
    uint32_t* pre_flags;   // [n_scen] AF_FLAG_DRAW_OVERFLOW
};

#if defined(__HIPCC__)
__device__ __forceinline__ double arrival_param(const ArrivalArgs& a, uint32_t param, uint32_t scen, double dflt) {
    for (uint32_t k = 0; k < a.n_ovr; ++k)
        if (a.ovr_param[k] == param && a.ovr_index[k] == 0u) dflt = a.ovr_values[(size_t)k * a.ovr_stride + scen];
    return dflt;
}
__device__ __noinline__ double users_draw_call(uint32_t dist, double mean, double sigma, uint64_t seed, uint32_t idx) {
    return users_draw(dist, mean, sigma, seed, idx);
}

// af_arrival_groups: a workgroup = up to 64 scenarios = one CHAIN wave (lane = scenario) + kProducers waves that work out
// unit variates for it and carry its arrival times to HBM.
//
// The chain wave is as long as ONE scenario's chain (~11 000 steps of eight draws, ~70 instructions each): it must not do
// anything else.  Two earlier forms of this kernel did -- a lane fetching its own variates from an HBM array (written by a
// separate, fully parallel kernel) and storing its own arrival times: every memory instruction of the wave touched 64 cache
// lines in 64 pages, 8.7 ms whatever the waits looked like (4.5 ms with all lanes reading ONE row: the address divergence
// was the cost); and the same with rows touched 256 contiguous bytes at a time and LDS doing the transposition: 8.3 ms, half
// of the wave's instructions were then data movement.  Here the variates never exist in HBM: inside a window all running
// lanes consume eight per step in lockstep, so while the chain wave sums round r (kRound steps) out of one LDS buffer the
// producer waves fill the other with the variates of round r + 1 -- draw indices d0[s] + 64 (r + 1) .. + 63 of every scenario
// s, d0 = where the scenario's window began -- and store the sums of round r - 1, which the chain wave left in LDS, to the
// scenarios' rows (a wave per row: 512 contiguous bytes).  One barrier per round.  The once-a-window user draw
// -- Poisson by chunked inversion, thousands of instructions -- is made by all lanes of the chain wave together: a lane that
// crossed its window end (or found nobody active) waits until every lane of the wave has.
//
// 10 000 scenarios = 250 workgroups of 40 (one per CU): the producers' ~110 instructions per variate are spread over the
// whole chip, the chain wave's round (~1 400 cycles) is what a round takes.
// (round 6: 15 producers -- four waves per SIMD, the workgroup's 1 024 threads -- instead of 11: the producers' Philox + logarithm
// are what a round takes, and a fourth wave per SIMD fills more of its issue slots.  Library variants on one box, 10 000 LB-2
// replicas, 7 / 11 / 15 producers: 6.25 / 5.15 / 4.97 ms at 40 scenarios per workgroup, - / 7.58 / 6.90 ms at 64:
// profiles/r06/ab_pregen_producers.txt)
#if !defined(AF_PREGEN_PRODUCERS)
#define AF_PREGEN_PRODUCERS 15   /* measurement builds: -DAF_PREGEN_PRODUCERS=n (<= 15: a workgroup is at most 16 waves) */
#endif
constexpr uint32_t kProducers = AF_PREGEN_PRODUCERS;
constexpr uint32_t kGroupThreads = 64u * (1u + kProducers);
constexpr uint32_t kIdle = 0xFFFFFFFFu;
typedef double d2_t __attribute__((ext_vector_type(2)));
struct GroupLds {
    double in[2][64][kChunk + 2u];     // [round parity][scenario]: the round's variates (+2: 16-byte reads spread over the banks)
    double sums[2][64][kChunk + 1u];   // [round parity][scenario]: the arrival times the round accepted (+1: banks)
    uint64_t seed[64];
    uint32_t d0[64];                   // draw index at which the scenario's current window began; kIdle: it does not run
    uint32_t sums_k[2][64], sums_n[2][64];   // where the round's arrival times go in the scenario's row, how many they are
    uint32_t k_final[64];
    uint32_t run[2];                   // [round parity] a lane still runs after the round
    uint32_t window[2];                // [window parity] 0: every scenario is done, 1: nobody runs in this window, 2: rounds follow
};
// (a wave's 64 variates of one pass belong to ONE scenario -- kChunk = 64 -- so seed and window start are scalars and the
// Philox key schedule is scalar work beside the vector instructions)
static_assert(kChunk == 64u, "group_produce / group_store: one wave, one scenario per pass");
__device__ __forceinline__ void group_produce(GroupLds& M, uint32_t buf, uint32_t round, uint32_t n_here, uint32_t ptid) {
    const uint32_t j = ptid & 63u;
    for (uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ptid >> 6)); s < n_here; s += kProducers) {
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.d0[s]);
        if (d0 == kIdle) continue;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)M.seed[s]);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(M.seed[s] >> 32));
        M.in[buf][s][j] = unit_variate(((uint64_t)hi << 32) | lo, d0 + round * kChunk + j);
    }
}
__device__ __forceinline__ void group_store(const GroupLds& M, uint32_t buf, double* rows, uint32_t stride, uint32_t n_here,
                                            uint32_t ptid) {
    const uint32_t j = ptid & 63u;
    for (uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ptid >> 6)); s < n_here; s += kProducers) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.sums_n[buf][s]);
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.sums_k[buf][s]);
        if (j < n) rows[(size_t)s * stride + k + j] = M.sums[buf][s][j];
    }
}
template <bool FAST_DIV>
__device__ __forceinline__ void group_chain_round(Lane& L, GroupLds& M, uint32_t buf, uint32_t lane, double T, uint32_t n_draw) {
    const uint32_t k0 = L.k;
    // (the variates of step st + 1 are asked for before step st is summed: an LDS latency per step otherwise)
    d2_t cur[kBatch / 2u], nxt[kBatch / 2u];
#pragma unroll
    for (uint32_t j = 0u; j < kBatch; j += 2u) cur[j / 2u] = *(const d2_t*)&M.in[buf][lane][j];
#pragma unroll
    for (uint32_t st = 0u; st < kRound; ++st) {
        if (st + 1u < kRound) {
#pragma unroll
            for (uint32_t j = 0u; j < kBatch; j += 2u) nxt[j / 2u] = *(const d2_t*)&M.in[buf][lane][(st + 1u) * kBatch + j];
        }
        double e[kBatch], G[kBatch], S[kBatch];
#pragma unroll
        for (uint32_t j = 0u; j < kBatch; j += 2u) {
            e[j] = cur[j / 2u].x;
            e[j + 1u] = cur[j / 2u].y;
        }
        lane_sums<FAST_DIV>(L, e, G, S);
        if (L.state == LANE_RUN) {
            // (all eight sums go to the line; those behind the accepted ones are written over by the next step's)
            double* o = &M.sums[buf][lane][L.k - k0];
#pragma unroll
            for (uint32_t j = 0u; j < kBatch; ++j) o[j] = S[j];
            if (lane_usual(L, G, T, n_draw)) lane_commit(L, G, S);
            else lane_slow(L, G, S, T, n_draw, nullptr, true);
        }
#pragma unroll
        for (uint32_t j = 0u; j < kBatch / 2u; ++j) cur[j] = nxt[j];
    }
    M.sums_k[buf][lane] = k0;
    M.sums_n[buf][lane] = L.k - k0;
    if (L.state != LANE_RUN) M.d0[lane] = kIdle;   // (the producers may or may not see it in time: variates nobody reads)
}
__global__ void __launch_bounds__(kGroupThreads) af_arrival_groups(const ArrivalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char group_smem[];
    GroupLds& M = *reinterpret_cast<GroupLds*>(group_smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const bool chain = tid < 64u;
    const uint32_t ptid = tid - 64u;   // (producers)
    const uint32_t slot0 = blockIdx.x * a.group;
    const uint32_t n_here = a.n_scen - slot0 < a.group ? a.n_scen - slot0 : a.group;
    const double T = a.total_time;
    double* rows = a.out + (size_t)slot0 * a.stride;
    // the chain wave's lanes: their scenarios
    const bool valid = chain && lane < n_here;
    const uint32_t slot = slot0 + (valid ? lane : 0u);
    const uint32_t scen = a.scen_map ? a.scen_map[slot] : slot;
    const uint64_t seed = a.seeds[scen];
    double users_mean = 0.0, users_sigma = 0.0, window_s = 0.0, rps_per_user = 0.0;
    Lane L;
    lane_init(L);
    L.state = LANE_DONE;
    if (chain) {
        users_mean = arrival_param(a, af::PARAM_GEN_USERS_MEAN, scen, a.users_mean);
        users_sigma = arrival_param(a, af::PARAM_GEN_USERS_SIGMA, scen, a.users_sigma);
        window_s = arrival_param(a, af::PARAM_GEN_WINDOW, scen, a.window_s);
        rps_per_user = arrival_param(a, af::PARAM_GEN_RPM_MEAN, scen, a.rpm) / 60.0;
        if (valid) L.state = LANE_WAIT;
        M.seed[lane] = seed;
        // (the chain wave is the critical path of the workgroup: it issues before the producers it shares its SIMD with)
        __builtin_amdgcn_s_setprio(3);
    }
    for (uint32_t w = 0u;; ++w) {   // windows
        if (chain) {
            if (L.state == LANE_WAIT)
                window_start(L, T, window_s, rps_per_user,
                             [&](uint32_t idx) { return users_draw_call(a.users_dist, users_mean, users_sigma, seed, idx); });
            M.d0[lane] = L.state == LANE_RUN ? L.draws : kIdle;
            const bool any_run = __any(L.state == LANE_RUN), any_left = __any(L.state != LANE_DONE);
            if (lane == 0u) M.window[w & 1u] = any_run ? 2u : any_left ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t what = M.window[w & 1u];
        if (what == 0u) break;
        if (what == 1u) continue;
        if (!chain) group_produce(M, 0u, 0u, n_here, ptid);
        const bool fast = chain ? !__any(L.state == LANE_RUN && !L.fast_div) : true;
        __syncthreads();
        uint32_t r = 0u;
        for (;;) {   // rounds
            const uint32_t b = r & 1u;
            if (chain) {
                if (fast) group_chain_round<true>(L, M, b, lane, T, a.n_draw);
                else group_chain_round<false>(L, M, b, lane, T, a.n_draw);
                const bool any_run = __any(L.state == LANE_RUN);
                if (lane == 0u) M.run[b] = any_run ? 1u : 0u;
            } else {
                group_produce(M, b ^ 1u, r + 1u, n_here, ptid);
                if (r > 0u) group_store(M, b ^ 1u, rows, a.stride, n_here, ptid);
            }
            __syncthreads();
            ++r;
            if (M.run[b] == 0u) break;
        }
        if (!chain) group_store(M, (r - 1u) & 1u, rows, a.stride, n_here, ptid);   // the last round's
    }
    // behind the last arrival: +inf
    if (chain) {
        M.k_final[lane] = L.k;
        if (valid) a.pre_flags[slot] = L.flags;
    }
    __syncthreads();
    for (uint32_t s = 0u; s < n_here; ++s) {
        double* o = rows + (size_t)s * a.stride;
        for (uint32_t i = M.k_final[s] + tid; i < a.n_draw; i += kGroupThreads) o[i] = af::AF_INF;
    }
}
#endif  // __HIPCC__


}  // namespace afp
