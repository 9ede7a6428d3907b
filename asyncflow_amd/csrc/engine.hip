// engine.hip -- MI355X (gfx950) batched discrete-event engine: kernels + C ABI.
//
// Kernels of one af_engine_run:
//   af_pregen_arrivals<G>                  arrival times of every scenario (the windowed sampler), 64 / G scenarios per
//                                          wave, into HBM: arrivals[scenario][index]
//   af_flow_kernel<IPL, FEAT>              stage-parallel kernel (af_flow.hpp): ONE WAVE PER SCENARIO moves up to 64
//                                          requests per step through the stations of a feed-forward request path;
//                                          every other random draw is produced where it is consumed.  Runs every plan
//                                          in its range (af_engine_flow_reason); scenarios it hands back get a second
//                                          chance on its most tolerant instantiation, then go to:
//   af_pregen_edges + af_des_kernel<LDS?, SimPy-order path?, log2 lanes, waves/SIMD>
//                                          the sequential next-event loop (af_core.hpp): one scenario per lane, one
//                                          wave per workgroup, only the first 2^KLOG lanes of a wave carry scenarios;
//                                          every edge draw pre-generated into HBM: draws[slot][stream][index]
//   af_summary_kernel / af_series_kernel   af_engine_summarize: the analyzer (af_summary.hpp)
//
// Memory plan
//   LDS  : flow kernel: [plan blob, patched per scenario][station lists, select scratch / server segments, rings,
//          counters, tick-difference ring] (aff::make_flow_layout); next-event kernels: [plan blob][per-lane state:
//          64-bit words, SoA [index][lane]] -> any per-lane index pattern is bank-conflict free
//   HBM  : arrival times / pre-generated draws (read once), outputs (rqs_clock, sampled series, counts), the scratch
//          of the shared-instant path; the next-event per-lane state too when it does not fit the 160 KiB LDS of a CU
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/asyncflow_hip.h"
#include "af_core.hpp"
#include "af_flow_host.hpp"
#include "af_plan_pack.hpp"
#include "af_pregen.hpp"
#include "af_summary.hpp"

#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr uint32_t kLdsLimit = 160u * 1024u;  // bytes per workgroup on gfx950
constexpr uint32_t kWave = 64u;

struct KArgs {
    double total_time, sample_period, gen_users_mean, gen_users_sigma, gen_rpm_mean, gen_window_s;
    uint32_t metrics_mask, gen_users_dist, gen_out_edge, client_out_edge;
    uint32_t n_edges, n_servers, lb_algo, n_lb_edges, n_rows, n_edge_marks, n_srv_marks, every_event_in_order;
    uint32_t off_edge, off_srv, off_ep, off_row, off_emark, off_smark, off_lb;  // word offsets in the blob
    uint32_t blob_bytes;  // multiple of 16
    const unsigned char* blob;
    af::Layout L;
    uint32_t n_scen;
    const uint64_t* seeds;
    uint32_t n_ovr;
    const uint32_t* ovr_param;
    const uint32_t* ovr_index;
    const double* ovr_values;  // [n_ovr][ovr_stride], already offset to this chunk's first scenario
    uint32_t ovr_stride;       // scenarios in the whole sweep
    double* clock;
    uint32_t clock_cap;
    uint32_t* samples;
    uint32_t tick_cap, n_series, series_pitch;
    uint32_t* counts;
    uint32_t* online_hist;  // [n_scen][online_hist_bins] or null
    uint32_t* online_rps;   // [n_scen][online_rps_buckets] or null
    uint32_t online_hist_bins, online_rps_buckets;
    double online_hist_scale;
    unsigned char* state;  // HBM-resident lane state (global-state mode only)
    uint64_t state_bytes_per_wave;
    double* draws;          // pre-generated draws [n_scen][1 + n_edges][n_draw]
    uint32_t n_draw;
    uint32_t* pre_flags;    // [n_scen] AF_FLAG_DRAW_OVERFLOW from the arrival pre-generation
    uint64_t* tie;          // [n_scen][L.tie_words] scratch of the shared-timestamp path (HBM)
    const uint32_t* scen_map;  // subset launches: lane j simulates scenario scen_map[j] (null = identity)
    const uint32_t* draw_slot; // where lane j's pre-generated draws / pre_flags / scratch live (null = draws_by_lane ? j : scenario)
    uint32_t draws_by_lane;
    uint32_t* n_shared;     // first pass: number of scenarios that met a shared instant
};

// Per-lane state memory, [index][lane] with 2^klog scenario lanes per wave.
// KLOG is a compile-time constant: the shift folds into the LDS instruction's immediate offset and
// into the address arithmetic of every state access (measured: -4.6 % kernel time against a
// run-time shift).
template <int KLOG>
struct MemLds {
    LDS_AS uint64_t* w;  // already offset by the lane
    __device__ __forceinline__ uint64_t ld(uint32_t i) const { return w[i << KLOG]; }
    __device__ __forceinline__ void st(uint32_t i, uint64_t v) const { w[i << KLOG] = v; }
};

template <int KLOG>
struct MemGlobal {
    uint64_t* w;
    __device__ __forceinline__ uint64_t ld(uint32_t i) const { return w[i << KLOG]; }
    __device__ __forceinline__ void st(uint32_t i, uint64_t v) const { w[i << KLOG] = v; }
};

__device__ __forceinline__ const LDS_AS uint64_t* lds_words(unsigned char* smem, uint32_t word_off) {
    return (const LDS_AS uint64_t*)smem + word_off;
}

// The wave's main loop.  Lanes free-run round().  In the variant that has the SimPy-order path, a
// lane that reports an instant shared by several timed events makes the wave leave the inner loop;
// that lane replays the instant in SimPy's event order and the wave re-enters.  The cold path sits
// OUTSIDE the inner loop and every lane's scalars go through memory around it, so that as little as
// possible stays live across it (measured on the 10k LB-2 sweep with the path never taken: 3 472 ms
// with the path inside the loop, 2 755 ms outside, 2 690 ms with the scalars parked; the lean
// variant, which has no such path at all, takes 2 085 ms -- the rest is register pressure the
// compiler still lets leak into the loop).
template <class LaneT, bool kFaithful>
__device__ __forceinline__ void run_lanes(LaneT& S, bool active) {
    if constexpr (!kFaithful) {  // lean variant: round() never reports ROUND_SHARED
        bool run = active;
        while (__any(run)) {
            if (run) run = S.round();
        }
    } else {
        uint32_t st = active ? LaneT::ROUND_MORE : LaneT::ROUND_STOP;
        for (;;) {
            do {
                if (st == LaneT::ROUND_MORE) st = S.round();
            } while (!__any(st == LaneT::ROUND_SHARED) && __any(st == LaneT::ROUND_MORE));
            if (!__any(st == LaneT::ROUND_SHARED)) break;
            volatile uint64_t* slot = S.D.tie + (S.L.tie_words - LaneT::PARK_WORDS);
            if (active) S.park_regs(slot);
            if (st == LaneT::ROUND_SHARED) {
                S.unpark_regs(slot);
                S.shared_instant();
                S.park_regs(slot);
                st = LaneT::ROUND_MORE;
            }
            if (active) S.unpark_regs(slot);
        }
    }
}

// (Measured on MI355X, profiles/r01: forcing more than the natural 3 waves/SIMD with a
// launch bound spills registers and is 1.5-3x slower; the lean variant keeps its natural 134 VGPRs.)
//
// kFaithful = false is the lean first pass: a scenario in which two timed events share an instant
// stops there (af_core.hpp) and is simulated again by the kFaithful = true variant, whose extra
// SimPy-order path costs ~30 % of kernel time through register pressure alone (measured).
//
// WPE = register budget as waves per SIMD: 3 (<= 168 VGPRs; the lean variant needs 134) or, for the
// SimPy-order variant only, 2 (it wants ~210 VGPRs: no spills, but a third fewer resident waves).
template <bool kLdsState, bool kFaithful, int KLOG>
__device__ __forceinline__ void des_body(const KArgs& a_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
#if defined(AF_JIT) && !defined(AF_FLOW_JIT)
    // Plan-specialised build (asyncflow_amd/jit.py): the plan's shape as compile-time constants --
    // state offsets fold into instruction immediates, loops over edges / servers / series unroll,
    // branches on the plan's shape disappear (measured on the 10k LB-2 sweep: 2 084 -> 1 912 ms).
    KArgs a = a_in;
    a.metrics_mask = AF_JIT_METRICS;
    a.gen_out_edge = AF_JIT_GEN_EDGE;
    a.client_out_edge = AF_JIT_CLIENT_EDGE;
    a.n_edges = AF_JIT_N_EDGES;
    a.n_servers = AF_JIT_N_SERVERS;
    a.lb_algo = AF_JIT_LB_ALGO;
    a.n_lb_edges = AF_JIT_N_LB;
    a.n_rows = AF_JIT_N_ROWS;
    a.n_edge_marks = AF_JIT_N_EMARKS;
    a.n_srv_marks = AF_JIT_N_SMARKS;
    a.every_event_in_order = AF_JIT_ORDER_ALL;
    a.off_edge = AF_JIT_OFF_EDGE;
    a.off_srv = AF_JIT_OFF_SRV;
    a.off_ep = AF_JIT_OFF_EP;
    a.off_row = AF_JIT_OFF_ROW;
    a.off_emark = AF_JIT_OFF_EMARK;
    a.off_smark = AF_JIT_OFF_SMARK;
    a.off_lb = AF_JIT_OFF_LB;
    a.blob_bytes = AF_JIT_BLOB_BYTES;
    a.L = af::make_layout(AF_JIT_CAP, AF_JIT_FCAP, AF_JIT_N_EDGES, AF_JIT_N_SERVERS, AF_JIT_N_LB, AF_JIT_N_ROWS, AF_JIT_OVR_MASK, AF_JIT_N_EMARKS, AF_JIT_N_SMARKS);
    // (clock / tick / draw capacities stay run-time values: they change with replicas, horizon and the runner's
    // auto-grow, and a key that contains them recompiles -- or silently misses -- for every such change)
    a.n_series = AF_JIT_N_EDGES + 3 * AF_JIT_N_SERVERS;
    a.series_pitch = (AF_JIT_N_EDGES + 3 * AF_JIT_N_SERVERS + 3) & ~3;
    if (AF_JIT_HAS_CLOCK) __builtin_assume(a.clock != nullptr); else a.clock = nullptr;
    if (AF_JIT_HAS_SAMPLES) __builtin_assume(a.samples != nullptr); else a.samples = nullptr;
    if (!AF_JIT_HAS_ONLINE) a.online_hist = a.online_rps = nullptr;
#else
    const KArgs& a = a_in;
#endif

    // stage the read-only plan into LDS (shared by the 64 scenarios of the wave)
    for (uint32_t i = lane * 16u; i < a.blob_bytes; i += kWave * 16u)
        *reinterpret_cast<uint4*>(smem + i) = *reinterpret_cast<const uint4*>(a.blob + i);
    __syncthreads();

    af::PlanView P;
    P.total_time = a.total_time;
    P.sample_period = a.sample_period;
    P.gen_users_mean = a.gen_users_mean;
    P.gen_users_sigma = a.gen_users_sigma;
    P.gen_rpm_mean = a.gen_rpm_mean;
    P.gen_window_s = a.gen_window_s;
    P.metrics_mask = a.metrics_mask;
    P.gen_users_dist = a.gen_users_dist;
    P.gen_out_edge = a.gen_out_edge;
    P.client_out_edge = a.client_out_edge;
    P.n_edges = a.n_edges;
    P.n_servers = a.n_servers;
    P.lb_algo = a.lb_algo;
    P.n_lb_edges = a.n_lb_edges;
    P.n_rows = a.n_rows;
    P.n_edge_marks = a.n_edge_marks;
    P.n_srv_marks = a.n_srv_marks;
    P.every_event_in_order = a.every_event_in_order;
    P.edge = lds_words(smem, a.off_edge);
    P.srv = lds_words(smem, a.off_srv);
    P.ep = lds_words(smem, a.off_ep);
    P.row = lds_words(smem, a.off_row);
    P.emark = lds_words(smem, a.off_emark);
    P.smark = lds_words(smem, a.off_smark);
    P.lb = lds_words(smem, a.off_lb);

    // Only the first 2^klog lanes of a wave carry scenarios.  With few scenarios per GPU the
    // sweep is spread over MANY narrow waves: fewer event kinds per round in each wave (less
    // divergence), every SIMD of the chip busy, several waves per SIMD hiding LDS latency.
    constexpr uint32_t kl = 1u << KLOG;
    const uint32_t scen = (blockIdx.x << KLOG) + lane;
    const bool active = lane < kl && scen < a.n_scen;
    const uint32_t sc = active ? (a.scen_map ? a.scen_map[scen] : scen) : 0u;

    af::LaneOut O;
    O.clock = a.clock ? a.clock + (size_t)sc * a.clock_cap * 2u : nullptr;
    O.samples = a.samples ? a.samples + (size_t)sc * a.series_pitch * a.tick_cap : nullptr;
    O.counts = a.counts + (size_t)sc * af::CNT_SLOTS;
    O.clock_cap = a.clock_cap;
    O.tick_cap = a.tick_cap;
    O.series_pitch = a.series_pitch;
    O.hist = a.online_hist ? a.online_hist + (size_t)sc * a.online_hist_bins : nullptr;
    O.rps = a.online_rps ? a.online_rps + (size_t)sc * a.online_rps_buckets : nullptr;
    O.hist_bins = a.online_hist_bins;
    O.rps_buckets = a.online_rps_buckets;
    O.hist_scale = a.online_hist_scale;

    const uint64_t seed = a.seeds[sc];
    auto ovr = [&](uint32_t k) { return a.ovr_values[(size_t)k * a.ovr_stride + sc]; };
    const uint32_t slot = !active ? 0u : a.draw_slot ? a.draw_slot[scen] : a.draws_by_lane ? scen : sc;
    af::PreDraws D;
    D.base = a.draws + (size_t)slot * (1u + a.n_edges) * a.n_draw;
    D.n_per_stream = a.n_draw;
    D.flags_in = a.pre_flags[slot];
    D.tie = a.tie + (size_t)slot * a.L.tie_words;

    if constexpr (kLdsState) {
        MemLds<KLOG> M;
        M.w = (LDS_AS uint64_t*)(smem + a.blob_bytes) + (lane & (kl - 1u));
        af::Lane<MemLds<KLOG>, kFaithful> S(P, a.L, M, O, D, seed);
        if (active) S.init(a.ovr_param, a.ovr_index, a.n_ovr, ovr);
        run_lanes<decltype(S), kFaithful>(S, active);
        if (active) S.write_counts();
        if (!kFaithful && active && (S.flags & af::FLAG_SHARED_INSTANT)) atomicAdd(a.n_shared, 1u);
    } else {
        unsigned char* base = a.state + (size_t)blockIdx.x * a.state_bytes_per_wave;
        MemGlobal<KLOG> M;
        M.w = reinterpret_cast<uint64_t*>(base) + (lane & (kl - 1u));
        af::Lane<MemGlobal<KLOG>, kFaithful> S(P, a.L, M, O, D, seed);
        if (active) S.init(a.ovr_param, a.ovr_index, a.n_ovr, ovr);
        run_lanes<decltype(S), kFaithful>(S, active);
        if (active) S.write_counts();
        if (!kFaithful && active && (S.flags & af::FLAG_SHARED_INSTANT)) atomicAdd(a.n_shared, 1u);
    }
}

#if !defined(AF_JIT) || defined(AF_FLOW_JIT)
// ---- stage-parallel kernel (af_flow.hpp): one wave per scenario -------------------------------------
struct WaveHip {
    static __device__ __forceinline__ uint32_t lane() { return threadIdx.x; }
    static __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
    static __device__ __forceinline__ bool any(bool p) { return __any(p) != 0; }
    static __device__ __forceinline__ uint32_t shfl32(uint32_t v, uint32_t src) {
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
    }
    static __device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src) {
        const uint32_t lo = shfl32((uint32_t)v, src), hi = shfl32((uint32_t)(v >> 32), src);
        return ((uint64_t)hi << 32) | lo;
    }
    // One wave per workgroup: its LDS instructions execute in order, so "every lane's LDS writes are visible to every
    // lane" needs no hardware wait beyond the data dependences the compiler tracks (lgkmcnt) -- only the COMPILER must
    // not move LDS accesses to other lanes' data across this point: wavefront-scope fences + a scheduling barrier.
    // (Workgroup-scope fences here cost an s_waitcnt vmcnt(0) each: ~30 times per round the wave waited for its own
    // rqs_clock / sample stores to reach memory.)
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    static __device__ __forceinline__ uint32_t lds_add(LDS_AS uint32_t* p, uint32_t v) {
        return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ void global_add(uint32_t* p, uint32_t v) {
        (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ uint32_t global_load(const uint32_t* p) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void global_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
    // the scenario's outputs are in memory (every lane's stores: a device-scope release fence per lane, then the wave barrier that
    // orders the lanes), then the flag (release, device scope) and the count (release, SYSTEM scope: host-coherent memory the
    // analyzer's stream waits on with hipStreamWaitValue32)
    static __device__ __forceinline__ void signal_done(uint32_t* flag, uint32_t* count) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0u) {
            __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    static __device__ __forceinline__ double rcp(double x) { return __builtin_amdgcn_rcp(x); }
    static __device__ __forceinline__ double fract(double x) { return __builtin_amdgcn_fract(x); }   // v_fract_f64: x - floor(x), exact for x >= 0
    static __device__ __forceinline__ uint32_t bcast32(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }
    static __device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t src) {
        const uint32_t lo = bcast32((uint32_t)v, src), hi = bcast32((uint32_t)(v >> 32), src);
        return ((uint64_t)hi << 32) | lo;
    }
    // wave64 inclusive prefix sum without LDS: Kogge-Stone inside each row of 16 lanes with DPP row shifts (lanes
    // shifted in from outside the row read 0), then the row totals with row_bcast:15 into rows 1, 3 and row_bcast:31
    // into rows 2, 3 (a lane whose row is masked off keeps `old` = 0)
    template <int CTRL, int ROW_MASK>
    static __device__ __forceinline__ uint32_t dpp(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    }
    static __device__ __forceinline__ uint32_t scan_incl_u32(uint32_t v) {
        v += dpp<0x111, 0xF>(v);   // row_shr:1
        v += dpp<0x112, 0xF>(v);   // row_shr:2
        v += dpp<0x114, 0xF>(v);   // row_shr:4
        v += dpp<0x118, 0xF>(v);   // row_shr:8
        v += dpp<0x142, 0xA>(v);   // row_bcast:15 -> rows 1 and 3
        v += dpp<0x143, 0xC>(v);   // row_bcast:31 -> rows 2 and 3
        return v;
    }
    static __device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    }
    static __device__ __forceinline__ unsigned long long clock() { return __builtin_amdgcn_s_memtime(); }   // shader cycles
};

// af_engine_run_summarized: the engine's counter block (FlowArgs::n_fallback) holds, at word kDoneWords, the addresses of the done
// flags and of the done counter, or zeros.  Read HERE, after run(), through a pointer the kernel keeps anyway: as two more
// kernel arguments they were two more scalar register pairs live across the whole kernel (general servers: 280 -> 295 ms).
constexpr uint32_t kDoneWords = 16u;
static __device__ __forceinline__ void signal_scenario_done(const uint32_t* counter_block, uint32_t sc) {
    if (counter_block == nullptr) return;
    const uint64_t* sig = reinterpret_cast<const uint64_t*>(counter_block + kDoneWords);
    uint32_t* flags = reinterpret_cast<uint32_t*>(sig[0]);
    if (flags != nullptr) WaveHip::signal_done(flags + sc, reinterpret_cast<uint32_t*>(sig[1]));
}

// Register budget as waves per SIMD (measured on MI355X, profiles/r02: see DESIGN.md section 4e).
#ifndef AF_FLOW_WPE
#define AF_FLOW_WPE 4
#endif
#endif
#ifndef AF_JIT
// (general servers: the LDS of that form admits ~2 waves per SIMD anyway, so its serial station gets the registers of a
// 3-waves-per-SIMD budget instead of spilling: 128 VGPRs + 112 B of scratch at 4)
template <uint32_t IPL, uint32_t FEAT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((FEAT & aff::FEAT_GENSRV) ? 3 : AF_FLOW_WPE))) af_flow_kernel(const aff::FlowArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= a.n_scen) return;
    const uint32_t sc = a.scen_map ? a.scen_map[blockIdx.x] : blockIdx.x;
    aff::Flow<WaveHip, IPL, FEAT> f(a);
    f.run((LDS_AS uint64_t*)smem, sc);
    signal_scenario_done(a.n_fallback, sc);
    if (threadIdx.x == 0u && a.n_fallback) {
        const uint32_t flags = a.counts[(size_t)sc * af::CNT_SLOTS + af::CNT_FLAGS];
        if (flags & aff::FLAG_FLOW_FALLBACK) {
            atomicAdd(a.n_fallback, 1u);
            if (flags & aff::FLOW_WHY_TIE) atomicAdd(a.n_fallback + 1, 1u);
            if (flags & aff::FLOW_WHY_LIST) atomicAdd(a.n_fallback + 2, 1u);
            if (flags & aff::FLOW_WHY_RING) atomicAdd(a.n_fallback + 3, 1u);
            if (flags & aff::FLOW_WHY_RAM) atomicAdd(a.n_fallback + 4, 1u);
        }
    }
}

template <bool kLdsState, bool kFaithful, int KLOG, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) af_des_kernel(const KArgs a) {
    des_body<kLdsState, kFaithful, KLOG>(a);
}
#endif

#ifdef AF_FLOW_JIT
}  // namespace

// Plan-specialised build of the stage-parallel kernel (asyncflow_amd/jit.py; the flags come from af_engine_jit_spec when
// the sweep would run on af_flow_kernel): the instantiation (IPL, FEAT), the plan's shape, the horizon / tick constants
// and the whole LDS layout are compile-time constants.  The generic kernel reads ~100 wave-uniform launch arguments
// that do not fit the 102 SGPRs (a quarter of its static VALU instructions are v_mov, 12 % SGPR<->VGPR-lane spills:
// DESIGN.md section 4e); here they are immediates, LDS offsets fold into the instructions, loops over servers / LB
// edges / step programs have constant trip counts.  Same source, same arithmetic: results are bit-identical.
// What stays a run-time argument: every pointer, the scenario count and the clock / tick / draw capacities (they
// change with replicas and the runner's auto-grow; a key that contained them would recompile for every such change).
#define AF_FJ_F64(bits) __builtin_bit_cast(double, (uint64_t)(bits))
// (the register budget follows the waves per SIMD the launch's LDS leaves room for -- AF_FJ_WPE, worked out by flow_jit_spec_string:
// BASELINE config 5's 20 KB per wave admit two waves per SIMD, and with the 256-register budget of two instead of the 128 of four
// its kernel takes 189 instead of 207 ms, round 5)
#ifndef AF_FJ_WPE
#define AF_FJ_WPE (((AF_FJ_FEAT) & aff::FEAT_GENSRV) ? 3 : AF_FLOW_WPE)
#endif
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AF_FJ_WPE))) af_flow_jit(const aff::FlowArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= a_in.n_scen) return;
    aff::FlowArgs a = a_in;
    a.total_time = AF_FJ_F64(AF_FJ_TOTAL_TIME);
    a.sample_period = AF_FJ_F64(AF_FJ_PERIOD);
    a.inv_period = AF_FJ_F64(AF_FJ_INV_PERIOD);
    a.tick_eps = AF_FJ_F64(AF_FJ_TICK_EPS);
    a.metrics_mask = AF_FJ_METRICS;
    a.gen_out_edge = AF_FJ_GEN_EDGE;
    a.client_out_edge = AF_FJ_CLIENT_EDGE;
    a.n_edges = AF_FJ_N_EDGES;
    a.n_servers = AF_FJ_N_SERVERS;
    a.has_lb = AF_FJ_HAS_LB;
    a.n_lb_edges = AF_FJ_N_LB;
    a.n_edge_marks = AF_FJ_N_EMARKS;
    a.n_srv_marks = AF_FJ_N_SMARKS;
    a.lb_least_connections = AF_FJ_LC;
    a.max_pre = AF_FJ_MAX_PRE;
    a.max_cpu = AF_FJ_MAX_CPU;
    a.max_post = AF_FJ_MAX_POST;
    a.ram_scale = AF_FJ_RAM_SCALE;
    a.ram_unit = 1.0 / AF_FJ_RAM_SCALE;
    a.off_edge = AF_FJ_OFF_EDGE;
    a.off_srv = AF_FJ_OFF_SRV;
    a.off_ep = AF_FJ_OFF_EP;
    a.off_row = AF_FJ_OFF_ROW;
    a.off_emark = AF_FJ_OFF_EMARK;
    a.off_smark = AF_FJ_OFF_SMARK;
    a.off_lb = AF_FJ_OFF_LB;
    a.blob_bytes = AF_FJ_BLOB_BYTES;
    a.n_ticks = AF_FJ_N_TICKS;
    a.L = aff::FlowLayout{AF_FJ_LAYOUT};
    if (AF_FJ_HAS_CLOCK) __builtin_assume(a.clock != nullptr); else a.clock = nullptr;
    if (AF_FJ_HAS_SAMPLES) __builtin_assume(a.samples != nullptr); else a.samples = nullptr;
    if (!AF_FJ_HAS_ONLINE) a.online_hist = a.online_rps = nullptr;
    if (!AF_FJ_HAS_OVR) a.n_ovr = 0u;
    const uint32_t sc = a.scen_map ? a.scen_map[blockIdx.x] : blockIdx.x;
    aff::Flow<WaveHip, AF_FJ_IPL, AF_FJ_FEAT> f(a);
    f.run((LDS_AS uint64_t*)smem, sc);
    signal_scenario_done(a.n_fallback, sc);
    if (threadIdx.x == 0u && a.n_fallback) {
        const uint32_t flags = a.counts[(size_t)sc * af::CNT_SLOTS + af::CNT_FLAGS];
        if (flags & aff::FLAG_FLOW_FALLBACK) {
            atomicAdd(a.n_fallback, 1u);
            if (flags & aff::FLOW_WHY_TIE) atomicAdd(a.n_fallback + 1, 1u);
            if (flags & aff::FLOW_WHY_LIST) atomicAdd(a.n_fallback + 2, 1u);
            if (flags & aff::FLOW_WHY_RING) atomicAdd(a.n_fallback + 3, 1u);
            if (flags & aff::FLOW_WHY_RAM) atomicAdd(a.n_fallback + 4, 1u);
        }
    }
}
#elif defined(AF_JIT)
}  // namespace

// Entry points of a plan-specialised code object (built by asyncflow_amd/jit.py with hipcc --genco,
// loaded by af_engine_set_kernels): the lean variant and the SimPy-order variant with its two
// register budgets, for ONE (state placement, lanes per wave) pair.
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) af_jit_lean(const KArgs a) {
    des_body<AF_JIT_LDS != 0, false, AF_JIT_KLOG>(a);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) af_jit_order3(const KArgs a) {
    des_body<AF_JIT_LDS != 0, true, AF_JIT_KLOG>(a);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) af_jit_order2(const KArgs a) {
    des_body<AF_JIT_LDS != 0, true, AF_JIT_KLOG>(a);
}
#else  // ------------------------------------------------------------------------------- AOT build

// ---- draw pre-generation (fully parallel, full occupancy) ---------------------------
__device__ __forceinline__ double ovr_or(const KArgs& a, uint32_t param, uint32_t index, uint32_t scen, double dflt) {
    for (uint32_t k = 0; k < a.n_ovr; ++k)
        if (a.ovr_param[k] == param && a.ovr_index[k] == index) dflt = a.ovr_values[(size_t)k * a.ovr_stride + scen];
    return dflt;
}

// Stream 0: the windowed arrival sampler (af::gen_next_gap is its sequential statement: samplers/
// poisson_poisson.py:51-82, gaussian_poisson.py:63-94) and the absolute arrival times
// t_k = t_{k-1} + gap_k (env.now + gap, rqs_generator.py:104).
//
// FOUR scenarios per wave, 16 lanes each.  The expensive part of a gap (Philox block, log, division) is
// evaluated for 16 consecutive draw indices of each scenario at once -- a draw is a pure function of its
// index.  The order-dependent part is two running f64 sums (the sampler's virtual clock and the
// simulation clock): lane l of a group adds the gaps 0..l of the batch to the group's sums ONE BY ONE, in
// draw order -- the sequential sampler's own additions, so every prefix is bit-identical to it -- while the
// 16 steps serve the four scenarios of the wave together.  The window / horizon tests are then evaluated
// on all prefixes at once; the first draw that crosses the window end (discarded, on to the next window:
// the next index is the new window's user draw) or the horizon ends the batch.
// Block b pre-generates for scenarios scen_map[4 b + g] (null = 4 b + g) into slots 4 b + g.  `stride` =
// doubles per slot (n_draw when only the arrivals are wanted: the flow kernel; (1 + n_edges) * n_draw otherwise).
// G = lanes per scenario (16, 12 or 8: 4, 5 or 8 scenarios per wave).  The kernel is bound by instruction issue per
// SIMD, so its time is (waves on the fullest SIMD) x (work of a wave): 2 500 waves of 4 scenarios on 1 024 SIMDs take
// the time of 3 waves, 2 000 waves of 5 scenarios the time of 2 slightly longer ones (af_engine_run picks G).
template <uint32_t G>
__global__ void __launch_bounds__(64) af_pregen_arrivals(const KArgs a, uint32_t stride) {
    constexpr uint32_t S = 64u / G;                     // scenarios per wave (lanes >= S * G idle)
    const uint32_t lane = threadIdx.x, grp = lane / G, l = lane - grp * G, gbase = grp * G;
    const uint32_t slot = blockIdx.x * S + grp;
    const bool valid = grp < S && slot < a.n_scen;
    const uint32_t scen = valid ? (a.scen_map ? a.scen_map[slot] : slot) : 0u;
    const uint64_t seed = a.seeds[scen];
    const double users_mean = ovr_or(a, af::PARAM_GEN_USERS_MEAN, 0u, scen, a.gen_users_mean);
    const double users_sigma = ovr_or(a, af::PARAM_GEN_USERS_SIGMA, 0u, scen, a.gen_users_sigma);
    const double rpm = ovr_or(a, af::PARAM_GEN_RPM_MEAN, 0u, scen, a.gen_rpm_mean);
    const double window_s = ovr_or(a, af::PARAM_GEN_WINDOW, 0u, scen, a.gen_window_s);
    const double rps_per_user = rpm / 60.0;
    const double T = a.total_time;
    double* out = a.draws + (size_t)slot * stride;  // stream 0 of this slot
    double g_now = 0.0, g_wend = 0.0, lam = 0.0, t = 0.0;  // identical in the G lanes of a group
    uint32_t draws = 0u, k = 0u, flags = 0u;
    bool run = valid;
    while (__any(run)) {
        run = run && g_now < T;
        if (run && g_now >= g_wend) {  // new window: the number of active users
            g_wend = g_now + window_s;
            const uint32_t idx = draws++;
            double users;
            if (a.gen_users_dist == af::DIST_NORMAL) {
                const double v = users_mean + users_sigma * af::af_norminv(af::uniform_j(seed, af::STREAM_GENERATOR, idx, 0u));
                users = v > 0.0 ? v : 0.0;
            } else {
                users = (double)af::af_poisson(users_mean, seed, af::STREAM_GENERATOR, idx, 0u);
            }
            lam = users * rps_per_user;
        }
        const bool idle = run && lam <= 0.0;   // nobody active in this window
        if (idle) g_now = g_wend;
        const bool draw = run && !idle;
        // G gaps of this scenario (lanes of scenarios that do not draw compute a harmless dummy)
        const af::U4 r = af::draw_block(seed, af::STREAM_GENERATOR, draws + l, 0u);
        double u = af::u53(r.x, r.y);
        if (u < 1e-15) u = 1e-15;
        const double dt = -af::af_log_unit(1.0 - u) / (draw ? lam : 1.0);
        // prefixes in draw order: Gs = sampler clock after gap l, Ss = simulation clock after gap l
        double Gs = g_now, Ss = t;
#pragma unroll
        for (uint32_t j = 0u; j < G; ++j) {
            const double dj = __shfl(dt, (int)((gbase + j) & 63u), 64);
            if (l >= j) {
                Gs += dj;
                Ss = Ss + dj;
            }
        }
        const bool over = Gs > T;                  // the sampler is exhausted at this draw
        const bool cross = !over && Gs >= g_wend;  // this draw crosses the window end: discarded
        const uint64_t stops = __ballot(draw && (over || cross));
        const uint32_t mine = (uint32_t)(stops >> (gbase & 63u)) & ((1u << G) - 1u);
        const uint32_t first = mine ? (uint32_t)__builtin_ctz(mine) : G;   // draws before it are arrivals
        uint32_t n_acc = first;
        bool full = false;
        if (draw && k + n_acc > a.n_draw) {  // more arrivals than the array holds
            n_acc = a.n_draw - k;
            full = true;
        }
        if (draw && l < n_acc) out[k + l] = Ss;
        // the group's state after the batch
        const double Glast = __shfl(Gs, (int)((gbase + G - 1u) & 63u), 64), Slast = __shfl(Ss, (int)((gbase + G - 1u) & 63u), 64);
        const double Sprev = __shfl(Ss, (int)((gbase + (first > 0u ? first - 1u : 0u)) & 63u), 64);
        const bool over_first = ((__ballot(over) >> (gbase & 63u)) >> (first % G)) & 1ull;
        if (draw) {
            k += n_acc;
            if (full) {
                flags = AF_FLAG_DRAW_OVERFLOW;
                run = false;
            } else if (first == G) {
                g_now = Glast;
                t = Slast;
                draws += G;
            } else {
                if (first > 0u) t = Sprev;
                draws += first + 1u;
                if (over_first) run = false;
                else g_now = g_wend;
            }
        }
    }
    if (valid) {
        for (uint32_t i = k + l; i < a.n_draw; i += G) out[i] = af::AF_INF;
        if (l == 0u) a.pre_flags[slot] = flags;
    }
}

// Round 3: the same kernel with ONE DPP ROW (16 lanes) per scenario.  The order-dependent part -- lane l adds gaps 0..l to
// the row's two running sums one by one -- used to fetch gap j of its group through the LDS crossbar (ds_bpermute, two
// per step and ~100+ cycles each, 12-16 dependent steps per batch: ~1 500 of a batch's ~2 000 cycles were this chain;
// the kernel was bound by that latency at 2 waves per SIMD, not by issue).  A group that IS a DPP row gets gap j as
// `v_mov_b32_dpp ... row_newbcast:j` (gfx90a+): register-file latency, no LDS.  Same draws, same f64 additions in the
// same order: bit-identical arrival times (tests/test_gpu_flow.py::test_arrival_pregeneration_group_widths_are_equivalent).
// The once-a-window user draw (Poisson by chunked inversion / normal quantile) sits behind a call so that the hot loop
// keeps <= 128 VGPRs: 2 500 waves of 4 scenarios are then all resident (4 per SIMD).
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
    const uint64_t u = af::d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0x150 + J, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0x150 + J, 0xF, 0xF, false);
    return af::u2d(((uint64_t)hi << 32) | lo);
}
__device__ __noinline__ double pregen_users_draw(uint32_t dist, double mean, double sigma, uint64_t seed, uint32_t idx) {
    if (dist == af::DIST_NORMAL) {
        const double v = mean + sigma * af::af_norminv(af::uniform_j(seed, af::STREAM_GENERATOR, idx, 0u));
        return v > 0.0 ? v : 0.0;
    }
    return (double)af::af_poisson(mean, seed, af::STREAM_GENERATOR, idx, 0u);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) af_pregen_arrivals_rows(const KArgs a, uint32_t stride) {
    constexpr uint32_t G = 16u, S = 4u;
    const uint32_t lane = threadIdx.x, grp = lane >> 4, l = lane & 15u, gbase = grp * G;
    const uint32_t slot = blockIdx.x * S + grp;
    const bool valid = slot < a.n_scen;
    const uint32_t scen = valid ? (a.scen_map ? a.scen_map[slot] : slot) : 0u;
    const uint64_t seed = a.seeds[scen];
    const double users_mean = ovr_or(a, af::PARAM_GEN_USERS_MEAN, 0u, scen, a.gen_users_mean);
    const double users_sigma = ovr_or(a, af::PARAM_GEN_USERS_SIGMA, 0u, scen, a.gen_users_sigma);
    const double rpm = ovr_or(a, af::PARAM_GEN_RPM_MEAN, 0u, scen, a.gen_rpm_mean);
    const double window_s = ovr_or(a, af::PARAM_GEN_WINDOW, 0u, scen, a.gen_window_s);
    const double rps_per_user = rpm / 60.0;
    const double T = a.total_time;
    double* out = a.draws + (size_t)slot * stride;
    double g_now = 0.0, g_wend = 0.0, lam = 0.0, t = 0.0;   // identical in the 16 lanes of a row
    uint32_t draws = 0u, k = 0u, flags = 0u;
    bool run = valid;
    while (__any(run)) {
        run = run && g_now < T;
        if (run && g_now >= g_wend) {  // new window: the number of active users
            g_wend = g_now + window_s;
            lam = pregen_users_draw(a.gen_users_dist, users_mean, users_sigma, seed, draws++) * rps_per_user;
        }
        const bool idle = run && lam <= 0.0;   // nobody active in this window
        if (idle) g_now = g_wend;
        const bool draw = run && !idle;
        const af::U4 r = af::draw_block(seed, af::STREAM_GENERATOR, draws + l, 0u);
        double u = af::u53(r.x, r.y);
        if (u < 1e-15) u = 1e-15;
        const double dt = -af::af_log_unit(1.0 - u) / (draw ? lam : 1.0);
        // prefixes in draw order: Gs = sampler clock after gap l, Ss = simulation clock after gap l (x + 0.0 == x exactly)
        double Gs = g_now, Ss = t;
#define AF_ROW_STEP(J) { const double dj = row_bcast<J>(dt); const double m = l >= (J) ? dj : 0.0; Gs += m; Ss = Ss + m; }
        AF_ROW_STEP(0) AF_ROW_STEP(1) AF_ROW_STEP(2) AF_ROW_STEP(3) AF_ROW_STEP(4) AF_ROW_STEP(5) AF_ROW_STEP(6) AF_ROW_STEP(7)
        AF_ROW_STEP(8) AF_ROW_STEP(9) AF_ROW_STEP(10) AF_ROW_STEP(11) AF_ROW_STEP(12) AF_ROW_STEP(13) AF_ROW_STEP(14) AF_ROW_STEP(15)
#undef AF_ROW_STEP
        const bool over = Gs > T;                  // the sampler is exhausted at this draw
        const bool cross = !over && Gs >= g_wend;  // this draw crosses the window end: discarded
        const uint64_t stops = __ballot(draw && (over || cross));
        const double Glast = row_bcast<15>(Gs), Slast = row_bcast<15>(Ss);
        if (stops == 0ull && !__any(draw && k + G > a.n_draw)) {   // the usual batch: 16 arrivals per drawing row, nothing else
            if (draw) {
                out[k + l] = Ss;
                k += G;
                g_now = Glast;
                t = Slast;
                draws += G;
            }
            continue;
        }
        const uint32_t mine = (uint32_t)(stops >> gbase) & 0xFFFFu;
        const uint32_t first = mine ? (uint32_t)__builtin_ctz(mine) : G;   // draws before it are arrivals
        uint32_t n_acc = first;
        bool full = false;
        if (draw && k + n_acc > a.n_draw) {  // more arrivals than the array holds
            n_acc = a.n_draw - k;
            full = true;
        }
        if (draw && l < n_acc) out[k + l] = Ss;
        const double Sprev = __shfl(Ss, (int)((gbase + (first > 0u ? first - 1u : 0u)) & 63u), 64);
        const bool over_first = ((__ballot(over) >> gbase) >> (first & 15u)) & 1ull;
        if (draw) {
            k += n_acc;
            if (full) {
                flags = AF_FLAG_DRAW_OVERFLOW;
                run = false;
            } else if (first == G) {
                g_now = Glast;
                t = Slast;
                draws += G;
            } else {
                if (first > 0u) t = Sprev;
                draws += first + 1u;
                if (over_first) run = false;
                else g_now = g_wend;
            }
        }
    }
    if (valid) {
        for (uint32_t i = k + l; i < a.n_draw; i += G) out[i] = af::AF_INF;
        if (l == 0u) a.pre_flags[slot] = flags;
    }
}

// ---- the grouped form (round 3, af_pregen.hpp: af_arrival_groups) -- one LANE per scenario for the order-dependent part,
// the variates worked out beside it by the other waves of the workgroup ------------------------------------------------
// (A first split -- the variates by a parallel kernel into HBM, the sums by 8 lanes and 8 draws per scenario and step, the
// same masked-sum formulation as above -- was built and removed: its chain kernel alone executed 2.9e9 VALU wave-
// instructions, 233 per 64 draws, 11.9 ms against 11.0 ms for the fused row kernel.  The serial bookkeeping is what the
// pre-generation costs, and it only goes away when a lane does nothing but its own scenario's additions: 16 adds + 24
// instructions of division per 64 draws.  af_pregen.hpp tells what that lane must NOT do: touch HBM.)

// launch helper: the group width with the shortest issue-bound time for `n` scenarios (see the kernel's comment).
// Measured (10 000 / 8 192 LB-2 replicas): 4 per wave 15.2 / 9.5 ms, 5 per wave 11.7 / -, 8 per wave - / 15.1 ms (one wave
// per SIMD: nothing hides its latencies).  More than 4 per wave only when the SIMDs still get ~2 waves each and the
// scenarios are alike (`uniform_load`: no per-scenario users / rpm column -- a wave is as slow as its heaviest scenario).
// `grouped`: af_arrival_groups instead (af_engine_run decides: sweeps of alike scenarios from a few thousand on, long
// sampling windows; AF_PREGEN_MODE=rows / groups overrides, AF_PREGEN_GROUP = scenarios per workgroup).
// scenarios per workgroup of af_arrival_groups: one workgroup per CU while that leaves it <= 64 scenarios -- the chain wave of a
// workgroup is as long as ONE scenario's chain however many lanes it has, the producers' work grows with them
static uint32_t pregen_group_width(uint32_t n) {
    uint32_t group = (n + 255u) / 256u < 64u ? (n + 255u) / 256u : 64u;
    if (const char* env = std::getenv("AF_PREGEN_GROUP")) {
        const int v = std::atoi(env);
        if (v >= 1 && v <= 64) group = (uint32_t)v;
    }
    return group;
}

static int launch_pregen_arrivals(const KArgs& a, uint32_t n, uint32_t stride, bool uniform_load, hipStream_t stream,
                                  bool grouped = false, uint32_t* group_out = nullptr) {
    if (grouped) {
        afp::ArrivalArgs g{};
        g.total_time = a.total_time;
        g.users_mean = a.gen_users_mean;
        g.users_sigma = a.gen_users_sigma;
        g.rpm = a.gen_rpm_mean;
        g.window_s = a.gen_window_s;
        g.users_dist = a.gen_users_dist;
        g.n_scen = n;
        g.n_draw = a.n_draw;
        g.stride = stride;
        g.group = pregen_group_width(n);
        if (group_out) *group_out = g.group;
        g.seeds = a.seeds;
        g.scen_map = a.scen_map;
        g.n_ovr = a.n_ovr;
        g.ovr_param = a.ovr_param;
        g.ovr_index = a.ovr_index;
        g.ovr_values = a.ovr_values;
        g.ovr_stride = a.ovr_stride;
        g.out = a.draws;
        g.pre_flags = a.pre_flags;
        // (set before every launch: the attribute is per device, engines of several devices run in threads of one process,
        // and the call costs microseconds next to a kernel of milliseconds -- ADVICE r3)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(afp::af_arrival_groups), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(afp::GroupLds)) != hipSuccess)
            return 1;
        hipLaunchKernelGGL(afp::af_arrival_groups, dim3((n + g.group - 1u) / g.group), dim3(afp::kGroupThreads),
                           sizeof(afp::GroupLds), stream, g);
        return 0;
    }
    double best = 1e300;
    uint32_t best_s = 4u;
    for (uint32_t s : {4u, 5u, 8u}) {
        const double waves = std::ceil((double)n / (double)s);
        if (s != 4u && (!uniform_load || waves / 1024.0 < 1.9)) continue;
        const double cost = std::ceil(waves / 1024.0) * (4.06 * (double)s + 8.0);
        if (cost < best - 1e-9) {
            best = cost;
            best_s = s;
        }
    }
    bool rows = true;   // round 3: one DPP row per scenario (af_pregen_arrivals_rows) unless an older variant is asked for
    if (const char* env = std::getenv("AF_PREGEN_SCEN_PER_WAVE")) {
        const uint32_t v = (uint32_t)std::atoi(env);
        if (v == 4u || v == 5u || v == 8u) {
            best_s = v;
            rows = false;
        }
    }
    if (rows) {
        hipLaunchKernelGGL(af_pregen_arrivals_rows, dim3((n + 3u) / 4u), dim3(64), 0, stream, a, stride);
        return 0;
    }
    const dim3 grid((n + best_s - 1u) / best_s);
    if (best_s == 4u) hipLaunchKernelGGL(af_pregen_arrivals<16>, grid, dim3(64), 0, stream, a, stride);
    else if (best_s == 5u) hipLaunchKernelGGL(af_pregen_arrivals<12>, grid, dim3(64), 0, stream, a, stride);
    else hipLaunchKernelGGL(af_pregen_arrivals<8>, grid, dim3(64), 0, stream, a, stride);
    return 0;
}

// second pass: the online counters of the scenarios that start over are cleared first
__global__ void af_zero_online(const KArgs a, uint32_t count) {
    const uint32_t j = blockIdx.x;
    if (j >= count) return;
    const uint32_t sc = a.scen_map[j];
    if (a.online_hist)
        for (uint32_t i = threadIdx.x; i < a.online_hist_bins; i += blockDim.x) a.online_hist[(size_t)sc * a.online_hist_bins + i] = 0u;
    if (a.online_rps)
        for (uint32_t i = threadIdx.x; i < a.online_rps_buckets; i += blockDim.x) a.online_rps[(size_t)sc * a.online_rps_buckets + i] = 0u;
}

// Streams 1 + e: block (x, scenario, edge) draws 256 consecutive messages (coalesced stores).
__global__ void __launch_bounds__(256) af_pregen_edges(const KArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_draw) return;
    const uint32_t slot = blockIdx.y;
    const uint32_t scen = a.scen_map ? a.scen_map[slot] : slot;
    const uint32_t e = blockIdx.z;
    const uint64_t* rec = reinterpret_cast<const uint64_t*>(a.blob) + a.off_edge + af::EREC * e;
    const double mean = ovr_or(a, af::PARAM_EDGE_MEAN, e, scen, af::u2d(rec[0]));
    const double sigma = ovr_or(a, af::PARAM_EDGE_SIGMA, e, scen, af::u2d(rec[1]));
    const double dropout = ovr_or(a, af::PARAM_EDGE_DROPOUT, e, scen, af::u2d(rec[2]));
    const uint32_t dist = (uint32_t)(rec[3] >> 16) & 0xFFu;
    a.draws[((size_t)slot * (1u + a.n_edges) + 1u + e) * a.n_draw + i] =
        af::pre_edge_draw(a.seeds[scen], e, i, dist, mean, sigma, dropout);
}

__global__ void af_probe_kernel(int kind, uint64_t seed, const double* in, const double* in2, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    const double y = in2 ? in2[i] : 0.0;
    double r;
    switch (kind) {
        case 0: {
            const uint32_t packed = (uint32_t)y;
            r = af::uniform_j(seed, packed >> 16, (uint32_t)x, packed & 0xFFFFu);
            break;
        }
        case 1: r = af::af_log(x); break;
        case 2: r = af::af_exp(x); break;
        case 3: r = af::af_norminv(x); break;
        case 4: r = af::af_sqrt(x); break;
        case 5: r = x / y; break;
        case 6: r = (double)af::af_poisson(x, seed, (uint32_t)y >> 16, (uint32_t)y & 0xFFFFu, 0u); break;
        case 7: r = af::af_log_unit(x); break;
        default: r = 0.0;
    }
    out[i] = r;
}

// ---- store-pattern probes (af_probe_store): calibration of rocprofv3's WRITE_SIZE on the store shapes of the
// stage-parallel kernel (MI355X_MICROARCH.md: the counter is calibrated for wide coalesced stores only).  Each wave owns
// a contiguous region and fills it exactly once with
//   wide     : 16 B per lane, 64 lanes = 1 KB per store instruction (the reference pattern);
//   pairs16  : the rqs_clock store of Flow::complete -- 16 B per lane, `lanes` consecutive lanes (a batch of n_sel <= 64
//              completions), the next batch right behind it: chunks of lanes x 16 B that are not 64-B aligned;
//   rows48   : the sample store of Flow::flush_ticks for a 12-word pitch -- 4 B per lane, 60 lanes = five 48-byte rows
//              = 240 B per store instruction, the next five rows right behind them.
__global__ void __launch_bounds__(64) af_probe_store_wide(uint32_t* out, size_t words_per_wave) {
    uint32_t* base = out + (size_t)blockIdx.x * words_per_wave;
    for (size_t w = (size_t)threadIdx.x * 4u; w + 4u <= words_per_wave; w += 256u) af::store4(base + w, 1u, 2u, 3u, 4u);
}
__global__ void __launch_bounds__(64) af_probe_store_pairs16(uint32_t* out, size_t words_per_wave, uint32_t lanes) {
    uint32_t* base = out + (size_t)blockIdx.x * words_per_wave;
    for (size_t w0 = 0; w0 + 4u * lanes <= words_per_wave; w0 += 4u * lanes)
        if (threadIdx.x < lanes) af::store4(base + w0 + 4u * threadIdx.x, 1u, 2u, 3u, 4u);
}
__global__ void __launch_bounds__(64) af_probe_store_rows48(uint32_t* out, size_t words_per_wave) {
    uint32_t* base = out + (size_t)blockIdx.x * words_per_wave;
    for (size_t w0 = 0; w0 + 60u <= words_per_wave; w0 += 60u)
        if (threadIdx.x < 60u) base[w0 + threadIdx.x] = (uint32_t)w0;
}

// ---- host side ---------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(AF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

uint32_t pow2_at_least(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

// kernel variant table: [LDS state][SimPy-order path][log2 lanes per wave][waves per SIMD]
template <bool kLds, bool kFaithful, int WPE>
const void* des_kernel_klog(uint32_t klog) {
    switch (klog) {
        case 0: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 0, WPE>);
        case 1: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 1, WPE>);
        case 2: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 2, WPE>);
        case 3: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 3, WPE>);
        case 4: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 4, WPE>);
        case 5: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 5, WPE>);
        default: return reinterpret_cast<const void*>(af_des_kernel<kLds, kFaithful, 6, WPE>);
    }
}
const void* des_kernel_for(bool lds, bool faithful, uint32_t klog, bool roomy) {
    if (!faithful) return lds ? des_kernel_klog<true, false, 3>(klog) : des_kernel_klog<false, false, 3>(klog);
    if (roomy) return lds ? des_kernel_klog<true, true, 2>(klog) : des_kernel_klog<false, true, 2>(klog);
    return lds ? des_kernel_klog<true, true, 3>(klog) : des_kernel_klog<false, true, 3>(klog);
}

struct af_engine {
    int device = 0;
    bool plan_only = false;   // created with AF_DEVICE_PLAN_ONLY: af_engine_jit_spec and the pure queries only
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    KArgs args{};
    unsigned char* d_blob = nullptr;
    unsigned char* d_state = nullptr;
    size_t state_cap = 0;
    void* d_sweep = nullptr;  // seeds + override tables
    size_t sweep_cap = 0;
    double* d_draws = nullptr;  // pre-generated draws
    size_t draws_cap = 0;
    uint32_t* d_pre_flags = nullptr;
    size_t pre_flags_cap = 0;
    uint64_t* d_tie = nullptr;
    size_t tie_cap = 0;
    uint32_t* d_n_shared = nullptr;
    uint32_t* d_map = nullptr;
    size_t map_cap = 0;
    uint32_t* d_order = nullptr;   // launch order of the stage-parallel kernel for sweeps over the load (heaviest scenario first)
    size_t order_cap = 0;
    // af_engine_run_summarized: the analyzer of the scenarios of the stage-parallel kernel's FULL residency rounds runs on a
    // second stream beside the kernel's last, partial round (round 6)
    const af_summary_t* fused_sum = nullptr;   // non-null while af_engine_run works for af_engine_run_summarized
    const af_outputs_t* fused_out = nullptr;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr;
    bool fused_done = false;                   // the sweep's summary was written during af_engine_run (but for the retry lists)
    uint32_t* d_done_flags = nullptr;          // [n] set by a wave of the stage-parallel kernel when its scenario's outputs are in memory
    size_t done_flags_cap = 0;
    uint32_t* h_done_count = nullptr;          // pinned, host-coherent: finished scenarios (the second stream waits on it: hipStreamWaitValue32)
    uint32_t* d_done_count = nullptr;          // ... its device address
    uint32_t* d_retry = nullptr;               // [2][n + 1]: count, then the scenarios the latency / the series kernel met unfinished
    size_t retry_cap = 0;
    int wait_value_ok = -1;                    // hipDeviceAttributeCanUseStreamWaitValue (-1: not asked yet)
    uint64_t done_ptrs[2] = {0, 0};            // host copy of d_fb[kDoneWords ..] (a member: an asynchronous copy reads it)
    bool shared_instants_likely = false;
    hipModule_t jit_module = nullptr;  // plan-specialised kernels (af_engine_set_kernels), valid for jit_spec only
    hipFunction_t jit_lean = nullptr, jit_order3 = nullptr, jit_order2 = nullptr;
    std::string jit_spec;
    hipModule_t flow_jit_module = nullptr;   // plan-specialised stage-parallel kernel, valid for flow_jit_spec only
    hipFunction_t flow_jit_fn = nullptr;
    std::string flow_jit_spec;
    hipEvent_t ev3 = nullptr, ev4 = nullptr;
    size_t draw_memory_bytes = 0;
    uint32_t request_capacity = 0, fifo_capacity = 0, force_global = 0, lanes_per_wave = 0;
    uint32_t n_lb_edges = 0;
    std::vector<uint32_t> row_of_step;
    af_stats_t stats{};
    // stage-parallel kernel (af_flow.hpp)
    bool flow_ok = false, flow_general_servers = false, flow_chain = false;   // flow_chain: servers feed servers (FEAT_CHAIN)
    uint32_t flow_levels = 1u;   // levels the servers form (1: no server feeds a server)
    uint32_t flow_lb_pos = 0u;   // the LB station runs in front of the servers of this level (0: right behind the client)
    std::string flow_reason;
    uint32_t flow_mode = 0, flow_list_entries = 0, flow_ring_rows = 0;
    aff::FlowArgs fargs{};
    aff::TickTable tick;
    double* d_tick = nullptr;
    double* d_arr = nullptr;       // arrival times of the chunk [n][n_draw]
    size_t arr_cap = 0;
    uint32_t* d_arr_flags = nullptr;
    size_t arr_flags_cap = 0;
    uint32_t* d_fb = nullptr;      // [32]: [0..9] hand-over counters of the first and of the second-chance launch; [16..19] two 64-bit
                                   // words: af_engine_run_summarized's done flags / done counter (kDoneWords), 0 = nobody waits;
                                   // [21..24], what the second chance (d_fb + 5) finds at that offset: always 0
    uint32_t* d_slot = nullptr;    // draw slots of a second pass over a subset
    size_t slot_cap = 0;
    // host copy of what the layout heuristics need
    uint32_t has_lb = 0, cores_max = 1, ram_slots_max = 1;
    std::vector<double> edge_mean, edge_sigma, edge_spike;   // per edge: latency law, largest cumulative spike
    std::vector<double> srv_ram_mb;
    std::vector<uint8_t> edge_dist;
    std::vector<int32_t> lb_edges, srv_out_edge;
    double service_max = 0.0, cpu_max = 0.0;
    int32_t gen_edge = 0, client_edge = 0;
    double rpm_mean = 0.0, users_mean = 0.0, users_sigma = 0.0;
    uint32_t users_dist = 0;
};

namespace {

int validate_plan(const af_plan_t* p) {
    if (!p) return fail(AF_ERR_INVALID, "plan is NULL");
    if (p->abi_version != AF_ABI_VERSION || p->struct_size != sizeof(af_plan_t))
        return fail(AF_ERR_ABI, "af_plan_t ABI mismatch (version or size)");
    if (!(p->total_time > 0.0) || !(p->sample_period > 0.0)) return fail(AF_ERR_INVALID, "bad horizon or sample period");
    if (p->n_edges == 0 || p->n_edges > 255 || p->n_servers > 255) return fail(AF_ERR_INVALID, "edge/server count out of range (1..255 edges, <=255 servers)");
    if (p->gen_out_edge < 0 || (uint32_t)p->gen_out_edge >= p->n_edges) return fail(AF_ERR_INVALID, "generator out edge invalid");
    if (p->client_out_edge < 0 || (uint32_t)p->client_out_edge >= p->n_edges) return fail(AF_ERR_INVALID, "client out edge invalid");
    if (p->has_lb && p->n_lb_edges == 0) return fail(AF_ERR_INVALID, "load balancer without out edges");
    for (uint32_t e = 0; e < p->n_edges; ++e) {
        if (p->edge_target_kind[e] > AF_NODE_SERVER) return fail(AF_ERR_INVALID, "edge target kind invalid");
        if (p->edge_target_kind[e] == AF_NODE_SERVER &&
            (p->edge_target_idx[e] < 0 || (uint32_t)p->edge_target_idx[e] >= p->n_servers))
            return fail(AF_ERR_INVALID, "edge targets unknown server");
        if (p->edge_target_kind[e] == AF_NODE_LB && !p->has_lb) return fail(AF_ERR_INVALID, "edge targets missing LB");
        if (p->edge_dist[e] > AF_DIST_UNIFORM) return fail(AF_ERR_INVALID, "edge distribution invalid");
    }
    for (uint32_t s = 0; s < p->n_servers; ++s) {
        if (p->srv_out_edge[s] < 0 || (uint32_t)p->srv_out_edge[s] >= p->n_edges) return fail(AF_ERR_INVALID, "server out edge invalid");
        if (p->srv_ep_begin[s + 1] <= p->srv_ep_begin[s]) return fail(AF_ERR_INVALID, "server without endpoints");
    }
    for (uint32_t i = 0; i < p->n_lb_edges; ++i)
        if (p->lb_edges[i] < 0 || (uint32_t)p->lb_edges[i] >= p->n_edges) return fail(AF_ERR_INVALID, "LB edge invalid");
    for (uint32_t i = 0; i < p->n_edge_marks; ++i)
        if (p->emark_edge[i] < 0 || (uint32_t)p->emark_edge[i] >= p->n_edges) return fail(AF_ERR_INVALID, "edge mark invalid");
    for (uint32_t i = 0; i < p->n_srv_marks; ++i)
        if (p->smark_lb_edge[i] >= (int32_t)p->n_edges) return fail(AF_ERR_INVALID, "server mark invalid");
    return AF_OK;
}

}  // namespace

namespace {

// Scenario lanes per wave and state placement.  The kernel is latency bound, so few scenarios are
// best spread over MANY narrow waves: fewer event kinds per round in a wave, every SIMD busy,
// several waves per SIMD hiding LDS latency.  Limits: <= 168 VGPRs -> 3 waves/SIMD = 12 per CU;
// LDS-resident state -> 160 KiB per CU.  Cost model fitted to MI355X measurements
// (profiles/r01/lanes_sweep.md): relative time of one wave-round f(lanes), x 2.3 when the state
// lives in HBM, x the number of residency batches.
void choose_lanes(const af_engine* e, uint32_t count, uint32_t blob_bytes, uint64_t bytes_per_lane, uint32_t& kl,
                  bool& lds_state) {
    static const double f_lanes[7] = {1.0, 1.4, 1.65, 2.2, 2.5, 2.4, 2.25};  // 1,2,4,...,64 lanes
    const double n_cu = 256.0, vgpr_waves_per_cu = 12.0;
    double best = 1e300;
    uint32_t best_kl = 4;
    bool best_lds = false;
    for (uint32_t k = 0; k < 7; ++k) {
        const uint32_t cand = 1u << k;
        if (e->lanes_per_wave != 0u && cand != e->lanes_per_wave) continue;
        const double waves_needed = (double)((count + cand - 1u) / cand);
        for (int lds = 1; lds >= 0; --lds) {
            if (lds && e->force_global) continue;
            double per_cu = vgpr_waves_per_cu;
            if (lds) {
                const uint64_t wg_bytes = (uint64_t)blob_bytes + bytes_per_lane * cand;
                if (wg_bytes > kLdsLimit) continue;
                const double fit = (double)(kLdsLimit / wg_bytes);
                per_cu = fit < per_cu ? fit : per_cu;
            }
            const double batches = waves_needed / (n_cu * per_cu);
            const double cost = (batches < 1.0 ? 1.0 : batches) * f_lanes[k] * (lds ? 1.0 : 2.3);
            if (cost < best) {
                best = cost;
                best_kl = cand;
                best_lds = lds != 0;
            }
        }
    }
    kl = best_kl;
    lds_state = best_lds;
}

// Everything a plan-specialised kernel bakes in, as hipcc -D flags (the JIT key).
std::string jit_spec_string(const KArgs& a, bool lds_state, uint32_t klog) {
    char buf[1024];
    std::snprintf(buf, sizeof buf,
                  "-DAF_JIT=1 -DAF_JIT_LDS=%d -DAF_JIT_KLOG=%u -DAF_JIT_METRICS=%u -DAF_JIT_GEN_EDGE=%u -DAF_JIT_CLIENT_EDGE=%u "
                  "-DAF_JIT_N_EDGES=%u -DAF_JIT_N_SERVERS=%u -DAF_JIT_LB_ALGO=%u -DAF_JIT_N_LB=%u -DAF_JIT_N_ROWS=%u "
                  "-DAF_JIT_N_EMARKS=%u -DAF_JIT_N_SMARKS=%u -DAF_JIT_ORDER_ALL=%u -DAF_JIT_OFF_EDGE=%u -DAF_JIT_OFF_SRV=%u "
                  "-DAF_JIT_OFF_EP=%u -DAF_JIT_OFF_ROW=%u -DAF_JIT_OFF_EMARK=%u -DAF_JIT_OFF_SMARK=%u -DAF_JIT_OFF_LB=%u "
                  "-DAF_JIT_BLOB_BYTES=%u -DAF_JIT_CAP=%u -DAF_JIT_FCAP=%u -DAF_JIT_OVR_MASK=%u "
                  "-DAF_JIT_HAS_CLOCK=%d -DAF_JIT_HAS_SAMPLES=%d -DAF_JIT_HAS_ONLINE=%d",
                  lds_state ? 1 : 0, klog, a.metrics_mask, a.gen_out_edge, a.client_out_edge, a.n_edges, a.n_servers, a.lb_algo,
                  a.n_lb_edges, a.n_rows, a.n_edge_marks, a.n_srv_marks, a.every_event_in_order, a.off_edge, a.off_srv, a.off_ep,
                  a.off_row, a.off_emark, a.off_smark, a.off_lb, a.blob_bytes, a.L.cap, a.L.fcap, a.L.ovr_mask,
                  a.clock ? 1 : 0, a.samples ? 1 : 0, (a.online_hist || a.online_rps) ? 1 : 0);
    return buf;
}

// Experiment hook (AF_WAVES_PER_CU=k): ask for more LDS per workgroup than it needs so that at most k waves fit a
// CU.  Measured on MI355X (10 000 LB-2 replicas): capping the stage-parallel kernel at 14 waves per CU -- three
// equal residency rounds instead of 16 + 16 + a partial one -- is SLOWER (81.6 vs 73.7 ms): the partial round already
// runs faster per wave (4 096 scenarios 30.1 ms, 8 192: 58.0 ms, 10 000: 73.7 ms), so the default pads nothing.
uint32_t spread_lds_bytes(uint32_t need_bytes) {
    if (const char* env = std::getenv("AF_WAVES_PER_CU")) {
        const uint32_t k = (uint32_t)std::atoi(env);
        if (k != 0u) {
            const uint32_t b = (kLdsLimit / k) & ~511u;
            return b > need_bytes ? b : need_bytes;
        }
    }
    return need_bytes;
}

uint32_t chunk_size(const af_engine* e, uint32_t n, size_t draw_bytes_per_scen, size_t mem_free) {
    // (one launch per chunk, and every launch ends with a latency-bound tail: chunks are a last resort)
    size_t budget = e->draw_memory_bytes ? e->draw_memory_bytes : (size_t)160 << 30;
    const size_t avail = mem_free + e->draws_cap;
    if (budget > avail / 10u * 6u) budget = avail / 10u * 6u;
    uint32_t chunk = (uint32_t)(budget / draw_bytes_per_scen < 65535u ? budget / draw_bytes_per_scen : 65535u);
    if (chunk == 0) return 0u;
    if (chunk > n) chunk = n;
    const uint32_t n_chunks = (n + chunk - 1u) / chunk;
    return (n + n_chunks - 1u) / n_chunks;  // equal chunks: no short, latency-bound tail launch
}


// ---- stage-parallel kernel: what one af_engine_run launches -------------------------------------------------
// List capacity and tick ring from what will be in flight at the heaviest point of the sweep, the instantiation
// (IPL, FEAT) that covers the launch.  Shared by af_engine_run and af_engine_jit_spec (the plan-specialised build
// bakes exactly this in).
// Does this sweep run on the stage-parallel kernel?  flow_mode 0: whenever the plan is in its range; 2 = the same, but a sweep
// it cannot be sized for is an error instead of a fall-back; 1 = never.
// (Round 3 sent plans with general servers -- several endpoints per server, core re-entry -- there only as sweeps of <= 8
// scenarios: the event-by-event station handed 42 % of the two-endpoint LB-2 scenarios back at T = 600 s and ran at 3 waves
// per CU.  Round 4: shared instants are resolved in the station (Flow::gs_instant: no hand-backs on that plan) and the first
// launch needs 17.6 instead of 41 KB of LDS per wave: 512 / 10 000 / 40 000 scenarios x 600 s take 0.30 / 1.52 / 5.46 s against
// 1.48 / 2.45 / 13.35 s on the next-event kernels -- profiles/r04/gensrv_*.json -- so they take it at every sweep size.)
static bool flow_wanted(const af_engine_t* e, uint32_t n_scenarios) {
    (void)n_scenarios;
    return e->flow_ok && e->flow_mode != 1u;
}

// Sweeps with a users / rpm column: does af_arrival_groups still pay?  Its workgroup is as slow as its heaviest scenario (the
// chain wave walks that scenario's draws; the lighter lanes idle, the producers skip them) and the kernel as slow as its slowest
// workgroup -- one per CU, all resident --, so its time follows the HEAVIEST scenario of the chunk whatever the others are,
// while the row kernel's follows the TOTAL number of draws at about twice the cost per draw.
// Measured (MI355X): config 3 (users 10 .. 1000, heaviest / mean = 1.98): rows 14.8 ms, groups 11.3 ms; config 4: 103 -> 90 ms.
// launch order of a chunk's scenarios for sweeps with a users / rpm column: heaviest first (stable)
static std::vector<uint32_t> heaviest_first(const af_engine_t* e, const af_sweep_t* sweep, uint32_t lo, uint32_t nc) {
    const double* users = nullptr;
    const double* rpm = nullptr;
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        const af_override_t& o = sweep->overrides[k];
        if (o.param == AF_PARAM_GEN_USERS_MEAN) users = o.values + lo;
        else if (o.param == AF_PARAM_GEN_RPM_MEAN) rpm = o.values + lo;
    }
    std::vector<double> load(nc);
    for (uint32_t i = 0; i < nc; ++i) load[i] = (users ? users[i] : e->users_mean) * (rpm ? rpm[i] : e->rpm_mean);
    std::vector<uint32_t> order(nc);
    for (uint32_t i = 0; i < nc; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return load[x] > load[y]; });
    return order;
}
static bool load_spread_suits_groups(const af_engine_t* e, const af_sweep_t* sweep, uint32_t lo, uint32_t nc) {
    const double* users = nullptr;
    const double* rpm = nullptr;
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        const af_override_t& o = sweep->overrides[k];
        if (o.param == AF_PARAM_GEN_USERS_MEAN) users = o.values + lo;
        else if (o.param == AF_PARAM_GEN_RPM_MEAN) rpm = o.values + lo;
    }
    double sum = 0.0, heaviest = 0.0;
    for (uint32_t i = 0; i < nc; ++i) {
        const double load = (users ? users[i] : e->users_mean) * (rpm ? rpm[i] : e->rpm_mean);
        heaviest = load > heaviest ? load : heaviest;
        sum += load;
    }
    return heaviest * (double)nc <= 2.5 * sum;
}

struct FlowPlan {
    aff::FlowLayout FL{}, FL2{};   // first launch; second chance (long lists with send times)
    bool big = false;              // the first launch already runs the long-list instantiation
    bool far = false;              // the tick ring does not reach the slowest message: a FEAT_FAR instantiation
    uint32_t big_caps[4] = {256u, 256u, 256u, 256u};
    uint32_t lds = 0;              // bytes of LDS per wave of the first launch
    uint32_t ipl = 1u, feat = 0u;  // instantiation of the first launch
    bool lean = false;
    bool gen_compact = false;      // general servers, first launch: long-list instantiation WITHOUT send times, lists sized by the load
};

int plan_flow(const af_engine* e, const KArgs& a, const af_sweep_t* sweep, const af_outputs_t* out, FlowPlan& P) {
    const uint32_t n = sweep->n_scenarios;
    aff::FlowLayout& FL = P.FL;
    aff::FlowLayout& FL2 = P.FL2;
    bool& flow_big = P.big;
    bool& flow_far = P.far;
    uint32_t (&big_caps)[4] = P.big_caps;
    uint32_t& flow_lds = P.lds;
    double users = e->users_mean, rpm = e->rpm_mean, ram_mb_scale = 1.0, spike_extra = 0.0;
    uint32_t cores_max = e->cores_max, cores_min = e->cores_max;
    std::vector<double> emean = e->edge_mean;
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        const af_override_t& o = sweep->overrides[k];
        double mx = o.values[0];
        for (uint32_t i = 1; i < n; ++i) mx = o.values[i] > mx ? o.values[i] : mx;
        if (o.param == AF_PARAM_GEN_USERS_MEAN) users = mx;
        else if (o.param == AF_PARAM_GEN_RPM_MEAN) rpm = mx;
        else if (o.param == AF_PARAM_EDGE_MEAN) emean[o.index] = mx;
        else if (o.param == AF_PARAM_SRV_CORES) {
            double mn = o.values[0];
            for (uint32_t i = 1; i < n; ++i) mn = o.values[i] < mn ? o.values[i] : mn;
            cores_max = std::max(cores_max, (uint32_t)mx);
            cores_min = std::min(cores_min, (uint32_t)mn);
        } else if (o.param == AF_PARAM_SRV_RAM_MB) ram_mb_scale = std::max(ram_mb_scale, mx / std::max(e->srv_ram_mb[o.index], 1e-9));
        else if (o.param == AF_PARAM_EMARK_DELTA && mx > 0.0) spike_extra += mx;   // (an upper bound: every swept spike on every hop)
    }
    std::vector<double> edge_spike = e->edge_spike;
    for (double& sp : edge_spike) sp += spike_extra;
    // per-server rings: core releases of the widest server of the sweep, departures of as many requests as its RAM admits
    uint32_t g_ring = e->fargs.L.g_ring, c_ring = std::max(e->fargs.L.c_ring, cores_max);
    if (ram_mb_scale > 1.0) {
        const double lim = a.n_servers <= 4u ? 256.0 : 128.0;
        const double want = std::min(lim, std::ceil((double)g_ring * ram_mb_scale));
        g_ring = aff::pow2_ge((uint32_t)want);
    }
    if (c_ring > 64u) return fail(AF_ERR_CAPACITY, "cpu_cores column: the stage-parallel kernel handles at most 64 cores per server");
    const double sd = e->users_dist == AF_DIST_POISSON ? std::sqrt(users > 0.0 ? users : 0.0) : e->users_sigma;
    const double rate = (users + 4.0 * sd) * rpm / 60.0 + 1e-9;
    auto lat_mean = [&](int32_t ed) {
        const double m = emean[ed], sg = e->edge_sigma[ed];
        switch (e->edge_dist[ed]) {
            case AF_DIST_LOG_NORMAL: return std::exp(m + 0.5 * sg * sg < 50.0 ? m + 0.5 * sg * sg : 50.0);
            case AF_DIST_NORMAL: return (m > 0.0 ? m : 0.0) + 0.4 * sg;
            case AF_DIST_UNIFORM: return 0.5;
            case AF_DIST_POISSON: return m + 0.5;   // whole seconds: what is in flight is at least the next second's worth
            default: return m;
        }
    };
    // the hops of the request path: generator edge, client edge, the slowest LB edge, the slowest server out-edge
    std::vector<int32_t> hops = {e->gen_edge, e->client_edge};
    auto slowest = [&](const std::vector<int32_t>& es) {
        int32_t best = -1;
        for (int32_t ed : es)
            if (best < 0 || lat_mean(ed) + edge_spike[ed] > lat_mean(best) + edge_spike[best]) best = ed;
        return best;
    };
    if (!e->lb_edges.empty()) hops.push_back(slowest(e->lb_edges));
    hops.push_back(slowest(e->srv_out_edge));
    // time in a server: service + the M/D/1 wait for a core at the heaviest load of the sweep
    const double n_active = e->has_lb ? (double)(e->lb_edges.size() > 1 && a.n_srv_marks ? e->lb_edges.size() - 1 : e->lb_edges.size()) : 1.0;
    const double rho = rate / n_active * e->cpu_max / (double)cores_min;
    const double wait = rho < 0.9 ? rho * e->cpu_max / (2.0 * (1.0 - rho)) : 20.0 * e->cpu_max + 1.0;
    const double in_server = e->service_max + 4.0 * wait;
    // messages pending at a station ~ rate x time in flight towards it (the completion list also holds the server time)
    // (a spiked edge: the station behind it runs ahead by the spike while it lasts -- Flow::send_floor -- but when it
    // ends, the messages still in flight and the ones sent after it interleave: rate x spike of them wait there, and
    // the servers then work off that burst, so as many wait for their departure in the completion list)
    double pend = 0.0, burst = 0.0;
    std::vector<double> pend_of(hops.size(), 0.0);
    for (size_t h = 0; h < hops.size(); ++h) {
        const double fly = lat_mean(hops[h]) + edge_spike[hops[h]] + (h + 1 == hops.size() ? in_server : 0.0);
        burst = std::fmax(burst, rate * edge_spike[hops[h]]);
        pend_of[h] = rate * fly;
        pend = std::fmax(pend, rate * fly);
    }
    if (burst > 16.0)   // (millisecond spikes -- BASELINE config 4 -- change nothing a 64-entry list would notice)
        pend_of.back() = std::fmax(pend_of.back(), burst + rate * (lat_mean(hops.back()) + in_server));
    pend = std::fmax(pend, pend_of.back());
    // capacities of the four station lists for the FEAT_BIGLIST instantiation (hops -> lists: generator edge -> 0,
    // client edge -> 1 with a load balancer else 2, LB edges -> 2, server out-edges -> 3)
    {
        for (uint32_t s = 0; s < 4u; ++s) big_caps[s] = (s == 1u && !e->has_lb) ? 64u : 256u;   // (no LB: its list stays empty)
        for (size_t h = 0; h < hops.size(); ++h) {
            const uint32_t s = h == 0 ? 0u : h + 1 == hops.size() ? 3u : (h == 1 && e->has_lb && e->flow_lb_pos == 0u) ? 1u : 2u;
            const double want = 1.5 * pend_of[h] + 128.0;
            const uint32_t c = want < 16384.0 ? ((uint32_t)want + 63u) & ~63u : 16384u;   // any multiple of 64
            if (c > big_caps[s]) big_caps[s] = c;
        }
        if (e->flow_lb_pos != 0u && big_caps[2] > big_caps[1]) big_caps[1] = big_caps[2];   // (servers in front of the LB: the same load passes its list)
    }
    // A list only has to leave ROOM: 64 - pending new messages fit per round.  Larger lists cost LDS (occupancy) and
    // ranking work on every round of every scenario (measured on the config-3 grid: 64 entries 122 ms, 128 entries
    // 144 ms, no hand-backs either way); an overflow costs one scenario a second run.  `pend` is the MEAN at the
    // heaviest point of the sweep: half a list of pending messages still leaves half a batch of room.
    uint32_t entries = e->flow_list_entries;
    if (entries == 0u) {
        entries = pend <= 32.0 ? 64u : pend <= 96.0 ? 128u : 256u;
        flow_big = pend > 200.0;   // more than register-resident lists hold: the whole launch on the long-list instantiation
    }
    // What the tick ring has to reach past its window.  With FEAT_FAR: the time a request spends INSIDE a server (its
    // queue / step / RAM intervals are entered when it arrives); a delivery the ring does not reach is entered by the
    // receiving station (af_flow.hpp, "Sampled series").  Without: the in-flight time of the slowest message of the
    // sweep (~1e-11 per request) -- the largest single hop at that quantile, the other hops at mean + 3 sd, spikes.
    auto lat_sd = [&](int32_t ed) {
        const double m = emean[ed], sg = e->edge_sigma[ed];
        switch (e->edge_dist[ed]) {
            case AF_DIST_LOG_NORMAL: {
                const double v = sg * sg < 50.0 ? sg * sg : 50.0;
                return lat_mean(ed) * std::sqrt(std::exp(v) - 1.0);
            }
            case AF_DIST_NORMAL: return sg;
            case AF_DIST_UNIFORM: return 0.29;
            case AF_DIST_POISSON: return std::sqrt(m > 0.0 ? m : 0.0);
            default: return m;
        }
    };
    auto lat_q = [&](int32_t ed) {   // a transit time one message in ~1e11 exceeds
        const double m = emean[ed], sg = e->edge_sigma[ed];
        switch (e->edge_dist[ed]) {
            case AF_DIST_LOG_NORMAL: return std::exp(m + 6.7 * sg < 50.0 ? m + 6.7 * sg : 50.0);
            case AF_DIST_NORMAL: return (m > 0.0 ? m : 0.0) + 6.7 * sg;
            case AF_DIST_UNIFORM: return 1.0;
            case AF_DIST_POISSON: return m + 7.0 * std::sqrt(m > 0.0 ? m : 0.0) + 12.0;   // (Poisson tail beyond 1e-11)
            default: return 25.3 * m;
        }
    };
    double tail_full = in_server;
    {
        size_t worst = 0;
        for (size_t h = 1; h < hops.size(); ++h)
            if (lat_q(hops[h]) > lat_q(hops[worst])) worst = h;
        for (size_t h = 0; h < hops.size(); ++h)
            tail_full += edge_spike[hops[h]] + (h == worst ? lat_q(hops[h]) : lat_mean(hops[h]) + 3.0 * lat_sd(hops[h]));
    }
    const double tail = in_server;
    uint32_t rows = e->flow_ring_rows, win_rows = 0u;
    const uint32_t pitch = a.series_pitch;
    const double tail_rows = std::ceil(tail / a.sample_period) + 2.0;
    if (out->samples == nullptr) {
        rows = 0u;    // no series: neither ring nor rows are touched
    } else if (rows == AF_FLOW_RING_IN_HBM) {
        rows = 0u;
    } else if (rows == 0u) {
        // the ring covers the generator's window (enough ticks for a full batch of arrivals at the heaviest load,
        // at least 8) + the server time; 8 KB of LDS keep 16 waves per CU, 24 KB are the limit before HBM takes over
        const double want_win = std::fmin(std::fmax(std::ceil(96.0 / rate / a.sample_period), 8.0), 4096.0);
        const uint32_t cap_pref = aff::pow2_ge(8u * 1024u / (pitch * 4u) + 1u) / 2u, cap_max = aff::pow2_ge(24u * 1024u / (pitch * 4u) + 1u) / 2u;
        if (tail_rows + 8.0 > (double)cap_max) {
            rows = 0u;   // a request stays in its server longer than any ring that fits reaches: differences in HBM
        } else {
            rows = aff::pow2_ge((uint32_t)(tail_rows + want_win));
            if (rows > cap_pref) rows = cap_pref >= aff::pow2_ge((uint32_t)(tail_rows + 8.0)) ? cap_pref : aff::pow2_ge((uint32_t)(tail_rows + 8.0));
            if (rows < 16u) rows = 16u;
            // a window of two or three batches binds less often: take it where it costs no occupancy (16 waves per CU
            // leave 10 KB each)
            if (!flow_big) {
                const uint32_t no_ring = a.blob_bytes + aff::make_flow_layout(entries, 0u, g_ring, c_ring, a.n_edges,
                                                                              a.n_servers, a.n_edge_marks).n_words * 8u;
                while (rows < aff::pow2_ge((uint32_t)(tail_rows + 2.0 * want_win)) && no_ring + 2u * rows * pitch * 4u <= 10u * 1024u) rows *= 2u;
            }
        }
    } else {
        rows = aff::pow2_ge(rows);
    }
    if (rows != 0u) {
        const double full_rows = std::ceil(tail_full / a.sample_period) + 2.0;
        flow_far = (double)rows - full_rows < std::fmin(8.0, (double)(rows / 2u)) || std::getenv("AF_FLOW_FORCE_FAR") != nullptr;   // (env: experiment hook)
        const double w = (double)rows - (flow_far ? tail_rows : full_rows);
        win_rows = w >= (double)(rows / 2u) ? (uint32_t)w : rows / 2u;   // an explicit small ring: half of it, overflow -> hand-back
    }
    // long lists have to fit the LDS of a compute unit next to everything else: halve the longest until they do
    const bool gen_srv = e->flow_general_servers;   // several endpoints per server / programs that come back to the core queue
    auto big_layout = [&](uint32_t ring) {
        aff::FlowLayout L = aff::make_flow_layout(0u, ring, g_ring, c_ring, a.n_edges, a.n_servers, a.n_edge_marks, true, big_caps, gen_srv);
        while (a.blob_bytes + L.n_words * 8u > kLdsLimit) {
            uint32_t m = 0;
            for (uint32_t s = 1; s < 4u; ++s)
                if (big_caps[s] > big_caps[m]) m = s;
            if (big_caps[m] <= 256u) break;
            big_caps[m] = (big_caps[m] / 2u + 63u) & ~63u;
            L = aff::make_flow_layout(0u, ring, g_ring, c_ring, a.n_edges, a.n_servers, a.n_edge_marks, true, big_caps, gen_srv);
        }
        return L;
    };
    // General servers, first launch (round 4): the station is one busy lane per server -- a chain of dependent LDS round trips --
    // so what the launch needs is WAVES per compute unit, i.e. little LDS per wave: lists of what the load needs (>= 128
    // entries: a batch of 64 must find room) without the send times of the second-chance form, instead of 4 x 256 x 3 words:
    // 41 -> ~20 KB per wave, 3 -> 7 waves per CU.  What overflows or ties is caught by the second chance (FL2), as ever.
    P.gen_compact = gen_srv && !flow_big && std::getenv("AF_FLOW_GEN_ROBUST_FIRST") == nullptr;   // (env: measurement hook)
    if (gen_srv) flow_big = true;   // (the general server station exists in the long-list instantiations only)
    if (P.gen_compact) {
        uint32_t caps1[4];
        // (64 entries where little is pending -- what the tandem kernel's lists hold for the same load --, 128 for the completion
        // list: it takes a whole round's departures of every server at once)
        for (uint32_t s = 0; s < 4u; ++s) caps1[s] = s == 3u ? 128u : 64u;
        for (size_t h = 0; h < hops.size(); ++h) {
            const uint32_t s = h == 0 ? 0u : h + 1 == hops.size() ? 3u : (h == 1 && e->has_lb && e->flow_lb_pos == 0u) ? 1u : 2u;
            const double want = pend_of[h] <= 24.0 ? 64.0 : 1.5 * pend_of[h] + 64.0;
            const uint32_t c = want < 1024.0 ? ((uint32_t)want + 63u) & ~63u : 1024u;
            if (c > caps1[s]) caps1[s] = c;
        }
        if (rows * pitch * 4u > 4u * 1024u) {
            rows = 0u;
            win_rows = 0u;
        }
        // (servers in front of the LB: its list takes a whole round's departures of the server that feeds it, like the completion list)
        // (servers that feed servers: the ONE server list holds the messages bound for every level)
        if (e->flow_chain) caps1[2] = std::max(caps1[2], 128u);
        if (e->flow_lb_pos != 0u) caps1[1] = std::max({caps1[1], caps1[2], 128u});
        FL = aff::make_flow_layout(0u, rows, g_ring, c_ring, a.n_edges, a.n_servers, a.n_edge_marks, false, caps1, true);
    } else if (flow_big) {
        if (rows * pitch * 4u > 8u * 1024u) {   // the lists need the LDS more than the tick ring does
            rows = 0u;
            win_rows = 0u;
        }
        FL = big_layout(rows);
    } else {
        FL = aff::make_flow_layout(entries, rows, g_ring, c_ring, a.n_edges, a.n_servers, a.n_edge_marks);
    }
    FL.win_rows = win_rows;
    FL2 = big_layout(0u);   // second chance: tick differences in HBM (no reach limit)
    FL2.win_rows = 0u;
    flow_lds = a.blob_bytes + FL.n_words * 8u;
    if (flow_lds > kLdsLimit) return fail(AF_ERR_CAPACITY, "flow kernel layout exceeds the LDS of a compute unit");

    // the leanest instantiation that covers this launch (a compiled-in feature costs wave-uniform registers)
    const bool lc = e->fargs.lb_least_connections != 0u;
    const bool has_online = out->online_hist != nullptr || out->online_rps != nullptr;
    const bool ring_ok = FL.ring_rows != 0u || out->samples == nullptr;
    const bool lean = a.n_edge_marks == 0u && a.n_srv_marks == 0u && !has_online && ring_ok;
    // injected spikes / outages, but neither the kernel-side summary nor tick differences in HBM (BASELINE config 4)
    const bool marks_only = !lean && !has_online && ring_ok;
    constexpr uint32_t kRobust = aff::FEAT_ALL | aff::FEAT_TIEBREAK | aff::FEAT_BIGLIST;
    const bool chain = e->flow_chain;
    P.lean = lean && !flow_big && !lc && !chain;
    if (P.gen_compact) {
        P.ipl = 1u;
        P.feat = aff::FEAT_ALL | aff::FEAT_BIGLIST | aff::FEAT_GENSRV | (lc ? (uint32_t)aff::FEAT_LC : 0u) | (chain ? (uint32_t)aff::FEAT_CHAIN : 0u);
    } else if (flow_big) {
        P.ipl = 1u;
        P.feat = kRobust | (lc ? (uint32_t)aff::FEAT_LC : 0u) | (gen_srv ? (uint32_t)aff::FEAT_GENSRV : 0u) | (chain ? (uint32_t)aff::FEAT_CHAIN : 0u);
    } else if (chain) {   // one generic instantiation per list length (every optional feature in); the plan-specialised build is the same FEAT
        P.ipl = FL.cap == 64u ? 1u : FL.cap == 128u ? 2u : 4u;
        P.feat = aff::FEAT_ALL | aff::FEAT_CHAIN | (lc ? (uint32_t)aff::FEAT_LC : 0u);
    } else if (lc) {
        P.ipl = FL.cap == 64u ? 1u : FL.cap == 128u ? 2u : 4u;
        P.feat = aff::FEAT_ALL | aff::FEAT_LC;
    } else if (FL.cap == 64u || FL.cap == 128u) {
        P.ipl = FL.cap / 64u;
        P.feat = lean ? (flow_far ? (uint32_t)aff::FEAT_FAR : 0u) : marks_only ? (uint32_t)(aff::FEAT_MARKS | aff::FEAT_FAR) : (uint32_t)aff::FEAT_ALL;
    } else {
        P.ipl = 4u;
        P.feat = aff::FEAT_ALL;
    }
    return AF_OK;
}

// the library's own (generic) instantiation for (ipl, feat); nullptr = not built
const void* flow_kernel_for(uint32_t ipl, uint32_t feat) {
    constexpr uint32_t kRobust = aff::FEAT_ALL | aff::FEAT_TIEBREAK | aff::FEAT_BIGLIST, kLC = aff::FEAT_LC, kAll = aff::FEAT_ALL;
    constexpr uint32_t kFar = aff::FEAT_FAR, kMarksFar = aff::FEAT_MARKS | aff::FEAT_FAR;
#define AF_FLOW_CASE(I, F) if (ipl == (I) && feat == (F)) return reinterpret_cast<const void*>(af_flow_kernel<(I), (F)>)
    AF_FLOW_CASE(1u, kRobust | kLC);
    AF_FLOW_CASE(1u, kRobust);
    AF_FLOW_CASE(1u, kRobust | kLC | (uint32_t)aff::FEAT_GENSRV);
    AF_FLOW_CASE(1u, kRobust | (uint32_t)aff::FEAT_GENSRV);
    AF_FLOW_CASE(1u, kAll | (uint32_t)aff::FEAT_BIGLIST | (uint32_t)aff::FEAT_GENSRV);          // general servers, first launch (compact lists)
    AF_FLOW_CASE(1u, kAll | (uint32_t)aff::FEAT_BIGLIST | (uint32_t)aff::FEAT_GENSRV | kLC);
    AF_FLOW_CASE(1u, kAll | (uint32_t)aff::FEAT_BIGLIST | (uint32_t)aff::FEAT_GENSRV | (uint32_t)aff::FEAT_CHAIN);   // general servers in tiers
    AF_FLOW_CASE(1u, kRobust | (uint32_t)aff::FEAT_GENSRV | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(1u, kAll | (uint32_t)aff::FEAT_BIGLIST | (uint32_t)aff::FEAT_GENSRV | kLC | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(1u, kRobust | (uint32_t)aff::FEAT_GENSRV | kLC | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(1u, kRobust | (uint32_t)aff::FEAT_CHAIN);   // servers that feed servers: the second-chance form and one per list length
    AF_FLOW_CASE(1u, kAll | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(2u, kAll | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(4u, kAll | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(1u, kRobust | kLC | (uint32_t)aff::FEAT_CHAIN);   // ... behind a least-connections LB
    AF_FLOW_CASE(1u, kAll | kLC | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(2u, kAll | kLC | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(4u, kAll | kLC | (uint32_t)aff::FEAT_CHAIN);
    AF_FLOW_CASE(1u, kAll | kLC);
    AF_FLOW_CASE(2u, kAll | kLC);
    AF_FLOW_CASE(4u, kAll | kLC);
    AF_FLOW_CASE(1u, kFar);
    AF_FLOW_CASE(1u, 0u);
    AF_FLOW_CASE(1u, kMarksFar);
    AF_FLOW_CASE(1u, kAll);
    AF_FLOW_CASE(2u, kFar);
    AF_FLOW_CASE(2u, 0u);
    AF_FLOW_CASE(2u, kMarksFar);
    AF_FLOW_CASE(2u, kAll);
    AF_FLOW_CASE(4u, kAll);
#undef AF_FLOW_CASE
    return nullptr;
}

// Everything the plan-specialised stage-parallel kernel bakes in, as hipcc -D flags (the JIT key).
std::string flow_jit_spec_string(const af_engine* e, const FlowPlan& P, const af_outputs_t* out, bool has_ovr) {
    const aff::FlowArgs& f = e->fargs;
    const aff::FlowLayout& L = P.FL;
    auto bits = [](double v) {
        uint64_t u;
        std::memcpy(&u, &v, 8);
        return (unsigned long long)u;
    };
    uint32_t dist_all = e->edge_dist.empty() ? 255u : e->edge_dist[0];     // the ONE latency law of every edge, else 255
    for (uint8_t d : e->edge_dist)
        if (d != dist_all) dist_all = 255u;
    // (measured, round 3: the constant pays for the laws behind the call -- BASELINE config 5, log-normal: 245.4 -> 235.5 ms per
    // 50 000 replicas -- but NOT for the exponential law, whose variate is inline anyway: without the call site the compiler
    // allocates 72 instead of 110 VGPRs and the kernel is 6 % SLOWER (48.5 -> 51.6 ms on config 2), five waves per SIMD included)
    if ((dist_all == AF_DIST_EXPONENTIAL && !std::getenv("AF_FLOW_DIST_CONST_EXP")) || std::getenv("AF_FLOW_NO_DIST_CONST")) dist_all = 255u;
    // waves per SIMD the LDS of this launch admits (160 KB per compute unit, four SIMDs): the register budget to compile for
    uint32_t wpe = 4u;
    if (const char* env = std::getenv("AF_FLOW_JIT_WPE")) {   // (measurement hook; anything outside 1..8 is ignored)
        const int v = std::atoi(env);
        if (v >= 1 && v <= 8) wpe = (uint32_t)v;
    } else {
        const uint32_t lds_alloc = (P.lds + 511u) & ~511u, per_cu = lds_alloc ? (160u * 1024u) / lds_alloc : 16u;
        wpe = (per_cu + 3u) / 4u;
        wpe = wpe < 2u ? 2u : wpe > 4u ? 4u : wpe;
        // general servers: the kernel also holds gen_servers_par's walk and top-c arrays -- 128 registers (four waves) would
        // spill them; the generic instantiation and round 4's measurements are at three
        if ((P.feat & (uint32_t)aff::FEAT_GENSRV) != 0u && wpe > 3u) wpe = 3u;
    }
    char buf[2048];
    std::snprintf(buf, sizeof buf,
                  "-DAF_JIT=1 -DAF_FLOW_JIT=1 -DAF_FJ_IPL=%u -DAF_FJ_FEAT=%u -DAF_FJ_TOTAL_TIME=0x%llxull -DAF_FJ_PERIOD=0x%llxull "
                  "-DAF_FJ_INV_PERIOD=0x%llxull -DAF_FJ_TICK_EPS=0x%llxull -DAF_FJ_METRICS=%u -DAF_FJ_GEN_EDGE=%u -DAF_FJ_CLIENT_EDGE=%u "
                  "-DAF_FJ_N_EDGES=%u -DAF_FJ_N_SERVERS=%u -DAF_FJ_HAS_LB=%u -DAF_FJ_N_LB=%u -DAF_FJ_N_EMARKS=%u -DAF_FJ_N_SMARKS=%u "
                  "-DAF_FJ_LC=%u -DAF_FJ_MAX_PRE=%u -DAF_FJ_MAX_CPU=%u -DAF_FJ_MAX_POST=%u -DAF_FJ_OFF_EDGE=%u -DAF_FJ_OFF_SRV=%u "
                  "-DAF_FJ_OFF_EP=%u -DAF_FJ_OFF_ROW=%u -DAF_FJ_OFF_EMARK=%u -DAF_FJ_OFF_SMARK=%u -DAF_FJ_OFF_LB=%u -DAF_FJ_BLOB_BYTES=%u "
                  "-DAF_FJ_N_TICKS=%u -DAF_FJ_HAS_CLOCK=%d -DAF_FJ_HAS_SAMPLES=%d -DAF_FJ_HAS_ONLINE=%d -DAF_FJ_HAS_OVR=%d -DAF_FJ_DIST_ALL=%u -DAF_FJ_RAM_SCALE=%.1f "
                  "-DAF_FJ_N_LEVELS=%u -DAF_FJ_LB_POS=%u -DAF_FJ_WPE=%u -DAF_FJ_LAYOUT=%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u",
                  P.ipl, P.feat | (std::getenv("AF_FLOW_PROF") ? (uint32_t)aff::FEAT_PROF : 0u), bits(f.total_time), bits(f.sample_period), bits(f.inv_period), bits(f.tick_eps), f.metrics_mask,
                  f.gen_out_edge, f.client_out_edge, f.n_edges, f.n_servers, f.has_lb, f.n_lb_edges, f.n_edge_marks, f.n_srv_marks,
                  f.lb_least_connections, f.max_pre, f.max_cpu, f.max_post, f.off_edge, f.off_srv, f.off_ep, f.off_row, f.off_emark,
                  f.off_smark, f.off_lb, f.blob_bytes, f.n_ticks, out->clock ? 1 : 0, out->samples ? 1 : 0,
                  (out->online_hist || out->online_rps) ? 1 : 0, has_ovr ? 1 : 0, dist_all, f.ram_scale, e->flow_levels, e->flow_lb_pos, wpe,
                  L.cap, L.ring_rows, L.win_rows, L.g_ring, L.c_ring, L.pitch, L.list_arrays, L.off_spike, L.off_list, L.off_aux, L.off_aux3,
                  L.off_out, L.off_sorted, L.off_hist, L.off_seg, L.off_fr, L.off_gr, L.off_cnt, L.off_ring, L.n_words, L.cap_of[0],
                  L.cap_of[1], L.cap_of[2], L.cap_of[3], L.off_list_of[0], L.off_list_of[1], L.off_list_of[2], L.off_list_of[3], L.off_eb, L.off_gsrv);
    std::string spec(buf);
    // Philox's key schedule per call site instead of hoisted (af_flow.hpp: Flow::edge_draw) where it was measured to pay -- the
    // lean form and the forms with timeline marks (configs 2 / 4: 38.2 -> 37.9, 670 -> 653 ms); it costs on the others (FAR alone,
    // config 3: 55.7 -> 56.4; general servers, config 6: 281 -> 293; config 5: nothing): profiles/r05/philox_keys_ab.txt
    const bool keys = (P.feat == 0u || (P.feat & aff::FEAT_MARKS) != 0u) && (P.feat & aff::FEAT_GENSRV) == 0u;
    if (keys && !std::getenv("AF_FLOW_KEYS_HOISTED")) spec += " -DAF_FJ_KEYS_PER_CALL=1";
    return spec;
}
}  // namespace

extern "C" {

int af_abi_version(void) { return AF_ABI_VERSION; }

const char* af_last_error(void) { return g_err.c_str(); }

uint32_t af_tick_count(double sample_period, double total_time) {
    if (!(sample_period > 0.0)) return 0;
    uint32_t n = 0;
    double t = 0.0 + sample_period;
    while (t < total_time) {
        n += 1;
        t = t + sample_period;
    }
    return n;
}

uint32_t af_series_count(const af_plan_t* plan) { return plan ? plan->n_edges + 3u * plan->n_servers : 0u; }

uint32_t af_series_pitch(const af_plan_t* plan) { return (af_series_count(plan) + 3u) & ~3u; }

int af_engine_create(const af_plan_t* plan, int device, const af_engine_options_t* opts, af_engine_t** out) {
    if (!out) return fail(AF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = validate_plan(plan)) return rc;
    // AF_DEVICE_PLAN_ONLY: a planning-only engine -- everything af_engine_jit_spec needs (plan facts, layout heuristics)
    // and no device state: it cannot run, and no HIP call is made on its behalf (build machines without a GPU)
    const bool plan_only = device == AF_DEVICE_PLAN_ONLY;
    if (!plan_only) {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return fail(AF_ERR_NO_DEVICE, "no HIP device visible: the asyncflow_amd engine has no CPU fallback");
        if (device < 0 || device >= n_dev) return fail(AF_ERR_NO_DEVICE, "device index out of range");
        HIP_TRY(hipSetDevice(device));
    }

    af_engine* e = new af_engine();
    e->device = device;
    e->plan_only = plan_only;
    KArgs& a = e->args;
    a.total_time = plan->total_time;
    a.sample_period = plan->sample_period;
    a.gen_users_mean = plan->gen_users_mean;
    a.gen_users_sigma = plan->gen_users_sigma;
    a.gen_rpm_mean = plan->gen_rpm_mean;
    a.gen_window_s = plan->gen_window_s;
    a.metrics_mask = plan->metrics_mask;
    a.gen_users_dist = plan->gen_users_dist;
    a.gen_out_edge = plan->gen_out_edge;
    a.client_out_edge = plan->client_out_edge;
    a.n_edges = plan->n_edges;
    a.n_servers = plan->n_servers;
    a.lb_algo = plan->lb_algo;
    a.n_lb_edges = plan->n_lb_edges;
    a.n_edge_marks = plan->n_edge_marks;
    a.n_srv_marks = plan->n_srv_marks;
    a.every_event_in_order = af::every_event_in_order(*plan) ? 1u : 0u;
    a.n_series = plan->n_edges + 3u * plan->n_servers;
    a.series_pitch = (a.n_series + 3u) & ~3u;

    af::PackedPlan pk;
    const std::string why = af::pack_plan(*plan, pk);
    if (!why.empty()) {
        delete e;
        return fail(AF_ERR_INVALID, why);
    }
    a.n_rows = pk.n_rows;
    a.off_edge = pk.off_edge;
    a.off_srv = pk.off_srv;
    a.off_ep = pk.off_ep;
    a.off_row = pk.off_row;
    a.off_emark = pk.off_emark;
    a.off_smark = pk.off_smark;
    a.off_lb = pk.off_lb;
    a.blob_bytes = (uint32_t)(pk.words.size() * 8u);
    e->row_of_step = pk.row_of_step;
    const std::vector<uint64_t>& blob = pk.words;

    e->request_capacity = opts && opts->request_capacity ? opts->request_capacity : 64u;
    e->fifo_capacity = pow2_at_least(opts && opts->fifo_capacity ? opts->fifo_capacity : 32u);
    e->force_global = opts ? opts->force_global_state : 0u;
    e->lanes_per_wave = opts ? opts->lanes_per_wave : 0u;
    e->draw_memory_bytes = opts ? (size_t)opts->draw_memory_mb << 20 : 0;
    e->shared_instants_likely = (opts && opts->expect_shared_instants != 0u) || a.every_event_in_order != 0u;
    if (e->lanes_per_wave & (e->lanes_per_wave - 1u)) {
        delete e;
        return fail(AF_ERR_INVALID, "lanes_per_wave must be 0 (auto) or a power of two <= 64");
    }
    if (e->lanes_per_wave > 64u) e->lanes_per_wave = 64u;
    // (a server's wait queues have a 32-bit word each since round 6: the cap is what a scenario's state may cost in HBM --
    // 32 bytes per waiter slot and server and queue pair; af_engine_run sizes its pieces by it)
    if (e->request_capacity > 65535u || e->fifo_capacity > AF_MAX_FIFO_CAPACITY) {
        delete e;
        return fail(AF_ERR_CAPACITY, "request_capacity must be <= 65535 and fifo_capacity <= 1048576");
    }
    if (a.blob_bytes > kLdsLimit / 2) {
        delete e;
        return fail(AF_ERR_CAPACITY, "plan too large for the LDS-resident plan blob");
    }

    // ---- stage-parallel kernel: is the plan in its range?  tick table, plan facts for the layout heuristics
    e->flow_mode = opts ? opts->flow_mode : 0u;
    e->flow_list_entries = opts ? opts->flow_list_entries : 0u;
    e->flow_ring_rows = opts ? opts->flow_ring_rows : 0u;
    if (e->flow_list_entries != 0u && e->flow_list_entries != 64u && e->flow_list_entries != 128u && e->flow_list_entries != 256u) {
        delete e;
        return fail(AF_ERR_INVALID, "flow_list_entries must be 0 (auto), 64, 128 or 256");
    }
    e->flow_reason = aff::flow_ineligible_reason(*plan);
    e->flow_ok = e->flow_reason.empty();
    e->flow_general_servers = e->flow_ok && aff::flow_needs_general_servers(*plan);
    e->flow_chain = e->flow_ok && aff::flow_needs_chain(*plan);
    e->flow_levels = e->flow_chain ? aff::flow_server_levels(*plan, nullptr) : 1u;
    e->flow_lb_pos = e->flow_chain ? aff::flow_lb_position(*plan) : 0u;
    e->has_lb = plan->has_lb;
    e->gen_edge = plan->gen_out_edge;
    e->client_edge = plan->client_out_edge;
    e->rpm_mean = plan->gen_rpm_mean;
    e->users_mean = plan->gen_users_mean;
    e->users_sigma = plan->gen_users_sigma;
    e->users_dist = plan->gen_users_dist;
    e->edge_mean.assign(plan->edge_mean, plan->edge_mean + plan->n_edges);
    e->edge_sigma.assign(plan->edge_sigma, plan->edge_sigma + plan->n_edges);
    e->edge_dist.assign(plan->edge_dist, plan->edge_dist + plan->n_edges);
    e->lb_edges.assign(plan->lb_edges, plan->lb_edges + plan->n_lb_edges);
    e->srv_out_edge.assign(plan->srv_out_edge, plan->srv_out_edge + plan->n_servers);
    e->srv_ram_mb.assign(plan->srv_ram_mb, plan->srv_ram_mb + plan->n_servers);
    e->edge_spike.assign(plan->n_edges, 0.0);
    {
        std::vector<double> acc(plan->n_edges, 0.0);
        for (uint32_t i = 0; i < plan->n_edge_marks; ++i) {
            const int32_t ed = plan->emark_edge[i];
            acc[ed] += plan->emark_delta[i];
            if (acc[ed] > e->edge_spike[ed]) e->edge_spike[ed] = acc[ed];
        }
    }
    for (uint32_t sv = 0; sv < plan->n_servers; ++sv) {
        if (plan->srv_cores[sv] > e->cores_max) e->cores_max = plan->srv_cores[sv];
        for (uint32_t ep = plan->srv_ep_begin[sv]; ep < plan->srv_ep_begin[sv + 1]; ++ep) {
            double svc = 0.0, cpu = 0.0;
            for (uint32_t i = plan->ep_step_begin[ep]; i < plan->ep_step_begin[ep + 1]; ++i) {
                svc += plan->step_time[i];
                if (plan->step_kind[i] == AF_STEP_CPU) cpu += plan->step_time[i];
            }
            if (svc > e->service_max) e->service_max = svc;
            if (cpu > e->cpu_max) e->cpu_max = cpu;
        }
    }
    if (e->flow_ok) {
        e->tick = aff::make_tick_table(plan->sample_period, plan->total_time);
        aff::FlowArgs& f = e->fargs;
        f.total_time = plan->total_time;
        f.sample_period = plan->sample_period;
        f.inv_period = e->tick.inv_period;
        f.tick_eps = e->tick.eps;
        f.metrics_mask = plan->metrics_mask;
        f.gen_out_edge = (uint32_t)plan->gen_out_edge;
        f.client_out_edge = (uint32_t)plan->client_out_edge;
        f.n_edges = plan->n_edges;
        f.n_servers = plan->n_servers;
        f.has_lb = plan->has_lb;
        f.n_lb_edges = plan->n_lb_edges;
        f.lb_least_connections = aff::lc_edges(*plan) != 0u ? 1u : 0u;
        f.n_edge_marks = plan->n_edge_marks;
        f.n_srv_marks = plan->n_srv_marks;
        aff::flow_step_maxima(*plan, f.max_pre, f.max_cpu, f.max_post);
        f.ram_scale = aff::flow_ram_scale(*plan);
        f.ram_unit = 1.0 / f.ram_scale;
        f.off_edge = pk.off_edge; f.off_srv = pk.off_srv; f.off_ep = pk.off_ep; f.off_row = pk.off_row;
        f.off_emark = pk.off_emark; f.off_smark = pk.off_smark; f.off_lb = pk.off_lb;
        f.blob_bytes = a.blob_bytes;
        f.n_ticks = (uint32_t)e->tick.t.size();
        f.L = aff::choose_flow_layout(*plan, 1u, 64u);   // g_ring / c_ring; entries and rows are chosen per run
    }

    if (plan_only) {
        *out = e;
        return AF_OK;
    }
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err == hipSuccess && e->flow_ok) err = hipMalloc((void**)&e->d_tick, (e->tick.t.size() + 1u) * 8u);
    if (err == hipSuccess && e->flow_ok && !e->tick.t.empty())
        err = hipMemcpy(e->d_tick, e->tick.t.data(), e->tick.t.size() * 8u, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc((void**)&e->d_fb, 32u * 4u);
    if (err == hipSuccess) err = hipEventCreate(&e->ev0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev1);
    if (err == hipSuccess) err = hipEventCreate(&e->ev2);
    if (err == hipSuccess) err = hipEventCreate(&e->ev3);
    if (err == hipSuccess) err = hipEventCreate(&e->ev4);
    if (err == hipSuccess) err = hipMalloc((void**)&e->d_n_shared, 4);
    if (err == hipSuccess) err = hipMalloc((void**)&e->d_blob, blob.size() * 8u);
    if (err == hipSuccess) err = hipMemcpy(e->d_blob, blob.data(), blob.size() * 8u, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        af_engine_destroy(e);
        return fail(AF_ERR_HIP, std::string("engine setup: ") + hipGetErrorString(err));
    }
    a.blob = e->d_blob;
    e->fargs.blob = e->d_blob;
    e->fargs.tick_t = e->d_tick;
    *out = e;
    return AF_OK;
}

constexpr int kSummaryWpe = 8;   // (af_summary.hpp: kWpe; BASELINE config 2, 10 000 scenarios: 4.67 -> 4.28 ms for the analyzer's pair of kernels)

// argument checks shared by af_engine_summarize and af_engine_run_summarized
static int check_summary_request(const af_engine_t* e, const af_outputs_t* out, const af_summary_t* sum) {
    if (!e || !out || !sum) return fail(AF_ERR_INVALID, "NULL argument");
    if (e->plan_only) return fail(AF_ERR_NO_DEVICE, "planning-only engine (AF_DEVICE_PLAN_ONLY)");
    if (sum->n_scenarios == 0) return fail(AF_ERR_INVALID, "empty summary request");
    if (!out->counts) return fail(AF_ERR_INVALID, "outputs.counts is required");
    const bool want_lat = sum->stats || sum->rps || sum->hist;
    const bool want_series = sum->series_mean || sum->series_max;
    if (want_lat && !sum->stats) return fail(AF_ERR_INVALID, "summary.stats is required with rps/hist");
    if (want_lat && (!out->clock || out->clock_capacity == 0)) return fail(AF_ERR_INVALID, "summary needs outputs.clock");
    if (want_series && (!out->samples || out->tick_capacity == 0)) return fail(AF_ERR_INVALID, "series summary needs outputs.samples");
    if (sum->hist && (sum->hist_bins == 0 || sum->hist_bins > 8192u || !(sum->hist_max > 0.0)))
        return fail(AF_ERR_INVALID, "hist_bins must be 1..8192 and hist_max > 0");
    if (sum->rps && sum->rps_buckets == 0) return fail(AF_ERR_INVALID, "rps buffer with zero buckets");
    const size_t dyn_bytes = ((size_t)(sum->rps ? sum->rps_buckets : 0u) + (sum->hist ? sum->hist_bins : 0u)) * 4u;
    if (dyn_bytes > 96u * 1024u) return fail(AF_ERR_CAPACITY, "rps_buckets + hist_bins exceed the LDS budget (24576 words)");
    if (want_series && e->args.series_pitch / 4u > (uint32_t)afs::kSeriesThreads) return fail(AF_ERR_CAPACITY, "too many sampled series for af_series_kernel");
    return AF_OK;
}

// The analyzer's two kernels over scenarios [sc0, sc0 + count) of `out` / `sum`, enqueued on `stream` (no synchronisation).
// `done_flags` / `retry` ([2][n + 1]: count + scenarios, latency kernel then series kernel): workgroups check their scenario's flag
// and put an unfinished one on their kernel's retry list; `map_lat` / `map_ser` + their counts: the retry launch over such a list.
static int launch_summary_kernels(af_engine_t* e, const af_outputs_t* out, const af_summary_t* sum, uint32_t sc0, uint32_t count, hipStream_t stream,
                                  const uint32_t* done_flags = nullptr, uint32_t* retry = nullptr, const uint32_t* map_lat = nullptr, uint32_t n_map_lat = 0u,
                                  const uint32_t* map_ser = nullptr, uint32_t n_map_ser = 0u) {
    if (count == 0u) return AF_OK;
    const bool by_map = map_lat != nullptr || map_ser != nullptr;
    const bool want_lat = sum->stats || sum->rps || sum->hist;
    const bool want_series = sum->series_mean || sum->series_max;
    const uint32_t rps_buckets = sum->rps ? sum->rps_buckets : 0u;
    const size_t dyn_bytes = ((size_t)rps_buckets + (sum->hist ? sum->hist_bins : 0u)) * 4u;
    const uint32_t n_series = e->args.n_edges + 3u * e->args.n_servers;
    if (want_lat) {
        afs::SumArgs s{};
        s.clock = out->clock + (size_t)sc0 * out->clock_capacity * 2u;
        s.counts = out->counts + (size_t)sc0 * AF_CNT_SLOTS;
        s.clock_cap = out->clock_capacity;
        s.cnt_completed_slot = AF_CNT_COMPLETED;
        s.stats = sum->stats + (size_t)sc0 * 8u;
        s.rps = sum->rps ? sum->rps + (size_t)sc0 * rps_buckets : nullptr;
        s.rps_buckets = rps_buckets;
        s.hist = sum->hist ? sum->hist + (size_t)sc0 * sum->hist_bins : nullptr;
        s.hist_bins = sum->hist ? sum->hist_bins : 0u;
        s.hist_scale = sum->hist ? (double)sum->hist_bins / sum->hist_max : 0.0;
        s.done_flags = done_flags;
        s.retry = retry;
        s.scen_map = map_lat;
        // register budget of the latency kernel (af_summary.hpp: kWpe); AF_SUMMARY_WPE=4|8: measurements
        int wpe = kSummaryWpe;
        if (const char* env = std::getenv("AF_SUMMARY_WPE")) wpe = std::atoi(env);
        const void* fn = wpe == 8 ? reinterpret_cast<const void*>(afs::af_summary_kernel<8>)
                                  : reinterpret_cast<const void*>(afs::af_summary_kernel<4>);
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes));
        void* kargs[] = {&s};
        const uint32_t grid = by_map ? n_map_lat : count;
        if (grid != 0u) HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(afs::kThreads), kargs, dyn_bytes, stream));
        HIP_TRY(hipGetLastError());
    }
    if (want_series) {
        afs::SeriesArgs s{};
        s.samples = out->samples + (size_t)sc0 * out->tick_capacity * e->args.series_pitch;
        s.counts = out->counts + (size_t)sc0 * AF_CNT_SLOTS;
        s.tick_cap = out->tick_capacity;
        s.pitch = e->args.series_pitch;
        s.n_series = n_series;
        s.cnt_ticks_slot = AF_CNT_TICKS;
        s.n_edges = e->args.n_edges;
        s.mean = sum->series_mean ? sum->series_mean + (size_t)sc0 * n_series : nullptr;
        s.maxv = sum->series_max ? sum->series_max + (size_t)sc0 * n_series : nullptr;
        s.done_flags = done_flags;
        s.retry = retry ? retry + (size_t)count + 1u : nullptr;
        s.scen_map = map_ser;
        const uint32_t grid = by_map ? n_map_ser : count;
        if (grid != 0u) hipLaunchKernelGGL(afs::af_series_kernel, dim3(grid), dim3(afs::kSeriesThreads), 0, stream, s);
        HIP_TRY(hipGetLastError());
    }
    return AF_OK;
}

int af_engine_run(af_engine_t* e, const af_sweep_t* sweep, const af_outputs_t* out) {
    if (!e || !sweep || !out) return fail(AF_ERR_INVALID, "NULL argument");
    if (sweep->n_scenarios == 0 || !sweep->seeds) return fail(AF_ERR_INVALID, "empty sweep");
    if (!out->counts) return fail(AF_ERR_INVALID, "outputs.counts is required");
    if (out->clock && out->clock_capacity == 0) return fail(AF_ERR_INVALID, "clock buffer with zero capacity");
    if (out->samples && out->tick_capacity == 0) return fail(AF_ERR_INVALID, "samples buffer with zero capacity");
    if (out->online_hist && (out->online_hist_bins == 0 || !(out->online_hist_max > 0.0)))
        return fail(AF_ERR_INVALID, "online_hist needs online_hist_bins > 0 and online_hist_max > 0");
    if (out->online_rps && out->online_rps_buckets == 0) return fail(AF_ERR_INVALID, "online_rps with zero buckets");
    if (e->plan_only) return fail(AF_ERR_NO_DEVICE, "planning-only engine (AF_DEVICE_PLAN_ONLY): it cannot run");
    HIP_TRY(hipSetDevice(e->device));
    KArgs a = e->args;
    const uint32_t n = sweep->n_scenarios;

    uint32_t mask = 0;
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        const af_override_t& o = sweep->overrides[k];
        if (o.param >= AF_PARAM_COUNT_ || !o.values) return fail(AF_ERR_INVALID, "bad override");
        const uint32_t lim = (o.param >= AF_PARAM_EDGE_MEAN && o.param <= AF_PARAM_EDGE_DROPOUT) ? a.n_edges
                             : (o.param == AF_PARAM_STEP_TIME)                                   ? (uint32_t)e->row_of_step.size()
                             : (o.param == AF_PARAM_SRV_CORES || o.param == AF_PARAM_SRV_RAM_MB)  ? a.n_servers
                             : (o.param >= AF_PARAM_EMARK_TIME && o.param <= AF_PARAM_EMARK_EDGE) ? a.n_edge_marks
                             : (o.param >= AF_PARAM_SMARK_TIME && o.param <= AF_PARAM_SMARK_DOWN) ? a.n_srv_marks
                                                                                                 : 1u;
        if (o.index >= lim) return fail(AF_ERR_INVALID, "override index out of range");
        // integral columns are checked here: the kernels only ever see values inside the plan's ranges
        for (uint32_t i = 0; i < n; ++i) {
            const double v = o.values[i];
            if (o.param == AF_PARAM_SRV_CORES && !(v >= 1.0 && v <= 65535.0 && v == std::floor(v)))
                return fail(AF_ERR_INVALID, "cpu_cores column: integral values within 1..65535");
            if (o.param == AF_PARAM_EMARK_EDGE && !(v >= 0.0 && v < (double)a.n_edges && v == std::floor(v)))
                return fail(AF_ERR_INVALID, "edge-mark column names an unknown edge");
            if (o.param == AF_PARAM_SMARK_LB_EDGE && !(v >= -1.0 && v < (double)a.n_edges && v == std::floor(v)))
                return fail(AF_ERR_INVALID, "server-mark column names an unknown LB edge");
            if (o.param == AF_PARAM_GEN_WINDOW && !(v > 0.0)) return fail(AF_ERR_INVALID, "user_sampling_window column must be positive");
            if (o.param == AF_PARAM_SRV_RAM_MB && !(v > 0.0)) return fail(AF_ERR_INVALID, "ram_mb column must be positive");
        }
        mask |= 1u << o.param;
    }
    a.L = af::make_layout(e->request_capacity, e->fifo_capacity, a.n_edges, a.n_servers, a.n_lb_edges, a.n_rows, mask, a.n_edge_marks, a.n_srv_marks);
    const uint64_t bytes_per_lane = af::layout_bytes_per_lane(a.L);

    // ---- upload seeds + override tables (one staging buffer) -------------------
    const size_t seeds_b = (size_t)n * 8;
    const size_t vals_b = (size_t)sweep->n_overrides * n * 8;
    const size_t tab_b = (size_t)(sweep->n_overrides ? sweep->n_overrides : 1) * 4;
    const size_t total = seeds_b + vals_b + 2 * ((tab_b + 7) & ~size_t(7));
    if (total > e->sweep_cap) {
        if (e->d_sweep) HIP_TRY(hipFree(e->d_sweep));
        e->d_sweep = nullptr;
        e->sweep_cap = 0;
        HIP_TRY(hipMalloc(&e->d_sweep, total));
        e->sweep_cap = total;
    }
    unsigned char* ds = static_cast<unsigned char*>(e->d_sweep);
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    HIP_TRY(hipMemcpyAsync(ds, sweep->seeds, seeds_b, hipMemcpyHostToDevice, e->stream));
    std::vector<uint32_t> params(sweep->n_overrides ? sweep->n_overrides : 1, 0u), idxs(params.size(), 0u);
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        params[k] = sweep->overrides[k].param;
        idxs[k] = params[k] == AF_PARAM_STEP_TIME ? e->row_of_step[sweep->overrides[k].index] : sweep->overrides[k].index;
        HIP_TRY(hipMemcpyAsync(ds + seeds_b + (size_t)k * n * 8, sweep->overrides[k].values, (size_t)n * 8,
                               hipMemcpyHostToDevice, e->stream));
    }
    unsigned char* d_params = ds + seeds_b + vals_b;
    unsigned char* d_idxs = d_params + ((tab_b + 7) & ~size_t(7));
    HIP_TRY(hipMemcpyAsync(d_params, params.data(), tab_b, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d_idxs, idxs.data(), tab_b, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // host staging vectors go out of scope below
    HIP_TRY(hipEventRecord(e->ev1, e->stream));

    a.n_ovr = sweep->n_overrides;
    a.ovr_param = reinterpret_cast<const uint32_t*>(d_params);
    a.ovr_index = reinterpret_cast<const uint32_t*>(d_idxs);
    a.ovr_stride = n;
    a.clock_cap = out->clock_capacity;
    a.tick_cap = out->tick_capacity;

    // ---- buffers ------------------------------------------------------------------------------------
    const uint32_t n_draw = sweep->draw_capacity ? sweep->draw_capacity : out->clock_capacity;
    if (n_draw == 0) return fail(AF_ERR_INVALID, "draw_capacity (or clock_capacity) must be > 0");
    const size_t draw_bytes_per_scen = (size_t)(1u + a.n_edges) * n_draw * sizeof(double);
    // (a sweep the stage-parallel kernel cannot be sized for -- a cpu_cores column above 64, lists that do not fit the LDS --
    // runs on the next-event kernels like a plan outside its range does, unless flow_mode 2 asked for that kernel: ADVICE r3)
    bool use_flow = flow_wanted(e, n);
    FlowPlan FP;
    if (use_flow) {
        const int rc = plan_flow(e, a, sweep, out, FP);
        if (rc == AF_ERR_CAPACITY && e->flow_mode != 2u) use_flow = false;
        else if (rc) return rc;
    }
    size_t mem_free = 0, mem_total = 0;
    HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
    // the stage-parallel kernel only needs the arrival times of a chunk; the next-event kernels every draw
    const uint32_t chunk = chunk_size(e, n, use_flow ? (size_t)n_draw * sizeof(double) : draw_bytes_per_scen,
                                      mem_free + (use_flow ? e->arr_cap : 0));
    if (chunk == 0) return fail(AF_ERR_CAPACITY, "draw_capacity too large for the device memory budget");
    auto grow = [&](void** ptr, size_t& cap, size_t need) -> int {
        if (need <= cap) return AF_OK;
        if (*ptr) HIP_TRY(hipFree(*ptr));
        *ptr = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(ptr, need));
        cap = need;
        return AF_OK;
    };
    a.n_draw = n_draw;
    a.n_shared = e->d_n_shared;
    a.scen_map = nullptr;
    a.draw_slot = nullptr;
    a.draws_by_lane = 0u;

    double ms_pregen = 0.0, ms_kernel = 0.0, ms_flow = 0.0;
    uint32_t kl = 0, waves = 0, lds_bytes = 0, n_chunks = 0, n_rerun = 0, n_jit = 0, n_jit_miss = 0;
    uint32_t fused_first = 0u;   // af_engine_run_summarized: scenarios whose analyzer ran beside the stage-parallel kernel's last round
    uint32_t fb_total[5] = {0, 0, 0, 0, 0}, flow_scen = 0;
    uint32_t flow_retried = 0, flow_to_next = 0;
    size_t draw_bytes = 0;
    uint32_t pregen_group = 0;
    bool lds_state = false;

    aff::FlowLayout& FL = FP.FL;
    aff::FlowLayout& FL2 = FP.FL2;
    const bool flow_big = FP.big;
    const uint32_t flow_lds = FP.lds;

    const size_t tie_words = a.L.tie_words;
    const bool hetero_load = (mask & ((1u << AF_PARAM_GEN_USERS_MEAN) | (1u << AF_PARAM_GEN_RPM_MEAN))) != 0u;
    // One launch of the next-event kernel over `count` scenarios (all of the chunk, or the ones listed in `map`).
    auto launch_des = [&](uint32_t count, bool faithful) -> int {
        choose_lanes(e, count, a.blob_bytes, bytes_per_lane, kl, lds_state);
        uint32_t klog = 0;
        while ((1u << klog) < kl) ++klog;
        a.n_scen = count;
        const uint64_t state_per_wave = bytes_per_lane * kl;
        waves = (count + kl - 1) / kl;
        lds_bytes = lds_state ? (uint32_t)(a.blob_bytes + state_per_wave) : a.blob_bytes;
        a.state = nullptr;
        a.state_bytes_per_wave = state_per_wave;
        if (!lds_state) {
            if (int rc = grow((void**)&e->d_state, e->state_cap, (size_t)state_per_wave * waves)) return rc;
            a.state = e->d_state;
        }
        // SimPy-order variant: when the waves do not all fit at 3 per SIMD anyway, the build with
        // the larger register budget (2 per SIMD, no spills) is the faster one (measured: 6 400
        // grid points 6.2 s -> 5.5 s; 2 500 waves that do fit: 2.7 s vs 4.3 s).  Sweeps over the load
        // (scenario lengths differ by orders of magnitude) do not run as lock-step batches: there the
        // per-wave speed decides and the roomy build wins even when everything would fit.
        const bool hetero = (mask & ((1u << AF_PARAM_GEN_USERS_MEAN) | (1u << AF_PARAM_GEN_RPM_MEAN))) != 0u;
        const bool roomy = faithful && (waves > 3072u || hetero);
        const void* fn = des_kernel_for(lds_state, faithful, klog, roomy);
        hipFunction_t jit_fn = nullptr;
        if (e->jit_module && lds_bytes <= 64u * 1024u) {
            if (jit_spec_string(a, lds_state, klog) == e->jit_spec) jit_fn = !faithful ? e->jit_lean : roomy ? e->jit_order2 : e->jit_order3;
            else n_jit_miss += 1u;
        }
        if (std::getenv("AF_DEBUG"))
            std::fprintf(stderr, "[af] launch: %u scenarios, %u waves x %u lanes, %s state, %s%s%s\n", count, waves, kl,
                         lds_state ? "LDS" : "HBM", faithful ? "SimPy-order" : "lean", roomy ? " (2 waves/SIMD build)" : "",
                         jit_fn ? ", plan-specialised" : "");
        void* kargs[] = {&a};
        if (jit_fn) {
            HIP_TRY(hipModuleLaunchKernel(jit_fn, waves, 1, 1, kWave, 1, 1, lds_bytes, e->stream, kargs, nullptr));
            n_jit += 1u;
        } else {
            if (lds_state) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            HIP_TRY(hipLaunchKernel(fn, dim3(waves), dim3(kWave), kargs, lds_bytes, e->stream));
        }
        return AF_OK;
    };

    // The next-event kernels over `count` scenarios of the current chunk: all of them (h_map == nullptr) or the
    // listed ones.  Draws are pre-generated per slot j = position in the list; pieces that fit the draw budget.
    auto run_sequential = [&](uint32_t count, const uint32_t* h_map, const KArgs& chunk_args) -> int {
        size_t mf = 0, mt = 0;
        HIP_TRY(hipMemGetInfo(&mf, &mt));
        // (per scenario: its draws, its scratch of the SimPy-order path and -- when the state does not fit the LDS: long wait
        // queues -- its state words in HBM)
        const size_t state_hbm = bytes_per_lane > 48u * 1024u ? (size_t)bytes_per_lane : 0u;
        uint32_t piece = chunk_size(e, count, draw_bytes_per_scen + tie_words * 8u + state_hbm, mf);
        if (piece == 0) return fail(AF_ERR_CAPACITY, "draw_capacity / fifo_capacity too large for the device memory budget");
        if (const char* env = std::getenv("AF_SEQ_PIECE")) {   // test hook: force a whole chunk to run in pieces
            const uint32_t v = (uint32_t)std::atoi(env);
            if (v != 0u && v < piece) piece = v;
        }
        // A whole chunk (no list) that does not fit the draw budget in one piece -- the free memory is measured again
        // here, after the scratch buffers of the chunk were allocated -- runs as pieces over an identity list: piece
        // p0 simulates scenarios [p0, p0 + cnt) with its draws in slots [0, cnt).  (Round 2 ran [0, cnt) again: ADVICE r2.)
        std::vector<uint32_t> ident;
        if (h_map == nullptr && piece < count) {
            ident.resize(count);
            for (uint32_t i = 0; i < count; ++i) ident[i] = i;
            h_map = ident.data();
        }
        const bool listed_subset = h_map != nullptr && ident.empty();
        for (uint32_t p0 = 0; p0 < count; p0 += piece) {
            const uint32_t cnt = count - p0 < piece ? count - p0 : piece;
            a = chunk_args;
            if (int rc = grow((void**)&e->d_draws, e->draws_cap, draw_bytes_per_scen * cnt)) return rc;
            if (int rc = grow((void**)&e->d_pre_flags, e->pre_flags_cap, (size_t)cnt * 4u)) return rc;
            if (int rc = grow((void**)&e->d_tie, e->tie_cap, (size_t)cnt * tie_words * 8u)) return rc;
            if (draw_bytes_per_scen * cnt > draw_bytes) draw_bytes = draw_bytes_per_scen * cnt;
            a.draws = e->d_draws;
            a.pre_flags = e->d_pre_flags;
            a.tie = e->d_tie;
            a.draw_slot = nullptr;
            if (h_map) {
                if (int rc = grow((void**)&e->d_map, e->map_cap, (size_t)cnt * 4u)) return rc;
                HIP_TRY(hipMemcpyAsync(e->d_map, h_map + p0, (size_t)cnt * 4u, hipMemcpyHostToDevice, e->stream));
                a.scen_map = e->d_map;
                a.draws_by_lane = 1u;
            } else {
                a.scen_map = nullptr;
                a.draws_by_lane = 0u;
            }
            a.n_scen = cnt;
            HIP_TRY(hipEventRecord(e->ev2, e->stream));
            if (launch_pregen_arrivals(a, cnt, (uint32_t)((1u + a.n_edges) * n_draw), !hetero_load, e->stream))
                return fail(AF_ERR_HIP, "af_pregen_arrivals: LDS attribute");
            hipLaunchKernelGGL(af_pregen_edges, dim3((n_draw + 255u) / 256u, cnt, a.n_edges), dim3(256), 0, e->stream, a);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(e->ev3, e->stream));
            if (listed_subset && (a.online_hist || a.online_rps)) {   // scenarios that start over: their online counters are cleared
                hipLaunchKernelGGL(af_zero_online, dim3(cnt), dim3(256), 0, e->stream, a, cnt);
                HIP_TRY(hipGetLastError());
            }
            // First pass: the lean kernel.  A scenario in which two timed events share an instant stops
            // there and is simulated again, from its start, by the kernel that has SimPy's event-by-event
            // path; engines whose plan keeps producing such scenarios go straight to that kernel.
            // (a handful of scenarios handed back by the stage-parallel kernel -- usually for a tie -- go straight to the
            // SimPy-order kernel: a second pass over them would cost a whole latency-bound launch more)
            const bool faithful_first = e->shared_instants_likely || (listed_subset && (uint64_t)count * 50u < chunk_args.n_scen);
            HIP_TRY(hipMemsetAsync(e->d_n_shared, 0, 4, e->stream));
            if (int rc = launch_des(cnt, faithful_first)) return rc;
            if (!faithful_first) {
                uint32_t n_shared = 0;
                HIP_TRY(hipMemcpyAsync(&n_shared, e->d_n_shared, 4, hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (n_shared > 0u) {
                    const uint32_t nc_all = chunk_args.n_scen;
                    std::vector<uint32_t> cnt_host((size_t)nc_all * AF_CNT_SLOTS);
                    HIP_TRY(hipMemcpy(cnt_host.data(), a.counts, cnt_host.size() * 4u, hipMemcpyDeviceToHost));
                    std::vector<uint32_t> map2, slot2;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const uint32_t scn = h_map ? h_map[p0 + j] : j;
                        if (cnt_host[(size_t)scn * AF_CNT_SLOTS + AF_CNT_FLAGS] & af::FLAG_SHARED_INSTANT) {
                            map2.push_back(scn);
                            slot2.push_back(j);
                        }
                    }
                    if (!map2.empty()) {
                        if (int rc = grow((void**)&e->d_map, e->map_cap, (size_t)nc_all * 4u)) return rc;
                        if (int rc = grow((void**)&e->d_slot, e->slot_cap, map2.size() * 4u)) return rc;
                        HIP_TRY(hipMemcpy(e->d_map, map2.data(), map2.size() * 4u, hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(e->d_slot, slot2.data(), slot2.size() * 4u, hipMemcpyHostToDevice));
                        a.scen_map = e->d_map;
                        a.draw_slot = e->d_slot;
                        if (a.online_hist || a.online_rps) {
                            hipLaunchKernelGGL(af_zero_online, dim3((uint32_t)map2.size()), dim3(256), 0, e->stream, a, (uint32_t)map2.size());
                            HIP_TRY(hipGetLastError());
                        }
                        if (int rc = launch_des((uint32_t)map2.size(), true)) return rc;
                        n_rerun += (uint32_t)map2.size();
                        // such plans keep producing them: later runs start with the SimPy-order kernel, which
                        // is cheaper than a second pass over a few long scenarios in narrow, latency-bound waves
                        e->shared_instants_likely = true;
                    }
                }
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(e->ev4, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            float ms_p = 0.f, ms_k = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms_p, e->ev2, e->ev3));
            HIP_TRY(hipEventElapsedTime(&ms_k, e->ev3, e->ev4));
            ms_pregen += ms_p;
            ms_kernel += ms_k;
        }
        return AF_OK;
    };

    for (uint32_t lo = 0; lo < n; lo += chunk) {
        const uint32_t nc = n - lo < chunk ? n - lo : chunk;
        n_chunks += 1u;
        a.n_scen = nc;
        a.scen_map = nullptr;
        a.draw_slot = nullptr;
        a.draws_by_lane = 0u;
        a.seeds = reinterpret_cast<const uint64_t*>(ds) + lo;
        a.ovr_values = reinterpret_cast<const double*>(ds + seeds_b) + lo;
        a.clock = out->clock ? out->clock + (size_t)lo * out->clock_capacity * 2u : nullptr;
        a.samples = out->samples ? out->samples + (size_t)lo * a.series_pitch * out->tick_capacity : nullptr;
        a.counts = out->counts + (size_t)lo * AF_CNT_SLOTS;
        a.online_hist = out->online_hist ? out->online_hist + (size_t)lo * out->online_hist_bins : nullptr;
        a.online_rps = out->online_rps ? out->online_rps + (size_t)lo * out->online_rps_buckets : nullptr;
        a.online_hist_bins = out->online_hist_bins;
        a.online_rps_buckets = out->online_rps_buckets;
        a.online_hist_scale = out->online_hist ? (double)out->online_hist_bins / out->online_hist_max : 0.0;
        const KArgs chunk_args = a;

        if (!use_flow) {
            if (int rc = run_sequential(nc, nullptr, chunk_args)) return rc;
            continue;
        }
        // ---- stage-parallel kernel over the whole chunk; what it hands back goes to the next-event kernels
        if (int rc = grow((void**)&e->d_arr, e->arr_cap, (size_t)nc * n_draw * sizeof(double))) return rc;
        if (int rc = grow((void**)&e->d_arr_flags, e->arr_flags_cap, (size_t)nc * 4u)) return rc;
        if ((size_t)nc * n_draw * sizeof(double) > draw_bytes) draw_bytes = (size_t)nc * n_draw * sizeof(double);
        a.draws = e->d_arr;
        a.pre_flags = e->d_arr_flags;
        // grouped pre-generation (af_pregen.hpp): sweeps of alike scenarios -- a workgroup is as slow as its heaviest one -- with
        // sampling windows long enough that a window start (a user draw by one wave, the pipeline refilled) does not count
        const char* pregen_mode = std::getenv("AF_PREGEN_MODE");   // rows | groups: tests and measurements
        const bool force_groups = pregen_mode && std::strcmp(pregen_mode, "groups") == 0;
        const bool force_rows = pregen_mode && std::strcmp(pregen_mode, "rows") == 0;
        // (sweeps over users / rpm -- BASELINE configs 3 / 4 -- as long as the heaviest scenario is within 2.5 x of the mean:
        // load_spread_suits_groups)
        const bool grouped = !force_rows && a.gen_window_s > 0.0 &&
                             (force_groups || (!(mask & (1u << AF_PARAM_GEN_WINDOW)) && nc >= 3072u &&
                                               (double)n_draw * a.gen_window_s >= 512.0 * a.total_time &&
                                               (!hetero_load || load_spread_suits_groups(e, sweep, lo, nc))));
        // Sweeps over the load: waves are dispatched in blockIdx order as slots free up, and a wave lasts as long as its scenario
        // has events.  A users-major grid (BASELINE configs 3 / 4: users ascending with the index) is then the WORST order --
        // the heaviest scenarios start last and the launch ends with a few long waves on an empty chip (list scheduling of
        // 10 000 waves with times ~ users on 4 096 slots: 3.54 mean wave times ascending, 2.73 heaviest first, 2.44 ideal).  Wave j
        // simulates scenario order[j]; everything the kernel reads or writes is indexed by the scenario.
        const bool ordered = hetero_load && std::getenv("AF_FLOW_ORDER_OFF") == nullptr;
        if (ordered) {
            std::vector<uint32_t> order = heaviest_first(e, sweep, lo, nc);
            if (int rc = grow((void**)&e->d_order, e->order_cap, (size_t)nc * 4u)) return rc;
            HIP_TRY(hipMemcpyAsync(e->d_order, order.data(), (size_t)nc * 4u, hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));   // (`order` leaves scope; 40 KB, before anything of this chunk is enqueued)
        }
        HIP_TRY(hipEventRecord(e->ev2, e->stream));
        if (launch_pregen_arrivals(a, nc, n_draw, !hetero_load, e->stream, grouped, &pregen_group)) return fail(AF_ERR_HIP, "af_arrival_groups: LDS attribute");
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e->ev3, e->stream));
        aff::FlowArgs f = e->fargs;
        f.L = FL;
        f.n_scen = nc;
        f.seeds = a.seeds;
        f.n_ovr = a.n_ovr;
        f.ovr_param = a.ovr_param;
        f.ovr_index = a.ovr_index;
        f.ovr_values = a.ovr_values;
        f.ovr_stride = a.ovr_stride;
        f.arrivals = e->d_arr;
        f.n_draw = n_draw;
        f.pre_flags = e->d_arr_flags;
        if (ordered) f.scen_map = e->d_order;
        f.clock = a.clock;
        f.clock_cap = a.clock_cap;
        f.samples = a.samples;
        f.tick_cap = a.tick_cap;
        f.counts = a.counts;
        f.online_hist = a.online_hist;
        f.online_rps = a.online_rps;
        f.online_hist_bins = a.online_hist_bins;
        f.online_rps_buckets = a.online_rps_buckets;
        f.online_hist_scale = a.online_hist_scale;
        f.n_fallback = e->d_fb;
        HIP_TRY(hipMemsetAsync(e->d_fb, 0, 32u * 4u, e->stream));
        // measurement hook (AF_FLOW_PROF=<file>, plan-specialised builds only): shader-clock time per section of Flow::run
        unsigned long long* d_prof = nullptr;
        const char* prof_path = std::getenv("AF_FLOW_PROF");
        if (prof_path != nullptr) {
            HIP_TRY(hipMalloc((void**)&d_prof, (size_t)nc * aff::kProfSections * 8u));
            HIP_TRY(hipMemsetAsync(d_prof, 0, (size_t)nc * aff::kProfSections * 8u, e->stream));
            f.prof = d_prof;
        }
        {
            const uint32_t flow_lds_launch = spread_lds_bytes(flow_lds);
            void* kargs[] = {&f};
            // plan-specialised build of exactly this launch (af_engine_set_kernels), else the library's generic instantiation
            const bool jit = e->flow_jit_fn != nullptr && flow_lds_launch <= 64u * 1024u &&
                             flow_jit_spec_string(e, FP, out, sweep->n_overrides != 0u) == e->flow_jit_spec;
            if (e->flow_jit_fn != nullptr && !jit) n_jit_miss += 1u;
            if (std::getenv("AF_DEBUG")) std::fprintf(stderr, "[af] flow jit spec: %s\n", flow_jit_spec_string(e, FP, out, sweep->n_overrides != 0u).c_str());
            if (std::getenv("AF_DEBUG"))
                std::fprintf(stderr, "[af] flow launch: %u scenarios, %u list entries%s, %u ring rows, %u B LDS per wave, af_flow_kernel<%u, %#x>%s%s\n", nc,
                             FL.cap, flow_big ? " (long-list instantiation)" : "", FL.ring_rows, flow_lds, FP.ipl, FP.feat,
                             FP.lean ? (FP.far ? ", lean instantiation with far edges" : ", lean instantiation") : "", jit ? ", plan-specialised" : "");
            const void* fn = jit ? nullptr : flow_kernel_for(FP.ipl, FP.feat);
            if (!jit && fn == nullptr) return fail(AF_ERR_INVALID, "no stage-parallel instantiation for this launch");
            if (!jit && flow_lds_launch > 48u * 1024u) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flow_lds_launch));
            // af_engine_run_summarized: the analyzer starts on a second stream once the scenarios of the kernel's full residency rounds
            // have finished (hipStreamWaitValue32 on a counter its waves bump) and runs beside the last, partial round, which leaves
            // most of the chip idle; the analyzer is HBM-bound where this kernel is VALU-bound.  A workgroup of the analyzer that
            // meets a scenario still being simulated (its done flag is clear) puts it on a retry list and leaves.
            // (Measured first, round 6: the kernel launched in two parts with the analyzer of the first beside the second --
            // 51.9 instead of 48.4 ms per step on BASELINE config 2: the second part cannot start before the LAST wave of the first
            // has ended and then lasts a whole scenario at low occupancy, 11 ms against the 7 ms of the natural tail.)
            uint32_t gate = 0u;
            if (e->fused_sum != nullptr && lo == 0u && nc == n && d_prof == nullptr) {
                if (e->wait_value_ok < 0) {
                    int v = 0;
                    if (hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, e->device) != hipSuccess) { (void)hipGetLastError(); v = 0; }
                    e->wait_value_ok = v;
                }
                int per_cu = 0;
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, e->device));
                if (jit) HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, e->flow_jit_fn, (int)kWave, flow_lds_launch));
                else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)kWave, flow_lds_launch));
                const uint32_t resident = (uint32_t)(per_cu > 0 ? per_cu : 0) * (uint32_t)prop.multiProcessorCount;
                const uint32_t rem = resident ? nc % resident : 0u;
                // (a tail of more than three quarters of a round leaves no room for the analyzer's workgroups; no tail, nothing to fill)
                // (and only where the partial round is a good part of the launch -- at most six rounds --: the analyzer of a sweep of
                // many rounds needs longer than the tail lasts, and beside a FULL round it only takes the kernel's slots: BASELINE
                // config 5, 12.2 rounds, 227 instead of 223 ms per step, measured)
                if (e->wait_value_ok > 0 && resident != 0u && nc > resident && nc <= 6u * resident && rem != 0u && rem * 4u <= resident * 3u) gate = nc - rem;
                if (std::getenv("AF_DEBUG")) std::fprintf(stderr, "[af] analyzer beside the last round: %u resident waves, gate at %u of %u scenarios\n", resident, gate, nc);
            }
            if (gate != 0u) {
                if (e->stream2 == nullptr) {
                    HIP_TRY(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreate(&e->ev_s0));
                    HIP_TRY(hipEventCreate(&e->ev_s1));
                    HIP_TRY(hipHostMalloc((void**)&e->h_done_count, 64, hipHostMallocCoherent | hipHostMallocMapped));
                    HIP_TRY(hipHostGetDevicePointer((void**)&e->d_done_count, e->h_done_count, 0));
                }
                if (int rc = grow((void**)&e->d_done_flags, e->done_flags_cap, (size_t)nc * 4u)) return rc;
                if (int rc = grow((void**)&e->d_retry, e->retry_cap, 2u * ((size_t)nc + 1u) * 4u)) return rc;
                *e->h_done_count = 0u;   // (nothing of this engine is in flight: af_engine_run is synchronous)
                HIP_TRY(hipMemsetAsync(e->d_done_flags, 0, (size_t)nc * 4u, e->stream));
                HIP_TRY(hipMemsetAsync(e->d_retry, 0, 2u * ((size_t)nc + 1u) * 4u, e->stream));
                // (where the kernel's entry point finds them: behind the hand-over counters it already has a pointer to -- two more
                // kernel arguments cost the general-server form 5 %, measured: its register allocation answers small changes chaotically)
                e->done_ptrs[0] = reinterpret_cast<uint64_t>(e->d_done_flags);
                e->done_ptrs[1] = reinterpret_cast<uint64_t>(e->d_done_count);
                HIP_TRY(hipMemcpyAsync(e->d_fb + kDoneWords, e->done_ptrs, 16u, hipMemcpyHostToDevice, e->stream));
            }
            f.n_scen = nc;
            if (jit) {
                HIP_TRY(hipModuleLaunchKernel(e->flow_jit_fn, nc, 1, 1, kWave, 1, 1, flow_lds_launch, e->stream, kargs, nullptr));
                n_jit += 1u;
            } else {
                HIP_TRY(hipLaunchKernel(fn, dim3(nc), dim3(kWave), kargs, flow_lds_launch, e->stream));
            }
            if (gate != 0u) {
                // (from here on the second stream waits for the counter: whatever goes wrong below must still let it through)
                auto release_side = [&]() { *e->h_done_count = 0xFFFFFFFFu; };
                hipError_t err = hipStreamWaitValue32(e->stream2, e->d_done_count, gate, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (err == hipSuccess) err = hipEventRecord(e->ev_s0, e->stream2);
                int rc = AF_OK;
                if (err == hipSuccess) rc = launch_summary_kernels(e, e->fused_out, e->fused_sum, 0u, nc, e->stream2, e->d_done_flags, e->d_retry, nullptr, 0u);
                if (err == hipSuccess && rc == AF_OK) err = hipEventRecord(e->ev_s1, e->stream2);
                if (err != hipSuccess || rc != AF_OK) {
                    release_side();
                    (void)hipStreamSynchronize(e->stream);
                    (void)hipStreamSynchronize(e->stream2);
                    if (rc != AF_OK) return rc;
                    HIP_TRY(err);
                }
            }
            fused_first = gate;
        }
        HIP_TRY(hipEventRecord(e->ev4, e->stream));
        uint32_t fb[5] = {0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(fb, e->d_fb, sizeof fb, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (fused_first != 0u) {
            HIP_TRY(hipStreamSynchronize(e->stream2));
            float ms_s = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms_s, e->ev_s0, e->ev_s1));
            e->stats.summary_beside_ms = ms_s;
            // (a scenario handed back is simulated again below and its outputs change: the sweep is then summarised again, in full)
            e->fused_done = fb[0] == 0u;
        }
        if (d_prof != nullptr) {
            std::vector<unsigned long long> hp((size_t)nc * aff::kProfSections);
            HIP_TRY(hipMemcpy(hp.data(), d_prof, hp.size() * 8u, hipMemcpyDeviceToHost));
            (void)hipFree(d_prof);
            static const char* names[aff::kProfSections] = {"setup+end", "generator batch", "select", "series: delivery end (FAR)", "station logic",
                                                            "servers_solve + claim", "server series", "draws (Philox, log)", "send: spike + series",
                                                            "append + send_floor", "complete", "flush_ticks + round end",
                                                            "general servers: lanes", "general servers: walk", "general servers: rank + relax", "general servers: state at the horizon",
                                                            "COUNT rounds solved at once", "COUNT relaxation sweeps", "COUNT lanes in solved rounds", "COUNT rounds walked by event",
                                                            "general servers: standing (ranks, RAM)", "general servers: commit walk (series)"};
            double acc[aff::kProfSections] = {}, total = 0.0;
            for (uint32_t i = 0; i < nc; ++i)
                for (uint32_t k = 0; k < aff::kProfSections; ++k) acc[k] += (double)hp[(size_t)i * aff::kProfSections + k];
            for (uint32_t k = 0; k < aff::kProfSections; ++k)
                if (!names[k] || std::strncmp(names[k], "COUNT", 5) != 0) total += acc[k];
            if (FILE* fp = std::fopen(prof_path, "a")) {
                std::fprintf(fp, "af_flow_kernel<%u, %#x> sections, %u waves, mean cycles per wave %.0f:\n", FP.ipl, FP.feat, nc, total / nc);
                for (uint32_t k = 0; k < aff::kProfSections; ++k)
                    if (names[k]) std::fprintf(fp, "  %-28s %6.2f %%  %12.0f cycles per wave\n", names[k], 100.0 * acc[k] / total, acc[k] / nc);
                std::fclose(fp);
            }
        }
        float ms_p = 0.f, ms_f = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms_p, e->ev2, e->ev3));
        HIP_TRY(hipEventElapsedTime(&ms_f, e->ev3, e->ev4));
        ms_pregen += ms_p;
        ms_flow += ms_f;
        flow_scen += nc;
        for (int k = 0; k < 5; ++k) fb_total[k] += fb[k];
        if (fb[0] > 0u) {
            std::vector<uint32_t> cnt_host((size_t)nc * AF_CNT_SLOTS);
            HIP_TRY(hipMemcpy(cnt_host.data(), a.counts, cnt_host.size() * 4u, hipMemcpyDeviceToHost));
            // Second chance on the stage-parallel kernel itself, in its most tolerant form: 256-entry lists, tick
            // differences in HBM (no reach limit), messages carry their send time so that two deliveries of one station
            // at the same instant are ordered the way SimPy orders them.  A wave costs tens of milliseconds; the same
            // scenario on a next-event kernel is a latency-bound chain of ~1 s (LB-2).  Only what this launch hands
            // back again -- or what can never fit its RAM model -- goes to the next-event kernels.
            std::vector<uint32_t> retry, rest;
            for (uint32_t i = 0; i < nc; ++i) {
                const uint32_t fl = cnt_host[(size_t)i * AF_CNT_SLOTS + AF_CNT_FLAGS];
                if (!(fl & aff::FLAG_FLOW_FALLBACK)) continue;
                // (flow_big: that WAS the most tolerant form; a tie the general server station gives up on stops the second chance at
                // the same instant: straight to the next-event kernels)
                if ((fl & (aff::FLOW_WHY_RAM | aff::FLOW_WHY_GEN_TIE)) || (flow_big && !FP.gen_compact)) rest.push_back(i);
                else retry.push_back(i);
            }
            if (!retry.empty()) {
                if (int rc = grow((void**)&e->d_map, e->map_cap, retry.size() * 4u)) return rc;
                HIP_TRY(hipMemcpyAsync(e->d_map, retry.data(), retry.size() * 4u, hipMemcpyHostToDevice, e->stream));
                aff::FlowArgs f2 = f;
                f2.L = FL2;
                f2.n_scen = (uint32_t)retry.size();
                f2.scen_map = e->d_map;
                f2.n_fallback = e->d_fb + 5;
                const uint32_t lds2 = a.blob_bytes + f2.L.n_words * 8u;
                if (lds2 > kLdsLimit) return fail(AF_ERR_CAPACITY, "flow kernel layout exceeds the LDS of a compute unit");
                if (a.online_hist || a.online_rps) {
                    a.scen_map = e->d_map;
                    hipLaunchKernelGGL(af_zero_online, dim3((uint32_t)retry.size()), dim3(256), 0, e->stream, a, (uint32_t)retry.size());
                    HIP_TRY(hipGetLastError());
                    a.scen_map = nullptr;
                }
                constexpr uint32_t kRobust = aff::FEAT_ALL | aff::FEAT_TIEBREAK | aff::FEAT_BIGLIST;
                const void* fn2 = (FP.gen_compact && e->flow_chain && f2.lb_least_connections) ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_GENSRV | aff::FEAT_LC | aff::FEAT_CHAIN>)
                                  : (FP.gen_compact && e->flow_chain) ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_GENSRV | aff::FEAT_CHAIN>)
                                  : (FP.gen_compact && f2.lb_least_connections) ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_LC | aff::FEAT_GENSRV>)
                                  : FP.gen_compact        ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_GENSRV>)
                                  : (f2.lb_least_connections && e->flow_chain) ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_LC | aff::FEAT_CHAIN>)
                                  : f2.lb_least_connections ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_LC>)
                                  : e->flow_chain         ? reinterpret_cast<const void*>(af_flow_kernel<1, kRobust | aff::FEAT_CHAIN>)
                                                          : reinterpret_cast<const void*>(af_flow_kernel<1, kRobust>);
                if (lds2 > 48u * 1024u) HIP_TRY(hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                HIP_TRY(hipEventRecord(e->ev3, e->stream));
                void* kargs2[] = {&f2};
                if (std::getenv("AF_DEBUG")) {
                    std::fprintf(stderr, "[af] flow second chance: %zu scenarios (lists of %u/%u/%u/%u entries with send times, differences in HBM), %u B LDS:",
                                 retry.size(), FL2.cap_of[0], FL2.cap_of[1], FL2.cap_of[2], FL2.cap_of[3], lds2);
                    for (size_t q = 0; q < retry.size() && q < 16; ++q)
                        std::fprintf(stderr, " %u(flags %#x)", lo + retry[q], cnt_host[(size_t)retry[q] * AF_CNT_SLOTS + AF_CNT_FLAGS]);
                    std::fprintf(stderr, "\n");
                }
                HIP_TRY(hipLaunchKernel(fn2, dim3((uint32_t)retry.size()), dim3(kWave), kargs2, lds2, e->stream));
                HIP_TRY(hipEventRecord(e->ev4, e->stream));
                uint32_t fb2[5] = {0, 0, 0, 0, 0};
                HIP_TRY(hipMemcpyAsync(fb2, e->d_fb + 5, sizeof fb2, hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                float ms2 = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms2, e->ev3, e->ev4));
                ms_flow += ms2;
                flow_retried += (uint32_t)retry.size();
                if (fb2[0] > 0u) {
                    HIP_TRY(hipMemcpy(cnt_host.data(), a.counts, cnt_host.size() * 4u, hipMemcpyDeviceToHost));
                    for (uint32_t i : retry)
                        if (cnt_host[(size_t)i * AF_CNT_SLOTS + AF_CNT_FLAGS] & aff::FLAG_FLOW_FALLBACK) rest.push_back(i);
                    std::sort(rest.begin(), rest.end());
                }
            }
            flow_to_next += (uint32_t)rest.size();
            if (!rest.empty())
                if (int rc = run_sequential((uint32_t)rest.size(), rest.data(), chunk_args)) return rc;
            // A plan whose scenarios mostly end on the next-event kernels anyway (general servers fed by edges with discrete
            // latencies: arrivals that coincide with step ends) pays a flow pass, a second chance AND a next-event pass
            // per scenario: once a chunk handed more than a quarter of its scenarios over, the chunks still to come go to the
            // next-event kernels directly (results are the same either way; flow_mode 2 = "always" keeps the stage-parallel kernel).
            if ((uint64_t)rest.size() * 4u > (uint64_t)nc && e->flow_mode != 2u) use_flow = false;
        }
    }

    float ms_h2d = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms_h2d, e->ev0, e->ev1));
    e->stats.kernel_ms = ms_kernel + ms_flow;
    e->stats.flow_kernel_ms = ms_flow;
    e->stats.flow_scenarios = flow_scen;
    e->stats.flow_fallback = fb_total[0];
    e->stats.flow_fallback_tie = fb_total[1];
    e->stats.flow_fallback_list = fb_total[2];
    e->stats.flow_fallback_ring = fb_total[3];
    e->stats.flow_fallback_ram = fb_total[4];
    e->stats.flow_retried = flow_retried;
    e->stats.flow_to_next_event = flow_to_next;
    e->stats.flow_list_entries = flow_scen ? FL.cap : 0u;
    e->stats.flow_ring_rows = flow_scen ? FL.ring_rows : 0u;
    e->stats.flow_lds_bytes = flow_lds;
    e->stats.jit_fallbacks = n_jit_miss;
    e->stats.pregen_group = pregen_group;
    e->stats.pregen_ms = ms_pregen;
    e->stats.h2d_ms = ms_h2d;
    e->stats.draw_bytes = draw_bytes;
    e->stats.state_bytes_per_scenario = bytes_per_lane;
    e->stats.state_in_lds = lds_state ? 1u : 0u;
    e->stats.lds_bytes_per_wave = lds_bytes;
    e->stats.waves = waves;
    e->stats.lanes_per_wave = kl;
    e->stats.chunks = n_chunks;
    e->stats.shared_instant_scenarios = n_rerun;
    e->stats.specialised_launches = n_jit;
    e->stats.request_capacity = e->request_capacity;
    e->stats.fifo_capacity = e->fifo_capacity;
    return AF_OK;
}

int af_engine_jit_spec(af_engine_t* e, const af_sweep_t* sweep, const af_outputs_t* out, char* buf, size_t cap) {
    if (!e || !sweep || !out || !buf || cap == 0) return fail(AF_ERR_INVALID, "NULL argument");
    if (sweep->n_scenarios == 0) return fail(AF_ERR_INVALID, "empty sweep");
    if (!e->plan_only) HIP_TRY(hipSetDevice(e->device));
    KArgs a = e->args;
    uint32_t mask = 0;
    for (uint32_t k = 0; k < sweep->n_overrides; ++k) {
        if (sweep->overrides[k].param >= AF_PARAM_COUNT_) return fail(AF_ERR_INVALID, "bad override");
        mask |= 1u << sweep->overrides[k].param;
    }
    a.L = af::make_layout(e->request_capacity, e->fifo_capacity, a.n_edges, a.n_servers, a.n_lb_edges, a.n_rows, mask, a.n_edge_marks, a.n_srv_marks);
    a.clock_cap = out->clock_capacity;
    a.tick_cap = out->tick_capacity;
    a.clock = out->clock;      // only their presence enters the spec
    a.samples = out->samples;
    a.online_hist = out->online_hist;
    a.online_rps = out->online_rps;
    a.n_draw = sweep->draw_capacity ? sweep->draw_capacity : out->clock_capacity;
    if (a.n_draw == 0) return fail(AF_ERR_INVALID, "draw_capacity (or clock_capacity) must be > 0");
    bool on_flow = flow_wanted(e, sweep->n_scenarios);
    FlowPlan FP;
    if (on_flow) {
        for (uint32_t k = 0; k < sweep->n_overrides; ++k)
            if (!sweep->overrides[k].values) return fail(AF_ERR_INVALID, "bad override");
        const int rc = plan_flow(e, a, sweep, out, FP);
        if (rc == AF_ERR_CAPACITY && e->flow_mode != 2u) on_flow = false;   // (af_engine_run: the next-event kernels then)
        else if (rc) return rc;
    }
    if (on_flow) {   // the sweep runs on the stage-parallel kernel: its spec
        const std::string spec = flow_jit_spec_string(e, FP, out, sweep->n_overrides != 0u);
        if (spec.size() + 1 > cap) return fail(AF_ERR_CAPACITY, "spec buffer too small");
        std::memcpy(buf, spec.c_str(), spec.size() + 1);
        return AF_OK;
    }
    // (the next-event kernels' spec depends on the chunking, i.e. on the free device memory: not known without a device)
    if (e->plan_only) return fail(AF_ERR_NO_DEVICE, "planning-only engine: only sweeps of the stage-parallel kernel have a device-independent spec");
    size_t mem_free = 0, mem_total = 0;
    HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
    const uint32_t chunk = chunk_size(e, sweep->n_scenarios, (size_t)(1u + a.n_edges) * a.n_draw * sizeof(double), mem_free);
    if (chunk == 0) return fail(AF_ERR_CAPACITY, "draw_capacity too large for the device memory budget");
    uint32_t kl = 0;
    bool lds_state = false;
    choose_lanes(e, chunk, a.blob_bytes, af::layout_bytes_per_lane(a.L), kl, lds_state);
    uint32_t klog = 0;
    while ((1u << klog) < kl) ++klog;
    const std::string spec = jit_spec_string(a, lds_state, klog);
    if (spec.size() + 1 > cap) return fail(AF_ERR_CAPACITY, "spec buffer too small");
    std::memcpy(buf, spec.c_str(), spec.size() + 1);
    return AF_OK;
}

int af_engine_set_kernels(af_engine_t* e, const char* spec, const void* image, size_t size) {
    if (!e) return fail(AF_ERR_INVALID, "NULL argument");
    if (e->plan_only) return fail(AF_ERR_NO_DEVICE, "planning-only engine (AF_DEVICE_PLAN_ONLY)");
    HIP_TRY(hipSetDevice(e->device));
    const bool unload_all = !spec || !image || size == 0;
    const bool is_flow = spec && std::strstr(spec, "-DAF_FLOW_JIT=1") != nullptr;
    if (e->flow_jit_module && (unload_all || is_flow)) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        (void)hipModuleUnload(e->flow_jit_module);
        e->flow_jit_module = nullptr;
        e->flow_jit_fn = nullptr;
        e->flow_jit_spec.clear();
    }
    if (is_flow && !unload_all) {   // a plan-specialised stage-parallel kernel (one entry point)
        hipModule_t mod = nullptr;
        HIP_TRY(hipModuleLoadData(&mod, image));
        hipFunction_t fn = nullptr;
        if (hipModuleGetFunction(&fn, mod, "af_flow_jit") != hipSuccess) {
            (void)hipModuleUnload(mod);
            return fail(AF_ERR_INVALID, "code object lacks af_flow_jit");
        }
        e->flow_jit_module = mod;
        e->flow_jit_fn = fn;
        e->flow_jit_spec = spec;
        return AF_OK;
    }
    if (e->jit_module) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        (void)hipModuleUnload(e->jit_module);
        e->jit_module = nullptr;
        e->jit_lean = e->jit_order3 = e->jit_order2 = nullptr;
        e->jit_spec.clear();
    }
    if (!spec || !image || size == 0) return AF_OK;  // back to the generic kernels
    hipModule_t mod = nullptr;
    HIP_TRY(hipModuleLoadData(&mod, image));
    hipFunction_t f0 = nullptr, f1 = nullptr, f2 = nullptr;
    if (hipModuleGetFunction(&f0, mod, "af_jit_lean") != hipSuccess || hipModuleGetFunction(&f1, mod, "af_jit_order3") != hipSuccess ||
        hipModuleGetFunction(&f2, mod, "af_jit_order2") != hipSuccess) {
        (void)hipModuleUnload(mod);
        return fail(AF_ERR_INVALID, "code object lacks af_jit_lean / af_jit_order3 / af_jit_order2");
    }
    e->jit_module = mod;
    e->jit_lean = f0;
    e->jit_order3 = f1;
    e->jit_order2 = f2;
    e->jit_spec = spec;
    return AF_OK;
}

int af_engine_summarize(af_engine_t* e, const af_outputs_t* out, const af_summary_t* sum) {
    if (int rc = check_summary_request(e, out, sum)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    // (Measured, round 5: the series kernel on a second stream beside the latency kernel gains nothing -- 4.68 vs 4.65 ms for the
    // pair, both are HBM-bound -- and neither does the analyzer of a finished part of a sweep under the stage-parallel kernel of the
    // next part when every part pays the arrival chain's 5 ms again.  profiles/r05/summary_wpe_ab.txt, overlap_probe.log.  Round 6:
    // af_engine_run_summarized overlaps it with the LAST, partial residency round of ONE launch sequence instead.)
    if (int rc = launch_summary_kernels(e, out, sum, 0u, sum->n_scenarios, e->stream)) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->stats.summary_ms = ms;
    e->stats.summary_overlapped = 0u;
    e->stats.summary_beside_ms = 0.0;
    return AF_OK;
}

// af_engine_run + af_engine_summarize in one call, with the same results.  Where the sweep runs as ONE launch sequence of the
// stage-parallel kernel over alike scenarios (no launch order), the kernel is launched in two parts -- the scenarios of its full
// residency rounds, then the rest -- and the analyzer of the first part runs on a second stream beside the second: the last,
// partial round leaves most of the chip idle (BASELINE config 2: 10 000 scenarios = 2.44 rounds of 4 096 waves; the 0.44 round
// takes 8.5 ms), the analyzer is HBM-bound and the simulation kernel VALU-bound.  Scenarios the kernel hands back are simulated
// again afterwards: their sweep is then summarised once more, in full.
int af_engine_run_summarized(af_engine_t* e, const af_sweep_t* sweep, const af_outputs_t* out, const af_summary_t* sum) {
    if (int rc = check_summary_request(e, out, sum)) return rc;
    if (!sweep || sum->n_scenarios != sweep->n_scenarios) return fail(AF_ERR_INVALID, "summary.n_scenarios must equal sweep.n_scenarios");
    // (a profiler that collects counters runs the kernels of a process ONE AT A TIME -- rocprofv3 --pmc sets
    // ROCPROF_COUNTER_COLLECTION --: nothing can run beside anything, and a dispatch held back behind a stream wait can stall
    // the tool's serialiser.  There, and with AF_NO_SUMMARY_OVERLAP set, the analyzer runs after the simulation.)
    const bool serialised = std::getenv("AF_NO_SUMMARY_OVERLAP") != nullptr || std::getenv("ROCPROF_COUNTER_COLLECTION") != nullptr;
    e->fused_sum = serialised ? nullptr : sum;
    e->fused_out = out;
    e->fused_done = false;
    e->stats.summary_beside_ms = 0.0;
    const int rc = af_engine_run(e, sweep, out);
    e->fused_sum = nullptr;
    e->fused_out = nullptr;
    if (rc != AF_OK) {
        // (whatever went wrong after the second stream began to wait for the kernel's counter: let it through and drain it)
        if (e->h_done_count != nullptr) *e->h_done_count = 0xFFFFFFFFu;
        if (e->stream2 != nullptr) (void)hipStreamSynchronize(e->stream2);
        (void)hipGetLastError();
        return rc;
    }
    const uint32_t n = sum->n_scenarios;
    uint32_t n_lat = 0u, n_ser = 0u;
    if (e->fused_done) {   // what the analyzer beside the kernel met unfinished
        HIP_TRY(hipMemcpy(&n_lat, e->d_retry, 4u, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&n_ser, e->d_retry + (size_t)n + 1u, 4u, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    if (e->fused_done) {
        if (int rc2 = launch_summary_kernels(e, out, sum, 0u, n, e->stream, nullptr, nullptr, e->d_retry + 1u, n_lat, e->d_retry + (size_t)n + 2u, n_ser)) return rc2;
    } else {
        if (int rc2 = launch_summary_kernels(e, out, sum, 0u, n, e->stream)) return rc2;
    }
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->stats.summary_ms = ms;                 // the part of the analyzer that was NOT hidden
    const bool want_lat = sum->stats || sum->rps || sum->hist;
    e->stats.summary_overlapped = e->fused_done ? n - (want_lat ? n_lat : n_ser) : 0u;
    return AF_OK;
}

const char* af_engine_flow_reason(const af_engine_t* e) { return e ? e->flow_reason.c_str() : ""; }

int af_engine_stats(const af_engine_t* e, af_stats_t* stats) {
    if (!e || !stats) return fail(AF_ERR_INVALID, "NULL argument");
    *stats = e->stats;
    return AF_OK;
}

void af_engine_destroy(af_engine_t* e) {
    if (!e) return;
    if (e->plan_only) {   // (owns no device state)
        delete e;
        return;
    }
    (void)hipSetDevice(e->device);
    if (e->d_blob) (void)hipFree(e->d_blob);
    if (e->d_state) (void)hipFree(e->d_state);
    if (e->d_sweep) (void)hipFree(e->d_sweep);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev2) (void)hipEventDestroy(e->ev2);
    if (e->ev3) (void)hipEventDestroy(e->ev3);
    if (e->ev4) (void)hipEventDestroy(e->ev4);
    if (e->ev_s0) (void)hipEventDestroy(e->ev_s0);
    if (e->ev_s1) (void)hipEventDestroy(e->ev_s1);
    if (e->stream2) (void)hipStreamDestroy(e->stream2);
    if (e->d_done_flags) (void)hipFree(e->d_done_flags);
    if (e->d_retry) (void)hipFree(e->d_retry);
    if (e->h_done_count) (void)hipHostFree(e->h_done_count);
    if (e->d_draws) (void)hipFree(e->d_draws);
    if (e->d_pre_flags) (void)hipFree(e->d_pre_flags);
    if (e->d_tie) (void)hipFree(e->d_tie);
    if (e->d_n_shared) (void)hipFree(e->d_n_shared);
    if (e->d_map) (void)hipFree(e->d_map);
    if (e->d_order) (void)hipFree(e->d_order);
    if (e->d_tick) (void)hipFree(e->d_tick);
    if (e->d_arr) (void)hipFree(e->d_arr);
    if (e->d_arr_flags) (void)hipFree(e->d_arr_flags);
    if (e->d_fb) (void)hipFree(e->d_fb);
    if (e->d_slot) (void)hipFree(e->d_slot);
    if (e->jit_module) (void)hipModuleUnload(e->jit_module);
    if (e->flow_jit_module) (void)hipModuleUnload(e->flow_jit_module);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// ---- RCCL, resolved at run time (include/asyncflow_hip.h: "the one collective") ------------------------
namespace {
struct nccl_id_t { char internal[AF_COMM_ID_BYTES]; };
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_id_t*) = nullptr;
    int (*CommInitRank)(void**, int, nccl_id_t, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;      // (optional: af_comm_count)
    int (*CommUserRank)(void*, int*) = nullptr;
    std::string origin;
    bool ok() const { return AllGather != nullptr; }
};
RcclApi g_rccl;

bool rccl_bind(void* h, const std::string& origin) {
    RcclApi r;
    r.handle = h;
    r.origin = origin;
    void* scope = h ? h : RTLD_DEFAULT;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(scope, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(scope, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(scope, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(scope, "ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(scope, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(scope, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(scope, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(scope, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(scope, "ncclCommUserRank"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) return false;
    g_rccl = r;
    return true;
}
int rccl_need() {
    if (g_rccl.ok()) return AF_OK;
    if (rccl_bind(nullptr, "already in the process")) return AF_OK;
    std::vector<std::string> cands;
    if (const char* env = std::getenv("ASYNCFLOW_RCCL_LIB")) cands.push_back(env);
    cands.push_back("librccl.so.1");
    cands.push_back("librccl.so");
    for (const std::string& c : cands)
        if (void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL))
            if (rccl_bind(h, c)) return AF_OK;
    return fail(AF_ERR_INVALID, "RCCL not found: call af_comm_load(path) or set ASYNCFLOW_RCCL_LIB");
}
#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        const int r_ = (expr);                                                                           \
        if (r_ != 0)                                                                                     \
            return fail(AF_ERR_HIP, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); \
    } while (0)
}  // namespace

int af_comm_load(const char* path) {
    if (!path) return rccl_need();
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(AF_ERR_INVALID, std::string("dlopen(") + path + "): " + dlerror());
    if (!rccl_bind(h, path)) return fail(AF_ERR_INVALID, std::string(path) + " does not export the RCCL entry points");
    return AF_OK;
}

int af_comm_unique_id(void* id_out) {
    if (!id_out) return fail(AF_ERR_INVALID, "NULL argument");
    if (int rc = rccl_need()) return rc;
    nccl_id_t id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return AF_OK;
}

int af_comm_init_rank(const void* id, int world_size, int rank, int device, void** comm_out) {
    if (!id || !comm_out || world_size <= 0 || rank < 0 || rank >= world_size) return fail(AF_ERR_INVALID, "bad communicator arguments");
    if (int rc = rccl_need()) return rc;
    HIP_TRY(hipSetDevice(device));
    nccl_id_t nid;
    std::memcpy(&nid, id, sizeof nid);
    void* comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world_size, nid, rank));
    *comm_out = comm;
    return AF_OK;
}

void af_comm_destroy(void* comm) {
    if (comm && g_rccl.ok()) (void)g_rccl.CommDestroy(comm);
}

int af_comm_count(void* comm, int* world_size_out, int* rank_out) {
    if (!comm) return fail(AF_ERR_INVALID, "NULL communicator");
    if (int rc = rccl_need()) return rc;
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(AF_ERR_INVALID, "this RCCL exports neither ncclCommCount nor ncclCommUserRank");
    int n = 0, r = 0;
    RCCL_TRY(g_rccl.CommCount(comm, &n));
    RCCL_TRY(g_rccl.CommUserRank(comm, &r));
    if (world_size_out) *world_size_out = n;
    if (rank_out) *rank_out = r;
    return AF_OK;
}

int af_engine_gather(af_engine_t* e, void* comm, int world_size, const af_summary_t* local, const af_summary_t* out) {
    if (!e || !comm || !local || !out || world_size <= 0) return fail(AF_ERR_INVALID, "NULL argument");
    if (local->n_scenarios == 0) return fail(AF_ERR_INVALID, "empty shard");
    if (out->n_scenarios != local->n_scenarios * (uint32_t)world_size)
        return fail(AF_ERR_INVALID, "gathered.n_scenarios must be world_size * local.n_scenarios");
    if (e->plan_only) return fail(AF_ERR_NO_DEVICE, "planning-only engine (AF_DEVICE_PLAN_ONLY)");
    if (int rc = rccl_need()) return rc;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = local->n_scenarios;
    const uint32_t n_series = e->args.n_series;
    struct Part { const void* src; void* dst; size_t row_bytes; const char* what; };
    const Part parts[] = {
        {local->stats, out->stats, 8u * sizeof(double), "stats"},
        {local->rps, out->rps, (size_t)local->rps_buckets * sizeof(float), "rps"},
        {local->hist, out->hist, (size_t)local->hist_bins * sizeof(uint32_t), "hist"},
        {local->series_mean, out->series_mean, (size_t)n_series * sizeof(double), "series_mean"},
        {local->series_max, out->series_max, (size_t)n_series * sizeof(uint32_t), "series_max"},
    };
    for (const Part& p : parts)
        if ((p.src == nullptr) != (p.dst == nullptr)) return fail(AF_ERR_INVALID, std::string("local / gathered disagree on ") + p.what);
    if (out->rps_buckets != local->rps_buckets || out->hist_bins != local->hist_bins)
        return fail(AF_ERR_INVALID, "local / gathered row lengths differ");
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    RCCL_TRY(g_rccl.GroupStart());
    // (an error inside the group must not leave it open: every later collective of this thread would be queued
    // behind a GroupEnd that never comes -- record the first failure, always close the group, then report)
    int first_err = 0;
    const char* first_what = "";
    for (const Part& p : parts)
        if (p.src && p.row_bytes && first_err == 0) {
            first_err = g_rccl.AllGather(p.src, p.dst, n * p.row_bytes, /* ncclInt8 */ 0, comm, e->stream);
            first_what = p.what;
        }
    const int end_err = g_rccl.GroupEnd();
    if (first_err != 0)
        return fail(AF_ERR_HIP, std::string("ncclAllGather(") + first_what + "): " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(first_err) : "rccl error"));
    RCCL_TRY(end_err);
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->stats.gather_ms = ms;
    return AF_OK;
}

int af_probe_store(int device, int pattern, uint32_t n_waves, uint64_t bytes_per_wave, uint32_t lanes, double* ms_out) {
    if (n_waves == 0 || bytes_per_wave < 1024u || pattern < 0 || pattern > 2) return fail(AF_ERR_INVALID, "bad probe arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(AF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n_dev) return fail(AF_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    const size_t words = (size_t)(bytes_per_wave / 4u) & ~size_t(3);
    uint32_t* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, (size_t)n_waves * words * 4u));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    if (pattern == 0) hipLaunchKernelGGL(af_probe_store_wide, dim3(n_waves), dim3(64), 0, 0, buf, words);
    else if (pattern == 1) hipLaunchKernelGGL(af_probe_store_pairs16, dim3(n_waves), dim3(64), 0, 0, buf, words, lanes ? (lanes > 64u ? 64u : lanes) : 57u);
    else hipLaunchKernelGGL(af_probe_store_rows48, dim3(n_waves), dim3(64), 0, 0, buf, words);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipDeviceSynchronize());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return AF_OK;
}

int af_probe_math(int device, int kind, uint64_t seed, const double* in, const double* in2, double* out, size_t n) {
    if (!in || !out || n == 0) return fail(AF_ERR_INVALID, "NULL argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(AF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n_dev) return fail(AF_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    double *d_in = nullptr, *d_in2 = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, n * 8));
    HIP_TRY(hipMalloc((void**)&d_out, n * 8));
    HIP_TRY(hipMemcpy(d_in, in, n * 8, hipMemcpyHostToDevice));
    if (in2) {
        HIP_TRY(hipMalloc((void**)&d_in2, n * 8));
        HIP_TRY(hipMemcpy(d_in2, in2, n * 8, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(af_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, kind, seed, d_in, d_in2,
                       d_out, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (d_in2) (void)hipFree(d_in2);
    return AF_OK;
}

}  // extern "C"

#endif  // AF_JIT
