// hostcheck.cpp -- TEST-ONLY host instantiation of the engine core.
//
// Compiles asyncflow_amd/csrc/af_core.hpp (the exact per-lane state machine the
// HIP kernel runs) with g++ for ONE lane, so that the CPU test-suite
// (-m "not gpu") can differential-test the kernel logic against the oracle
// without a GPU.  It is built into tests/hostcheck/libaf_hostcheck.so by
// tests/hostcheck/build.py, loaded only by tests/, and is NOT a fallback: the
// asyncflow_amd package neither builds, loads nor knows about it, and raises if
// the HIP library or a GPU is missing.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../asyncflow_amd/csrc/af_plan_pack.hpp"
#include "../../asyncflow_amd/csrc/af_flow_host.hpp"
#include "../../asyncflow_amd/csrc/af_pregen.hpp"
#include "wave_emul.hpp"

namespace {

struct MemHost {
    uint64_t* w;
    uint64_t ld(uint32_t i) const { return w[i]; }
    void st(uint32_t i, uint64_t v) const { w[i] = v; }
};

int g_two_pass = 0, g_reruns = 0;
uint32_t *g_online_hist = nullptr, *g_online_rps = nullptr;
uint32_t g_online_bins = 0, g_online_buckets = 0;
double g_online_scale = 0.0;

}  // namespace

extern "C" int hc_simulate(const af_plan_t* p, uint64_t seed, uint32_t n_ovr, const uint32_t* ovr_param,
                           const uint32_t* ovr_index, const double* ovr_value, uint32_t cap, uint32_t fcap,
                           uint32_t clock_cap, double* clock, uint32_t tick_cap, uint32_t* samples,
                           uint32_t* counts, uint32_t draw_cap) {
    if (!p || p->abi_version != AF_ABI_VERSION || p->struct_size != sizeof(af_plan_t)) return AF_ERR_ABI;
    if (fcap == 0 || (fcap & (fcap - 1)) != 0 || fcap > AF_MAX_FIFO_CAPACITY) return AF_ERR_INVALID;
    af::PackedPlan pk;
    if (!af::pack_plan(*p, pk).empty()) return AF_ERR_INVALID;
    af::PlanView V{};
    af::fill_view_scalars(*p, pk, V);
    V.edge = pk.words.data() + pk.off_edge;
    V.srv = pk.words.data() + pk.off_srv;
    V.ep = pk.words.data() + pk.off_ep;
    V.row = pk.words.data() + pk.off_row;
    V.emark = pk.words.data() + pk.off_emark;
    V.smark = pk.words.data() + pk.off_smark;
    V.lb = pk.words.data() + pk.off_lb;

    uint32_t mask = 0;
    std::vector<uint32_t> idx(n_ovr ? n_ovr : 1, 0u);
    double users_mean = p->gen_users_mean, users_sigma = p->gen_users_sigma, rpm = p->gen_rpm_mean, window_s = p->gen_window_s;
    std::vector<double> e_mean(p->edge_mean, p->edge_mean + p->n_edges), e_sig(p->edge_sigma, p->edge_sigma + p->n_edges),
        e_drop(p->edge_dropout, p->edge_dropout + p->n_edges);
    for (uint32_t k = 0; k < n_ovr; ++k) {
        mask |= 1u << ovr_param[k];
        idx[k] = ovr_param[k] == AF_PARAM_STEP_TIME ? pk.row_of_step[ovr_index[k]] : ovr_index[k];
        switch (ovr_param[k]) {
            case AF_PARAM_GEN_USERS_MEAN: users_mean = ovr_value[k]; break;
            case AF_PARAM_GEN_USERS_SIGMA: users_sigma = ovr_value[k]; break;
            case AF_PARAM_GEN_RPM_MEAN: rpm = ovr_value[k]; break;
            case AF_PARAM_EDGE_MEAN: e_mean[ovr_index[k]] = ovr_value[k]; break;
            case AF_PARAM_EDGE_SIGMA: e_sig[ovr_index[k]] = ovr_value[k]; break;
            case AF_PARAM_EDGE_DROPOUT: e_drop[ovr_index[k]] = ovr_value[k]; break;
            case AF_PARAM_GEN_WINDOW: window_s = ovr_value[k]; break;
            default: break;
        }
    }
    // pre-generation, exactly what af_pregen_arrivals / af_pregen_edges do on the GPU
    const uint32_t n_draw = draw_cap;
    std::vector<double> draws((size_t)(1 + p->n_edges) * (n_draw ? n_draw : 1), 0.0);
    uint32_t flags_in = 0;
    {
        af::GenState g;
        double t = 0.0;
        uint32_t k = 0;
        for (; k < n_draw; ++k) {
            const double gap = af::gen_next_gap(g, seed, p->gen_users_dist, users_mean, users_sigma, rpm,
                                                window_s, p->total_time);
            if (gap < 0.0) break;
            t = t + gap;
            draws[k] = t;
        }
        if (k == n_draw) {
            if (af::gen_next_gap(g, seed, p->gen_users_dist, users_mean, users_sigma, rpm, window_s,
                                 p->total_time) >= 0.0)
                flags_in |= AF_FLAG_DRAW_OVERFLOW;
        }
        for (; k < n_draw; ++k) draws[k] = af::AF_INF;
    }
    for (uint32_t e = 0; e < p->n_edges; ++e)
        for (uint32_t i = 0; i < n_draw; ++i)
            draws[(size_t)(1 + e) * n_draw + i] = af::pre_edge_draw(seed, e, i, p->edge_dist[e], e_mean[e], e_sig[e], e_drop[e]);

    const af::Layout L = af::make_layout(cap, fcap, p->n_edges, p->n_servers, p->n_lb_edges, pk.n_rows, mask, p->n_edge_marks, p->n_srv_marks);
    std::vector<uint64_t> w(L.n_words ? L.n_words : 1, 0ull);
    const uint32_t pitch = (p->n_edges + 3u * p->n_servers + 3u) & ~3u;
    af::LaneOut O{clock, samples, counts, clock_cap, tick_cap, pitch, g_online_hist, g_online_rps, g_online_bins, g_online_buckets,
                  g_online_scale};
    if (g_online_hist) std::memset(g_online_hist, 0, 4u * g_online_bins);
    if (g_online_rps) std::memset(g_online_rps, 0, 4u * g_online_buckets);
    std::vector<uint64_t> tie(L.tie_words, 0ull);
    af::PreDraws D{draws.data(), n_draw, flags_in, tie.data()};
    if (g_two_pass && !V.every_event_in_order) {  // what af_engine_run does: lean variant first, SimPy-order variant on demand
        af::Lane<MemHost, false> lean(V, L, MemHost{w.data()}, O, D, seed);
        lean.init(ovr_param, idx.data(), n_ovr, [&](uint32_t k) { return ovr_value[k]; });
        while (lean.round() != lean.ROUND_STOP) {
        }
        lean.write_counts();
        g_reruns += (lean.flags & af::FLAG_SHARED_INSTANT) ? 1 : 0;
        if (!(lean.flags & af::FLAG_SHARED_INSTANT)) return 0;
        std::fill(w.begin(), w.end(), 0xDEADBEEFDEADBEEFull);  // the second pass starts over
        if (g_online_hist) std::memset(g_online_hist, 0, 4u * g_online_bins);
        if (g_online_rps) std::memset(g_online_rps, 0, 4u * g_online_buckets);
    }
    af::Lane<MemHost, true> lane(V, L, MemHost{w.data()}, O, D, seed);
    lane.init(ovr_param, idx.data(), n_ovr, [&](uint32_t k) { return ovr_value[k]; });
    for (;;) {
        const auto st = lane.round();
        if (st == lane.ROUND_STOP) break;
        if (st == lane.ROUND_SHARED) lane.shared_instant();
    }
    lane.write_counts();
    return 0;
}

extern "C" void hc_set_two_pass(int on) { g_two_pass = on; }
extern "C" void hc_set_test_quantum(int bits) { af::g_test_quantum_bits = bits; }   // af_math.hpp: test-only tie generator
extern "C" void hc_set_online(uint32_t* hist, uint32_t bins, double hist_max, uint32_t* rps, uint32_t buckets) {
    g_online_hist = hist;
    g_online_bins = bins;
    g_online_scale = hist ? (double)bins / hist_max : 0.0;
    g_online_rps = rps;
    g_online_buckets = buckets;
}
extern "C" int hc_reruns(void) { return g_reruns; }

extern "C" uint64_t hc_bytes_per_lane(uint32_t cap, uint32_t fcap, uint32_t n_edges, uint32_t n_servers,
                                      uint32_t n_lb, uint32_t n_rows, uint32_t mask) {
    return af::layout_bytes_per_lane(af::make_layout(cap, fcap, n_edges, n_servers, n_lb, n_rows, mask));
}

// ---- the stage-parallel kernel (af_flow.hpp) on the 64-fibre wave emulator ---------------------------
namespace {
std::string g_flow_reason;
}

extern "C" const char* hc_flow_reason(void) { return g_flow_reason.c_str(); }

// Returns 0 when the scenario ran (counts[AF_CNT_FLAGS] may carry FLAG_FLOW_FALLBACK = 1 << 8 and the reason
// bits 9..12: the product then re-runs the scenario on the sequential kernels), 1 when the plan is not
// eligible for the flow kernel (hc_flow_reason() says why), < 0 on invalid arguments.
extern "C" int hc_flow_simulate(const af_plan_t* p, uint64_t seed, uint32_t n_ovr, const uint32_t* ovr_param,
                                const uint32_t* ovr_index, const double* ovr_value, uint32_t ipl, uint32_t ring_rows,
                                uint32_t clock_cap, double* clock, uint32_t tick_cap, uint32_t* samples, uint32_t* counts,
                                uint32_t draw_cap) {
    if (!p || p->abi_version != AF_ABI_VERSION || p->struct_size != sizeof(af_plan_t)) return AF_ERR_ABI;
    // the second-chance instantiation (FEAT_TIEBREAK | FEAT_BIGLIST): send times + lists of their own lengths;
    // bits 16..31 = entries of the long list(s) (0: 256), bits 12..14 = which list is long (0: all four; else index + 1, the others hold 256)
    const bool gen_srv = aff::flow_needs_general_servers(*p);   // (several endpoints per server, ...: the long-list instantiation with the general server station)
    const bool robust = (ipl & 0x100u) != 0u || gen_srv;
    const bool near_only = (ipl & 0x200u) != 0u;   // the lean instantiations without FEAT_FAR (the sender enters both ends of every message)
    const uint32_t big_cap = (ipl >> 16) ? (ipl >> 16) : 256u, big_which = (ipl >> 12) & 7u;
    ipl &= 0xFFu;
    if (robust) ipl = 4;
    if (ipl != 1 && ipl != 2 && ipl != 4) return AF_ERR_INVALID;
    g_flow_reason = aff::flow_ineligible_reason(*p);
    if (!g_flow_reason.empty()) return 1;
    af::PackedPlan pk;
    if (!af::pack_plan(*p, pk).empty()) return AF_ERR_INVALID;

    double users_mean = p->gen_users_mean, users_sigma = p->gen_users_sigma, rpm = p->gen_rpm_mean, window_s = p->gen_window_s;
    std::vector<uint32_t> idx(n_ovr ? n_ovr : 1, 0u);
    for (uint32_t k = 0; k < n_ovr; ++k) {
        idx[k] = ovr_param[k] == AF_PARAM_STEP_TIME ? pk.row_of_step[ovr_index[k]] : ovr_index[k];
        if (ovr_param[k] == AF_PARAM_GEN_WINDOW) window_s = ovr_value[k];
        if (ovr_param[k] == AF_PARAM_GEN_USERS_MEAN) users_mean = ovr_value[k];
        if (ovr_param[k] == AF_PARAM_GEN_USERS_SIGMA) users_sigma = ovr_value[k];
        if (ovr_param[k] == AF_PARAM_GEN_RPM_MEAN) rpm = ovr_value[k];
    }
    // arrivals, exactly what af_pregen_arrivals does on the GPU
    const uint32_t n_draw = draw_cap ? draw_cap : 1u;
    std::vector<double> arrivals(n_draw, af::AF_INF);
    uint32_t flags_in = 0;
    {
        af::GenState g;
        double t = 0.0;
        uint32_t k = 0;
        for (; k < n_draw; ++k) {
            const double gap = af::gen_next_gap(g, seed, p->gen_users_dist, users_mean, users_sigma, rpm, window_s, p->total_time);
            if (gap < 0.0) break;
            t = t + gap;
            arrivals[k] = t;
        }
        if (k == n_draw && af::gen_next_gap(g, seed, p->gen_users_dist, users_mean, users_sigma, rpm, window_s, p->total_time) >= 0.0)
            flags_in |= AF_FLAG_DRAW_OVERFLOW;
    }
    const aff::TickTable tt = aff::make_tick_table(p->sample_period, p->total_time);

    aff::FlowArgs a{};
    a.total_time = p->total_time;
    a.sample_period = p->sample_period;
    a.inv_period = tt.inv_period;
    a.tick_eps = tt.eps;
    a.metrics_mask = p->metrics_mask;
    a.gen_out_edge = (uint32_t)p->gen_out_edge;
    a.client_out_edge = (uint32_t)p->client_out_edge;
    a.n_edges = p->n_edges;
    a.n_servers = p->n_servers;
    a.has_lb = p->has_lb;
    a.n_lb_edges = p->n_lb_edges;
    a.lb_least_connections = aff::lc_edges(*p) != 0u ? 1u : 0u;
    a.n_edge_marks = p->n_edge_marks;
    a.n_srv_marks = p->n_srv_marks;
    aff::flow_step_maxima(*p, a.max_pre, a.max_cpu, a.max_post);
    a.ram_scale = aff::flow_ram_scale(*p);
    a.ram_unit = 1.0 / a.ram_scale;
    a.off_edge = pk.off_edge; a.off_srv = pk.off_srv; a.off_ep = pk.off_ep; a.off_row = pk.off_row;
    a.off_emark = pk.off_emark; a.off_smark = pk.off_smark; a.off_lb = pk.off_lb;
    a.blob_bytes = (uint32_t)(pk.words.size() * 8u);
    a.blob = reinterpret_cast<const unsigned char*>(pk.words.data());
    a.L = aff::choose_flow_layout(*p, ipl, ring_rows);
    {   // per-scenario server resources (AF_PARAM_SRV_CORES / _RAM_MB): the rings must hold the overridden sizes (af_engine_run: plan_flow)
        uint32_t c_ring = a.L.c_ring, g_ring = a.L.g_ring;
        for (uint32_t k = 0; k < n_ovr; ++k) {
            if (ovr_param[k] == AF_PARAM_SRV_CORES && (uint32_t)ovr_value[k] > c_ring) c_ring = (uint32_t)ovr_value[k];
            if (ovr_param[k] == AF_PARAM_SRV_RAM_MB) {
                const double scale = ovr_value[k] / p->srv_ram_mb[ovr_index[k]];
                const double want = std::min(256.0, std::ceil((double)a.L.g_ring * scale));
                if (want > (double)g_ring) g_ring = aff::pow2_ge((uint32_t)want);
            }
        }
        if (c_ring != a.L.c_ring || g_ring != a.L.g_ring)
            a.L = aff::make_flow_layout(a.L.cap, a.L.ring_rows, g_ring, c_ring, p->n_edges, p->n_servers, p->n_edge_marks);
    }
    if (gen_srv && !robust) {   // general servers, first launch (engine.hip: plan_flow, gen_compact): 128-entry lists without send times
        uint32_t caps4[4] = {128u, 128u, 128u, 128u};
        a.L = aff::make_flow_layout(0u, a.L.ring_rows, a.L.g_ring, a.L.c_ring, p->n_edges, p->n_servers, p->n_edge_marks, false, caps4, true);
        a.L.win_rows = a.L.ring_rows / 2u;
    }
    if (robust) {
        uint32_t caps4[4];
        for (uint32_t s = 0; s < 4u; ++s) caps4[s] = (big_which == 0u || big_which == s + 1u) ? big_cap : 256u;
        a.L = aff::make_flow_layout(0u, a.L.ring_rows, a.L.g_ring, a.L.c_ring, p->n_edges, p->n_servers, p->n_edge_marks, true, caps4, gen_srv);
        a.L.win_rows = a.L.ring_rows / 2u;
    }
    a.tick_t = tt.t.data();
    a.n_ticks = (uint32_t)tt.t.size();
    a.n_scen = 1;
    a.seeds = &seed;
    a.n_ovr = n_ovr;
    a.ovr_param = ovr_param;
    a.ovr_index = idx.data();
    a.ovr_values = ovr_value;
    a.ovr_stride = 1;
    a.arrivals = arrivals.data();
    a.n_draw = n_draw;
    a.pre_flags = &flags_in;
    a.clock = clock;
    a.clock_cap = clock_cap;
    a.samples = samples;
    a.tick_cap = tick_cap;
    a.counts = counts;
    a.online_hist = g_online_hist;
    a.online_rps = g_online_rps;
    a.online_hist_bins = g_online_bins;
    a.online_rps_buckets = g_online_buckets;
    a.online_hist_scale = g_online_scale;
    if (g_online_hist) std::memset(g_online_hist, 0, 4u * g_online_bins);
    if (g_online_rps) std::memset(g_online_rps, 0, 4u * g_online_buckets);
    a.n_fallback = nullptr;

    std::vector<uint64_t> lds(pk.words.size() + a.L.n_words + 2u, 0xDEADBEEFDEADBEEFull);
    // the instantiation the engine would launch: the lean one when the launch needs none of the optional features
    const bool lean = p->n_edge_marks == 0 && p->n_srv_marks == 0 && !g_online_hist && !g_online_rps && (a.L.ring_rows != 0 || !samples);
    const bool lc = a.lb_least_connections != 0u;
    const bool marks_only = !lean && !g_online_hist && !g_online_rps && (a.L.ring_rows != 0 || !samples);   // (engine.hip: config-4-like launches)
    constexpr uint32_t kAll = aff::FEAT_ALL, kRobust = aff::FEAT_ALL | aff::FEAT_TIEBREAK | aff::FEAT_BIGLIST, kLC = aff::FEAT_LC;
    constexpr uint32_t kGen = aff::FEAT_GENSRV;
    const bool chain = aff::flow_needs_chain(*p);   // servers feed servers: the FEAT_CHAIN instantiations (engine.hip: plan_flow)
    constexpr uint32_t kChain = aff::FEAT_CHAIN;
    auto body = [&]() {
        if (chain && gen_srv && lc && !robust) { aff::Flow<emu::WaveEmu, 1, kAll | aff::FEAT_BIGLIST | kGen | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && gen_srv && lc) { aff::Flow<emu::WaveEmu, 1, kRobust | kGen | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && gen_srv && !robust) { aff::Flow<emu::WaveEmu, 1, kAll | aff::FEAT_BIGLIST | kGen | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && gen_srv) { aff::Flow<emu::WaveEmu, 1, kRobust | kGen | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && lc && robust) { aff::Flow<emu::WaveEmu, 1, kRobust | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && lc && ipl == 1) { aff::Flow<emu::WaveEmu, 1, kAll | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && lc && ipl == 2) { aff::Flow<emu::WaveEmu, 2, kAll | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && lc) { aff::Flow<emu::WaveEmu, 4, kAll | kLC | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && robust) { aff::Flow<emu::WaveEmu, 1, kRobust | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && ipl == 1) { aff::Flow<emu::WaveEmu, 1, kAll | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain && ipl == 2) { aff::Flow<emu::WaveEmu, 2, kAll | kChain> f(a); f.run(lds.data(), 0u); }
        else if (chain) { aff::Flow<emu::WaveEmu, 4, kAll | kChain> f(a); f.run(lds.data(), 0u); }
        else if (gen_srv && lc && !robust) { aff::Flow<emu::WaveEmu, 1, kAll | aff::FEAT_BIGLIST | kLC | kGen> f(a); f.run(lds.data(), 0u); }
        else if (gen_srv && !robust) { aff::Flow<emu::WaveEmu, 1, kAll | aff::FEAT_BIGLIST | kGen> f(a); f.run(lds.data(), 0u); }
        else if (gen_srv && lc) { aff::Flow<emu::WaveEmu, 1, kRobust | kLC | kGen> f(a); f.run(lds.data(), 0u); }
        else if (gen_srv) { aff::Flow<emu::WaveEmu, 1, kRobust | kGen> f(a); f.run(lds.data(), 0u); }
        else if (robust && lc) { aff::Flow<emu::WaveEmu, 1, kRobust | kLC> f(a); f.run(lds.data(), 0u); }
        else if (robust) { aff::Flow<emu::WaveEmu, 1, kRobust> f(a); f.run(lds.data(), 0u); }
        else if (lc && ipl == 1) { aff::Flow<emu::WaveEmu, 1, kAll | kLC> f(a); f.run(lds.data(), 0u); }
        else if (lc && ipl == 2) { aff::Flow<emu::WaveEmu, 2, kAll | kLC> f(a); f.run(lds.data(), 0u); }
        else if (lc) { aff::Flow<emu::WaveEmu, 4, kAll | kLC> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 1 && lean && near_only) { aff::Flow<emu::WaveEmu, 1, 0u> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 1 && lean) { aff::Flow<emu::WaveEmu, 1, aff::FEAT_FAR> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 1 && marks_only) { aff::Flow<emu::WaveEmu, 1, aff::FEAT_MARKS | aff::FEAT_FAR> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 1) { aff::Flow<emu::WaveEmu, 1> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 2 && lean && near_only) { aff::Flow<emu::WaveEmu, 2, 0u> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 2 && lean) { aff::Flow<emu::WaveEmu, 2, aff::FEAT_FAR> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 2 && marks_only) { aff::Flow<emu::WaveEmu, 2, aff::FEAT_MARKS | aff::FEAT_FAR> f(a); f.run(lds.data(), 0u); }
        else if (ipl == 2) { aff::Flow<emu::WaveEmu, 2> f(a); f.run(lds.data(), 0u); }
        else { aff::Flow<emu::WaveEmu, 4> f(a); f.run(lds.data(), 0u); }
    };
    emu::run_wave(body);
    return 0;
}

extern "C" uint64_t hc_flow_lds_bytes(const af_plan_t* p, uint32_t ipl, uint32_t ring_rows) {
    af::PackedPlan pk;
    if (!p || !af::pack_plan(*p, pk).empty()) return 0;
    return 8ull * (pk.words.size() + aff::choose_flow_layout(*p, ipl, ring_rows).n_words);
}

// ---- elementary functions of af_math.hpp on the host (the device build of the same header is probed by af_probe_math) ----
// kind 1 log, 2 exp, 3 normal quantile, 7 log of a normal number in (0, 1]
extern "C" void hc_math(int kind, const double* in, double* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i)
        out[i] = kind == 1 ? af::af_log(in[i]) : kind == 2 ? af::af_exp(in[i]) : kind == 3 ? af::af_norminv(in[i]) : af::af_log_unit(in[i]);
}

// ---- the arrival sampler (af_pregen.hpp) ------------------------------------------------------------------------------
// which = 0: af::gen_next_gap, the sequential statement (what the oracle follows); which = 1 / 2: the per-lane functions of
// af_arrival_groups driven the way the kernel drives ONE lane (window_start, then steps of eight unit variates), with the
// hoisted-reciprocal division (1) or the plain one (2).  Returns the number of arrivals.
extern "C" int64_t hc_arrivals(int which, uint64_t seed, uint32_t dist, double mean, double sigma, double rpm, double window_s,
                               double T, uint32_t n_draw, double* out, uint32_t* flags) {
    *flags = 0u;
    for (uint32_t i = 0; i < n_draw; ++i) out[i] = af::AF_INF;
    if (which == 0) {
        af::GenState g;
        double t = 0.0;
        uint32_t k = 0;
        for (; k < n_draw; ++k) {
            const double gap = af::gen_next_gap(g, seed, dist, mean, sigma, rpm, window_s, T);
            if (gap < 0.0) break;
            t = t + gap;
            out[k] = t;
        }
        if (k == n_draw && af::gen_next_gap(g, seed, dist, mean, sigma, rpm, window_s, T) >= 0.0) *flags |= AF_FLAG_DRAW_OVERFLOW;
        return k;
    }
    afp::Lane L;
    afp::lane_init(L);
    const double rps_per_user = rpm / 60.0;
    while (L.state != afp::LANE_DONE) {
        if (L.state == afp::LANE_WAIT) {
            afp::window_start(L, T, window_s, rps_per_user,
                              [&](uint32_t idx) { return afp::users_draw(dist, mean, sigma, seed, idx); });
            continue;
        }
        double e[afp::kBatch];
        for (uint32_t j = 0; j < afp::kBatch; ++j) e[j] = afp::unit_variate(seed, L.draws + j);
        if (which == 1 && L.fast_div) afp::lane_step<true>(L, e, T, n_draw, out);
        else afp::lane_step<false>(L, e, T, n_draw, out);
    }
    *flags = L.flags;
    return L.k;
}
