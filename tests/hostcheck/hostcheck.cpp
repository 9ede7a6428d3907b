// hostcheck.cpp -- TEST-ONLY host instantiation of the engine core.
//
// Compiles asyncflow_amd/csrc/af_core.hpp (the exact per-lane state machine the
// HIP kernel runs) with g++ for ONE lane, so that the CPU test-suite
// (-m "not gpu") can differential-test the kernel logic against the oracle
// without a GPU.  It is built into tests/hostcheck/libaf_hostcheck.so by
// tests/hostcheck/build.py, loaded only by tests/, and is NOT a fallback: the
// asyncflow_amd package neither builds, loads nor knows about it, and raises if
// the HIP library or a GPU is missing.
#include <cstring>
#include <vector>

#include "../../asyncflow_amd/csrc/af_core.hpp"
#include "../../include/asyncflow_hip.h"

namespace {

struct MemHost {
    double* d;
    uint32_t* w;
    double ld64(uint32_t i) const { return d[i]; }
    void st64(uint32_t i, double v) const { d[i] = v; }
    uint32_t ld32(uint32_t i) const { return w[i]; }
    void st32(uint32_t i, uint32_t v) const { w[i] = v; }
};

template <class T>
std::vector<uint32_t> widen(const T* p, size_t n) {
    std::vector<uint32_t> v(n ? n : 1);
    for (size_t i = 0; i < n; ++i) v[i] = (uint32_t)p[i];
    return v;
}

}  // namespace

extern "C" int hc_simulate(const af_plan_t* p, uint64_t seed, uint32_t n_ovr, const uint32_t* ovr_param,
                           const uint32_t* ovr_index, const double* ovr_value, uint32_t cap, uint32_t fcap,
                           uint32_t clock_cap, double* clock, uint32_t tick_cap, uint32_t* samples,
                           uint32_t* counts) {
    if (!p || p->abi_version != AF_ABI_VERSION || p->struct_size != sizeof(af_plan_t)) return AF_ERR_ABI;
    if (fcap == 0 || (fcap & (fcap - 1)) != 0) return AF_ERR_INVALID;
    auto tk = widen(p->edge_target_kind, p->n_edges);
    auto ed = widen(p->edge_dist, p->n_edges);
    auto sk = widen(p->step_kind, p->n_steps);
    auto sd = widen(p->smark_down, p->n_srv_marks);

    af::PlanView V{};
    V.total_time = p->total_time;
    V.sample_period = p->sample_period;
    V.gen_users_mean = p->gen_users_mean;
    V.gen_users_sigma = p->gen_users_sigma;
    V.gen_rpm_mean = p->gen_rpm_mean;
    V.gen_window_s = p->gen_window_s;
    V.metrics_mask = p->metrics_mask;
    V.gen_users_dist = p->gen_users_dist;
    V.gen_out_edge = p->gen_out_edge;
    V.client_out_edge = p->client_out_edge;
    V.n_edges = p->n_edges;
    V.n_servers = p->n_servers;
    V.lb_algo = p->lb_algo;
    V.n_lb_edges = p->n_lb_edges;
    V.n_endpoints = p->n_endpoints;
    V.n_steps = p->n_steps;
    V.n_edge_marks = p->n_edge_marks;
    V.n_srv_marks = p->n_srv_marks;
    V.e_mean = p->edge_mean;
    V.e_sigma = p->edge_sigma;
    V.e_drop = p->edge_dropout;
    V.s_ram = p->srv_ram_mb;
    V.ep_ram = p->ep_ram;
    V.st_time = p->step_time;
    V.em_time = p->emark_time;
    V.em_delta = p->emark_delta;
    V.sm_time = p->smark_time;
    V.lb_edges = p->lb_edges;
    V.e_tkind = tk.data();
    V.e_tidx = p->edge_target_idx;
    V.e_dist = ed.data();
    V.s_cores = p->srv_cores;
    V.s_out = p->srv_out_edge;
    V.s_epb = p->srv_ep_begin;
    V.ep_stepb = p->ep_step_begin;
    V.st_kind = sk.data();
    V.em_edge = p->emark_edge;
    V.sm_edge = p->smark_lb_edge;
    V.sm_down = sd.data();

    uint32_t mask = 0;
    for (uint32_t k = 0; k < n_ovr; ++k) mask |= 1u << ovr_param[k];
    const af::Layout L = af::make_layout(cap, fcap, p->n_edges, p->n_servers, p->n_lb_edges, p->n_steps, mask);
    std::vector<double> d(L.n_d ? L.n_d : 1, 0.0);
    std::vector<uint32_t> w(L.n_w ? L.n_w : 1, 0u);
    af::LaneOut O{clock, samples, counts, clock_cap, tick_cap};
    af::Lane<MemHost> lane(V, L, MemHost{d.data(), w.data()}, O, seed);
    lane.init(ovr_param, ovr_index, n_ovr, [&](uint32_t k) { return ovr_value[k]; });
    while (lane.round()) {
    }
    lane.write_counts();
    return 0;
}

extern "C" uint64_t hc_bytes_per_lane(uint32_t cap, uint32_t fcap, uint32_t n_edges, uint32_t n_servers,
                                      uint32_t n_lb, uint32_t n_steps, uint32_t mask) {
    return af::layout_bytes_per_lane(af::make_layout(cap, fcap, n_edges, n_servers, n_lb, n_steps, mask));
}
