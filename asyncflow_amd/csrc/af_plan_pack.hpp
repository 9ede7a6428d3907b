// af_plan_pack.hpp -- HOST-side packing of an `af_plan_t` (include/asyncflow_hip.h) into the
// 64-bit record arrays the kernel reads (af_core.hpp::PlanView).  Plain C++, no HIP.
#pragma once

#include <cstring>
#include <string>
#include <vector>

#include "../../include/asyncflow_hip.h"
#include "af_core.hpp"

namespace af {

struct PackedPlan {
    std::vector<uint64_t> words;
    uint32_t off_edge = 0, off_srv = 0, off_ep = 0, off_row = 0, off_emark = 0, off_smark = 0, off_lb = 0;
    uint32_t n_rows = 0;
    std::vector<uint32_t> row_of_step;  // plan step index -> step row (END rows interleaved)
};

// Returns an empty string on success, else the reason the plan cannot be represented.
inline std::string pack_plan(const af_plan_t& p, PackedPlan& out) {
    if (p.n_edges == 0 || p.n_edges > 255) return "1..255 edges supported";
    if (p.n_servers > 255) return "at most 255 servers supported";
    if (p.n_endpoints > 65535) return "at most 65535 endpoints supported";
    if ((uint64_t)p.n_steps + p.n_endpoints > 65535) return "at most 65535 step rows supported";
    auto& w = out.words;
    w.clear();
    auto d = [](double x) { return d2u(x); };

    out.off_edge = (uint32_t)w.size();
    for (uint32_t e = 0; e < p.n_edges; ++e) {
        w.push_back(d(p.edge_mean[e]));
        w.push_back(d(p.edge_sigma[e]));
        w.push_back(d(p.edge_dropout[e]));
        const uint32_t tidx = p.edge_target_kind[e] == AF_NODE_SERVER ? (uint32_t)p.edge_target_idx[e] : 0u;
        w.push_back((uint64_t)p.edge_target_kind[e] | ((uint64_t)tidx << 8) | ((uint64_t)p.edge_dist[e] << 16));
    }
    out.off_srv = (uint32_t)w.size();
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        if (p.srv_cores[s] == 0 || p.srv_cores[s] > 65535) return "cpu_cores must be within 1..65535";
        const uint64_t n_ep = p.srv_ep_begin[s + 1] - p.srv_ep_begin[s];
        w.push_back(d(p.srv_ram_mb[s]));
        w.push_back((uint64_t)p.srv_cores[s] | ((uint64_t)(uint32_t)p.srv_out_edge[s] << 16) |
                    ((uint64_t)p.srv_ep_begin[s] << 32) | (n_ep << 48));
    }
    // step rows first (to know each endpoint's first row), then endpoint records
    std::vector<uint64_t> rows;
    std::vector<uint32_t> first_row(p.n_endpoints ? p.n_endpoints : 1, 0u);
    out.row_of_step.assign(p.n_steps ? p.n_steps : 1, 0u);
    for (uint32_t ep = 0; ep < p.n_endpoints; ++ep) {
        first_row[ep] = (uint32_t)(rows.size() / TREC);
        for (uint32_t i = p.ep_step_begin[ep]; i < p.ep_step_begin[ep + 1]; ++i) {
            out.row_of_step[i] = (uint32_t)(rows.size() / TREC);
            rows.push_back(d(p.step_time[i]));
            rows.push_back(d(p.ep_ram[ep]));
            rows.push_back(p.step_kind[i] == AF_STEP_CPU ? (uint64_t)STEP_CPU : (uint64_t)STEP_IO);
        }
        rows.push_back(d(0.0));
        rows.push_back(d(p.ep_ram[ep]));
        rows.push_back((uint64_t)STEP_END);
    }
    out.n_rows = (uint32_t)(rows.size() / TREC);
    out.off_ep = (uint32_t)w.size();
    for (uint32_t ep = 0; ep < p.n_endpoints; ++ep) {
        w.push_back(d(p.ep_ram[ep]));
        w.push_back((uint64_t)first_row[ep]);
    }
    out.off_row = (uint32_t)w.size();
    w.insert(w.end(), rows.begin(), rows.end());
    out.off_emark = (uint32_t)w.size();
    for (uint32_t i = 0; i < p.n_edge_marks; ++i) {
        w.push_back(d(p.emark_time[i]));
        w.push_back(d(p.emark_delta[i]));
        w.push_back((uint64_t)(uint32_t)p.emark_edge[i]);
    }
    out.off_smark = (uint32_t)w.size();
    for (uint32_t i = 0; i < p.n_srv_marks; ++i) {
        w.push_back(d(p.smark_time[i]));
        w.push_back((uint64_t)(uint32_t)(p.smark_lb_edge[i] + 1) | ((uint64_t)(p.smark_down[i] ? 1u : 0u) << 32));
    }
    out.off_lb = (uint32_t)w.size();
    for (uint32_t i = 0; i < p.n_lb_edges; ++i) w.push_back((uint64_t)(uint32_t)p.lb_edges[i]);
    if (w.size() % 2) w.push_back(0);  // keep the blob a multiple of 16 bytes
    if (w.empty()) w.assign(2, 0);
    return std::string();
}

// Fill the scalar part of a PlanView from the plan (pointers are set by the caller).
// A server whose out-edge leads to another server or to the load balancer AND can deliver with zero
// latency (an atom at 0: poisson, or normal truncated at 0): the zero-delay Timeout of such a
// delivery is queued between the zero-time steps of the sending server's cascade, which the inline
// cascades cannot express.  Such plans (none of the reference's own examples) run every request event
// through the SimPy-order path.
// So do plans with a RAM need that is not a multiple of 1/256 MB (100.3, 64.7): the f64 sums of such needs round, simpy's
// `Container._do_put` (`if capacity - level >= amount`) can then refuse a put by one rounding, and the response waits for the
// next RAM get of that server (server.py:270-276) -- modelled by the SimPy-order path only (af_core.hpp::m_srv_finish).  Whole
// megabytes and multiples of 1/256 MB (every example of the reference) have exact sums and never meet it.
inline bool every_event_in_order(const af_plan_t& p) {
    for (uint32_t ep = 0; ep < p.n_endpoints; ++ep) {
        const double fine = p.ep_ram[ep] * 256.0;
        if (fine != (double)(int64_t)fine || p.ep_ram[ep] > 4194304.0) return true;
    }
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        const double fine = p.srv_ram_mb[s] * 256.0;
        if (fine != (double)(int64_t)fine) return true;
    }
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        const int32_t e = p.srv_out_edge[s];
        if (e < 0) continue;
        const bool atom_at_zero = p.edge_dist[e] == AF_DIST_POISSON || p.edge_dist[e] == AF_DIST_NORMAL;
        if (p.edge_target_kind[e] != AF_NODE_CLIENT && atom_at_zero) return true;
    }
    return false;
}

inline void fill_view_scalars(const af_plan_t& p, const PackedPlan& pk, PlanView& V) {
    V.total_time = p.total_time;
    V.sample_period = p.sample_period;
    V.gen_users_mean = p.gen_users_mean;
    V.gen_users_sigma = p.gen_users_sigma;
    V.gen_rpm_mean = p.gen_rpm_mean;
    V.gen_window_s = p.gen_window_s;
    V.metrics_mask = p.metrics_mask;
    V.gen_users_dist = p.gen_users_dist;
    V.gen_out_edge = (uint32_t)p.gen_out_edge;
    V.client_out_edge = (uint32_t)p.client_out_edge;
    V.n_edges = p.n_edges;
    V.n_servers = p.n_servers;
    V.lb_algo = p.lb_algo;
    V.n_lb_edges = p.n_lb_edges;
    V.n_rows = pk.n_rows;
    V.n_edge_marks = p.n_edge_marks;
    V.n_srv_marks = p.n_srv_marks;
    V.every_event_in_order = every_event_in_order(p) ? 1u : 0u;
}

}  // namespace af
