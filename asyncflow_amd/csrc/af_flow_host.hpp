// af_flow_host.hpp -- HOST-side helpers of the stage-parallel kernel (af_flow.hpp): which plans it can
// run, the tick-time table, the LDS layout.  Plain C++, no HIP (also used by the test-only wave emulator).
#pragma once

#include <cmath>
#include <string>
#include <vector>

#include "../../include/asyncflow_hip.h"
#include "af_flow.hpp"

namespace aff {

// Servers the tandem recurrence of Flow::servers_solve cannot express (round 3, FEAT_GENSRV): several endpoints per server
// (server.py:101), or a step program that is not IO* CPU* IO* (it comes back to the core queue after an I/O step).  Such plans
// run on the instantiation whose server station simulates each server event by event (Flow::gen_servers).
inline bool flow_needs_general_servers(const af_plan_t& p) {
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        if (p.srv_ep_begin[s + 1] - p.srv_ep_begin[s] != 1u) return true;
        const uint32_t ep = p.srv_ep_begin[s];
        uint32_t phase = 0;  // 0 leading IO, 1 CPU, 2 trailing IO
        for (uint32_t i = p.ep_step_begin[ep]; i < p.ep_step_begin[ep + 1]; ++i) {
            const bool cpu = p.step_kind[i] == AF_STEP_CPU;
            if (phase == 0 && cpu) phase = 1;
            else if (phase == 1 && !cpu) phase = 2;
            else if (phase == 2 && cpu) return true;
        }
    }
    return false;
}

// Servers that feed servers (FEAT_CHAIN, round 4): level 0 = fed by the client / the load balancer only, level k = its deepest
// feeding server is of level k - 1.  Round 5: a server may feed the LOAD BALANCER (client -> server chain -> LB -> servers ->
// client): every server behind the LB is then deeper than that server.  Returns the number of levels (1: no server feeds a server),
// 0 when the servers feed each other in a cycle; `level` (may be null) gets each server's level.  (Flow::run works the same levels
// out on the device.)
inline uint32_t flow_server_levels(const af_plan_t& p, uint32_t* level) {
    std::vector<uint32_t> lv(p.n_servers, 0u);
    uint32_t deepest = 0u;
    auto raise = [&](uint32_t w, uint32_t to, bool& changed) {
        if (lv[w] < to) {
            lv[w] = to;
            changed = true;
            deepest = lv[w] > deepest ? lv[w] : deepest;
        }
    };
    for (uint32_t pass = 0; pass <= p.n_servers; ++pass) {
        bool changed = false;
        for (uint32_t s = 0; s < p.n_servers; ++s) {
            const int32_t e = p.srv_out_edge[s];
            if (p.edge_target_kind[e] == AF_NODE_LB) {
                for (uint32_t i = 0; i < p.n_lb_edges; ++i)
                    if (p.edge_target_kind[p.lb_edges[i]] == AF_NODE_SERVER) raise((uint32_t)p.edge_target_idx[p.lb_edges[i]], lv[s] + 1u, changed);
                continue;
            }
            if (p.edge_target_kind[e] != AF_NODE_SERVER) continue;
            raise((uint32_t)p.edge_target_idx[e], lv[s] + 1u, changed);
        }
        if (!changed) break;
        if (pass == p.n_servers) return 0u;   // still growing after n_servers passes: a cycle
    }
    if (level)
        for (uint32_t s = 0; s < p.n_servers; ++s) level[s] = lv[s];
    return deepest + 1u;
}
inline bool flow_needs_chain(const af_plan_t& p) {
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        const uint32_t k = p.edge_target_kind[p.srv_out_edge[s]];
        if (k == AF_NODE_SERVER || k == AF_NODE_LB) return true;
    }
    return false;
}
// the level of the servers behind the LB (Flow::lb_pos): 0 when the client feeds the LB, else 1 + the level of the server that does
inline uint32_t flow_lb_position(const af_plan_t& p) {
    if (!p.has_lb) return 0u;
    std::vector<uint32_t> lv(p.n_servers, 0u);
    if (flow_server_levels(p, lv.data()) == 0u) return 0u;
    uint32_t pos = 0u;
    for (uint32_t s = 0; s < p.n_servers; ++s)
        if (p.edge_target_kind[p.srv_out_edge[s]] == AF_NODE_LB) pos = lv[s] + 1u;
    return pos;
}

// Empty string: the plan's request path is the feed-forward chain the flow kernel implements.
// Otherwise the reason it is not (the sequential next-event kernels run such plans).
inline std::string flow_ineligible_reason(const af_plan_t& p) {
    if (p.n_servers == 0) return "no server";
    if (p.n_servers > kSrvSlots) return "more than 16 servers";
    if (p.n_edges > 64u) return "more than 64 edges";   // (an edge's send counter lives in the register of the edge's lane)
    if (p.n_edges + 3u * p.n_servers > 128u) return "more than 128 sampled series";   // (a lane carries two series: flush_ticks)
    if (p.edge_target_kind[p.gen_out_edge] != AF_NODE_CLIENT) return "generator does not feed the client";
    const uint32_t ck = p.edge_target_kind[p.client_out_edge];
    uint32_t feeds_lb = 0u;   // servers whose out-edge leads to the load balancer
    for (uint32_t s = 0; s < p.n_servers; ++s) feeds_lb += p.edge_target_kind[p.srv_out_edge[s]] == AF_NODE_LB ? 1u : 0u;
    if (p.has_lb) {
        // (round 5) client -> LB, or client -> server chain -> LB: exactly one edge leads into the load balancer
        if (ck == AF_NODE_LB && feeds_lb != 0u) return "the client and a server both feed the load balancer";
        if (ck != AF_NODE_LB && (ck != AF_NODE_SERVER || feeds_lb != 1u)) return "neither the client nor exactly one server feeds the load balancer";
        if (p.n_lb_edges == 0 || p.n_lb_edges > 16u) return "load balancer fan-out outside 1..16";
        if (p.lb_algo == AF_LB_LEAST_CONNECTIONS && (p.n_lb_edges > kMaxServers || p.n_servers > kMaxServers)) return "least-connections fan-out above 16";
        for (uint32_t i = 0; i < p.n_lb_edges; ++i)
            if (p.edge_target_kind[p.lb_edges[i]] != AF_NODE_SERVER) return "load balancer edge does not lead to a server";
    } else if (ck != AF_NODE_SERVER) {
        return "client does not feed a server";
    }
    // (Poisson -- integer-second -- latencies are inside since round 3: a delivery is send + k with a CONTINUOUS send time, so two
    // deliveries of one station tie only when their send times differ by an integer, as rarely as with any other law; a
    // zero latency is an event of the next station at the send instant, like a truncated normal's.  What they need is
    // room: rate x 1 s messages wait at a station -- plan_flow sizes the lists for it, an overflow is handed back.
    // 150 fuzzed payloads with 1-3 Poisson edges on the wave emulator: 0 mismatches, 1 tie: tests/test_flow_hostcheck.py.)
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        if (p.edge_target_kind[p.srv_out_edge[s]] == AF_NODE_LB && !p.has_lb) return "a server feeds a load balancer the plan does not have";
        if (p.srv_cores[s] > 64u) return "more than 64 cores";
        // (the tick ring holds integer differences: RAM needs in whole MB, or in multiples of 1/256 MB -- flow_ram_scale)
        for (uint32_t ep = p.srv_ep_begin[s]; ep < p.srv_ep_begin[s + 1]; ++ep) {
            const double ram = p.ep_ram[ep], fine = ram * 256.0;
            if (fine != std::floor(fine) || ram < 0.0 || ram > 4194304.0) return "RAM need is not a multiple of 1/256 MB";
        }
    }
    if (flow_needs_chain(p)) {   // (round 4: the server station once per level -- af_flow.hpp, FEAT_CHAIN)
        const uint32_t levels = flow_server_levels(p, nullptr);
        if (levels == 0u) return "servers feed each other in a cycle";
        if (levels > kMaxLevels) return "server chain deeper than five levels";
    }
    if (flow_needs_general_servers(p) && p.n_endpoints + p.n_steps > 65535u) return "more step rows than a request record addresses";
    return std::string();
}

struct TickTable {
    std::vector<double> t;   // t[k] = time of tick k+1
    double inv_period = 0.0, eps = 0.0;
};
// tick times exactly as the collector produces them (0 + p + p + ..., collector.py:50-53) and the
// band around k * period inside which a time has to be compared with the table itself
inline TickTable make_tick_table(double period, double total_time) {
    TickTable tt;
    tt.inv_period = 1.0 / period;
    double t = 0.0 + period, dev = 0.0;
    while (t < total_time) {
        tt.t.push_back(t);
        const double q = t * tt.inv_period;
        const double d = std::fabs(q - (double)tt.t.size());
        if (d > dev) dev = d;
        t = t + period;
    }
    {   // one more tick position (the first one NOT taken) also bounds the fast path
        const double d = std::fabs(t * tt.inv_period - (double)(tt.t.size() + 1));
        if (d > dev) dev = d;
    }
    tt.eps = 4.0 * dev + 1e-9;
    if (tt.eps > 0.25) tt.eps = 0.5;   // hopeless drift: every lookup goes through the table
    return tt;
}

// 1 when every endpoint needs whole MB of RAM, else 256 (dyadic fractions; flow_ineligible_reason refuses anything finer)
inline double flow_ram_scale(const af_plan_t& p) {
    for (uint32_t ep = 0; ep < p.n_endpoints; ++ep)
        if (p.ep_ram[ep] != std::floor(p.ep_ram[ep])) return 256.0;
    return 1.0;
}

// longest leading-I/O / CPU / trailing-I/O run over the servers' endpoints (eligible plans: IO* CPU* IO*)
inline void flow_step_maxima(const af_plan_t& p, uint32_t& max_pre, uint32_t& max_cpu, uint32_t& max_post) {
    max_pre = max_cpu = max_post = 0;
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        const uint32_t ep = p.srv_ep_begin[s];
        uint32_t cnt[3] = {0, 0, 0}, phase = 0;
        for (uint32_t i = p.ep_step_begin[ep]; i < p.ep_step_begin[ep + 1]; ++i) {
            const bool cpu = p.step_kind[i] == AF_STEP_CPU;
            if (phase == 0 && cpu) phase = 1;
            else if (phase == 1 && !cpu) phase = 2;
            cnt[phase] += 1;
        }
        if (cnt[0] > max_pre) max_pre = cnt[0];
        if (cnt[1] > max_cpu) max_cpu = cnt[1];
        if (cnt[2] > max_post) max_post = cnt[2];
    }
}

inline uint32_t pow2_ge(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// LB out-edges of a least-connections load balancer (0: round robin / no LB)
inline uint32_t lc_edges(const af_plan_t& p) { return (p.has_lb && p.lb_algo == AF_LB_LEAST_CONNECTIONS) ? p.n_lb_edges : 0u; }

// list capacity (64 * ipl), tick-ring rows: from the expected number of messages in flight per
// station and the time a round spans (host estimates; overflow is detected by the kernel)
inline FlowLayout choose_flow_layout(const af_plan_t& p, uint32_t ipl, uint32_t ring_rows) {
    uint32_t cmax = 1, gmax = 1;
    for (uint32_t s = 0; s < p.n_servers; ++s) {
        if (p.srv_cores[s] > cmax) cmax = p.srv_cores[s];
        const double ram = p.ep_ram[p.srv_ep_begin[s]];
        if (ram > 0.0) {
            const double slots = std::floor(p.srv_ram_mb[s] / ram);
            // (the ring remembers this many departures per server: 8 KB of LDS at most; servers with more slots than
            // that run here as long as fewer requests than the ring holds are inside at once)
            const double lim = p.n_servers <= 4u ? 256.0 : 128.0;
            const uint32_t want = slots > lim ? (uint32_t)lim : (uint32_t)slots;
            if (want > gmax) gmax = want;
        }
    }
    return make_flow_layout(64u * ipl, ring_rows ? pow2_ge(ring_rows) : 0u, pow2_ge(gmax), cmax, p.n_edges, p.n_servers, p.n_edge_marks);
}

}  // namespace aff
