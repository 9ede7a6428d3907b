// af_core.hpp -- per-scenario next-event state machine of the MI355X engine.
//
// One scenario per lane.  Replaces, for that scenario, everything that
// `SimulationRunner.run()` -> `env.run(until=T)` executes in the reference
// (/root/reference/src/asyncflow/runtime/simulation_runner.py:349-376): the
// SimPy heap, the generator / edge / client / load-balancer / server coroutines
// (runtime/actors/*.py), the event-injection timelines (runtime/events/
// injection.py) and the metric collector (metrics/collector.py).
//
// Execution model ("atomic cascades"):
//   * only TIMED events live in the per-scenario priority queue -- a binary heap
//     of (time f64, request slot), ties broken by a per-scenario push sequence;
//     the generator, sampler and injection timers are register-resident
//     "special" sources with fixed class order on ties;
//   * every zero-time SimPy step following a timed event is executed inline in
//     the order SimPy runs it when no other timed event shares the timestamp.
//
// State lives in `Mem`, a word-addressed per-lane memory laid out [index][lane]
// (SoA across the 64 lanes of a wave): LDS when it fits (bank-conflict-free for
// arbitrary per-lane indices), HBM otherwise (coalesced for equal indices).
//
// This header is device code under hipcc and plain C++ under g++: the latter is
// the TEST-ONLY host instantiation built by tests/hostcheck/ (never shipped,
// never reachable from the asyncflow_amd package).
#pragma once

#include <stdint.h>

#include "af_math.hpp"

#if defined(__HIPCC__)
#define AF_CORE __device__ __forceinline__
#define AF_CORE_NOINLINE __device__ __noinline__
#define AF_PLAN_AS __attribute__((address_space(3)))
#else
#define AF_CORE inline
#define AF_CORE_NOINLINE inline
#define AF_PLAN_AS
#endif

namespace af {

// ---- must match include/asyncflow_hip.h ---------------------------------
enum : uint32_t { NODE_CLIENT = 0, NODE_LB = 1, NODE_SERVER = 2 };
enum : uint32_t { LB_ROUND_ROBIN = 0, LB_LEAST_CONNECTIONS = 1 };
enum : uint32_t { STEP_CPU = 0, STEP_IO = 1 };
enum : uint32_t { METRIC_READY = 1, METRIC_IO = 2, METRIC_RAM = 4, METRIC_EDGE = 8 };
enum : uint32_t {
    FLAG_POOL_OVERFLOW = 1u << 0,
    FLAG_FIFO_OVERFLOW = 1u << 1,
    FLAG_CLOCK_OVERFLOW = 1u << 2,
    FLAG_TICK_OVERFLOW = 1u << 3,
    FLAG_RAM_STARVED = 1u << 4,
    FLAG_TIME_TIE = 1u << 5,
};
enum : uint32_t {
    CNT_GENERATED = 0, CNT_COMPLETED, CNT_DROPPED, CNT_EVENTS, CNT_TICKS, CNT_FLAGS, CNT_MAX_LIVE, CNT_MARKS, CNT_SLOTS
};
enum : uint32_t {
    PARAM_GEN_USERS_MEAN = 0, PARAM_GEN_USERS_SIGMA, PARAM_GEN_RPM_MEAN, PARAM_EDGE_MEAN, PARAM_EDGE_SIGMA,
    PARAM_EDGE_DROPOUT, PARAM_STEP_TIME, PARAM_COUNT
};

// ---- request state word ----------------------------------------------------
// RST : kind[0:2] | idx8[3:10] (edge when in transit, server otherwise) | hops[11:18] | in_io[19]
// RST2: endpoint[0:15] | absolute step index[16:31]
enum : uint32_t { RK_TRANSIT = 0, RK_CPU = 1, RK_IO = 2, RK_WAIT_RAM = 3, RK_WAIT_CPU = 4 };
AF_HD uint32_t rst_pack(uint32_t kind, uint32_t idx, uint32_t hops, uint32_t in_io) {
    return kind | (idx << 3) | ((hops > 255u ? 255u : hops) << 11) | (in_io << 19);
}

// ---- the lowered plan as seen by device code (LDS-resident copy) -----------
struct PlanView {
    double total_time, sample_period;
    double gen_users_mean, gen_users_sigma, gen_rpm_mean, gen_window_s;
    uint32_t metrics_mask, gen_users_dist;
    int32_t gen_out_edge, client_out_edge;
    uint32_t n_edges, n_servers, lb_algo, n_lb_edges, n_endpoints, n_steps, n_edge_marks, n_srv_marks;
    const AF_PLAN_AS double* e_mean;
    const AF_PLAN_AS double* e_sigma;
    const AF_PLAN_AS double* e_drop;
    const AF_PLAN_AS double* s_ram;
    const AF_PLAN_AS double* ep_ram;
    const AF_PLAN_AS double* st_time;
    const AF_PLAN_AS double* em_time;
    const AF_PLAN_AS double* em_delta;
    const AF_PLAN_AS double* sm_time;
    const AF_PLAN_AS int32_t* lb_edges;
    const AF_PLAN_AS uint32_t* e_tkind;
    const AF_PLAN_AS int32_t* e_tidx;
    const AF_PLAN_AS uint32_t* e_dist;
    const AF_PLAN_AS uint32_t* s_cores;
    const AF_PLAN_AS int32_t* s_out;
    const AF_PLAN_AS uint32_t* s_epb;
    const AF_PLAN_AS uint32_t* ep_stepb;
    const AF_PLAN_AS uint32_t* st_kind;
    const AF_PLAN_AS int32_t* em_edge;
    const AF_PLAN_AS int32_t* sm_edge;
    const AF_PLAN_AS uint32_t* sm_down;
};

// ---- per-lane state layout (word offsets; computed by the host) -------------
struct Layout {
    uint32_t cap;       // live requests == heap capacity
    uint32_t fcap;      // per-server wait-queue capacity (power of two)
    uint32_t ovr_mask;  // bit p set: af_param class p has a per-lane column
    // f64 region, offsets in doubles
    uint32_t d_hk, d_t0, d_spike, d_ramfree, d_ramuse, d_emean, d_esig, d_edrop, d_stime, n_d;
    // u32 region, offsets in words
    uint32_t w_hs, w_rst, w_rst2, w_rseq, w_free, w_conn, w_sends, w_cpufree, w_ready, w_io, w_arr, w_rblk;
    uint32_t w_cqh, w_cqn, w_rqh, w_rqn, w_cq, w_rq, w_lb, n_w;
};

AF_HD Layout make_layout(uint32_t cap, uint32_t fcap, uint32_t n_edges, uint32_t n_servers, uint32_t n_lb,
                         uint32_t n_steps, uint32_t ovr_mask) {
    Layout L{};
    L.cap = cap;
    L.fcap = fcap;
    L.ovr_mask = ovr_mask;
    uint32_t d = 0;
    L.d_hk = d; d += cap;
    L.d_t0 = d; d += cap;
    L.d_spike = d; d += n_edges;
    L.d_ramfree = d; d += n_servers;
    L.d_ramuse = d; d += n_servers;
    L.d_emean = d; if (ovr_mask & (1u << PARAM_EDGE_MEAN)) d += n_edges;
    L.d_esig = d; if (ovr_mask & (1u << PARAM_EDGE_SIGMA)) d += n_edges;
    L.d_edrop = d; if (ovr_mask & (1u << PARAM_EDGE_DROPOUT)) d += n_edges;
    L.d_stime = d; if (ovr_mask & (1u << PARAM_STEP_TIME)) d += n_steps;
    L.n_d = d;
    uint32_t w = 0;
    L.w_hs = w; w += cap;
    L.w_rst = w; w += cap;
    L.w_rst2 = w; w += cap;
    L.w_rseq = w; w += cap;
    L.w_free = w; w += cap;
    L.w_conn = w; w += n_edges;
    L.w_sends = w; w += n_edges;
    L.w_cpufree = w; w += n_servers;
    L.w_ready = w; w += n_servers;
    L.w_io = w; w += n_servers;
    L.w_arr = w; w += n_servers;
    L.w_rblk = w; w += n_servers;
    L.w_cqh = w; w += n_servers;
    L.w_cqn = w; w += n_servers;
    L.w_rqh = w; w += n_servers;
    L.w_rqn = w; w += n_servers;
    L.w_cq = w; w += n_servers * fcap;
    L.w_rq = w; w += n_servers * fcap;
    L.w_lb = w; w += n_lb;
    L.n_w = w;
    return L;
}
AF_HD uint64_t layout_bytes_per_lane(const Layout& L) { return 8ull * L.n_d + 4ull * L.n_w; }

// ---- outputs of one scenario -------------------------------------------------
struct LaneOut {
    double* clock;      // [clock_cap][2] or nullptr
    uint32_t* samples;  // [n_series][tick_cap] or nullptr
    uint32_t* counts;   // [CNT_SLOTS]
    uint32_t clock_cap, tick_cap;
};

constexpr double AF_INF = __builtin_huge_val();

template <class Mem>
struct Lane {
    const PlanView& P;
    const Layout& L;
    Mem M;
    LaneOut O;
    uint64_t seed;

    // register-resident scalars
    double now, t_gen, g_now, g_wend, g_lam, t_tick;
    double users_mean, users_sigma, rpm;
    uint32_t g_draws, heap_n, seq, bump, free_top, live, max_live, lb_n, emark_i, smark_i;
    uint32_t n_gen, n_comp, n_drop, n_events, n_ticks, n_marks, flags, rounds;
    int32_t pending_grant_sv;
    bool hole;

    AF_CORE Lane(const PlanView& p, const Layout& l, Mem m, LaneOut o, uint64_t s) : P(p), L(l), M(m), O(o), seed(s) {}

    // ---- parameter accessors (plan value or per-scenario column) -------------
    AF_CORE double edge_mean(uint32_t e) const {
        return (L.ovr_mask & (1u << PARAM_EDGE_MEAN)) ? M.ld64(L.d_emean + e) : P.e_mean[e];
    }
    AF_CORE double edge_sigma(uint32_t e) const {
        return (L.ovr_mask & (1u << PARAM_EDGE_SIGMA)) ? M.ld64(L.d_esig + e) : P.e_sigma[e];
    }
    AF_CORE double edge_dropout(uint32_t e) const {
        return (L.ovr_mask & (1u << PARAM_EDGE_DROPOUT)) ? M.ld64(L.d_edrop + e) : P.e_drop[e];
    }
    AF_CORE double step_time(uint32_t i) const {
        return (L.ovr_mask & (1u << PARAM_STEP_TIME)) ? M.ld64(L.d_stime + i) : P.st_time[i];
    }

    // ---- priority queue: binary heap of (time, slot), ties by push sequence ----
    AF_CORE bool ev_less(double ka, uint32_t sa, double kb, uint32_t sb) const {
        if (ka != kb) return ka < kb;
        return M.ld32(L.w_rseq + sa) < M.ld32(L.w_rseq + sb);
    }
    AF_CORE void sift_down(uint32_t pos, double key, uint32_t slot, uint32_t n) {
        for (;;) {
            uint32_t c = 2u * pos + 1u;
            if (c >= n) break;
            double kc = M.ld64(L.d_hk + c);
            uint32_t sc = M.ld32(L.w_hs + c);
            if (c + 1u < n) {
                const double k2 = M.ld64(L.d_hk + c + 1u);
                const uint32_t s2 = M.ld32(L.w_hs + c + 1u);
                if (ev_less(k2, s2, kc, sc)) {
                    c += 1u;
                    kc = k2;
                    sc = s2;
                }
            }
            if (!ev_less(kc, sc, key, slot)) break;
            M.st64(L.d_hk + pos, kc);
            M.st32(L.w_hs + pos, sc);
            pos = c;
        }
        M.st64(L.d_hk + pos, key);
        M.st32(L.w_hs + pos, slot);
    }
    AF_CORE void sift_up(uint32_t pos, double key, uint32_t slot) {
        while (pos > 0u) {
            const uint32_t p = (pos - 1u) >> 1;
            const double kp = M.ld64(L.d_hk + p);
            const uint32_t sp = M.ld32(L.w_hs + p);
            if (!ev_less(key, slot, kp, sp)) break;
            M.st64(L.d_hk + pos, kp);
            M.st32(L.w_hs + pos, sp);
            pos = p;
        }
        M.st64(L.d_hk + pos, key);
        M.st32(L.w_hs + pos, slot);
    }
    // schedule the (single) pending timed event of request `slot`
    AF_CORE void push(double t, uint32_t slot) {
        M.st32(L.w_rseq + slot, seq++);
        if (hole) {  // the event popped this round left the root free: replace-top
            hole = false;
            sift_down(0u, t, slot, heap_n);
        } else {
            sift_up(heap_n++, t, slot);
        }
    }
    AF_CORE void remove_root() {
        hole = false;
        heap_n -= 1u;
        if (heap_n > 0u) {
            const double k = M.ld64(L.d_hk + heap_n);
            const uint32_t s = M.ld32(L.w_hs + heap_n);
            sift_down(0u, k, s, heap_n);
        }
    }

    // ---- request pool ----------------------------------------------------------
    AF_CORE uint32_t alloc_slot() {
        uint32_t s;
        if (free_top > 0u) {
            s = M.ld32(L.w_free + --free_top);
        } else if (bump < L.cap) {
            s = bump++;
        } else {
            flags |= FLAG_POOL_OVERFLOW;
            return 0xFFFFFFFFu;
        }
        live += 1u;
        if (live > max_live) max_live = live;
        return s;
    }
    AF_CORE void free_slot(uint32_t s) {
        M.st32(L.w_free + free_top++, s);
        live -= 1u;
    }

    // ---- server wait queues (rings) ---------------------------------------------
    AF_CORE bool q_push(uint32_t w_q, uint32_t w_h, uint32_t w_n, uint32_t sv, uint32_t slot) {
        const uint32_t n = M.ld32(w_n + sv);
        if (n >= L.fcap) {
            flags |= FLAG_FIFO_OVERFLOW;
            return false;
        }
        const uint32_t h = M.ld32(w_h + sv);
        M.st32(w_q + sv * L.fcap + ((h + n) & (L.fcap - 1u)), slot);
        M.st32(w_n + sv, n + 1u);
        return true;
    }
    AF_CORE uint32_t q_front(uint32_t w_q, uint32_t w_h, uint32_t sv) const {
        return M.ld32(w_q + sv * L.fcap + M.ld32(w_h + sv));
    }
    AF_CORE uint32_t q_pop(uint32_t w_q, uint32_t w_h, uint32_t w_n, uint32_t sv) {
        const uint32_t h = M.ld32(w_h + sv);
        const uint32_t slot = M.ld32(w_q + sv * L.fcap + h);
        M.st32(w_h + sv, (h + 1u) & (L.fcap - 1u));
        M.st32(w_n + sv, M.ld32(w_n + sv) - 1u);
        return slot;
    }

    // ---- generator: samplers/poisson_poisson.py:51-82, gaussian_poisson.py:63-94 ----
    AF_CORE double next_gap() {
        const double T = P.total_time;
        const double rps_per_user = rpm / 60.0;
        while (g_now < T) {
            if (g_now >= g_wend) {
                g_wend = g_now + P.gen_window_s;
                const uint32_t idx = g_draws++;
                double users;
                if (P.gen_users_dist == DIST_NORMAL) {
                    const double v = users_mean + users_sigma * af_norminv(uniform_j(seed, STREAM_GENERATOR, idx, 0u));
                    users = v > 0.0 ? v : 0.0;
                } else {
                    users = (double)af_poisson(users_mean, seed, STREAM_GENERATOR, idx, 0u);
                }
                g_lam = users * rps_per_user;
            }
            if (g_lam <= 0.0) {
                g_now = g_wend;
                continue;
            }
            double u = uniform_j(seed, STREAM_GENERATOR, g_draws++, 0u);
            if (u < 1e-15) u = 1e-15;
            const double dt = -af_log(1.0 - u) / g_lam;
            if (g_now + dt > T) break;
            if (g_now + dt >= g_wend) {
                g_now = g_wend;
                continue;
            }
            g_now += dt;
            return dt;
        }
        g_now = T + 1.0;
        return -1.0;
    }

    // ---- edge: EdgeRuntime.transport/_deliver up to the timeout (edge.py:73-107) ----
    AF_CORE void edge_send(uint32_t slot, uint32_t e, uint32_t hops) {
        const uint32_t idx = M.ld32(L.w_sends + e);
        M.st32(L.w_sends + e, idx + 1u);
        const uint32_t stream = stream_edge(e);
        const U4 r = draw_block(seed, stream, idx, 0u);
        if (u53(r.x, r.y) < edge_dropout(e)) {  // dropped: no latency draw (edge.py:78-86)
            n_drop += 1u;
            free_slot(slot);
            return;
        }
        M.st32(L.w_conn + e, M.ld32(L.w_conn + e) + 1u);
        const double transit = variate_from_u1(P.e_dist[e], edge_mean(e), edge_sigma(e), u53(r.z, r.w), seed, stream, idx);
        const double effective = transit + M.ld64(L.d_spike + e);  // spike read at SEND time (edge.py:94-106)
        M.st32(L.w_rst + slot, rst_pack(RK_TRANSIT, e, hops, 0u));
        push(now + effective, slot);
    }

    // ---- server: ServerRuntime._handle_request (server.py:79-276) ----------------
    // a CPU token became free: first waiter (Container FIFO) or 0xFFFFFFFF
    AF_CORE uint32_t cpu_release(uint32_t sv) {
        if (M.ld32(L.w_cqn + sv) > 0u) return q_pop(L.w_cq, L.w_cqh, L.w_cqn, sv);  // token handed over
        M.st32(L.w_cpufree + sv, M.ld32(L.w_cpufree + sv) + 1u);
        return 0xFFFFFFFFu;
    }
    // the waiter's `yield cpu_req` returns (server.py:220-231)
    AF_CORE void cpu_granted(uint32_t w, uint32_t sv) {
        M.st32(L.w_ready + sv, M.ld32(L.w_ready + sv) - 1u);
        const uint32_t st = M.ld32(L.w_rst + w);
        const uint32_t step = M.ld32(L.w_rst2 + w) >> 16;
        M.st32(L.w_rst + w, (st & ~7u) | RK_CPU);
        push(now + step_time(step), w);
    }

    // The for-loop of _handle_request from `step` until the next timed event,
    // a wait, or the end of the endpoint.  Non-recursive: RAM grants caused by a
    // finishing request are left to the caller via pending_grant_sv.
    AF_CORE void advance(uint32_t slot, uint32_t sv, uint32_t ep, uint32_t step, uint32_t hops, bool core_locked,
                         bool in_io) {
        const uint32_t end = P.ep_stepb[ep + 1u];
        const uint32_t rst2 = ep | (step << 16);
        if (step < end) {
            if (P.st_kind[step] == STEP_CPU) {  // server.py:199-231
                if (in_io) {
                    in_io = false;
                    M.st32(L.w_io + sv, M.ld32(L.w_io + sv) - 1u);
                }
                if (!core_locked) {
                    const uint32_t cf = M.ld32(L.w_cpufree + sv);
                    if (M.ld32(L.w_cqn + sv) == 0u && cf > 0u) {
                        M.st32(L.w_cpufree + sv, cf - 1u);  // granted at once: not in the ready queue
                    } else {
                        M.st32(L.w_rst + slot, rst_pack(RK_WAIT_CPU, sv, hops, 0u));
                        M.st32(L.w_rst2 + slot, rst2);
                        if (q_push(L.w_cq, L.w_cqh, L.w_cqn, sv, slot)) {
                            M.st32(L.w_ready + sv, M.ld32(L.w_ready + sv) + 1u);
                        } else {
                            free_slot(slot);
                        }
                        return;
                    }
                }
                M.st32(L.w_rst + slot, rst_pack(RK_CPU, sv, hops, 0u));
                M.st32(L.w_rst2 + slot, rst2);
                push(now + step_time(step), slot);
                return;
            }
            // I/O step, server.py:235-255
            uint32_t granted = 0xFFFFFFFFu;
            if (core_locked) {
                granted = cpu_release(sv);
                if (!in_io) {
                    in_io = true;
                    M.st32(L.w_io + sv, M.ld32(L.w_io + sv) + 1u);
                }
            } else if (!in_io) {
                in_io = true;
                M.st32(L.w_io + sv, M.ld32(L.w_io + sv) + 1u);
            }
            M.st32(L.w_rst + slot, rst_pack(RK_IO, sv, hops, 1u));
            M.st32(L.w_rst2 + slot, rst2);
            push(now + step_time(step), slot);                     // own Timeout first ...
            if (granted != 0xFFFFFFFFu) cpu_granted(granted, sv);  // ... then the waiter's
            return;
        }
        // endpoint finished, server.py:257-276
        if (core_locked) {
            const uint32_t granted = cpu_release(sv);
            if (granted != 0xFFFFFFFFu) cpu_granted(granted, sv);  // waiter's get is processed first
        }
        if (in_io) M.st32(L.w_io + sv, M.ld32(L.w_io + sv) - 1u);
        const double ram = P.ep_ram[ep];
        if (ram > 0.0) {
            M.st64(L.d_ramuse + sv, M.ld64(L.d_ramuse + sv) - ram);
            M.st64(L.d_ramfree + sv, M.ld64(L.d_ramfree + sv) + ram);
            pending_grant_sv = (int32_t)sv;
        }
        edge_send(slot, (uint32_t)P.s_out[sv], hops);
    }

    // Container._trigger_get on the RAM container: FIFO, head-of-line blocking
    AF_CORE void drain_ram_grants() {
        while (pending_grant_sv >= 0) {
            const uint32_t sv = (uint32_t)pending_grant_sv;
            pending_grant_sv = -1;
            while (M.ld32(L.w_rqn + sv) > 0u) {
                const uint32_t w = q_front(L.w_rq, L.w_rqh, sv);
                const uint32_t rst2 = M.ld32(L.w_rst2 + w);
                const uint32_t ep = rst2 & 0xFFFFu;
                const double need = P.ep_ram[ep];
                const double free_ram = M.ld64(L.d_ramfree + sv);
                if (free_ram < need) break;
                q_pop(L.w_rq, L.w_rqh, L.w_rqn, sv);
                M.st64(L.d_ramfree + sv, free_ram - need);
                M.st64(L.d_ramuse + sv, M.ld64(L.d_ramuse + sv) + need);
                const uint32_t hops = (M.ld32(L.w_rst + w) >> 11) & 0xFFu;
                advance(w, sv, ep, rst2 >> 16, hops, false, false);
            }
        }
    }

    // _dispatcher + head of _handle_request (server.py:303-313, 79-149)
    AF_CORE void server_arrival(uint32_t slot, uint32_t sv, uint32_t hops) {
        hops += 1u;  // record_hop(SERVER)
        const uint32_t epb = P.s_epb[sv];
        const uint32_t n_ep = P.s_epb[sv + 1u] - epb;
        const uint32_t idx = M.ld32(L.w_arr + sv);
        M.st32(L.w_arr + sv, idx + 1u);
        uint32_t pick = 0u;
        if (n_ep > 1u) {  // rng.integers(0, n_ep), server.py:101
            const U4 r = draw_block(seed, stream_server(sv), idx, 0u);
            pick = (uint32_t)(((uint64_t)r.x * n_ep) >> 32);
        }
        const uint32_t ep = epb + pick;
        const uint32_t step0 = P.ep_stepb[ep];
        const double ram = P.ep_ram[ep];
        if (ram > 0.0) {  // server.py:146-149
            if (ram > P.s_ram[sv] || M.ld32(L.w_rblk + sv)) {
                // can never be served: it (and everything queued behind it) waits
                // forever in the reference, observable nowhere -> dropped here.
                flags |= FLAG_RAM_STARVED;
                M.st32(L.w_rblk + sv, 1u);
                free_slot(slot);
                return;
            }
            const double free_ram = M.ld64(L.d_ramfree + sv);
            if (M.ld32(L.w_rqn + sv) == 0u && free_ram >= ram) {
                M.st64(L.d_ramfree + sv, free_ram - ram);
                M.st64(L.d_ramuse + sv, M.ld64(L.d_ramuse + sv) + ram);
            } else {
                M.st32(L.w_rst + slot, rst_pack(RK_WAIT_RAM, sv, hops, 0u));
                M.st32(L.w_rst2 + slot, ep | (step0 << 16));
                if (!q_push(L.w_rq, L.w_rqh, L.w_rqn, sv, slot)) free_slot(slot);
                return;
            }
        }
        advance(slot, sv, ep, step0, hops, false, false);
    }

    // EdgeRuntime._deliver after the timeout (edge.py:110-116) + the target node
    AF_CORE void deliver(uint32_t slot, uint32_t e, uint32_t hops) {
        hops += 1u;  // record_hop(NETWORK_CONNECTION)
        M.st32(L.w_conn + e, M.ld32(L.w_conn + e) - 1u);
        const uint32_t tk = P.e_tkind[e];
        if (tk == NODE_CLIENT) {  // ClientRuntime._forwarder, client.py:46-71
            hops += 1u;
            if (hops > 3u) {
                if (O.clock != nullptr) {
                    if (n_comp < O.clock_cap) {
                        O.clock[2u * n_comp] = M.ld64(L.d_t0 + slot);
                        O.clock[2u * n_comp + 1u] = now;
                    } else {
                        flags |= FLAG_CLOCK_OVERFLOW;
                    }
                }
                n_comp += 1u;
                free_slot(slot);
            } else {
                edge_send(slot, (uint32_t)P.client_out_edge, hops);
            }
        } else if (tk == NODE_LB) {  // LoadBalancerRuntime._forwarder, load_balancer.py:60-72
            hops += 1u;
            uint32_t out;
            if (P.lb_algo == LB_LEAST_CONNECTIONS) {  // lb_algorithms.py:10-20
                out = M.ld32(L.w_lb);
                uint32_t best = M.ld32(L.w_conn + out);
                for (uint32_t i = 1u; i < lb_n; ++i) {
                    const uint32_t cand = M.ld32(L.w_lb + i);
                    const uint32_t c = M.ld32(L.w_conn + cand);
                    if ((int32_t)c < (int32_t)best) {
                        best = c;
                        out = cand;
                    }
                }
            } else {  // round_robin: first key, move_to_end (lb_algorithms.py:22-36)
                out = M.ld32(L.w_lb);
                for (uint32_t i = 1u; i < lb_n; ++i) M.st32(L.w_lb + i - 1u, M.ld32(L.w_lb + i));
                M.st32(L.w_lb + lb_n - 1u, out);
            }
            edge_send(slot, out, hops);
        } else {
            server_arrival(slot, (uint32_t)P.e_tidx[e], hops);
        }
    }

    // ---- event injection (runtime/events/injection.py:167-226) -------------------
    AF_CORE void apply_emarks() {
        for (;;) {
            const uint32_t i = emark_i++;
            const uint32_t e = (uint32_t)P.em_edge[i];
            M.st64(L.d_spike + e, M.ld64(L.d_spike + e) + P.em_delta[i]);
            n_marks += 1u;
            if (emark_i >= P.n_edge_marks || P.em_time[emark_i] > now) break;
        }
    }
    AF_CORE void apply_smarks() {
        for (;;) {
            const uint32_t i = smark_i++;
            const int32_t e = P.sm_edge[i];
            n_marks += 1u;
            if (e >= 0) {
                uint32_t pos = 0xFFFFFFFFu;
                for (uint32_t k = 0u; k < lb_n; ++k)
                    if (M.ld32(L.w_lb + k) == (uint32_t)e) pos = k;
                if (P.sm_down[i]) {  // lb_out_edges.pop(edge_id, None)
                    if (pos != 0xFFFFFFFFu) {
                        for (uint32_t k = pos + 1u; k < lb_n; ++k) M.st32(L.w_lb + k - 1u, M.ld32(L.w_lb + k));
                        lb_n -= 1u;
                    }
                } else {  // re-insert + move_to_end
                    if (pos != 0xFFFFFFFFu) {
                        for (uint32_t k = pos + 1u; k < lb_n; ++k) M.st32(L.w_lb + k - 1u, M.ld32(L.w_lb + k));
                        M.st32(L.w_lb + lb_n - 1u, (uint32_t)e);
                    } else {
                        M.st32(L.w_lb + lb_n, (uint32_t)e);
                        lb_n += 1u;
                    }
                }
            }
            if (smark_i >= P.n_srv_marks || P.sm_time[smark_i] > now) break;
        }
    }

    // ---- sampler tick (metrics/collector.py:50-66) --------------------------------
    AF_CORE void sample_tick() {
        if (O.samples != nullptr) {
            if (n_ticks < O.tick_cap) {
                const uint32_t k = n_ticks;
                if (P.metrics_mask & METRIC_EDGE)
                    for (uint32_t e = 0u; e < P.n_edges; ++e) O.samples[e * O.tick_cap + k] = M.ld32(L.w_conn + e);
                constexpr uint32_t all = METRIC_READY | METRIC_IO | METRIC_RAM;
                if ((P.metrics_mask & all) == all)
                    for (uint32_t v = 0u; v < P.n_servers; ++v) {
                        const uint32_t base = (P.n_edges + 3u * v) * O.tick_cap + k;
                        O.samples[base] = M.ld32(L.w_ready + v);
                        O.samples[base + O.tick_cap] = M.ld32(L.w_io + v);
                        O.samples[base + 2u * O.tick_cap] = __builtin_bit_cast(uint32_t, (float)M.ld64(L.d_ramuse + v));
                    }
            } else {
                flags |= FLAG_TICK_OVERFLOW;
            }
        }
        n_ticks += 1u;
    }

    // ---- life cycle -----------------------------------------------------------------
    // `ovr` : per-lane reader of the override columns, ovr(k) -> double, k-th column
    template <class OvrFn>
    AF_CORE void init(const uint32_t* ovr_param, const uint32_t* ovr_index, uint32_t n_ovr, OvrFn ovr) {
        now = 0.0;
        g_now = 0.0;
        g_wend = 0.0;
        g_lam = 0.0;
        g_draws = heap_n = seq = bump = free_top = live = max_live = emark_i = smark_i = 0u;
        n_gen = n_comp = n_drop = n_events = n_ticks = n_marks = flags = rounds = 0u;
        pending_grant_sv = -1;
        hole = false;
        users_mean = P.gen_users_mean;
        users_sigma = P.gen_users_sigma;
        rpm = P.gen_rpm_mean;
        for (uint32_t e = 0u; e < P.n_edges; ++e) {
            M.st64(L.d_spike + e, 0.0);
            M.st32(L.w_conn + e, 0u);
            M.st32(L.w_sends + e, 0u);
            if (L.ovr_mask & (1u << PARAM_EDGE_MEAN)) M.st64(L.d_emean + e, P.e_mean[e]);
            if (L.ovr_mask & (1u << PARAM_EDGE_SIGMA)) M.st64(L.d_esig + e, P.e_sigma[e]);
            if (L.ovr_mask & (1u << PARAM_EDGE_DROPOUT)) M.st64(L.d_edrop + e, P.e_drop[e]);
        }
        if (L.ovr_mask & (1u << PARAM_STEP_TIME))
            for (uint32_t i = 0u; i < P.n_steps; ++i) M.st64(L.d_stime + i, P.st_time[i]);
        for (uint32_t v = 0u; v < P.n_servers; ++v) {  // build_containers: init full
            M.st64(L.d_ramfree + v, P.s_ram[v]);
            M.st64(L.d_ramuse + v, 0.0);
            M.st32(L.w_cpufree + v, P.s_cores[v]);
            M.st32(L.w_ready + v, 0u);
            M.st32(L.w_io + v, 0u);
            M.st32(L.w_arr + v, 0u);
            M.st32(L.w_rblk + v, 0u);
            M.st32(L.w_cqh + v, 0u);
            M.st32(L.w_cqn + v, 0u);
            M.st32(L.w_rqh + v, 0u);
            M.st32(L.w_rqn + v, 0u);
        }
        lb_n = P.n_lb_edges;
        for (uint32_t i = 0u; i < lb_n; ++i) M.st32(L.w_lb + i, (uint32_t)P.lb_edges[i]);
        for (uint32_t k = 0u; k < n_ovr; ++k) {
            const double v = ovr(k);
            const uint32_t idx = ovr_index[k];
            switch (ovr_param[k]) {
                case PARAM_GEN_USERS_MEAN: users_mean = v; break;
                case PARAM_GEN_USERS_SIGMA: users_sigma = v; break;
                case PARAM_GEN_RPM_MEAN: rpm = v; break;
                case PARAM_EDGE_MEAN: M.st64(L.d_emean + idx, v); break;
                case PARAM_EDGE_SIGMA: M.st64(L.d_esig + idx, v); break;
                case PARAM_EDGE_DROPOUT: M.st64(L.d_edrop + idx, v); break;
                case PARAM_STEP_TIME: M.st64(L.d_stime + idx, v); break;
                default: break;
            }
        }
        const double gap = next_gap();
        t_gen = gap >= 0.0 ? 0.0 + gap : AF_INF;
        t_tick = 0.0 + P.sample_period;
    }

    // One next-event round.  Returns false once the scenario reached the horizon.
    AF_CORE bool round() {
        // next event among {heap, arrival, tick, server marks, edge marks}; on equal
        // times the LATER test wins: edge marks < server marks < tick < arrival < heap.
        uint32_t cls = 4u;
        double t = heap_n > 0u ? M.ld64(L.d_hk) : AF_INF;
        if (t_gen <= t) { cls = 3u; t = t_gen; }
        if (t_tick <= t) { cls = 2u; t = t_tick; }
        if (smark_i < P.n_srv_marks) {
            const double ts = P.sm_time[smark_i];
            if (ts <= t) { cls = 1u; t = ts; }
        }
        if (emark_i < P.n_edge_marks) {
            const double te = P.em_time[emark_i];
            if (te <= t) { cls = 0u; t = te; }
        }
        if (!(t < P.total_time)) return false;  // the stop event is URGENT at T
        if (rounds > 0u && t == now) flags |= FLAG_TIME_TIE;
        rounds += 1u;
        now = t;

        if (cls == 4u) {
            const uint32_t slot = M.ld32(L.w_hs);
            hole = true;
            const uint32_t st = M.ld32(L.w_rst + slot);
            const uint32_t kind = st & 7u;
            const uint32_t idx = (st >> 3) & 0xFFu;
            const uint32_t hops = (st >> 11) & 0xFFu;
            n_events += 1u;
            if (kind == RK_TRANSIT) {
                deliver(slot, idx, hops);
            } else {  // CPU or I/O step finished: the for-loop moves to the next step
                const uint32_t rst2 = M.ld32(L.w_rst2 + slot);
                advance(slot, idx, rst2 & 0xFFFFu, (rst2 >> 16) + 1u, hops, kind == RK_CPU, (st >> 19) & 1u);
            }
            drain_ram_grants();
            if (hole) remove_root();
        } else if (cls == 3u) {  // RqsGeneratorRuntime._event_arrival (rqs_generator.py:101-119)
            n_gen += 1u;
            n_events += 1u;
            const double gap = next_gap();
            const double t_arrival = now;
            t_gen = gap >= 0.0 ? now + gap : AF_INF;
            const uint32_t slot = alloc_slot();
            if (slot != 0xFFFFFFFFu) {
                M.st64(L.d_t0 + slot, t_arrival);
                edge_send(slot, (uint32_t)P.gen_out_edge, 1u);  // hops = 1: record_hop(GENERATOR)
            }
        } else if (cls == 2u) {
            sample_tick();
            t_tick = now + P.sample_period;
        } else if (cls == 1u) {
            apply_smarks();
        } else {
            apply_emarks();
        }
        return true;
    }

    AF_CORE void write_counts() const {
        O.counts[CNT_GENERATED] = n_gen;
        O.counts[CNT_COMPLETED] = n_comp;
        O.counts[CNT_DROPPED] = n_drop;
        O.counts[CNT_EVENTS] = n_events;
        O.counts[CNT_TICKS] = n_ticks;
        O.counts[CNT_FLAGS] = flags;
        O.counts[CNT_MAX_LIVE] = max_live;
        O.counts[CNT_MARKS] = n_marks;
    }
};

}  // namespace af
