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
#define AF_CORE_NOINLINE __attribute__((noinline))
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
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

// Cold helpers kept out of line so that the hot loop stays small (the gfx950
// instruction cache is shared by two CUs; the round body must fit in it).
AF_CORE_NOINLINE double cold_variate(uint32_t dist, double mean, double sigma, double u1, uint64_t seed,
                                     uint32_t stream, uint32_t index) {
    return variate_from_u1(dist, mean, sigma, u1, seed, stream, index);
}
AF_CORE_NOINLINE double cold_users_draw(uint32_t dist, double mean, double sigma, uint64_t seed, uint32_t idx) {
    if (dist == DIST_NORMAL) {  // gaussian_poisson.py:72-76 + common_helpers.py:32-33
        const double v = mean + sigma * af_norminv(uniform_j(seed, STREAM_GENERATOR, idx, 0u));
        return v > 0.0 ? v : 0.0;
    }
    return (double)af_poisson(mean, seed, STREAM_GENERATOR, idx, 0u);  // poisson_poisson.py:60
}
AF_CORE_NOINLINE uint32_t cold_endpoint_pick(uint64_t seed, uint32_t sv, uint32_t idx, uint32_t n_ep) {
    const U4 r = draw_block(seed, stream_server(sv), idx, 0u);  // rng.integers(0, n_ep), server.py:101
    return (uint32_t)(((uint64_t)r.x * n_ep) >> 32);
}

template <class Mem>
struct Lane {
    const PlanView& P;
    const Layout& L;
    Mem M;
    LaneOut O;
    uint64_t seed;

    // register-resident scalars
    double now, t_gen, g_now, g_wend, g_lam, t_tick, t_emark, t_smark;
    double users_mean, users_sigma, rpm;
    uint32_t g_draws, heap_n, seq, bump, free_top, live, max_live, lb_n, emark_i, smark_i;
    uint32_t n_gen, n_comp, n_drop, n_events, n_ticks, n_marks, flags, rounds;

    // per-round work registers ("follow-ups" of the timed event being handled)
    bool hole;            // the popped event left the heap root free
    uint32_t pend_count;  // pushes buffered this pass (<= 2): the heap code exists once
    double pend_t0, pend_t1;
    uint32_t pend_slot0, pend_slot1;
    bool gen_first;       // the generator's Initialize has not run yet
    uint32_t fu_grant;  // waiter that received a CPU token (NONE32 = none)
    uint32_t fu_grant_sv;
    bool fu_send;       // a message has to be put on an edge
    uint32_t send_slot, send_edge, send_hops;
    bool fu_adv;        // a request (re)enters the endpoint step loop
    uint32_t adv_slot, adv_sv, adv_ep, adv_step, adv_hops;
    bool adv_core, adv_io;
    int32_t fu_ram_sv;  // RAM was released on this server: serve its wait queue

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
    // Apply the buffered pushes of this pass: the first one takes the root when the
    // popped event left it free (replace-top), otherwise they are appended; if
    // nothing replaced the popped event the root is removed.  ONE sift_down and ONE
    // sift_up instance serve every case.
    AF_CORE void heap_commit(bool final_pass) {
        uint32_t k = 0u;
        if (hole && (pend_count > 0u || final_pass)) {
            double key;
            uint32_t slot;
            if (pend_count > 0u) {
                key = pend_t0;
                slot = pend_slot0;
                k = 1u;
            } else {
                heap_n -= 1u;
                key = M.ld64(L.d_hk + heap_n);
                slot = M.ld32(L.w_hs + heap_n);
            }
            hole = false;
            if (heap_n > 0u && (k == 1u || heap_n > 0u)) sift_down(0u, key, slot, heap_n);
        }
        for (; k < pend_count; ++k) sift_up(heap_n++, k == 0u ? pend_t0 : pend_t1, k == 0u ? pend_slot0 : pend_slot1);
        pend_count = 0u;
    }
    // schedule the (single) pending timed event of request `slot` (caller guarantees room)
    AF_CORE void emit(double t, uint32_t slot) {
        M.st32(L.w_rseq + slot, seq++);
        if (pend_count == 0u) {
            pend_t0 = t;
            pend_slot0 = slot;
        } else {
            pend_t1 = t;
            pend_slot1 = slot;
        }
        pend_count += 1u;
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
            return NONE32;
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
                g_lam = cold_users_draw(P.gen_users_dist, users_mean, users_sigma, seed, g_draws++) * rps_per_user;
            }
            if (g_lam <= 0.0) {
                g_now = g_wend;
                continue;
            }
            const U4 r = draw_block(seed, STREAM_GENERATOR, g_draws++, 0u);
            double u = u53(r.x, r.y);
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

    // ---- SEND stage: EdgeRuntime.transport/_deliver up to the timeout (edge.py:73-107) ----
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
        const double u1 = u53(r.z, r.w);
        const uint32_t dist = P.e_dist[e];
        const double mean = edge_mean(e);
        double transit;
        if (dist == DIST_EXPONENTIAL) {
            transit = -(mean * af_log(1.0 - u1));
        } else {
            transit = cold_variate(dist, mean, edge_sigma(e), u1, seed, stream, idx);
        }
        const double effective = transit + M.ld64(L.d_spike + e);  // spike read at SEND time (edge.py:94-106)
        M.st32(L.w_rst + slot, rst_pack(RK_TRANSIT, e, hops, 0u));
        emit(now + effective, slot);
    }

    // ---- GRANT stage: the waiter's `yield cpu_req` returns (server.py:220-231) ----
    AF_CORE void cpu_granted(uint32_t w, uint32_t sv) {
        M.st32(L.w_ready + sv, M.ld32(L.w_ready + sv) - 1u);
        const uint32_t st = M.ld32(L.w_rst + w);
        const uint32_t step = M.ld32(L.w_rst2 + w) >> 16;
        M.st32(L.w_rst + w, (st & ~7u) | RK_CPU);
        emit(now + step_time(step), w);
    }
    // a CPU token became free: hand it to the first waiter (Container FIFO)
    AF_CORE void cpu_release(uint32_t sv) {
        if (M.ld32(L.w_cqn + sv) > 0u) {
            fu_grant = q_pop(L.w_cq, L.w_cqh, L.w_cqn, sv);
            fu_grant_sv = sv;
        } else {
            M.st32(L.w_cpufree + sv, M.ld32(L.w_cpufree + sv) + 1u);
        }
    }

    // ---- ADV stage: the for-loop of _handle_request (server.py:197-276) from `step`
    // until the next timed event, a wait, or the end of the endpoint.
    AF_CORE void advance(uint32_t slot, uint32_t sv, uint32_t ep, uint32_t step, uint32_t hops, bool core_locked,
                         bool in_io) {
        const uint32_t end = P.ep_stepb[ep + 1u];
        const uint32_t rst2 = ep | (step << 16);
        if (step < end) {
            uint32_t kind_bits;
            if (P.st_kind[step] == STEP_CPU) {  // server.py:199-231
                if (in_io) M.st32(L.w_io + sv, M.ld32(L.w_io + sv) - 1u);
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
                kind_bits = rst_pack(RK_CPU, sv, hops, 0u);
            } else {  // I/O step, server.py:235-255
                if (core_locked) cpu_release(sv);  // the waiter's Timeout is created AFTER this one
                if (!in_io) M.st32(L.w_io + sv, M.ld32(L.w_io + sv) + 1u);
                kind_bits = rst_pack(RK_IO, sv, hops, 1u);
            }
            M.st32(L.w_rst + slot, kind_bits);
            M.st32(L.w_rst2 + slot, rst2);
            emit(now + step_time(step), slot);
            return;
        }
        // endpoint finished, server.py:257-276: waiter's grant, then transport(), then RAM waiters
        if (core_locked) cpu_release(sv);
        if (in_io) M.st32(L.w_io + sv, M.ld32(L.w_io + sv) - 1u);
        const double ram = P.ep_ram[ep];
        if (ram > 0.0) {
            M.st64(L.d_ramuse + sv, M.ld64(L.d_ramuse + sv) - ram);
            M.st64(L.d_ramfree + sv, M.ld64(L.d_ramfree + sv) + ram);
            if (M.ld32(L.w_rqn + sv) > 0u) fu_ram_sv = (int32_t)sv;
        }
        fu_send = true;
        send_slot = slot;
        send_edge = (uint32_t)P.s_out[sv];
        send_hops = hops;
    }

    // ---- RAM stage: Container._trigger_get, FIFO with head-of-line blocking ------
    AF_CORE void ram_stage() {
        const uint32_t sv = (uint32_t)fu_ram_sv;
        if (M.ld32(L.w_rqn + sv) == 0u) {
            fu_ram_sv = -1;
            return;
        }
        const uint32_t w = q_front(L.w_rq, L.w_rqh, sv);
        const uint32_t rst2 = M.ld32(L.w_rst2 + w);
        const uint32_t ep = rst2 & 0xFFFFu;
        const double need = P.ep_ram[ep];
        const double free_ram = M.ld64(L.d_ramfree + sv);
        if (free_ram < need) {
            fu_ram_sv = -1;
            return;
        }
        q_pop(L.w_rq, L.w_rqh, L.w_rqn, sv);
        M.st64(L.d_ramfree + sv, free_ram - need);
        M.st64(L.d_ramuse + sv, M.ld64(L.d_ramuse + sv) + need);
        fu_adv = true;  // keep fu_ram_sv: the queue is looked at again after this waiter
        adv_slot = w;
        adv_sv = sv;
        adv_ep = ep;
        adv_step = rst2 >> 16;
        adv_hops = (M.ld32(L.w_rst + w) >> 11) & 0xFFu;
        adv_core = false;
        adv_io = false;
    }

    // _dispatcher + head of _handle_request (server.py:303-313, 79-149)
    AF_CORE void server_arrival(uint32_t slot, uint32_t sv, uint32_t hops) {
        hops += 1u;  // record_hop(SERVER)
        const uint32_t epb = P.s_epb[sv];
        const uint32_t n_ep = P.s_epb[sv + 1u] - epb;
        const uint32_t idx = M.ld32(L.w_arr + sv);
        M.st32(L.w_arr + sv, idx + 1u);
        const uint32_t pick = n_ep > 1u ? cold_endpoint_pick(seed, sv, idx, n_ep) : 0u;
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
        fu_adv = true;
        adv_slot = slot;
        adv_sv = sv;
        adv_ep = ep;
        adv_step = step0;
        adv_hops = hops;
        adv_core = false;
        adv_io = false;
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
                fu_send = true;
                send_slot = slot;
                send_edge = (uint32_t)P.client_out_edge;
                send_hops = hops;
            }
        } else if (tk == NODE_LB) {  // LoadBalancerRuntime._forwarder, load_balancer.py:60-72
            hops += 1u;
            uint32_t out = M.ld32(L.w_lb);
            if (P.lb_algo == LB_LEAST_CONNECTIONS) {  // lb_algorithms.py:10-20
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
                for (uint32_t i = 1u; i < lb_n; ++i) M.st32(L.w_lb + i - 1u, M.ld32(L.w_lb + i));
                M.st32(L.w_lb + lb_n - 1u, out);
            }
            fu_send = true;
            send_slot = slot;
            send_edge = out;
            send_hops = hops;
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
        t_emark = emark_i < P.n_edge_marks ? P.em_time[emark_i] : AF_INF;
    }
    AF_CORE void apply_smarks() {
        for (;;) {
            const uint32_t i = smark_i++;
            const int32_t e = P.sm_edge[i];
            n_marks += 1u;
            if (e >= 0) {
                uint32_t pos = NONE32;
                for (uint32_t k = 0u; k < lb_n; ++k)
                    if (M.ld32(L.w_lb + k) == (uint32_t)e) pos = k;
                if (P.sm_down[i]) {  // lb_out_edges.pop(edge_id, None)
                    if (pos != NONE32) {
                        for (uint32_t k = pos + 1u; k < lb_n; ++k) M.st32(L.w_lb + k - 1u, M.ld32(L.w_lb + k));
                        lb_n -= 1u;
                    }
                } else {  // re-insert + move_to_end
                    if (pos != NONE32) {
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
        t_smark = smark_i < P.n_srv_marks ? P.sm_time[smark_i] : AF_INF;
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
        hole = fu_send = fu_adv = false;
        pend_count = 0u;
        gen_first = true;
        fu_grant = NONE32;
        fu_ram_sv = -1;
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
        t_gen = 0.0;  // the generator's Initialize runs as the first "arrival" round (gen_first)
        t_tick = 0.0 + P.sample_period;
        t_emark = P.n_edge_marks ? P.em_time[0] : AF_INF;
        t_smark = P.n_srv_marks ? P.sm_time[0] : AF_INF;
    }

    // One next-event round.  Returns false once the scenario reached the horizon.
    //
    // Structure (every heavy block appears ONCE so the code stays I-cache sized and
    // lanes re-converge at each stage):
    //   select -> decode (light, per event kind) -> [ADV -> GRANT -> SEND -> RAM]* -> heap commit
    // The stage order is the order SimPy creates the corresponding Timeouts in
    // (server.py:235-276): own I/O timer before the CPU waiter's; on endpoint end
    // the CPU waiter's timer, then transport(), then the RAM waiters.
    AF_CORE bool round() {
        // next event among {heap, arrival, tick, server marks, edge marks}; on equal
        // times the LATER test wins: edge marks < server marks < tick < arrival < heap.
        uint32_t cls = 4u;
        double t = heap_n > 0u ? M.ld64(L.d_hk) : AF_INF;
        if (t_gen <= t) { cls = 3u; t = t_gen; }
        if (t_tick <= t) { cls = 2u; t = t_tick; }
        if (t_smark <= t) { cls = 1u; t = t_smark; }
        if (t_emark <= t) { cls = 0u; t = t_emark; }
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
                fu_adv = true;
                adv_slot = slot;
                adv_sv = idx;
                adv_ep = rst2 & 0xFFFFu;
                adv_step = (rst2 >> 16) + 1u;
                adv_hops = hops;
                adv_core = kind == RK_CPU;
                adv_io = (st >> 19) & 1u;
            }
        } else if (cls == 3u) {  // RqsGeneratorRuntime._event_arrival (rqs_generator.py:101-119)
            const bool first = gen_first;  // Initialize: only draws the first gap
            gen_first = false;
            const double t_arrival = now;
            const double gap = next_gap();
            t_gen = gap >= 0.0 ? now + gap : AF_INF;
            uint32_t slot = NONE32;
            if (!first) {
                n_gen += 1u;
                n_events += 1u;
                slot = alloc_slot();
            }
            if (slot != NONE32) {
                M.st64(L.d_t0 + slot, t_arrival);
                fu_send = true;
                send_slot = slot;
                send_edge = (uint32_t)P.gen_out_edge;
                send_hops = 1u;  // record_hop(GENERATOR)
            }
        } else if (cls == 2u) {
            sample_tick();
            t_tick = now + P.sample_period;
        } else if (cls == 1u) {
            apply_smarks();
        } else {
            apply_emarks();
        }

        for (;;) {
            // each stage pushes at most one event; the buffer holds two, so a pass
            // stops early (stage order preserved) when it is full -- rare.
            if (fu_adv) {
                fu_adv = false;
                advance(adv_slot, adv_sv, adv_ep, adv_step, adv_hops, adv_core, adv_io);
            }
            bool room = pend_count < 2u;
            if (fu_grant != NONE32 && room) {
                const uint32_t w = fu_grant;
                fu_grant = NONE32;
                cpu_granted(w, fu_grant_sv);
                room = pend_count < 2u;
            }
            if (fu_send && fu_grant == NONE32 && room) {
                fu_send = false;
                edge_send(send_slot, send_edge, send_hops);
            }
            if (fu_ram_sv >= 0 && fu_grant == NONE32 && !fu_send) ram_stage();
            const bool more = fu_adv || fu_grant != NONE32 || fu_send || fu_ram_sv >= 0;
            heap_commit(!more);
            if (!more) break;
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
