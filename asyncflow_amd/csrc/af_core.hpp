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
//     whose entries carry the whole request: (time, start time, seq|state); there
//     is no request pool and no dependent state lookup after a pop;
//     ties are broken by a per-scenario push sequence; the generator, sampler and
//     injection timers are register-resident "special" sources with a fixed class
//     order on ties;
//   * every zero-time SimPy step following a timed event is executed inline in
//     the order SimPy runs it when no other timed event shares the timestamp;
//   * when two or more timed events DO share a timestamp, SimPy interleaves their
//     zero-time steps breadth-first (every step is a NORMAL heap event behind the
//     other timed events of that instant, process Initialize events are URGENT).
//     Such instants are detected before anything is processed and run through
//     `micro_mode`, a cold path that replays SimPy's event-by-event order on the
//     same state (FIFO of zero-time events in a per-scenario HBM scratch).
//
// State lives in `Mem`, a per-lane memory of 64-bit words laid out [index][lane]
// (SoA across the 64 lanes of a wave): LDS when it fits (ds_read_b64 of
// [i][lane] is bank-conflict free for ANY per-lane index), HBM otherwise
// (coalesced for equal indices).
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
enum : uint32_t { METRIC_READY = 1, METRIC_IO = 2, METRIC_RAM = 4, METRIC_EDGE = 8 };
enum : uint32_t {
    FLAG_POOL_OVERFLOW = 1u << 0,
    FLAG_FIFO_OVERFLOW = 1u << 1,
    FLAG_CLOCK_OVERFLOW = 1u << 2,
    FLAG_TICK_OVERFLOW = 1u << 3,
    FLAG_RAM_STARVED = 1u << 4,
    FLAG_TIME_TIE = 1u << 5,
    FLAG_DRAW_OVERFLOW = 1u << 6,
    FLAG_SHARED_INSTANT = 1u << 7,  // internal: never visible in the outputs of af_engine_run
    FLAG_NEGATIVE_DELAY = 1u << 13, // transit + spike < 0 at a send (the reference raises "Negative delay")
};
enum : uint32_t {
    CNT_GENERATED = 0, CNT_COMPLETED, CNT_DROPPED, CNT_EVENTS, CNT_TICKS, CNT_FLAGS, CNT_MAX_LIVE, CNT_MARKS, CNT_SLOTS
};
enum : uint32_t {
    PARAM_GEN_USERS_MEAN = 0, PARAM_GEN_USERS_SIGMA, PARAM_GEN_RPM_MEAN, PARAM_EDGE_MEAN, PARAM_EDGE_SIGMA,
    PARAM_EDGE_DROPOUT, PARAM_STEP_TIME,
    // round 3: sampling window, server resources, the injected events' timelines
    PARAM_GEN_WINDOW, PARAM_SRV_CORES, PARAM_SRV_RAM_MB, PARAM_EMARK_TIME, PARAM_EMARK_DELTA, PARAM_EMARK_EDGE,
    PARAM_SMARK_TIME, PARAM_SMARK_LB_EDGE, PARAM_SMARK_DOWN, PARAM_COUNT
};
constexpr uint32_t kParamMarkBits = (1u << PARAM_EMARK_TIME) | (1u << PARAM_EMARK_DELTA) | (1u << PARAM_EMARK_EDGE) |
                                    (1u << PARAM_SMARK_TIME) | (1u << PARAM_SMARK_LB_EDGE) | (1u << PARAM_SMARK_DOWN);

// ---- request state (low 32 bits of a heap / queue entry's B word) -----------
//   kind[0:1] | idx[2:9] (edge while in transit, server otherwise) | hops[10:12] (saturating:
//   only `len(history) > 3` is ever tested, client.py:62) | in_io[13] | step row[14:29]
enum : uint32_t { RK_TRANSIT = 0, RK_CPU = 1, RK_IO = 2, RK_WAIT = 3 };
AF_HD uint32_t st_pack(uint32_t kind, uint32_t idx, uint32_t hops, uint32_t in_io, uint32_t step) {
    return kind | (idx << 2) | ((hops > 7u ? 7u : hops) << 10) | (in_io << 13) | (step << 14);
}
AF_HD uint32_t st_kind(uint32_t s) { return s & 3u; }
AF_HD uint32_t st_idx(uint32_t s) { return (s >> 2) & 0xFFu; }
AF_HD uint32_t st_hops(uint32_t s) { return (s >> 10) & 7u; }
AF_HD uint32_t st_io(uint32_t s) { return (s >> 13) & 1u; }
AF_HD uint32_t st_step(uint32_t s) { return (s >> 14) & 0xFFFFu; }

// ---- the lowered plan as seen by device code: arrays of 64-bit records -------
// (LDS-resident copy under hipcc; packed on the host by af_plan_pack.hpp)
//   edge  record [4]: mean, sigma, dropout, meta = tkind | tidx<<8 | dist<<16
//   server record[2]: ram_mb, meta = cores | out_edge<<16 | first_endpoint<<32 | n_endpoints<<48
//   endpoint rec [2]: ram, first step row
//   step row     [3]: time, ram of its endpoint, kind (STEP_CPU | STEP_IO | STEP_END);
//                     every endpoint's rows end with one STEP_END row
//   edge mark    [3]: time, delta, edge ; server mark [2]: time, meta = (lb_edge+1) | down<<32
enum : uint32_t { STEP_CPU = 0, STEP_IO = 1, STEP_END = 2 };
enum : uint32_t { EREC = 4, SREC = 2, PREC = 2, TREC = 3, MREC = 3, NREC = 2 };

struct PlanView {
    double total_time, sample_period;
    double gen_users_mean, gen_users_sigma, gen_rpm_mean, gen_window_s;
    uint32_t metrics_mask, gen_users_dist;
    uint32_t gen_out_edge, client_out_edge;
    uint32_t n_edges, n_servers, lb_algo, n_lb_edges, n_rows, n_edge_marks, n_srv_marks;
    uint32_t every_event_in_order;  // 1: every request event goes through the SimPy-order path (af_plan_pack.hpp)
    const AF_PLAN_AS uint64_t* edge;
    const AF_PLAN_AS uint64_t* srv;
    const AF_PLAN_AS uint64_t* ep;
    const AF_PLAN_AS uint64_t* row;
    const AF_PLAN_AS uint64_t* emark;
    const AF_PLAN_AS uint64_t* smark;
    const AF_PLAN_AS uint64_t* lb;  // [n_lb_edges] out-edge indices in payload order
};

AF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
AF_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }

// ---- per-lane state layout (offsets in 64-bit words; computed by the host) ----
//   heap K/A/B [cap] each; edge [E][2] = {conn[0:15] | ring_ahead[16:23] | sends<<32, spike};
//   draw rings [1+E][RING] (stream 0 = arrival times, 1+e = edge e transit times, staged
//   from the pre-generated HBM arrays); optional per-scenario step-time column; server
//   [S][6] = {cpu_free | ready<<32, io | arrivals<<32, ram_free, ram_in_use,
//   cq_head | cq_n<<32, rq_head | rq_n<<32 | blocked<<63}; wait queues
//   [S][fcap][2] = {start time, state}; LB order [n_lb] (only used when n_lb > 8).
//   (round 6: a queue word per queue -- rounds 1-5 packed both into one, 16- and 15-bit counts, which capped a server's
//   queue at 16 384 waiters where the reference's simpy Container queue has no bound, server.py:146-149, 210-227)
struct Layout {
    uint32_t cap;       // pending timed events (== requests in flight) per scenario
    uint32_t fcap;      // per-server wait-queue capacity (power of two, <= AF_MAX_FIFO_CAPACITY = 2^20)
    uint32_t ovr_mask;  // bit p set: af_param class p is overridden per scenario
    uint32_t hk, ha, hb, edge, ring, stime, srv, cq, rq, lb, n_words;
    uint32_t srvram, marks;   // per-scenario ram_mb [S] / timeline marks [MREC n_emarks + NREC n_smarks] (only when overridden)
    // per-scenario HBM scratch of the shared-timestamp path (64-bit words, plain array):
    //   zero-time event FIFO [tcap][2] | node inbox items [tcap][2] | refused RAM puts [tcap][2] (they outlive the instant:
    //   Lane::pb_n) | forwarder-busy bits
    uint32_t tcap, tie_words;
};
#ifndef AF_RING_LOW
#define AF_RING_LOW 1
#endif
enum : uint32_t { LEDGE = 2, LSRV = 6, RING = 4, RING_LOW = AF_RING_LOW };

AF_HD Layout make_layout(uint32_t cap, uint32_t fcap, uint32_t n_edges, uint32_t n_servers, uint32_t n_lb,
                         uint32_t n_rows, uint32_t ovr_mask, uint32_t n_emarks = 0u, uint32_t n_smarks = 0u) {
    Layout L{};
    L.cap = cap;
    L.fcap = fcap;
    L.ovr_mask = ovr_mask;
    uint32_t w = 0;
    L.hk = w; w += cap;
    L.ha = w; w += cap;
    L.hb = w; w += cap;
    L.edge = w; w += LEDGE * n_edges;
    L.ring = w; w += RING * (1u + n_edges);
    L.stime = w; if (ovr_mask & (1u << PARAM_STEP_TIME)) w += n_rows;
    L.srv = w; w += LSRV * n_servers;
    L.cq = w; w += 2u * n_servers * fcap;
    L.rq = w; w += 2u * n_servers * fcap;
    L.lb = w; w += n_lb > 8u ? n_lb : 0u;
    L.srvram = w; if (ovr_mask & (1u << PARAM_SRV_RAM_MB)) w += n_servers;
    L.marks = w; if (ovr_mask & kParamMarkBits) w += 3u * n_emarks + 2u * n_smarks;   // MREC, NREC
    L.n_words = w;
    uint32_t tc = 16u;
    while (tc < 2u * cap + 8u) tc <<= 1;
    L.tcap = tc;
    L.tie_words = 6u * tc + (2u + n_servers + 63u) / 64u + 17u;  // + Lane::PARK_WORDS at the end
    return L;
}
AF_HD uint64_t layout_bytes_per_lane(const Layout& L) { return 8ull * L.n_words; }

// ---- outputs of one scenario -------------------------------------------------
struct LaneOut {
    double* clock;      // [clock_cap][2] or nullptr
    uint32_t* samples;  // [tick_cap][series_pitch] or nullptr (16-byte aligned rows)
    uint32_t* counts;   // [CNT_SLOTS]
    uint32_t clock_cap, tick_cap, series_pitch;
    // optional summary accumulated by the kernel itself (af_outputs_t.online_*): integer counters
    // bumped with fire-and-forget atomics, for sweeps whose per-request clock would not fit
    uint32_t* hist;     // [hist_bins] latency histogram or nullptr
    uint32_t* rps;      // [rps_buckets] completions per 1-s window (k-1, k] or nullptr
    uint32_t hist_bins, rps_buckets;
    double hist_scale;  // hist_bins / hist_max
};

#if defined(__HIP_DEVICE_COMPILE__)
#define AF_BUMP(p) ((void)atomicAdd((p), 1u))  // result unused: the no-return form, never waited for
#else
#define AF_BUMP(p) ((void)(*(p) += 1u))
#endif

// one 16-byte store (rows of the sample array are 16-byte aligned)
AF_HD void store4(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#else
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}

constexpr double AF_INF = __builtin_huge_val();

// ---- pre-generated random draws --------------------------------------------------
// Every random variate of a scenario is a pure function of (seed, stream, draw index)
// and of the scenario's parameters -- never of the simulation state.  They are therefore
// produced up front by fully parallel kernels (engine.hip: af_pregen_*), at full GPU
// occupancy, into HBM:  draws[scenario][stream][index]  (f64; a scenario's next draws of a
// stream are contiguous, so the 4-entry top-ups below touch one or two 32 B sectors)
//   stream 0     : absolute arrival times of the generator (AF_INF after the last one)
//   stream 1 + e : transit time of the index-th message sent on edge e, or -1.0 if that
//                  message is dropped (edge.py:78-86)
// The sequential next-event kernel only stages them through small LDS rings: a ring is
// topped up (4 loads issued at the start of a round, stored at its end -- never waited
// for) in the round after one of its entries was consumed.
struct PreDraws {
    const double* base;     // this scenario's block: [1 + n_edges][n_per_stream]
    uint32_t n_per_stream;  // entries per stream
    uint32_t flags_in;      // AF_FLAG_DRAW_OVERFLOW if the arrival stream did not fit
    uint64_t* tie;          // this scenario's scratch for shared timestamps: [Layout::tie_words]
    AF_HD double entry(uint32_t stream, uint32_t index) const {
        return base[(size_t)stream * n_per_stream + index];
    }
};

// One message on one edge: EdgeRuntime._deliver's dropout test + latency draw
// (edge.py:78-90, samplers/common_helpers.py:49-89).
AF_HD double pre_edge_draw(uint64_t seed, uint32_t e, uint32_t idx, uint32_t dist, double mean, double sigma,
                           double dropout) {
    const uint32_t stream = stream_edge(e);
    const U4 r = draw_block(seed, stream, idx, 0u);
    if (u53(r.x, r.y) < dropout) return -1.0;  // dropped: no latency draw
    return test_quant(variate_from_u1(dist, mean, sigma, u53(r.z, r.w), seed, stream, idx));
}

// The windowed arrival sampler: samplers/poisson_poisson.py:51-82, gaussian_poisson.py:63-94.
struct GenState {
    double g_now = 0.0, g_wend = 0.0, g_lam = 0.0;
    uint32_t draws = 0;
};
AF_HD double gen_next_gap(GenState& g, uint64_t seed, uint32_t users_dist, double users_mean, double users_sigma,
                          double rpm, double window_s, double T) {
    const double rps_per_user = rpm / 60.0;
    while (g.g_now < T) {
        if (g.g_now >= g.g_wend) {
            g.g_wend = g.g_now + window_s;
            const uint32_t idx = g.draws++;
            double users;
            if (users_dist == DIST_NORMAL) {  // gaussian_poisson.py:72-76 + common_helpers.py:32-33
                const double v = users_mean + users_sigma * af_norminv(uniform_j(seed, STREAM_GENERATOR, idx, 0u));
                users = v > 0.0 ? v : 0.0;
            } else {  // poisson_poisson.py:60
                users = (double)af_poisson(users_mean, seed, STREAM_GENERATOR, idx, 0u);
            }
            g.g_lam = users * rps_per_user;
        }
        if (g.g_lam <= 0.0) {
            g.g_now = g.g_wend;
            continue;
        }
        const U4 r = draw_block(seed, STREAM_GENERATOR, g.draws++, 0u);
        double u = u53(r.x, r.y);
        if (u < 1e-15) u = 1e-15;
        const double dt = test_quant(-af_log_unit(1.0 - u) / g.g_lam);
        if (g.g_now + dt > T) break;
        if (g.g_now + dt >= g.g_wend) {
            g.g_now = g.g_wend;
            continue;
        }
        g.g_now += dt;
        return dt;
    }
    g.g_now = T + 1.0;
    return -1.0;
}

AF_CORE_NOINLINE uint32_t cold_endpoint_pick(uint64_t seed, uint32_t sv, uint32_t idx, uint32_t n_ep) {
    const U4 r = draw_block(seed, stream_server(sv), idx, 0u);  // rng.integers(0, n_ep), server.py:101
    return (uint32_t)(((uint64_t)r.x * n_ep) >> 32);
}

// Every scalar of a scenario that survives from one round to the next.
struct LaneRegs {
    double now, t_gen, t_tick, t_emark, t_smark;
    uint64_t lb_list;    // LB out-edge order, 8 bits per entry (n_lb_edges <= 8), else in Mem
    // creation sequence of the pending generator / sampler / timeline timers: together with the
    // sequence in every heap entry this is SimPy's event-id order among TIMED events
    uint32_t q_gen, q_tick, q_emark, q_smark;
    uint32_t arr_ahead;  // arrival times staged in the ring beyond the next one
    int32_t dirty_edge;  // edge whose ring was consumed from last (-1 = none): topped up next round
    uint32_t fl;
    uint32_t heap_n, seq, live, max_live, lb_n, emark_i, smark_i;
    uint32_t n_gen, n_comp, n_drop, n_events, n_ticks, n_marks, flags;
    uint32_t pb_n;       // RAM puts simpy's Container refused and that still wait (micro_mode; 0 unless needs are fractional)
};

template <bool kFaithful> struct RoundType { using type = uint32_t; };
template <> struct RoundType<false> { using type = bool; };

// kFaithful = false builds the lean variant used for the first pass of a sweep: it has no
// SimPy-order path; a scenario that meets a shared instant stops with FLAG_SHARED_INSTANT and
// is simulated again, from the start, by the kFaithful = true variant (engine.hip).
template <class Mem, bool kFaithful = true>
struct Lane : LaneRegs {
    const PlanView& P;
    const Layout& L;
    Mem M;
    LaneOut O;
    PreDraws D;
    uint64_t seed;

    // Per-lane boolean state lives in ONE register `fl` (a bool per lane costs a 64-bit
    // lane mask in scalar registers and a pile of mask arithmetic at every branch).
    enum : uint32_t {
        F_HOLE = 1u,       // the popped event left the heap root free
        F_GRANT = 2u,      // a waiter received a CPU token
        F_SEND = 4u,       // a message has to be put on an edge
        F_ADV = 8u,        // a request (re)enters the endpoint step loop
        F_ADV_CORE = 16u,  // ... holding a CPU core
        F_ADV_IO = 32u,    // ... counted in the I/O queue
        F_DIRTY_ARR = 64u,   // an arrival time was consumed: top its ring up next round
        F_SEND_FIRST = 128u  // this pass: the message leaves before the CPU waiter resumes
    };

    // per-round work registers ("follow-ups" of the timed event being handled)
    uint32_t pend_count;  // pushes buffered this pass (<= 2): the heap code exists once
    double pend_k0, pend_k1;
    uint64_t pend_a0, pend_a1, pend_b0, pend_b1;
    uint64_t grant_a;
    uint32_t grant_st;
    uint64_t send_a;
    uint32_t send_edge, send_hops;
    uint64_t adv_a;
    uint32_t adv_sv, adv_step, adv_hops;
    int32_t fu_ram_sv;  // RAM was released on this server: serve its wait queue

    AF_CORE Lane(const PlanView& p, const Layout& l, Mem m, LaneOut o, PreDraws d, uint64_t s)
        : P(p), L(l), M(m), O(o), D(d), seed(s) {}

    // ---- parameter accessor (plan value or per-scenario column) ---------------
    // per-scenario server RAM / timeline marks when the sweep has such columns, else the plan's
    AF_CORE double ram_cap(uint32_t sv) const {
        return (L.ovr_mask & (1u << PARAM_SRV_RAM_MB)) ? u2d(M.ld(L.srvram + sv)) : u2d(P.srv[SREC * sv]);
    }
    AF_CORE uint64_t emark_w(uint32_t i, uint32_t k) const {
        return (L.ovr_mask & kParamMarkBits) ? M.ld(L.marks + MREC * i + k) : P.emark[MREC * i + k];
    }
    AF_CORE uint64_t smark_w(uint32_t i, uint32_t k) const {
        return (L.ovr_mask & kParamMarkBits) ? M.ld(L.marks + MREC * P.n_edge_marks + NREC * i + k) : P.smark[NREC * i + k];
    }
    AF_CORE double row_time(uint32_t r) const {
        return (L.ovr_mask & (1u << PARAM_STEP_TIME)) ? u2d(M.ld(L.stime + r)) : u2d(P.row[TREC * r]);
    }

    // ---- priority queue: binary heap of (time, start, seq|state) ------------------
    // (time, push sequence) order; the sequence sits in the high half of B
    AF_CORE static bool ev_less(double ka, uint64_t ba, double kb, uint64_t bb) {
        return ka < kb || (ka == kb && (ba >> 32) < (bb >> 32));
    }
    AF_CORE void sift_down(uint32_t pos, double key, uint64_t a, uint64_t b, uint32_t n) {
        for (;;) {
            uint32_t c = 2u * pos + 1u;
            if (c >= n) break;
            // both children are fetched in full (6 independent LDS reads, one latency)
            double kc = u2d(M.ld(L.hk + c));
            uint64_t ac = M.ld(L.ha + c);
            uint64_t bc = M.ld(L.hb + c);
            if (c + 1u < n) {
                const double k2 = u2d(M.ld(L.hk + c + 1u));
                const uint64_t a2 = M.ld(L.ha + c + 1u);
                const uint64_t b2 = M.ld(L.hb + c + 1u);
                if (ev_less(k2, b2, kc, bc)) {
                    c += 1u;
                    kc = k2;
                    ac = a2;
                    bc = b2;
                }
            }
            if (!ev_less(kc, bc, key, b)) break;
            M.st(L.hk + pos, d2u(kc));
            M.st(L.ha + pos, ac);
            M.st(L.hb + pos, bc);
            pos = c;
        }
        M.st(L.hk + pos, d2u(key));
        M.st(L.ha + pos, a);
        M.st(L.hb + pos, b);
    }
    AF_CORE void sift_up(uint32_t pos, double key, uint64_t a, uint64_t b) {
        while (pos > 0u) {
            const uint32_t p = (pos - 1u) >> 1;
            const double kp = u2d(M.ld(L.hk + p));
            const uint64_t bp = M.ld(L.hb + p);
            if (!ev_less(key, b, kp, bp)) break;
            M.st(L.hk + pos, d2u(kp));
            M.st(L.ha + pos, M.ld(L.ha + p));
            M.st(L.hb + pos, bp);
            pos = p;
        }
        M.st(L.hk + pos, d2u(key));
        M.st(L.ha + pos, a);
        M.st(L.hb + pos, b);
    }
    // Apply the buffered pushes of this pass: the first one takes the root when the
    // popped event left it free (replace-top), otherwise they are appended; if
    // nothing replaced the popped event the root is removed.  ONE sift_down and ONE
    // sift_up instance serve every case.
    AF_CORE void heap_commit(bool final_pass) {
        uint32_t k = 0u;
        if ((fl & F_HOLE) && (pend_count > 0u || final_pass)) {
            double key;
            uint64_t a, b;
            if (pend_count > 0u) {
                key = pend_k0;
                a = pend_a0;
                b = pend_b0;
                k = 1u;
            } else {
                heap_n -= 1u;
                key = u2d(M.ld(L.hk + heap_n));
                a = M.ld(L.ha + heap_n);
                b = M.ld(L.hb + heap_n);
            }
            fl &= ~F_HOLE;
            if (heap_n > 0u) sift_down(0u, key, a, b, heap_n);
        }
        for (; k < pend_count; ++k) {
            if (heap_n >= L.cap) {  // more requests in flight than request_capacity
                flags |= FLAG_POOL_OVERFLOW;
                live -= 1u;
                continue;
            }
            sift_up(heap_n++, k == 0u ? pend_k0 : pend_k1, k == 0u ? pend_a0 : pend_a1, k == 0u ? pend_b0 : pend_b1);
        }
        pend_count = 0u;
    }
    // schedule the (single) pending timed event of a request (caller guarantees room)
    AF_CORE void emit(double t, uint64_t a, uint32_t state) {
        // A zero-delay Timeout belongs BEHIND the zero-time steps still queued at this instant and
        // AHEAD of the steps those trigger (SimPy FIFO).  The inline cascade runs all of them first.
        // That is the same thing when nothing else is pending, and it commutes when the event is a
        // delivery to the client (completion bookkeeping only); anything else is reported.
        if (t == now) {
            const bool pending = (fl & (F_ADV | F_GRANT | F_SEND)) != 0u || fu_ram_sv >= 0;
            const bool to_client =
                st_kind(state) == RK_TRANSIT && ((uint32_t)P.edge[EREC * st_idx(state) + 3u] & 0xFFu) == NODE_CLIENT;
            if (pending && !to_client) flags |= FLAG_TIME_TIE;
        }
        const uint64_t b = ((uint64_t)(seq++) << 32) | state;
        if (pend_count == 0u) {
            pend_k0 = t;
            pend_a0 = a;
            pend_b0 = b;
        } else {
            pend_k1 = t;
            pend_a1 = a;
            pend_b1 = b;
        }
        pend_count += 1u;
    }

    // ---- server wait queues: rings of (start time, state) ---------------------------
    // queue words: [4] cq_head[0:31] | cq_n[32:63]; [5] rq_head[0:31] | rq_n[32:62] | ram_blocked[63]
    AF_CORE static uint32_t q_head(uint64_t q) { return (uint32_t)q; }
    AF_CORE static uint32_t q_n(uint64_t q) { return (uint32_t)(q >> 32) & 0x7FFFFFFFu; }
    AF_CORE uint64_t q_popped(uint64_t q) const {   // head + 1 (ring), n - 1; bit 63 stays
        return (q & (1ull << 63)) | (uint64_t)((q_head(q) + 1u) & (L.fcap - 1u)) | ((uint64_t)(q_n(q) - 1u) << 32);
    }
    AF_CORE bool q_push(uint32_t base, uint32_t sv, uint32_t head, uint32_t n, uint64_t a, uint32_t state) {
        if (n >= L.fcap) {
            flags |= FLAG_FIFO_OVERFLOW;
            live -= 1u;
            return false;
        }
        const uint32_t at = base + 2u * (sv * L.fcap + ((head + n) & (L.fcap - 1u)));
        M.st(at, a);
        M.st(at + 1u, (uint64_t)state);
        return true;
    }

    // ---- draw rings: stage the pre-generated draws through LDS ---------------------------
    // ring row of stream s, draw index i: L.ring + RING * s + (i & (RING - 1))
    struct TopUp {  // four draws in flight between the start and the end of a round
        double v0, v1, v2, v3;
        uint32_t stream, c0;
        bool active;
    };
    AF_CORE double draw_or(uint32_t stream, uint32_t i, double none) const {
        return i < D.n_per_stream ? D.entry(stream, i) : none;
    }
    AF_CORE TopUp topup_begin(bool want, uint32_t stream, uint32_t c0) const {
        TopUp t;
        t.active = want;
        t.stream = stream;
        t.c0 = c0;
        const double none = stream == 0u ? AF_INF : -1.0;
        t.v0 = t.v1 = t.v2 = t.v3 = none;
        if (want) {  // four independent global loads; nobody waits for them until topup_end
            t.v0 = draw_or(stream, c0, none);
            t.v1 = draw_or(stream, c0 + 1u, none);
            t.v2 = draw_or(stream, c0 + 2u, none);
            t.v3 = draw_or(stream, c0 + 3u, none);
        }
        return t;
    }
    AF_CORE void topup_store(const TopUp& t) {
        const uint32_t row = L.ring + RING * t.stream;
        M.st(row + (t.c0 & (RING - 1u)), d2u(t.v0));
        M.st(row + ((t.c0 + 1u) & (RING - 1u)), d2u(t.v1));
        M.st(row + ((t.c0 + 2u) & (RING - 1u)), d2u(t.v2));
        M.st(row + ((t.c0 + 3u) & (RING - 1u)), d2u(t.v3));
    }
    AF_CORE void topup_end_arrivals(const TopUp& t) {
        if (!t.active) return;
        topup_store(t);
        arr_ahead = t.c0 + RING - 1u - n_gen;  // ring now holds [c0, c0 + RING); t_gen is entry n_gen
    }
    AF_CORE void topup_end_edge(const TopUp& t) {
        if (!t.active) return;
        topup_store(t);
        const uint32_t at = L.edge + LEDGE * (t.stream - 1u);
        const uint64_t cs = M.ld(at);
        const uint32_t sends = (uint32_t)(cs >> 32);  // may have advanced past the staged window in a burst
        const uint32_t ahead = t.c0 + RING > sends ? t.c0 + RING - sends : 0u;
        M.st(at, (cs & ~(0xFFull << 16)) | ((uint64_t)ahead << 16));
    }

    // ---- SEND stage: EdgeRuntime.transport/_deliver up to the timeout (edge.py:73-107) ----
    template <bool kMicro = false>
    AF_CORE void edge_send(uint64_t a, uint32_t e, uint32_t hops) {
        const uint32_t at = L.edge + LEDGE * e;
        const uint64_t cs = M.ld(at);  // conn[0:15] | ring_ahead[16:23] | sends<<32
        const double spike = P.n_edge_marks != 0u ? u2d(M.ld(at + 1u)) : 0.0;  // no timeline: never anything but +0.0
        const uint32_t idx = (uint32_t)(cs >> 32);
        const uint32_t ahead = (uint32_t)(cs >> 16) & 0xFFu;
        // transit time of this edge's idx-th message (pre-drawn; -1 = dropped, edge.py:78-86)
        double transit;
        uint64_t ncs = cs + (1ull << 32);  // sends += 1
        if (ahead > 0u) {
            transit = u2d(M.ld(L.ring + RING * (1u + e) + (idx & (RING - 1u))));
            ncs -= 1ull << 16;
        } else {
            transit = draw_or(1u + e, idx, -1.0);  // ring empty: straight from HBM (rare)
            if (idx >= D.n_per_stream) flags |= FLAG_DRAW_OVERFLOW;
        }
        // ask for a top-up only at the low-water mark: one top-up then serves three sends instead of one
        if (ahead <= RING_LOW + 1u) dirty_edge = (int32_t)e;
        if (transit < 0.0) {
            M.st(at, ncs);
            n_drop += 1u;
            live -= 1u;
            return;
        }
        M.st(at, ncs + 1ull);  // conn += 1 (edge.py:88)
        const double effective = transit + spike;  // spike read at SEND time (edge.py:94-106)
        if (effective < 0.0) flags |= FLAG_NEGATIVE_DELAY;   // env.timeout(effective) raises in the reference (edge.py:107)
        if constexpr (kMicro) m_emit(now + effective, a, st_pack(RK_TRANSIT, e, hops, 0u, 0u), MK_EDGE_TIMEOUT);
        else emit(now + effective, a, st_pack(RK_TRANSIT, e, hops, 0u, 0u));
    }

    // ---- GRANT stage: the waiter's `yield cpu_req` returns (server.py:220-231) ----
    AF_CORE void cpu_granted() {
        const uint32_t sv = st_idx(grant_st);
        const uint32_t at = L.srv + LSRV * sv;
        M.st(at, M.ld(at) - (1ull << 32));  // ready -= 1
        const uint32_t step = st_step(grant_st);
        emit(now + row_time(step), grant_a, st_pack(RK_CPU, sv, st_hops(grant_st), 0u, step));
    }
    // a CPU token became free: hand it to the first waiter (Container FIFO)
    AF_CORE void cpu_release(uint32_t sv) {
        const uint32_t at = L.srv + LSRV * sv;
        const uint64_t q = M.ld(at + 4u);
        if (q_n(q) > 0u) {
            const uint32_t from = L.cq + 2u * (sv * L.fcap + q_head(q));
            grant_a = M.ld(from);
            grant_st = (uint32_t)M.ld(from + 1u);
            fl |= F_GRANT;
            M.st(at + 4u, q_popped(q));
        } else {
            M.st(at, M.ld(at) + 1ull);  // cpu_free += 1
        }
    }

    // ---- ADV stage: the for-loop of _handle_request (server.py:197-276) at step row
    // `step`: a CPU step, an I/O step, or the END row of the endpoint.
    AF_CORE void advance(uint64_t a, uint32_t sv, uint32_t step, uint32_t hops, bool core_locked, bool in_io) {
        const uint32_t at = L.srv + LSRV * sv;
        const uint32_t kind = (uint32_t)P.row[TREC * step + 2u];
        if (kind == STEP_CPU) {  // server.py:199-231
            if (in_io) M.st(at + 1u, M.ld(at + 1u) - 1ull);  // io -= 1
            if (!core_locked) {
                const uint64_t w0 = M.ld(at);  // cpu_free | ready<<32
                const uint64_t q = M.ld(at + 4u);
                const uint32_t cqn = q_n(q);
                if (cqn == 0u && (uint32_t)w0 > 0u) {
                    M.st(at, w0 - 1ull);  // granted at once: not in the ready queue
                } else {
                    if (q_push(L.cq, sv, q_head(q), cqn, a, st_pack(RK_WAIT, sv, hops, 0u, step))) {
                        M.st(at + 4u, q + (1ull << 32));
                        M.st(at, w0 + (1ull << 32));  // ready += 1
                    }
                    return;
                }
            }
            emit(now + row_time(step), a, st_pack(RK_CPU, sv, hops, 0u, step));
            return;
        }
        if (kind == STEP_IO) {  // server.py:235-255
            if (core_locked) cpu_release(sv);  // the waiter's Timeout is created AFTER this one
            if (!in_io) M.st(at + 1u, M.ld(at + 1u) + 1ull);  // io += 1
            emit(now + row_time(step), a, st_pack(RK_IO, sv, hops, 1u, step));
            return;
        }
        // endpoint finished, server.py:257-276: waiter's grant, then transport(), then RAM waiters
        if (core_locked) cpu_release(sv);
        if (in_io) M.st(at + 1u, M.ld(at + 1u) - 1ull);
        const double ram = u2d(P.row[TREC * step + 1u]);
        if (ram > 0.0) {
            M.st(at + 3u, d2u(u2d(M.ld(at + 3u)) - ram));  // ram_in_use
            // Container._do_put (`if capacity - level >= amount`) cannot refuse here: plans whose needs are not multiples
            // of 1/256 MB -- the only ones whose sums round -- run every request event through micro_mode
            // (af_plan_pack.hpp::every_event_in_order), which models the waiting put (m_srv_finish).
            M.st(at + 2u, d2u(u2d(M.ld(at + 2u)) + ram));  // ram container level
            if (q_n(M.ld(at + 5u)) > 0u) fu_ram_sv = (int32_t)sv;
        }
        // with RAM to give back, `yield RAM.put` lets the CPU waiter resume first (server.py:273-276);
        // without, transport() runs in the same step and its Timeout is created first
        fl |= ram > 0.0 ? F_SEND : (F_SEND | F_SEND_FIRST);
        send_a = a;
        send_edge = (uint32_t)(P.srv[SREC * sv + 1u] >> 16) & 0xFFFFu;
        send_hops = hops;
    }

    // ---- RAM stage: Container._trigger_get, FIFO with head-of-line blocking ------
    AF_CORE void ram_stage() {
        const uint32_t sv = (uint32_t)fu_ram_sv;
        const uint32_t at = L.srv + LSRV * sv;
        const uint64_t q = M.ld(at + 5u);
        if (q_n(q) == 0u) {
            fu_ram_sv = -1;
            return;
        }
        const uint32_t from = L.rq + 2u * (sv * L.fcap + q_head(q));
        const uint64_t a = M.ld(from);
        const uint32_t st = (uint32_t)M.ld(from + 1u);
        const uint32_t step = st_step(st);
        const double need = u2d(P.row[TREC * step + 1u]);
        const double free_ram = u2d(M.ld(at + 2u));
        if (free_ram < need) {
            fu_ram_sv = -1;
            return;
        }
        M.st(at + 5u, q_popped(q));
        M.st(at + 2u, d2u(free_ram - need));
        M.st(at + 3u, d2u(u2d(M.ld(at + 3u)) + need));
        fl = (fl & ~(F_ADV_CORE | F_ADV_IO)) | F_ADV;  // keep fu_ram_sv: the queue is looked at again after this waiter
        adv_a = a;
        adv_sv = sv;
        adv_step = step;
        adv_hops = st_hops(st);
    }

    // _dispatcher + head of _handle_request (server.py:303-313, 79-149)
    AF_CORE void server_arrival(uint64_t a, uint32_t sv, uint32_t hops) {
        hops += 1u;  // record_hop(SERVER)
        const uint32_t at = L.srv + LSRV * sv;
        const uint64_t meta = P.srv[SREC * sv + 1u];
        const uint32_t epb = (uint32_t)(meta >> 32) & 0xFFFFu;
        const uint32_t n_ep = (uint32_t)(meta >> 48);
        const uint64_t w1 = M.ld(at + 1u);  // io | arrivals<<32
        M.st(at + 1u, w1 + (1ull << 32));
        const uint32_t pick = n_ep > 1u ? cold_endpoint_pick(seed, sv, (uint32_t)(w1 >> 32), n_ep) : 0u;
        const uint32_t ep = epb + pick;
        const double ram = u2d(P.ep[PREC * ep]);
        const uint32_t step0 = (uint32_t)P.ep[PREC * ep + 1u];
        if (ram > 0.0) {  // server.py:146-149
            const uint64_t q = M.ld(at + 5u);
            if (ram > ram_cap(sv) || (q >> 63)) {
                // can never be served: it (and everything queued behind it) waits
                // forever in the reference, observable nowhere -> dropped here.
                flags |= FLAG_RAM_STARVED;
                M.st(at + 5u, q | (1ull << 63));
                live -= 1u;
                return;
            }
            const double free_ram = u2d(M.ld(at + 2u));
            const uint32_t rqn = q_n(q);
            if (rqn == 0u && free_ram >= ram) {
                M.st(at + 2u, d2u(free_ram - ram));
                M.st(at + 3u, d2u(u2d(M.ld(at + 3u)) + ram));
            } else {
                if (q_push(L.rq, sv, q_head(q), rqn, a, st_pack(RK_WAIT, sv, hops, 0u, step0)))
                    M.st(at + 5u, q + (1ull << 32));
                return;
            }
        }
        fl = (fl & ~(F_ADV_CORE | F_ADV_IO)) | F_ADV;
        adv_a = a;
        adv_sv = sv;
        adv_step = step0;
        adv_hops = hops;
    }

    // ---- load balancer order list (lb_algorithms.py, injection.py:201-226) ----------
    AF_CORE uint32_t lb_get(uint32_t i) const {
        return P.n_lb_edges <= 8u ? (uint32_t)(lb_list >> (8u * i)) & 0xFFu : (uint32_t)M.ld(L.lb + i);
    }
    AF_CORE void lb_set(uint32_t i, uint32_t e) {
        if (P.n_lb_edges <= 8u) {
            lb_list = (lb_list & ~(0xFFull << (8u * i))) | ((uint64_t)e << (8u * i));
        } else {
            M.st(L.lb + i, (uint64_t)e);
        }
    }
    AF_CORE uint32_t lb_pick() {
        const uint32_t lb_n = P.n_srv_marks != 0u ? LaneRegs::lb_n : P.n_lb_edges;  // only outages change the membership
        uint32_t out = lb_get(0u);
        if (P.lb_algo == LB_LEAST_CONNECTIONS) {  // lb_algorithms.py:10-20: first minimum in current order
            uint32_t best = (uint32_t)M.ld(L.edge + LEDGE * out) & 0xFFFFu;
            for (uint32_t i = 1u; i < lb_n; ++i) {
                const uint32_t cand = lb_get(i);
                const uint32_t c = (uint32_t)M.ld(L.edge + LEDGE * cand) & 0xFFFFu;
                if (c < best) {
                    best = c;
                    out = cand;
                }
            }
        } else if (P.n_lb_edges <= 8u) {  // round_robin: first key, move_to_end (lb_algorithms.py:22-36)
            const uint32_t sh = 8u * (lb_n - 1u);  // bytes above the live entries may be stale: mask them
            lb_list = ((lb_list >> 8) & ((1ull << sh) - 1ull)) | ((uint64_t)out << sh);
        } else {
            for (uint32_t i = 1u; i < lb_n; ++i) lb_set(i - 1u, lb_get(i));
            lb_set(lb_n - 1u, out);
        }
        return out;
    }

    // the client's second visit (client.py:62-69): the request is complete
    AF_CORE void complete(uint64_t a) {
        if (O.clock != nullptr) {
            if (n_comp < O.clock_cap) {
                O.clock[2u * n_comp] = u2d(a);
                O.clock[2u * n_comp + 1u] = now;
            } else {
                flags |= FLAG_CLOCK_OVERFLOW;
            }
        }
        if (O.hist != nullptr) {  // same binning as af_summary_t.hist
            const double bf = (now - u2d(a)) * O.hist_scale;
            AF_BUMP(O.hist + (bf >= (double)(O.hist_bins - 1u) ? O.hist_bins - 1u : (uint32_t)bf));
        }
        if (O.rps != nullptr) {  // window (k-1, k], analyzer.py:112-121
            const double kf = __builtin_ceil(now);
            const uint32_t k = kf < 1.0 ? 1u : (uint32_t)kf;
            if (k <= O.rps_buckets) AF_BUMP(O.rps + (k - 1u));
        }
        n_comp += 1u;
        live -= 1u;
    }

    // EdgeRuntime._deliver after the timeout (edge.py:110-116) + the target node
    AF_CORE void deliver(uint64_t a, uint32_t e, uint32_t hops) {
        hops += 1u;  // record_hop(NETWORK_CONNECTION)
        const uint32_t at = L.edge + LEDGE * e;
        M.st(at, M.ld(at) - 1ull);  // conn -= 1
        const uint32_t meta = (uint32_t)P.edge[EREC * e + 3u];
        const uint32_t tk = meta & 0xFFu;
        if (tk == NODE_CLIENT) {  // ClientRuntime._forwarder, client.py:46-71
            hops += 1u;
            if (hops > 3u) {
                complete(a);
            } else {
                fl |= F_SEND;
                send_a = a;
                send_edge = P.client_out_edge;
                send_hops = hops;
            }
        } else if (tk == NODE_LB) {  // LoadBalancerRuntime._forwarder, load_balancer.py:60-72
            fl |= F_SEND;
            send_a = a;
            send_edge = lb_pick();
            send_hops = hops + 1u;
        } else {
            server_arrival(a, (meta >> 8) & 0xFFu, hops);
        }
    }

    // ---- event injection (runtime/events/injection.py:167-226) -------------------
    AF_CORE void apply_emarks() {
        for (;;) {
            const uint32_t i = emark_i++;
            const uint32_t at = L.edge + LEDGE * (uint32_t)emark_w(i, 2u) + 1u;
            M.st(at, d2u(u2d(M.ld(at)) + u2d(emark_w(i, 1u))));
            n_marks += 1u;
            if (emark_i >= P.n_edge_marks || u2d(emark_w(emark_i, 0u)) > now) break;
        }
        t_emark = emark_i < P.n_edge_marks ? u2d(emark_w(emark_i, 0u)) : AF_INF;
        q_emark = seq++;
    }
    AF_CORE void apply_smarks() {
        for (;;) {
            const uint32_t i = smark_i++;
            const uint64_t meta = smark_w(i, 1u);
            const uint32_t e1 = (uint32_t)meta;  // lb edge + 1, 0 = server not behind the LB
            n_marks += 1u;
            if (e1 != 0u) {
                const uint32_t e = e1 - 1u;
                uint32_t pos = 0xFFFFFFFFu;
                for (uint32_t k = 0u; k < lb_n; ++k)
                    if (lb_get(k) == e) pos = k;
                if (pos != 0xFFFFFFFFu) {  // remove (pop() and the re-insert both start by removing)
                    for (uint32_t k = pos + 1u; k < lb_n; ++k) lb_set(k - 1u, lb_get(k));
                    lb_n -= 1u;
                }
                if (!(meta >> 32)) {  // SERVER_UP: append at the end (move_to_end)
                    lb_set(lb_n, e);
                    lb_n += 1u;
                }
            }
            if (smark_i >= P.n_srv_marks || u2d(smark_w(smark_i, 0u)) > now) break;
        }
        t_smark = smark_i < P.n_srv_marks ? u2d(smark_w(smark_i, 0u)) : AF_INF;
        q_smark = seq++;
    }

    // ---- sampler tick (metrics/collector.py:50-66) --------------------------------
    // One tick = one contiguous row [series_pitch] of 4-byte words (series order: edges,
    // then ready / io / ram per server), written with 16-byte stores: a row is 1-2 HBM
    // sectors instead of one partial write per series.
    AF_CORE void sample_tick() {
        if (O.samples != nullptr) {
            if (n_ticks < O.tick_cap) {
                const uint32_t n_series = P.n_edges + 3u * P.n_servers;
                uint32_t* row = O.samples + (size_t)n_ticks * O.series_pitch;
                const bool edges_on = (P.metrics_mask & METRIC_EDGE) != 0u;
                constexpr uint32_t all = METRIC_READY | METRIC_IO | METRIC_RAM;
                const bool servers_on = (P.metrics_mask & all) == all;
                for (uint32_t s0 = 0u; s0 < n_series; s0 += 4u) {
                    uint32_t v[4];
#pragma unroll
                    for (uint32_t k = 0u; k < 4u; ++k) {
                        const uint32_t sidx = s0 + k;
                        uint32_t val = 0u;
                        if (sidx < P.n_edges) {
                            if (edges_on) val = (uint32_t)M.ld(L.edge + LEDGE * sidx) & 0xFFFFu;
                        } else if (sidx < n_series && servers_on) {
                            const uint32_t r = sidx - P.n_edges;
                            const uint32_t sv = r / 3u, which = r - 3u * sv;
                            const uint32_t at = L.srv + LSRV * sv;
                            if (which == 0u) val = (uint32_t)(M.ld(at) >> 32);            // ready_queue_len
                            else if (which == 1u) val = (uint32_t)M.ld(at + 1u);          // event_loop_io_sleep
                            else val = __builtin_bit_cast(uint32_t, (float)u2d(M.ld(at + 3u)));  // ram_in_use
                        }
                        v[k] = val;
                    }
                    store4(row + s0, v[0], v[1], v[2], v[3]);
                }
            } else {
                flags |= FLAG_TICK_OVERFLOW;
            }
        }
        n_ticks += 1u;
    }


    // =====================================================================================
    // Shared timestamps: SimPy's own event-by-event order (cold path, see DESIGN.md "Ties").
    //
    // At an instant `now` SimPy's heap order (time, priority, event id) means: the TIMED events
    // already scheduled for `now` run first, by creation order; every zero-time step they
    // trigger is a NORMAL event appended behind them (FIFO); a process Initialize (edge
    // `_deliver`, server `_handle_request`) is URGENT and runs right after the step that
    // spawned it.  The handlers below are the reference's coroutine fragments between two
    // yields, keyed by the SimPy event whose processing resumes them:
    //   EDGE_TIMEOUT  edge.py:110-116          STORE_PUT / STORE_GET  simpy.Store + the forwarders
    //   STEP_TIMEOUT  server.py:231,255        (client.py:46-71, load_balancer.py:60-72, server.py:303-313)
    //   CPU_GOT       server.py:220-231        CBOX_PUT   client.py:69 (completed_box.put)
    //   CPU_PUT_IO    server.py:241-255        RAM_GOT    server.py:149
    //   CPU_PUT_END   server.py:258-259        RAM_PUT    server.py:273-276
    // A request waiting in a node inbox (simpy.Store) or as a FIFO entry carries its whole state
    // (start time, hops, server, in_io, step row), like a heap entry does.
    enum : uint32_t {
        MK_NONE = 0, MK_EDGE_TIMEOUT, MK_STEP_TIMEOUT, MK_STORE_PUT, MK_STORE_GET, MK_CBOX_PUT,
        MK_RAM_GOT, MK_CPU_GOT, MK_CPU_PUT_IO, MK_CPU_PUT_END, MK_RAM_PUT
    };
    enum : uint32_t { UK_NONE = 0, UK_EDGE_INIT, UK_SRV_INIT };
    uint32_t mq_head, mq_n, bx_n;  // FIFO ring / inbox items in the HBM scratch D.tie
    uint32_t uk, u_idx, u_hops;    // the (single) pending URGENT event
    uint64_t ua;

    AF_CORE void mq_push(uint32_t mk, uint64_t a, uint32_t st, uint32_t node = 0u, uint32_t waiting = 0u) {
        if (mq_n >= L.tcap) {
            flags |= FLAG_POOL_OVERFLOW;
            return;
        }
        const uint32_t at = 2u * ((mq_head + mq_n) & (L.tcap - 1u));
        D.tie[at] = a;
        D.tie[at + 1u] = (uint64_t)st | ((uint64_t)mk << 32) | ((uint64_t)node << 40) | ((uint64_t)waiting << 56);
        mq_n += 1u;
    }
    AF_CORE void bx_push(uint32_t node, uint64_t a, uint32_t st) {  // StorePut: append (unbounded store)
        if (bx_n >= L.tcap) {
            flags |= FLAG_POOL_OVERFLOW;
            return;
        }
        const uint32_t at = 2u * L.tcap + 2u * bx_n;
        D.tie[at] = a;
        D.tie[at + 1u] = (uint64_t)st | ((uint64_t)node << 40);
        bx_n += 1u;
    }
    AF_CORE bool bx_pop(uint32_t node, uint64_t& a, uint32_t& st) {  // first item of this node's inbox
        const uint32_t base = 2u * L.tcap;
        for (uint32_t i = 0u; i < bx_n; ++i) {
            const uint64_t b = D.tie[base + 2u * i + 1u];
            if ((uint32_t)(b >> 40) != node) continue;
            a = D.tie[base + 2u * i];
            st = (uint32_t)b;
            for (uint32_t k = i + 1u; k < bx_n; ++k) {
                D.tie[base + 2u * (k - 1u)] = D.tie[base + 2u * k];
                D.tie[base + 2u * (k - 1u) + 1u] = D.tie[base + 2u * k + 1u];
            }
            bx_n -= 1u;
            return true;
        }
        return false;
    }
    // forwarder of a node is between `box.get()` succeeding and its next `box.get()`
    AF_CORE bool busy(uint32_t node) const { return (D.tie[6u * L.tcap + (node >> 6)] >> (node & 63u)) & 1ull; }
    AF_CORE void set_busy(uint32_t node, bool v) {
        uint64_t& w = D.tie[6u * L.tcap + (node >> 6)];
        w = v ? (w | (1ull << (node & 63u))) : (w & ~(1ull << (node & 63u)));
    }
    AF_CORE void m_urgent(uint32_t k, uint64_t a, uint32_t idx, uint32_t hops) {
        uk = k;
        ua = a;
        u_idx = idx;
        u_hops = hops;
    }
    // a Timeout: into the heap, or -- zero delay -- behind the steps queued at this instant
    AF_CORE void m_emit(double t, uint64_t a, uint32_t state, uint32_t mk) {
        const uint32_t sq = seq++;
        if (t == now) {
            mq_push(mk, a, state);
            return;
        }
        if (heap_n >= L.cap) {
            flags |= FLAG_POOL_OVERFLOW;
            live -= 1u;
            return;
        }
        sift_up(heap_n++, t, a, ((uint64_t)sq << 32) | state);
    }
    AF_CORE void m_heap_pop() {
        heap_n -= 1u;
        if (heap_n > 0u) {
            const double key = u2d(M.ld(L.hk + heap_n));
            const uint64_t a = M.ld(L.ha + heap_n);
            const uint64_t b = M.ld(L.hb + heap_n);
            sift_down(0u, key, a, b, heap_n);
        }
    }
    // StoreGet.__init__ -> _trigger_get: the forwarder asks for its next message
    AF_CORE void m_forwarder_get(uint32_t node) {
        uint64_t a;
        uint32_t st;
        if (bx_pop(node, a, st)) {
            set_busy(node, true);
            mq_push(MK_STORE_GET, a, st, node);
        } else {
            set_busy(node, false);
        }
    }
    // Container._trigger_get of the CPU container; entries beyond `n_old` were appended by the
    // caller's own get() and have not been counted in the ready queue.  Returns #grants.
    AF_CORE uint32_t m_cpu_trigger(uint32_t sv, uint32_t n_old) {
        const uint32_t at = L.srv + LSRV * sv;
        uint64_t w0 = M.ld(at);
        uint64_t q = M.ld(at + 4u);
        uint32_t g = 0u;
        while (q_n(q) > 0u && (uint32_t)w0 > 0u) {
            const uint32_t from = L.cq + 2u * (sv * L.fcap + q_head(q));
            mq_push(MK_CPU_GOT, M.ld(from), (uint32_t)M.ld(from + 1u), 0u, g < n_old ? 1u : 0u);
            q = q_popped(q);
            w0 -= 1ull;  // level -= 1
            g += 1u;
        }
        M.st(at, w0);
        M.st(at + 4u, q);
        return g;
    }
    AF_CORE void m_ram_trigger(uint32_t sv) {  // head-of-line blocking FIFO
        const uint32_t at = L.srv + LSRV * sv;
        for (;;) {
            const uint64_t q = M.ld(at + 5u);
            if (q_n(q) == 0u) break;
            const uint32_t from = L.rq + 2u * (sv * L.fcap + q_head(q));
            const uint64_t a = M.ld(from);
            const uint32_t st = (uint32_t)M.ld(from + 1u);
            const double need = u2d(P.row[TREC * st_step(st) + 1u]);
            const double free_ram = u2d(M.ld(at + 2u));
            if (free_ram < need) break;
            M.st(at + 5u, q_popped(q));
            M.st(at + 2u, d2u(free_ram - need));
            mq_push(MK_RAM_GOT, a, st);
        }
    }
    // ---- the RAM container's PUT queue (simpy BaseResource.put_queue; only ever non-empty with fractional needs) ----
    // `Container._do_put` succeeds `if capacity - level >= amount`.  In exact arithmetic capacity - level is the sum of what is
    // held and the test cannot fail; with a need like 100.3 MB it fails by ONE ROUNDING (2048 - fl(2048 - 100.3) < 100.3) and
    // `yield RAM.put(total_ram)` (server.py:273) WAITS: the put stays in the queue, every later put of that server queues
    // behind it (`_trigger_put` walks from the head and stops at the first refusal), and the queue is walked again by every
    // new put and when a RAM get of that server is PROCESSED (`_trigger_put` is the get's first callback) -- the level that
    // get took lets the put through, and the response leaves then.  Entries: (start time, state) of the request, all servers
    // in ONE list in arrival order (a server's subsequence is its FIFO), in the scenario's HBM scratch.
    AF_CORE bool pb_has(uint32_t sv) const {
        for (uint32_t i = 0u; i < pb_n; ++i)
            if (st_idx((uint32_t)D.tie[4u * L.tcap + 2u * i + 1u]) == sv) return true;
        return false;
    }
    // BaseResource._trigger_put of server sv's RAM container
    AF_CORE void m_put_trigger(uint32_t sv) {
        if (pb_n == 0u) return;
        const uint32_t at = L.srv + LSRV * sv;
        const uint32_t base = 4u * L.tcap;
        uint32_t i = 0u;
        while (i < pb_n) {
            const uint32_t st = (uint32_t)D.tie[base + 2u * i + 1u];
            if (st_idx(st) != sv) {
                i += 1u;
                continue;
            }
            const double amount = u2d(P.row[TREC * st_step(st) + 1u]);
            const double level = u2d(M.ld(at + 2u));
            if (!(ram_cap(sv) - level >= amount)) break;  // `if not proceed: break` (head-of-line)
            M.st(at + 2u, d2u(level + amount));
            mq_push(MK_RAM_PUT, D.tie[base + 2u * i], st);  // event.succeed()
            for (uint32_t k = i + 1u; k < pb_n; ++k) {
                D.tie[base + 2u * (k - 1u)] = D.tie[base + 2u * k];
                D.tie[base + 2u * (k - 1u) + 1u] = D.tie[base + 2u * k + 1u];
            }
            pb_n -= 1u;
        }
    }
    // A refused put at the head of the put queue AND a waiter at the head of the get queue that does not fit: the level can
    // never change again (puts and gets are both head-of-line blocked, and nothing else moves it), so every later RAM request
    // of this server waits for good, like behind a starved one -- same bit, same report; they need not be kept.
    AF_CORE void m_ram_deadlock_check(uint32_t sv) {
        const uint32_t at = L.srv + LSRV * sv;
        const uint64_t q = M.ld(at + 5u);
        if ((q >> 63) || q_n(q) == 0u || pb_n == 0u) return;
        const double level = u2d(M.ld(at + 2u));
        const uint32_t hst = (uint32_t)M.ld(L.rq + 2u * (sv * L.fcap + q_head(q)) + 1u);
        if (level >= u2d(P.row[TREC * st_step(hst) + 1u])) return;   // the get at the head fits (it is served when a put is processed)
        for (uint32_t i = 0u; i < pb_n; ++i) {
            const uint32_t st = (uint32_t)D.tie[4u * L.tcap + 2u * i + 1u];
            if (st_idx(st) != sv) continue;
            if (ram_cap(sv) - level >= u2d(P.row[TREC * st_step(st) + 1u])) return;   // the put at the head goes through at the next walk
            flags |= FLAG_RAM_STARVED;
            M.st(at + 5u, q | (1ull << 63));
            return;
        }
    }
    // tail of _handle_request once the core is back (server.py:261-276)
    AF_CORE void m_srv_finish(uint64_t a, uint32_t sv, uint32_t step, uint32_t hops, bool in_io) {
        const uint32_t at = L.srv + LSRV * sv;
        if (in_io) M.st(at + 1u, M.ld(at + 1u) - 1ull);
        const double ram = u2d(P.row[TREC * step + 1u]);
        if (ram > 0.0) {
            M.st(at + 3u, d2u(u2d(M.ld(at + 3u)) - ram));
            // ContainerPut.__init__: append to the put queue, _trigger_put(None)
            const double level = u2d(M.ld(at + 2u));
            if ((pb_n != 0u && pb_has(sv)) || !(ram_cap(sv) - level >= ram)) {
                if (pb_n >= L.tcap) {
                    flags |= FLAG_POOL_OVERFLOW;
                    return;
                }
                D.tie[4u * L.tcap + 2u * pb_n] = a;
                D.tie[4u * L.tcap + 2u * pb_n + 1u] = (uint64_t)st_pack(RK_WAIT, sv, hops, 0u, step);
                pb_n += 1u;
                m_put_trigger(sv);   // (the walk a new put starts: the head may fit by now)
                m_ram_deadlock_check(sv);
                return;
            }
            M.st(at + 2u, d2u(level + ram));
            mq_push(MK_RAM_PUT, a, st_pack(RK_WAIT, sv, hops, 0u, step));
            return;
        }
        m_urgent(UK_EDGE_INIT, a, (uint32_t)(P.srv[SREC * sv + 1u] >> 16) & 0xFFFFu, hops);
    }
    // the for-loop of _handle_request from step row `step` up to its next yield (server.py:197-259)
    AF_CORE void m_srv_continue(uint64_t a, uint32_t sv, uint32_t step, uint32_t hops, bool core_locked, bool in_io) {
        const uint32_t at = L.srv + LSRV * sv;
        const uint32_t kind = (uint32_t)P.row[TREC * step + 2u];
        if (kind == STEP_CPU) {
            if (in_io) M.st(at + 1u, M.ld(at + 1u) - 1ull);
            if (!core_locked) {  // cpu_req = CPU.get(1): append, trigger, `if not cpu_req.triggered`
                const uint64_t q = M.ld(at + 4u);
                const uint32_t cqn = q_n(q);
                if (!q_push(L.cq, sv, q_head(q), cqn, a, st_pack(RK_WAIT, sv, hops, 0u, step))) return;
                M.st(at + 4u, q + (1ull << 32));
                if (m_cpu_trigger(sv, cqn) <= cqn) M.st(at, M.ld(at) + (1ull << 32));  // still queued: ready += 1
                return;
            }
            m_emit(now + row_time(step), a, st_pack(RK_CPU, sv, hops, 0u, step), MK_STEP_TIMEOUT);
            return;
        }
        if (kind == STEP_IO) {
            if (core_locked) {  // yield CPU.put(1)
                M.st(at, M.ld(at) + 1ull);
                mq_push(MK_CPU_PUT_IO, a, st_pack(RK_WAIT, sv, hops, in_io ? 1u : 0u, step));
                return;
            }
            if (!in_io) M.st(at + 1u, M.ld(at + 1u) + 1ull);
            m_emit(now + row_time(step), a, st_pack(RK_IO, sv, hops, 1u, step), MK_STEP_TIMEOUT);
            return;
        }
        if (core_locked) {  // endpoint finished holding the core: yield CPU.put(1)
            M.st(at, M.ld(at) + 1ull);
            mq_push(MK_CPU_PUT_END, a, st_pack(RK_WAIT, sv, hops, in_io ? 1u : 0u, step));
            return;
        }
        m_srv_finish(a, sv, step, hops, in_io);
    }
    // head of _handle_request, run by its Initialize event (server.py:79-149)
    AF_CORE void m_srv_init(uint64_t a, uint32_t sv, uint32_t hops) {
        hops += 1u;
        const uint32_t at = L.srv + LSRV * sv;
        const uint64_t meta = P.srv[SREC * sv + 1u];
        const uint32_t epb = (uint32_t)(meta >> 32) & 0xFFFFu;
        const uint32_t n_ep = (uint32_t)(meta >> 48);
        const uint64_t w1 = M.ld(at + 1u);
        M.st(at + 1u, w1 + (1ull << 32));
        const uint32_t pick = n_ep > 1u ? cold_endpoint_pick(seed, sv, (uint32_t)(w1 >> 32), n_ep) : 0u;
        const uint32_t ep = epb + pick;
        const double ram = u2d(P.ep[PREC * ep]);
        const uint32_t step0 = (uint32_t)P.ep[PREC * ep + 1u];
        if (ram > 0.0) {
            const uint64_t q = M.ld(at + 5u);
            if (ram > ram_cap(sv) || (q >> 63)) {  // same treatment as server_arrival
                flags |= FLAG_RAM_STARVED;
                M.st(at + 5u, q | (1ull << 63));
                live -= 1u;
                return;
            }
            if (q_push(L.rq, sv, q_head(q), q_n(q), a, st_pack(RK_WAIT, sv, hops, 0u, step0))) {
                M.st(at + 5u, q + (1ull << 32));
                m_ram_trigger(sv);
                m_ram_deadlock_check(sv);
            }
            return;  // yield RAM.get(total_ram)
        }
        m_srv_continue(a, sv, step0, hops, false, false);
    }

    AF_CORE void micro_mode() {
        mq_head = mq_n = bx_n = 0u;
        uk = UK_NONE;
        for (;;) {
            // next event of this instant: timed ones by creation order, then the FIFO
            uint32_t cls = 5u, best = 0xFFFFFFFFu;
            if (heap_n > 0u && u2d(M.ld(L.hk)) == now) {
                best = (uint32_t)(M.ld(L.hb) >> 32);
                cls = 4u;
            }
            if (t_gen == now && q_gen < best) { best = q_gen; cls = 3u; }
            if (t_tick == now && q_tick < best) { best = q_tick; cls = 2u; }
            if (t_smark == now && q_smark < best) { best = q_smark; cls = 1u; }
            if (t_emark == now && q_emark < best) { best = q_emark; cls = 0u; }
            uint64_t ra = 0ull;
            uint32_t rst = 0u, mk = MK_NONE, node = 0u, waiting = 0u;
            if (cls == 4u) {
                ra = M.ld(L.ha);
                rst = (uint32_t)M.ld(L.hb);
                m_heap_pop();
                mk = st_kind(rst) == RK_TRANSIT ? MK_EDGE_TIMEOUT : MK_STEP_TIMEOUT;
            } else if (cls == 3u) {  // rqs_generator.py:101-119
                n_gen += 1u;
                n_events += 1u;
                live += 1u;
                if (live > max_live) max_live = live;
                m_urgent(UK_EDGE_INIT, d2u(now), P.gen_out_edge, 1u);
                if (arr_ahead > 0u) {
                    arr_ahead -= 1u;
                    t_gen = u2d(M.ld(L.ring + (n_gen & (RING - 1u))));
                } else {
                    t_gen = draw_or(0u, n_gen, AF_INF);
                }
                q_gen = seq++;
                if (arr_ahead <= RING_LOW) fl |= F_DIRTY_ARR;
            } else if (cls == 2u) {
                sample_tick();
                t_tick = now + P.sample_period;
                q_tick = seq++;
            } else if (cls == 1u) {
                apply_smarks();
            } else if (cls == 0u) {
                apply_emarks();
            } else {
                if (mq_n == 0u) break;
                const uint32_t at = 2u * mq_head;
                ra = D.tie[at];
                const uint64_t b = D.tie[at + 1u];
                mq_head = (mq_head + 1u) & (L.tcap - 1u);
                mq_n -= 1u;
                rst = (uint32_t)b;
                mk = (uint32_t)(b >> 32) & 0xFFu;
                node = (uint32_t)(b >> 40) & 0xFFFFu;
                waiting = (uint32_t)(b >> 56) & 1u;
            }
            const uint32_t sv = st_idx(rst);
            const uint32_t hops = st_hops(rst);
            const uint32_t step = st_step(rst);
            switch (mk) {
                case MK_EDGE_TIMEOUT: {  // edge.py:110-116, then StorePut on the target's inbox
                    n_events += 1u;
                    const uint32_t e = st_idx(rst);
                    const uint32_t at = L.edge + LEDGE * e;
                    M.st(at, M.ld(at) - 1ull);
                    const uint32_t meta = (uint32_t)P.edge[EREC * e + 3u];
                    const uint32_t tk = meta & 0xFFu;
                    const uint32_t target = tk == NODE_CLIENT ? 0u : tk == NODE_LB ? 1u : 2u + ((meta >> 8) & 0xFFu);
                    bx_push(target, ra, st_pack(RK_TRANSIT, e, hops + 1u, 0u, 0u));
                    mq_push(MK_STORE_PUT, 0ull, 0u, target);
                    break;
                }
                case MK_STORE_PUT: {  // the put is processed: Store._trigger_get is its first callback
                    uint64_t a;
                    uint32_t st;
                    if (!busy(node) && bx_pop(node, a, st)) {
                        set_busy(node, true);
                        mq_push(MK_STORE_GET, a, st, node);
                    }
                    break;
                }
                case MK_STORE_GET: {  // a forwarder resumes with a message
                    if (node == 0u) {  // client.py:46-71
                        const uint32_t h = hops + 1u;
                        if (h > 3u) {
                            complete(ra);
                            mq_push(MK_CBOX_PUT, 0ull, 0u);  // yield completed_box.put(state)
                        } else {
                            m_urgent(UK_EDGE_INIT, ra, P.client_out_edge, h);
                            m_forwarder_get(0u);
                        }
                    } else if (node == 1u) {  // load_balancer.py:60-72
                        m_urgent(UK_EDGE_INIT, ra, lb_pick(), hops + 1u);
                        m_forwarder_get(1u);
                    } else {  // server.py:303-313
                        m_urgent(UK_SRV_INIT, ra, node - 2u, hops);
                        m_forwarder_get(node);
                    }
                    break;
                }
                case MK_CBOX_PUT: m_forwarder_get(0u); break;
                case MK_RAM_GOT: {  // server.py:149-150; the get's first callback is the container's _trigger_put
                    m_put_trigger(sv);
                    const uint32_t at = L.srv + LSRV * sv;
                    M.st(at + 3u, d2u(u2d(M.ld(at + 3u)) + u2d(P.row[TREC * step + 1u])));
                    m_srv_continue(ra, sv, step, hops, false, false);
                    break;
                }
                case MK_CPU_GOT: {  // server.py:220-231
                    if (waiting) {
                        const uint32_t at = L.srv + LSRV * sv;
                        M.st(at, M.ld(at) - (1ull << 32));
                    }
                    m_emit(now + row_time(step), ra, st_pack(RK_CPU, sv, hops, 0u, step), MK_STEP_TIMEOUT);
                    break;
                }
                case MK_STEP_TIMEOUT:  // the for-loop moves on to the next step row
                    n_events += 1u;
                    m_srv_continue(ra, sv, step + 1u, hops, st_kind(rst) == RK_CPU, st_io(rst) != 0u);
                    break;
                case MK_CPU_PUT_IO: {  // server.py:241-255: put processed -> grants, then the process resumes
                    m_cpu_trigger(sv, 0xFFFFFFFFu);
                    const uint32_t at = L.srv + LSRV * sv;
                    if (!st_io(rst)) M.st(at + 1u, M.ld(at + 1u) + 1ull);
                    m_emit(now + row_time(step), ra, st_pack(RK_IO, sv, hops, 1u, step), MK_STEP_TIMEOUT);
                    break;
                }
                case MK_CPU_PUT_END:  // server.py:258-259
                    m_cpu_trigger(sv, 0xFFFFFFFFu);
                    m_srv_finish(ra, sv, step, hops, st_io(rst) != 0u);
                    break;
                case MK_RAM_PUT:  // server.py:273-276
                    m_ram_trigger(sv);
                    m_urgent(UK_EDGE_INIT, ra, (uint32_t)(P.srv[SREC * sv + 1u] >> 16) & 0xFFFFu, hops);
                    break;
                default: break;
            }
            while (uk != UK_NONE) {  // process Initialize events are URGENT
                const uint32_t k = uk;
                uk = UK_NONE;
                if (k == UK_EDGE_INIT) edge_send<true>(ua, u_idx, u_hops);
                else m_srv_init(ua, u_idx, u_hops);
            }
        }
    }

    // ---- life cycle -----------------------------------------------------------------
    // `ovr` : per-lane reader of the override columns, ovr(k) -> double, k-th column;
    // ovr_index of a STEP_TIME column is a step ROW.
    template <class OvrFn>
    AF_CORE void init(const uint32_t* ovr_param, const uint32_t* ovr_index, uint32_t n_ovr, OvrFn ovr) {
        now = 0.0;
        heap_n = live = max_live = emark_i = smark_i = 0u;
        n_gen = n_comp = n_drop = n_events = n_ticks = n_marks = 0u;
        // the Initialize events run in process start order (simulation_runner.py:364-366):
        // edge timeline, server timeline, generator, ..., collector
        q_emark = 0u;
        q_smark = 1u;
        q_gen = 2u;
        q_tick = 3u;
        seq = 4u;
        for (uint32_t i = 0u; i < (2u + P.n_servers + 63u) / 64u; ++i) D.tie[6u * L.tcap + i] = 0ull;
        pb_n = 0u;
        flags = D.flags_in;
        fl = 0u;
        pend_count = 0u;
        fu_ram_sv = -1;
        for (uint32_t e = 0u; e < P.n_edges; ++e) {
            M.st(L.edge + LEDGE * e, 0ull);
            M.st(L.edge + LEDGE * e + 1u, d2u(0.0));
        }
        if (L.ovr_mask & (1u << PARAM_STEP_TIME))
            for (uint32_t r = 0u; r < P.n_rows; ++r) M.st(L.stime + r, P.row[TREC * r]);
        for (uint32_t v = 0u; v < P.n_servers; ++v) {  // build_containers: init full
            const uint32_t at = L.srv + LSRV * v;
            M.st(at, P.srv[SREC * v + 1u] & 0xFFFFull);  // cpu_free = cores, ready = 0
            M.st(at + 1u, 0ull);
            M.st(at + 2u, P.srv[SREC * v]);  // RAM level = ram_mb
            M.st(at + 3u, d2u(0.0));
            M.st(at + 4u, 0ull);
            M.st(at + 5u, 0ull);
        }
        lb_n = P.n_lb_edges;
        lb_list = 0ull;
        for (uint32_t i = 0u; i < lb_n; ++i) lb_set(i, (uint32_t)P.lb[i]);
        if (L.ovr_mask & (1u << PARAM_SRV_RAM_MB))
            for (uint32_t v = 0u; v < P.n_servers; ++v) M.st(L.srvram + v, P.srv[SREC * v]);
        if (L.ovr_mask & kParamMarkBits) {
            for (uint32_t i = 0u; i < MREC * P.n_edge_marks; ++i) M.st(L.marks + i, P.emark[i]);
            for (uint32_t i = 0u; i < NREC * P.n_srv_marks; ++i) M.st(L.marks + MREC * P.n_edge_marks + i, P.smark[i]);
        }
        for (uint32_t k = 0u; k < n_ovr; ++k) {
            const double v = ovr(k);
            const uint32_t idx = ovr_index[k], prm = ovr_param[k];
            if (prm == PARAM_STEP_TIME) M.st(L.stime + idx, d2u(v));
            else if (prm == PARAM_SRV_CORES) M.st(L.srv + LSRV * idx, (uint64_t)(uint32_t)v);            // cpu_free = cores, ready = 0
            else if (prm == PARAM_SRV_RAM_MB) { M.st(L.srvram + idx, d2u(v)); M.st(L.srv + LSRV * idx + 2u, d2u(v)); }   // capacity, level
            else if (prm == PARAM_EMARK_TIME) M.st(L.marks + MREC * idx, d2u(v));
            else if (prm == PARAM_EMARK_DELTA) M.st(L.marks + MREC * idx + 1u, d2u(v));
            else if (prm == PARAM_EMARK_EDGE) M.st(L.marks + MREC * idx + 2u, (uint64_t)(uint32_t)v);
            else if (prm == PARAM_SMARK_TIME) M.st(L.marks + MREC * P.n_edge_marks + NREC * idx, d2u(v));
            else if (prm == PARAM_SMARK_LB_EDGE) {   // value = LB out-edge index, -1 = the server is not behind the LB
                const uint32_t at = L.marks + MREC * P.n_edge_marks + NREC * idx + 1u;
                M.st(at, (M.ld(at) & ~0xFFFFFFFFull) | (uint64_t)(uint32_t)((int32_t)v + 1));
            } else if (prm == PARAM_SMARK_DOWN) {
                const uint32_t at = L.marks + MREC * P.n_edge_marks + NREC * idx + 1u;
                M.st(at, (M.ld(at) & 0xFFFFFFFFull) | ((uint64_t)(v != 0.0 ? 1u : 0u) << 32));
            }   // (the others only shape the draws)
        }
        dirty_edge = -1;
        {   // blocking first fill of every ring (once per scenario)
            const TopUp t = topup_begin(true, 0u, 0u);
            topup_end_arrivals(t);
            t_gen = u2d(M.ld(L.ring));
            for (uint32_t e = 0u; e < P.n_edges; ++e) topup_end_edge(topup_begin(true, 1u + e, 0u));
        }
        t_tick = 0.0 + P.sample_period;
        t_emark = P.n_edge_marks ? u2d(emark_w(0u, 0u)) : AF_INF;
        t_smark = P.n_srv_marks ? u2d(smark_w(0u, 0u)) : AF_INF;
    }

    // One next-event round.  Returns false once the scenario reached the horizon.
    //
    // Structure (every heavy block appears ONCE so the code stays I-cache sized and
    // lanes re-converge at each stage):
    //   select -> decode (light, per event kind) -> [ADV -> GRANT -> SEND -> RAM]* -> heap commit
    // The stage order is the order SimPy creates the corresponding Timeouts in
    // (server.py:235-276): own I/O timer before the CPU waiter's; on endpoint end
    // the CPU waiter's timer, then transport(), then the RAM waiters.
    // Returns ROUND_STOP once the scenario reached the horizon, ROUND_SHARED when the next instant is
    // shared by several timed events (nothing has been processed; the caller runs shared_instant()
    // OUTSIDE its hot loop, so that the cold path's registers do not weigh on the loop).
    // (the lean variant keeps a plain bool: a lane mask in scalar registers instead of a vector value)
    using RoundT = typename RoundType<kFaithful>::type;
    static constexpr RoundT ROUND_STOP = RoundT(0), ROUND_MORE = RoundT(1), ROUND_SHARED = RoundT(kFaithful ? 2 : 1);
    AF_CORE RoundT round() {
        // top up the rings consumed from in the previous round: loads are issued now and
        // stored at the end of this round, so their HBM latency hides behind the round
        const bool want_arr = (fl & F_DIRTY_ARR) != 0u;
        const bool want_edge = dirty_edge >= 0;
        const uint32_t tu_e = want_edge ? (uint32_t)dirty_edge : 0u;
        fl &= ~F_DIRTY_ARR;
        dirty_edge = -1;
        const TopUp tu_arr = topup_begin(want_arr, 0u, n_gen);
        const TopUp tu_edge =
            topup_begin(want_edge, 1u + tu_e, want_edge ? (uint32_t)(M.ld(L.edge + LEDGE * tu_e) >> 32) : 0u);
        // the root entry is fetched in full up front: three independent LDS reads
        const double t_heap = heap_n > 0u ? u2d(M.ld(L.hk)) : AF_INF;
        const uint64_t root_a = M.ld(L.ha);
        const uint32_t root_st = (uint32_t)M.ld(L.hb);
        // the root's children: an entry sharing the root's timestamp is always reachable through
        // equal keys, so the two children tell whether the instant is shared inside the heap
        // (prefetching them in full for the sift-down that follows was measured: no gain, the
        // kernel is issue bound at 2-3 waves per SIMD)
        const double t_c1 = u2d(M.ld(L.hk + 1u));
        const double t_c2 = u2d(M.ld(L.hk + 2u));
        // next event among {heap, arrival, tick, server marks, edge marks}; on equal
        // times the LATER test wins: edge marks < server marks < tick < arrival < heap.
        uint32_t cls = 4u;
        double t = t_heap;
        if (t_gen <= t) { cls = 3u; t = t_gen; }
        if (t_tick <= t) { cls = 2u; t = t_tick; }
        // (plans without timelines -- a compile-time fact in plan-specialised builds -- skip all of it)
        if (P.n_srv_marks != 0u && t_smark <= t) { cls = 1u; t = t_smark; }
        if (P.n_edge_marks != 0u && t_emark <= t) { cls = 0u; t = t_emark; }
        if (!(t < P.total_time)) return ROUND_STOP;  // the stop event is URGENT at T (pending top-ups are moot)
        now = t;
        // Two or more timed events at this instant: SimPy interleaves their zero-time steps.  On equal
        // times the selection above prefers the timers, so: a request event (cls 4) shares its instant
        // iff a child of the root has the same key; an arrival (cls 3) iff the heap root has; a tick or
        // timeline mark needs care iff a request event or an arrival shares its instant (ticks and
        // marks commute with each other, the fixed class order above is as good as any).
        // (plans with P.every_event_in_order are never given to the lean variant: engine.hip)
        const bool shared = cls == 4u ? ((heap_n > 1u && t_c1 == t) || (heap_n > 2u && t_c2 == t) ||
                                         (kFaithful && P.every_event_in_order != 0u))
                                      : (t_heap == t || (cls < 3u && t_gen == t));
        if (__builtin_expect(shared, 0)) {
            if constexpr (kFaithful) {
                // the top-ups this round had started are simply asked for again in the next one
                if (want_arr) fl |= F_DIRTY_ARR;
                if (want_edge) dirty_edge = (int32_t)tu_e;
                return ROUND_SHARED;
            } else {
                // Simulated again, from t = 0, by the variant that has the SimPy-order path.  (Handing
                // the state over instead was measured: keeping it restorable costs the lean kernel
                // 3.5 % on every sweep, and an engine that saw one such scenario starts its later
                // sweeps with the other variant anyway.)
                flags |= FLAG_SHARED_INSTANT;
                return ROUND_STOP;
            }
        }

        if (cls == 4u) {
            fl |= F_HOLE;
            n_events += 1u;
            const uint32_t kind = st_kind(root_st);
            if (kind == RK_TRANSIT) {
                deliver(root_a, st_idx(root_st), st_hops(root_st));
            } else {  // CPU or I/O step finished: the for-loop moves to the next step row
                fl |= F_ADV | (kind == RK_CPU ? F_ADV_CORE : 0u) | (st_io(root_st) ? F_ADV_IO : 0u);
                adv_a = root_a;
                adv_sv = st_idx(root_st);
                adv_step = st_step(root_st) + 1u;
                adv_hops = st_hops(root_st);
            }
        } else if (cls == 3u) {  // RqsGeneratorRuntime._event_arrival (rqs_generator.py:101-119)
            n_gen += 1u;
            n_events += 1u;
            live += 1u;
            if (live > max_live) max_live = live;
            q_gen = seq++;  // next(time_gaps) + the new Timeout precede the edge process (rqs_generator.py:104-119)
            fl |= F_SEND;
            send_a = d2u(now);
            send_edge = P.gen_out_edge;
            send_hops = 1u;  // record_hop(GENERATOR)
            // next arrival time: pre-generated (the sampler never looks at the simulation state)
            if (arr_ahead > 0u) {
                arr_ahead -= 1u;
                t_gen = u2d(M.ld(L.ring + (n_gen & (RING - 1u))));
            } else {
                t_gen = draw_or(0u, n_gen, AF_INF);  // ring empty: straight from HBM (rare)
            }
            if (arr_ahead <= RING_LOW) fl |= F_DIRTY_ARR;
        } else if (cls == 2u) {
            sample_tick();
            t_tick = now + P.sample_period;
            q_tick = seq++;
        } else if (P.n_srv_marks != 0u && cls == 1u) {
            apply_smarks();
        } else if (P.n_edge_marks != 0u) {
            apply_emarks();
        }

        for (;;) {
            // each stage pushes at most one event; the buffer holds two, so a pass
            // stops early (stage order preserved) when it is full -- rare.
            if (fl & F_ADV) {
                const uint32_t f = fl;
                fl &= ~(F_ADV | F_ADV_CORE | F_ADV_IO);
                advance(adv_a, adv_sv, adv_step, adv_hops, (f & F_ADV_CORE) != 0u, (f & F_ADV_IO) != 0u);
            }
            if ((fl & (F_GRANT | F_SEND_FIRST)) == F_GRANT && pend_count < 2u) {
                fl &= ~F_GRANT;
                cpu_granted();
            }
            if ((fl & F_SEND) && (fl & (F_GRANT | F_SEND_FIRST)) != F_GRANT && pend_count < 2u) {
                fl &= ~(F_SEND | F_SEND_FIRST);
                edge_send(send_a, send_edge, send_hops);
            }
            if (fu_ram_sv >= 0 && !(fl & (F_GRANT | F_SEND))) ram_stage();
            const bool more = (fl & (F_ADV | F_GRANT | F_SEND)) != 0u || fu_ram_sv >= 0;
            heap_commit(!more);
            if (!more) break;
        }
        topup_end_arrivals(tu_arr);
        topup_end_edge(tu_edge);
        return ROUND_MORE;
    }
    // the instant `now` that round() reported as shared, in SimPy's event order
    AF_CORE void shared_instant() { micro_mode(); }

    // All scalars of the lane through memory (volatile: the compiler must not forward the stored
    // values to the loads).  The kernel does this around shared_instant() for every lane of the
    // wave, so that NO value is live across the cold path and the hot loop keeps its own register
    // allocation (engine.hip: run_lanes).
    enum : uint32_t { PARK_WORDS = 17 };
    AF_CORE void park_regs(volatile uint64_t* dst) const {
        dst[0] = d2u(now); dst[1] = d2u(t_gen); dst[2] = d2u(t_tick); dst[3] = d2u(t_emark); dst[4] = d2u(t_smark);
        dst[5] = lb_list;
        dst[6] = q_gen | ((uint64_t)q_tick << 32);
        dst[7] = q_emark | ((uint64_t)q_smark << 32);
        dst[8] = arr_ahead | ((uint64_t)(uint32_t)dirty_edge << 32);
        dst[9] = fl | ((uint64_t)heap_n << 32);
        dst[10] = seq | ((uint64_t)live << 32);
        dst[11] = max_live | ((uint64_t)lb_n << 32);
        dst[12] = emark_i | ((uint64_t)smark_i << 32);
        dst[13] = n_gen | ((uint64_t)n_comp << 32);
        dst[14] = n_drop | ((uint64_t)n_events << 32);
        dst[15] = n_ticks | ((uint64_t)n_marks << 32);
        dst[16] = flags | ((uint64_t)pb_n << 32);
    }
    AF_CORE void unpark_regs(const volatile uint64_t* src) {
        now = u2d(src[0]); t_gen = u2d(src[1]); t_tick = u2d(src[2]); t_emark = u2d(src[3]); t_smark = u2d(src[4]);
        lb_list = src[5];
        const uint64_t w6 = src[6], w7 = src[7], w8 = src[8], w9 = src[9], w10 = src[10], w11 = src[11], w12 = src[12],
                       w13 = src[13], w14 = src[14], w15 = src[15];
        q_gen = (uint32_t)w6; q_tick = (uint32_t)(w6 >> 32);
        q_emark = (uint32_t)w7; q_smark = (uint32_t)(w7 >> 32);
        arr_ahead = (uint32_t)w8; dirty_edge = (int32_t)(uint32_t)(w8 >> 32);
        fl = (uint32_t)w9; heap_n = (uint32_t)(w9 >> 32);
        seq = (uint32_t)w10; live = (uint32_t)(w10 >> 32);
        max_live = (uint32_t)w11; lb_n = (uint32_t)(w11 >> 32);
        emark_i = (uint32_t)w12; smark_i = (uint32_t)(w12 >> 32);
        n_gen = (uint32_t)w13; n_comp = (uint32_t)(w13 >> 32);
        n_drop = (uint32_t)w14; n_events = (uint32_t)(w14 >> 32);
        n_ticks = (uint32_t)w15; n_marks = (uint32_t)(w15 >> 32);
        const uint64_t w16 = src[16];
        flags = (uint32_t)w16; pb_n = (uint32_t)(w16 >> 32);
        pend_count = 0u;
        fu_ram_sv = -1;
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
