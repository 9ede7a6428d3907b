/* des_oracle_atomic.c -- ORACLE / TEST INFRASTRUCTURE ONLY.
 *
 * Second CPU restatement of the same hot path, stating the HIP ENGINE's
 * execution semantics ("atomic cascades"):
 *
 *   - only TIMED events (SimPy Timeouts) are queued, ordered by (time, push seq);
 *   - every zero-time SimPy step that follows a timed event (Initialize, Store
 *     put/get, Container put/get) is executed inline, in the order SimPy runs
 *     them when no other timed event shares the timestamp.
 *
 * Relationship (tests/test_oracle_golden.py, tests/test_reference_live.py):
 *   reference (Python, /root/reference) == des_oracle.c (SimPy-faithful)  ALWAYS;
 *   des_oracle.c == this file  whenever no two timed events share a timestamp
 *   (`ties == 0`; continuous latencies make ties a measure-zero event, but
 *   deterministic step times under queueing produce them -- DESIGN.md "Ties");
 *   HIP engine == this file  ALWAYS, bit for bit (tests/test_gpu_parity.py).
 *
 * Each function cites the reference lines it follows.  Never linked into
 * asyncflow_amd.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/asyncflow_hip.h" /* data format of the lowered plan only */
#include "oracle_rng.h"

/* ---------------------------------------------------------------- events */
enum { EV_ARRIVAL, EV_TICK, EV_EMARK, EV_SMARK, EV_DELIVER, EV_CPU_DONE, EV_IO_DONE };

typedef struct {
    double t;
    uint64_t seq;
    int kind;
    int req;
} ev_t;

typedef struct {
    ev_t* a;
    size_t n, cap;
    uint64_t next_seq;
} heap_t;

/* Engine order: (time, class, push seq).  Classes: edge-timeline mark <
 * server-timeline mark < sampler tick < arrival < request events.  Ties between
 * classes are measure-zero and commute in their observable effects except
 * tick-vs-request (see DESIGN.md "Ties"); the fixed order keeps the HIP engine's
 * register-resident "special" timers out of its heap. */
static int ev_class(int kind) {
    switch (kind) {
        case EV_EMARK: return 0;
        case EV_SMARK: return 1;
        case EV_TICK: return 2;
        case EV_ARRIVAL: return 3;
        default: return 4;
    }
}
static int ev_less(const ev_t* x, const ev_t* y) {
    if (x->t != y->t) return x->t < y->t;
    int cx = ev_class(x->kind), cy = ev_class(y->kind);
    if (cx != cy) return cx < cy;
    return x->seq < y->seq;
}

static void heap_push(heap_t* h, double t, int kind, int req) {
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 64;
        h->a = (ev_t*)realloc(h->a, h->cap * sizeof(ev_t));
    }
    ev_t e = {t, h->next_seq++, kind, req};
    size_t i = h->n++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!ev_less(&e, &h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = e;
}

static ev_t heap_pop(heap_t* h) {
    ev_t top = h->a[0];
    ev_t last = h->a[--h->n];
    size_t i = 0;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && ev_less(&h->a[c + 1], &h->a[c])) c += 1;
        if (!ev_less(&h->a[c], &last)) break;
        h->a[i] = h->a[c];
        i = c;
    }
    if (h->n) h->a[i] = last;
    return top;
}

/* ---------------------------------------------------------------- state */
typedef struct { /* RequestState (runtime/rqs_state.py:21-51) + _handle_request locals */
    double t0;       /* initial_time                                  */
    double ram;      /* total_ram of the selected endpoint            */
    int hops;        /* len(history)                                  */
    int edge;        /* edge currently carrying the message           */
    int server, ep;  /* server / endpoint being executed              */
    uint32_t step;   /* absolute index of the next/current step       */
    int core_locked, in_io;
    int next_free;
} req_t;

typedef struct { int* a; size_t head, n, cap; } fifo_t;

static void fifo_push(fifo_t* f, int v) {
    if (f->head + f->n == f->cap) {
        if (f->head > 0) {
            memmove(f->a, f->a + f->head, f->n * sizeof(int));
            f->head = 0;
        } else {
            f->cap = f->cap ? 2 * f->cap : 16;
            f->a = (int*)realloc(f->a, f->cap * sizeof(int));
        }
    }
    f->a[f->head + f->n++] = v;
}
static int fifo_pop(fifo_t* f) { f->n--; return f->a[f->head++]; }
static int fifo_front(const fifo_t* f) { return f->a[f->head]; }

typedef struct { /* ServerRuntime + ServerContainers (server.py, server_containers.py:34-68) */
    int cpu_free;        /* CPU container level      */
    double ram_free;     /* RAM container level      */
    int ready, io;       /* _el_ready_queue_len, _el_io_queue_len */
    double ram_in_use;   /* _ram_in_use              */
    fifo_t cpu_wait, ram_wait;
    uint32_t arrivals;
} srv_t;

typedef struct { int conn; double spike; uint32_t sends; } edge_t;

typedef struct {
    const af_plan_t* p;
    uint64_t seed;
    double now;
    heap_t heap;
    req_t* reqs; int n_reqs, cap_reqs, free_head, live, max_live;
    srv_t* srv;
    edge_t* edge;
    int* lb_order; int lb_n;
    /* generator: poisson_poisson.py:51-82 / gaussian_poisson.py:63-94 */
    double g_now, g_window_end, g_lam; uint32_t g_draws;
    uint32_t emark_i, smark_i;
    /* outputs */
    uint64_t n_generated, n_completed, n_dropped, n_events, n_ticks, n_marks, flags, n_ties;
    double* clock; uint64_t clock_cap;
    uint32_t* samples; uint64_t tick_cap; uint32_t n_series;
} sim_t;

static int req_alloc(sim_t* s) {
    int r;
    if (s->free_head >= 0) {
        r = s->free_head;
        s->free_head = s->reqs[r].next_free;
    } else {
        if (s->n_reqs == s->cap_reqs) {
            s->cap_reqs = s->cap_reqs ? 2 * s->cap_reqs : 64;
            s->reqs = (req_t*)realloc(s->reqs, (size_t)s->cap_reqs * sizeof(req_t));
        }
        r = s->n_reqs++;
    }
    memset(&s->reqs[r], 0, sizeof(req_t));
    s->live += 1;
    if (s->live > s->max_live) s->max_live = s->live;
    return r;
}
static void req_free(sim_t* s, int r) {
    s->reqs[r].next_free = s->free_head;
    s->free_head = r;
    s->live -= 1;
}

/* ------------------------------------------------ generator (samplers/) */
/* Returns the next inter-arrival gap or -1 when the sampler is exhausted.
 * Follows poisson_poisson.py:55-82 line by line (gaussian_poisson.py:67-94 is
 * identical except for the users draw). */
static double next_gap(sim_t* s) {
    const af_plan_t* p = s->p;
    const double T = p->total_time;
    const double rps_per_user = p->gen_rpm_mean / 60.0; /* TimeDefaults.MIN_TO_SEC */
    while (s->g_now < T) {
        if (s->g_now >= s->g_window_end) {
            s->g_window_end = s->g_now + p->gen_window_s;
            double users;
            uint32_t idx = s->g_draws++;
            if (p->gen_users_dist == AF_DIST_NORMAL) { /* rqs_generator.py:83-90 */
                double v = p->gen_users_mean +
                           p->gen_users_sigma *
                               orc_norminv(orc_uniform(s->seed, ORC_STREAM_GENERATOR, idx, 0));
                users = v > 0.0 ? v : 0.0; /* common_helpers.py:32-33 */
            } else {
                users = (double)orc_poisson(p->gen_users_mean, s->seed, ORC_STREAM_GENERATOR, idx, 0);
            }
            s->g_lam = users * rps_per_user;
        }
        if (s->g_lam <= 0.0) {
            s->g_now = s->g_window_end;
            continue;
        }
        double u = orc_uniform(s->seed, ORC_STREAM_GENERATOR, s->g_draws++, 0);
        if (u < 1e-15) u = 1e-15; /* max(u, 1e-15) */
        double dt = -orc_log(1.0 - u) / s->g_lam;
        if (s->g_now + dt > T) break;
        if (s->g_now + dt >= s->g_window_end) {
            s->g_now = s->g_window_end;
            continue;
        }
        s->g_now += dt;
        return dt;
    }
    s->g_now = T + 1.0; /* exhausted for good */
    return -1.0;
}

/* -------------------------------------------------- edge (actors/edge.py) */
/* EdgeRuntime.transport/_deliver up to the `yield env.timeout` (edge.py:73-107) */
static void edge_send(sim_t* s, int r, int e) {
    const af_plan_t* p = s->p;
    edge_t* ed = &s->edge[e];
    uint32_t idx = ed->sends++;
    double u = orc_uniform(s->seed, ORC_STREAM_EDGE(e), idx, 0);
    if (u < p->edge_dropout[e]) { /* edge.py:78-86: dropped, no latency draw */
        s->n_dropped += 1;
        req_free(s, r);
        return;
    }
    ed->conn += 1; /* edge.py:88 */
    double transit = orc_variate(p->edge_dist[e], p->edge_mean[e], p->edge_sigma[e], s->seed,
                                 ORC_STREAM_EDGE(e), idx, 1);
    double effective = transit + ed->spike; /* edge.py:94-106, spike read at SEND time */
    s->reqs[r].edge = e;
    heap_push(&s->heap, s->now + effective, EV_DELIVER, r);
}

/* ----------------------------------------------- server (actors/server.py) */
static void run_steps(sim_t* s, int r);

/* A CPU token became free: hand it to the first waiter (Container FIFO).
 * Returns the waiter (its Timeout is scheduled by the caller, AFTER the
 * releasing request's own, see the ordering note in run_steps) or -1. */
static int cpu_release(sim_t* s, int sv) {
    srv_t* S = &s->srv[sv];
    S->cpu_free += 1;
    if (S->cpu_wait.n > 0) {
        int w = fifo_pop(&S->cpu_wait);
        S->cpu_free -= 1;
        return w;
    }
    return -1;
}

/* server.py:220-231: the waiter's `yield cpu_req` returns */
static void cpu_granted(sim_t* s, int w) {
    req_t* W = &s->reqs[w];
    srv_t* S = &s->srv[W->server];
    S->ready -= 1; /* waiting_cpu -> False */
    W->core_locked = 1;
    heap_push(&s->heap, s->now + s->p->step_time[W->step], EV_CPU_DONE, w);
}

/* The for-loop of _handle_request (server.py:197-276) from step `req.step`. */
static void run_steps(sim_t* s, int r) {
    const af_plan_t* p = s->p;
    req_t* R = &s->reqs[r];
    const int sv = R->server;
    srv_t* S = &s->srv[sv];
    const uint32_t end = p->ep_step_begin[R->ep + 1];

    if (R->step < end) {
        if (p->step_kind[R->step] == AF_STEP_CPU) { /* server.py:199-231 */
            if (R->in_io) {
                R->in_io = 0;
                S->io -= 1;
            }
            if (!R->core_locked) {
                if (S->cpu_wait.n == 0 && S->cpu_free > 0) {
                    S->cpu_free -= 1; /* cpu_req.triggered: not counted in ready */
                    R->core_locked = 1;
                } else {
                    fifo_push(&S->cpu_wait, r); /* waiting_cpu = True */
                    S->ready += 1;
                    return;
                }
            }
            heap_push(&s->heap, s->now + p->step_time[R->step], EV_CPU_DONE, r);
            return;
        }
        /* I/O step, server.py:235-255 */
        int granted = -1;
        if (R->core_locked) {
            granted = cpu_release(s, sv);
            R->core_locked = 0;
            if (!R->in_io) {
                R->in_io = 1;
                S->io += 1;
            }
        } else if (!R->in_io) {
            R->in_io = 1;
            S->io += 1;
        }
        /* SimPy order: the put event's first callback grants the waiter (its get
         * event is only *scheduled*), then this process resumes and creates its
         * own Timeout; the waiter's Timeout is created later. */
        heap_push(&s->heap, s->now + p->step_time[R->step], EV_IO_DONE, r);
        if (granted >= 0) cpu_granted(s, granted);
        return;
    }

    /* endpoint finished, server.py:257-276 */
    if (R->core_locked) {
        int granted = cpu_release(s, sv);
        R->core_locked = 0;
        /* here the waiter's get event is processed BEFORE the RAM put below is,
         * hence before this request's transport(): waiter first. */
        if (granted >= 0) cpu_granted(s, granted);
    }
    if (R->in_io) {
        R->in_io = 0;
        S->io -= 1;
    }
    const double ram = R->ram;
    if (ram > 0.0) { /* `if total_ram:` */
        S->ram_in_use -= ram;
        S->ram_free += ram;
    }
    /* transport() -> Initialize is URGENT: runs before the RAM waiters' get events */
    edge_send(s, r, p->srv_out_edge[sv]);
    if (ram > 0.0) {
        /* Container._trigger_get: FIFO with head-of-line blocking */
        while (S->ram_wait.n > 0) {
            int w = fifo_front(&S->ram_wait);
            double need = s->reqs[w].ram;
            if (S->ram_free < need) break;
            fifo_pop(&S->ram_wait);
            S->ram_free -= need;
            S->ram_in_use += need; /* server.py:149 */
            run_steps(s, w);
            S = &s->srv[sv];
        }
    }
}

/* _dispatcher + head of _handle_request (server.py:303-313, 79-149) */
static void server_arrival(sim_t* s, int r, int sv) {
    const af_plan_t* p = s->p;
    srv_t* S = &s->srv[sv];
    req_t* R = &s->reqs[r];
    R->hops += 1; /* record_hop(SERVER) */
    uint32_t n_ep = p->srv_ep_begin[sv + 1] - p->srv_ep_begin[sv];
    uint32_t idx = S->arrivals++;
    uint32_t pick = 0;
    if (n_ep > 1) { /* rng.integers(0, n_ep), server.py:101 */
        pick = (uint32_t)(((uint64_t)orc_word0(s->seed, ORC_STREAM_SERVER(sv), idx) * n_ep) >> 32);
    }
    R->server = sv;
    R->ep = (int)(p->srv_ep_begin[sv] + pick);
    R->ram = p->ep_ram[R->ep];
    R->step = p->ep_step_begin[R->ep];
    R->core_locked = 0;
    R->in_io = 0;
    if (R->ram > 0.0) { /* server.py:146-149 */
        if (R->ram > p->srv_ram_mb[sv]) s->flags |= AF_FLAG_RAM_STARVED;
        if (S->ram_wait.n == 0 && S->ram_free >= R->ram) {
            S->ram_free -= R->ram;
            S->ram_in_use += R->ram;
        } else {
            fifo_push(&S->ram_wait, r); /* blocked on RAM: in neither queue */
            return;
        }
    }
    run_steps(s, r);
}

/* EdgeRuntime._deliver after the timeout (edge.py:110-116) + the target node */
static void deliver(sim_t* s, int r) {
    const af_plan_t* p = s->p;
    req_t* R = &s->reqs[r];
    const int e = R->edge;
    R->hops += 1; /* record_hop(NETWORK_CONNECTION) */
    s->edge[e].conn -= 1;
    switch (p->edge_target_kind[e]) {
        case AF_NODE_CLIENT: /* ClientRuntime._forwarder, client.py:46-71 */
            R->hops += 1;
            if (R->hops > 3) {
                if (s->n_completed < s->clock_cap) {
                    s->clock[2 * s->n_completed] = R->t0;
                    s->clock[2 * s->n_completed + 1] = s->now;
                } else {
                    s->flags |= AF_FLAG_CLOCK_OVERFLOW;
                }
                s->n_completed += 1;
                req_free(s, r);
            } else {
                edge_send(s, r, p->client_out_edge);
            }
            break;
        case AF_NODE_LB: { /* LoadBalancerRuntime._forwarder, load_balancer.py:60-72 */
            R->hops += 1;
            int out;
            if (p->lb_algo == AF_LB_LEAST_CONNECTIONS) { /* lb_algorithms.py:10-20 */
                int best = 0;
                for (int i = 1; i < s->lb_n; ++i)
                    if (s->edge[s->lb_order[i]].conn < s->edge[s->lb_order[best]].conn) best = i;
                out = s->lb_order[best];
            } else { /* round_robin, lb_algorithms.py:22-36: first key, move_to_end */
                out = s->lb_order[0];
                for (int i = 1; i < s->lb_n; ++i) s->lb_order[i - 1] = s->lb_order[i];
                s->lb_order[s->lb_n - 1] = out;
            }
            edge_send(s, r, out);
            break;
        }
        default:
            server_arrival(s, r, p->edge_target_idx[e]);
    }
}

/* ------------------------------------- event injection (events/injection.py) */
static void apply_emarks(sim_t* s) { /* _assign_edges_spike, injection.py:167-198 */
    const af_plan_t* p = s->p;
    for (;;) {
        uint32_t i = s->emark_i++;
        s->edge[p->emark_edge[i]].spike += p->emark_delta[i];
        s->n_marks += 1;
        if (s->emark_i >= p->n_edge_marks) return;
        if (p->emark_time[s->emark_i] > s->now) break; /* dt > 0: new Timeout */
    }
    heap_push(&s->heap, p->emark_time[s->emark_i], EV_EMARK, -1);
}

static void apply_smarks(sim_t* s) { /* _assign_server_state, injection.py:201-226 */
    const af_plan_t* p = s->p;
    for (;;) {
        uint32_t i = s->smark_i++;
        int e = p->smark_lb_edge[i];
        s->n_marks += 1;
        if (e >= 0) {
            int pos = -1;
            for (int k = 0; k < s->lb_n; ++k)
                if (s->lb_order[k] == e) pos = k;
            if (p->smark_down[i]) { /* lb_out_edges.pop(edge_id, None) */
                if (pos >= 0) {
                    for (int k = pos + 1; k < s->lb_n; ++k) s->lb_order[k - 1] = s->lb_order[k];
                    s->lb_n -= 1;
                }
            } else { /* re-insert + move_to_end */
                if (pos >= 0) {
                    for (int k = pos + 1; k < s->lb_n; ++k) s->lb_order[k - 1] = s->lb_order[k];
                    s->lb_order[s->lb_n - 1] = e;
                } else {
                    s->lb_order[s->lb_n++] = e;
                }
            }
        }
        if (s->smark_i >= p->n_srv_marks) return;
        if (p->smark_time[s->smark_i] > s->now) break;
    }
    heap_push(&s->heap, p->smark_time[s->smark_i], EV_SMARK, -1);
}

/* ----------------------------- sampler tick (metrics/collector.py:50-66) */
static void sample_tick(sim_t* s) {
    const af_plan_t* p = s->p;
    if (s->samples && s->n_ticks < s->tick_cap) {
        const uint64_t k = s->n_ticks;
        if (p->metrics_mask & AF_METRIC_EDGE_CONCURRENT_CONNECTION)
            for (uint32_t e = 0; e < p->n_edges; ++e)
                s->samples[(uint64_t)e * s->tick_cap + k] = (uint32_t)s->edge[e].conn;
        const uint32_t all = AF_METRIC_READY_QUEUE_LEN | AF_METRIC_EVENT_LOOP_IO_SLEEP |
                             AF_METRIC_RAM_IN_USE;
        if ((p->metrics_mask & all) == all)
            for (uint32_t v = 0; v < p->n_servers; ++v) {
                uint64_t base = (uint64_t)(p->n_edges + 3 * v) * s->tick_cap + k;
                float ram = (float)s->srv[v].ram_in_use;
                uint32_t ram_bits;
                memcpy(&ram_bits, &ram, 4);
                s->samples[base] = (uint32_t)s->srv[v].ready;
                s->samples[base + s->tick_cap] = (uint32_t)s->srv[v].io;
                s->samples[base + 2 * s->tick_cap] = ram_bits;
            }
    } else if (s->samples) {
        s->flags |= AF_FLAG_TICK_OVERFLOW;
    }
    s->n_ticks += 1;
}

/* ------------------------------------------------------------- top level */
static uint64_t g_atomic_ties;
uint64_t orc_atomic_last_ties(void) { return g_atomic_ties; }

/* counts: uint64[8] indexed by af_count_slot.  clock: [clock_cap][2] f64 or
 * NULL.  samples: [n_series][tick_cap] 4-byte words or NULL.  Returns 0. */
int orc_simulate_atomic(const af_plan_t* plan, uint64_t seed, uint64_t clock_cap, double* clock,
                 uint64_t tick_cap, uint32_t* samples, uint64_t* counts) {
    if (!plan || plan->abi_version != AF_ABI_VERSION || plan->struct_size != sizeof(af_plan_t))
        return AF_ERR_ABI;
    sim_t S;
    memset(&S, 0, sizeof(S));
    sim_t* s = &S;
    s->p = plan;
    s->seed = seed;
    s->free_head = -1;
    s->clock = clock;
    s->clock_cap = clock ? clock_cap : 0;
    s->samples = samples;
    s->tick_cap = tick_cap;
    s->n_series = plan->n_edges + 3 * plan->n_servers;
    s->srv = (srv_t*)calloc(plan->n_servers ? plan->n_servers : 1, sizeof(srv_t));
    s->edge = (edge_t*)calloc(plan->n_edges ? plan->n_edges : 1, sizeof(edge_t));
    s->lb_order = (int*)calloc(plan->n_lb_edges ? plan->n_lb_edges : 1, sizeof(int));
    for (uint32_t v = 0; v < plan->n_servers; ++v) { /* build_containers: init full */
        s->srv[v].cpu_free = (int)plan->srv_cores[v];
        s->srv[v].ram_free = plan->srv_ram_mb[v];
    }
    s->lb_n = (int)plan->n_lb_edges;
    for (int i = 0; i < s->lb_n; ++i) s->lb_order[i] = plan->lb_edges[i];

    /* process start order (simulation_runner.py:364-366): events, generator,
     * ..., collector.  Marks at t == 0 are applied inside the Initialize. */
    if (plan->n_edge_marks) {
        if (plan->emark_time[0] > 0.0) heap_push(&s->heap, plan->emark_time[0], EV_EMARK, -1);
        else apply_emarks(s);
    }
    if (plan->n_srv_marks) {
        if (plan->smark_time[0] > 0.0) heap_push(&s->heap, plan->smark_time[0], EV_SMARK, -1);
        else apply_smarks(s);
    }
    {
        double gap = next_gap(s);
        if (gap >= 0.0) heap_push(&s->heap, 0.0 + gap, EV_ARRIVAL, -1);
    }
    heap_push(&s->heap, 0.0 + plan->sample_period, EV_TICK, -1);

    const double T = plan->total_time;
    int have_prev = 0;
    while (s->heap.n > 0) {
        if (!(s->heap.a[0].t < T)) break; /* stop event is URGENT at T */
        ev_t ev = heap_pop(&s->heap);
        if (have_prev && ev.t == s->now) s->n_ties += 1;
        have_prev = 1;
        s->now = ev.t;
        switch (ev.kind) {
            case EV_ARRIVAL: { /* RqsGeneratorRuntime._event_arrival, rqs_generator.py:101-119 */
                s->n_generated += 1;
                s->n_events += 1;
                int r = req_alloc(s);
                s->reqs[r].t0 = s->now;
                s->reqs[r].hops = 1; /* record_hop(GENERATOR) */
                /* the generator's next Timeout is created before the edge process starts */
                double gap = next_gap(s);
                if (gap >= 0.0) heap_push(&s->heap, s->now + gap, EV_ARRIVAL, -1);
                edge_send(s, r, plan->gen_out_edge);
                break;
            }
            case EV_TICK:
                sample_tick(s);
                heap_push(&s->heap, s->now + plan->sample_period, EV_TICK, -1);
                break;
            case EV_EMARK: apply_emarks(s); break;
            case EV_SMARK: apply_smarks(s); break;
            case EV_DELIVER:
                s->n_events += 1;
                deliver(s, ev.req);
                break;
            case EV_CPU_DONE:
                s->n_events += 1;
                s->reqs[ev.req].step += 1;
                run_steps(s, ev.req);
                break;
            case EV_IO_DONE:
                s->n_events += 1;
                s->reqs[ev.req].step += 1;
                run_steps(s, ev.req);
                break;
        }
    }

    if (counts) {
        counts[AF_CNT_GENERATED] = s->n_generated;
        counts[AF_CNT_COMPLETED] = s->n_completed;
        counts[AF_CNT_DROPPED] = s->n_dropped;
        counts[AF_CNT_EVENTS] = s->n_events;
        counts[AF_CNT_TICKS] = s->n_ticks;
        counts[AF_CNT_FLAGS] = s->flags;
        counts[AF_CNT_MAX_LIVE] = (uint64_t)s->max_live;
        counts[AF_CNT_MARKS] = s->n_marks;
    }
    g_atomic_ties = s->n_ties;
    for (uint32_t v = 0; v < plan->n_servers; ++v) {
        free(s->srv[v].cpu_wait.a);
        free(s->srv[v].ram_wait.a);
    }
    free(s->srv);
    free(s->edge);
    free(s->lb_order);
    free(s->reqs);
    free(s->heap.a);
    return 0;
}

