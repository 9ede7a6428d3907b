/* des_oracle.c -- ORACLE / TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, one scenario at a time, unbounded dynamic state)
 * of the reference hot path
 *
 *     SimulationRunner.run() -> env.run(until=T)
 *     (/root/reference/src/asyncflow/runtime/simulation_runner.py:349-376)
 *
 * i.e. the reference actors AND the SimPy 4.1.1 scheduling rules they run on.
 * Every SimPy event the reference creates -- Timeouts, but also the zero-time
 * Initialize / Store put+get / Container put+get events -- is an entry of ONE
 * heap keyed (time, priority, eid) exactly like simpy.Environment's, and each
 * reference generator is a small state machine resumed at its `yield` points.
 * (Events nobody waits on -- process-termination events -- are not created:
 * dropping an eid does not change the relative order of the others.)
 * This matters: with deterministic step times, queued requests finish at
 * EXACTLY equal timestamps and SimPy then interleaves the zero-time steps of the
 * tied cascades breadth-first; tests/golden/overload_t30 pins that behaviour.
 * Random variates come from the counter-based spec in oracle_rng.h (see there
 * for why not numpy's PCG64).
 *
 * Pinning: tests/test_oracle_golden.py checks this file bit for bit against
 * tests/golden/ (npz files), which oracle/make_golden.py produced by running the
 * UNMODIFIED reference actors (imported from /root/reference) on the SimPy
 * stand-in with oracle/rng_adapter.py injected at the reference's own rng seam.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library; asyncflow_amd never links or loads it.
 *
 * counts[AF_CNT_MARKS+...]: see orc_simulate; `ties` (returned through
 * orc_last_ties) counts timed events popped at the timestamp of the previous
 * one, i.e. the instants at which SimPy interleaves the zero-time steps of
 * several cascades (DESIGN.md "Ties"); the tests use it to prove that a case
 * exercises that regime.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/asyncflow_hip.h" /* data format of the lowered plan only */
#include "oracle_rng.h"

/* ---------------------------------------------------------------- events */
#define PRIO_URGENT 0
#define PRIO_NORMAL 1

enum {
    /* timed (simpy.Timeout) */
    EV_GEN_TIMEOUT,   /* rqs_generator.py:104  yield env.timeout(gap)           */
    EV_TICK,          /* collector.py:53       yield env.timeout(period)        */
    EV_EMARK,         /* injection.py:187      yield env.timeout(dt)            */
    EV_SMARK,         /* injection.py:210                                       */
    EV_EDGE_TIMEOUT,  /* edge.py:107           yield env.timeout(effective)     */
    EV_STEP_TIMEOUT,  /* server.py:231,255     yield env.timeout(cpu|io time)   */
    /* zero-time */
    EV_EDGE_INIT,     /* edge.py:124           env.process(_deliver) Initialize, URGENT */
    EV_SRV_INIT,      /* server.py:313         env.process(_handle_request) Initialize, URGENT */
    EV_STORE_PUT,     /* edge.py:116           yield target_box.put(state)      */
    EV_STORE_GET,     /* client.py:48 / load_balancer.py:63 / server.py:310  yield box.get() */
    EV_CBOX_PUT,      /* client.py:69          yield completed_box.put(state)   */
    EV_RAM_GOT,       /* server.py:148         yield RAM.get(total_ram)         */
    EV_CPU_GOT,       /* server.py:220         yield cpu_req                    */
    EV_CPU_PUT_IO,    /* server.py:241         yield CPU.put(1)  (before an I/O step) */
    EV_CPU_PUT_END,   /* server.py:258         yield CPU.put(1)  (endpoint finished)  */
    EV_RAM_PUT        /* server.py:273         yield RAM.put(total_ram)         */
};

static int is_timed(int kind) { return kind <= EV_STEP_TIMEOUT; }

typedef struct {
    double t;
    uint64_t eid;
    int prio;
    int kind;
    int a; /* request id, or node id for store events */
    int b; /* node id for EV_STORE_GET */
} ev_t;

typedef struct {
    ev_t* a;
    size_t n, cap;
    uint64_t next_eid;
} heap_t;

/* simpy.Environment heap order: (time, priority, eid) */
static int ev_less(const ev_t* x, const ev_t* y) {
    if (x->t != y->t) return x->t < y->t;
    if (x->prio != y->prio) return x->prio < y->prio;
    return x->eid < y->eid;
}

static void heap_push(heap_t* h, double t, int prio, int kind, int a, int b) {
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 64;
        h->a = (ev_t*)realloc(h->a, h->cap * sizeof(ev_t));
    }
    ev_t e = {t, h->next_eid++, prio, kind, a, b};
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
    uint32_t step;   /* absolute index of the current step            */
    int core_locked, in_io, waiting_cpu;
    int got;         /* scratch: "get event triggered" marker         */
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

/* simpy.Store used as a node inbox + the single forwarder process reading it */
typedef struct {
    fifo_t items;
    int getter_waiting; /* forwarder is blocked in `yield box.get()` */
} box_t;

typedef struct { /* ServerRuntime + ServerContainers (server.py, server_containers.py:34-68) */
    int cpu_level;       /* CPU container level      */
    double ram_level;    /* RAM container level      */
    int ready, io;       /* _el_ready_queue_len, _el_io_queue_len */
    double ram_in_use;   /* _ram_in_use              */
    fifo_t cpu_q, ram_q; /* Container.get_queue      */
    fifo_t ram_pq;       /* RAM Container.put_queue: puts that `_do_put` refused (fractional needs, see ram_trigger_put) */
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
    box_t* box;          /* [0]=client, [1]=lb, [2+s]=server s */
    int* lb_order; int lb_n;
    /* generator: poisson_poisson.py:51-82 / gaussian_poisson.py:63-94 */
    double g_now, g_window_end, g_lam; uint32_t g_draws;
    uint32_t emark_i, smark_i;
    /* outputs */
    uint64_t n_generated, n_completed, n_dropped, n_events, n_ticks, n_marks, flags;
    uint64_t n_heap_events, n_ties, n_put_waits;
    double* clock; uint64_t clock_cap;
    uint32_t* samples; uint64_t tick_cap; uint32_t n_series;
} sim_t;

#define BOX_CLIENT 0
#define BOX_LB 1
#define BOX_SERVER(s) (2 + (s))

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

static void sched(sim_t* s, double delay, int prio, int kind, int a, int b) {
    heap_push(&s->heap, s->now + delay, prio, kind, a, b); /* Environment.schedule */
    s->n_heap_events += 1;
}

/* Timeline marks: the plan already holds the clock value `now + dt` the
 * reference's relative wait lands on (asyncflow_amd/plan.py::_env_times). */
static void sched_abs(sim_t* s, double t, int kind) {
    heap_push(&s->heap, t, PRIO_NORMAL, kind, -1, 0);
    s->n_heap_events += 1;
}

/* ------------------------------------------------ generator (samplers/) */
/* Returns the next inter-arrival gap or -1 when the sampler is exhausted.
 * Follows poisson_poisson.py:55-82 line by line (gaussian_poisson.py:67-94 is
 * identical except for the users draw). */
/* TEST-ONLY: latencies and arrival gaps rounded down to a multiple of 2^-bits s (0 = off), the same hook as
 * asyncflow_amd/csrc/af_math.hpp::test_quant in the host builds of the engine core: exact timestamp ties by the
 * thousand, to pin tie handling on SimPy's event order (tests/test_flow_hostcheck.py). */
static int g_test_quantum_bits = 0;
void orc_set_test_quantum(int bits) { g_test_quantum_bits = bits; }
static double test_quant(double x) {
    if (g_test_quantum_bits <= 0 || !(x > 0.0)) return x;
    const double sc = (double)(1ull << g_test_quantum_bits);
    return floor(x * sc) / sc;
}

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
        double dt = test_quant(-orc_log(1.0 - u) / s->g_lam);
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

/* ------------------------------------------------ simpy.Store (node inbox) */
/* StoreGet.__init__ -> _trigger_get: succeed at once if an item is queued */
static void forwarder_get(sim_t* s, int node) {
    box_t* B = &s->box[node];
    if (B->items.n > 0) {
        int item = fifo_pop(&B->items);
        B->getter_waiting = 0;
        sched(s, 0.0, PRIO_NORMAL, EV_STORE_GET, item, node);
    } else {
        B->getter_waiting = 1;
    }
}

/* -------------------------------------------------- edge (actors/edge.py) */
/* EdgeRuntime.transport (edge.py:119-124): spawn the _deliver process */
static void transport(sim_t* s, int r, int e) {
    s->reqs[r].edge = e;
    sched(s, 0.0, PRIO_URGENT, EV_EDGE_INIT, r, 0);
}

/* _deliver up to its first yield (edge.py:73-107) */
static void edge_init(sim_t* s, int r) {
    const af_plan_t* p = s->p;
    const int e = s->reqs[r].edge;
    edge_t* ed = &s->edge[e];
    uint32_t idx = ed->sends++;
    double u = orc_uniform(s->seed, ORC_STREAM_EDGE(e), idx, 0);
    if (u < p->edge_dropout[e]) { /* edge.py:78-86: dropped, no latency draw */
        s->n_dropped += 1;
        req_free(s, r);
        return;
    }
    ed->conn += 1; /* edge.py:88 */
    double transit = test_quant(orc_variate(p->edge_dist[e], p->edge_mean[e], p->edge_sigma[e], s->seed,
                                            ORC_STREAM_EDGE(e), idx, 1));
    double effective = transit + ed->spike; /* edge.py:94-106, spike read at SEND time */
    if (effective < 0.0) s->flags |= AF_FLAG_NEGATIVE_DELAY; /* edge.py:107 -> simpy raises ValueError("Negative delay") */
    sched(s, effective, PRIO_NORMAL, EV_EDGE_TIMEOUT, r, 0);
}

/* _deliver after the timeout (edge.py:110-116) */
static void edge_timeout(sim_t* s, int r) {
    const af_plan_t* p = s->p;
    req_t* R = &s->reqs[r];
    const int e = R->edge;
    R->hops += 1; /* record_hop(NETWORK_CONNECTION) */
    s->edge[e].conn -= 1;
    int node = p->edge_target_kind[e] == AF_NODE_CLIENT ? BOX_CLIENT
             : p->edge_target_kind[e] == AF_NODE_LB     ? BOX_LB
                                                        : BOX_SERVER(p->edge_target_idx[e]);
    /* StorePut.__init__: unbounded store -> append + succeed immediately */
    fifo_push(&s->box[node].items, r);
    sched(s, 0.0, PRIO_NORMAL, EV_STORE_PUT, node, 0);
}

/* the put event is processed: its first callback is Store._trigger_get */
static void store_put_processed(sim_t* s, int node) {
    box_t* B = &s->box[node];
    if (B->getter_waiting && B->items.n > 0) {
        int item = fifo_pop(&B->items);
        B->getter_waiting = 0;
        sched(s, 0.0, PRIO_NORMAL, EV_STORE_GET, item, node);
    }
    /* then the edge process resumes and terminates (nobody waits on it) */
}

/* ----------------------------------------------- server (actors/server.py) */
/* Container._trigger_get for the CPU container (amount is always 1) */
static void cpu_trigger_get(sim_t* s, int sv) {
    srv_t* S = &s->srv[sv];
    while (S->cpu_q.n > 0) {
        if (S->cpu_level < 1) break; /* _do_get fails: head-of-line blocking */
        int w = fifo_pop(&S->cpu_q);
        S->cpu_level -= 1;
        s->reqs[w].got = 1;
        sched(s, 0.0, PRIO_NORMAL, EV_CPU_GOT, w, 0);
    }
}

/* Container._trigger_get for the RAM container */
static void ram_trigger_get(sim_t* s, int sv) {
    srv_t* S = &s->srv[sv];
    while (S->ram_q.n > 0) {
        int w = fifo_front(&S->ram_q);
        double need = s->reqs[w].ram;
        if (S->ram_level < need) break;
        fifo_pop(&S->ram_q);
        S->ram_level -= need;
        s->reqs[w].got = 1;
        sched(s, 0.0, PRIO_NORMAL, EV_RAM_GOT, w, 0);
    }
}

/* BaseResource._trigger_put for the RAM container: walk the put queue from its head, stop at the first put that
 * `Container._do_put` refuses (`if self._capacity - self._level >= event.amount`) -- head-of-line blocking, like the gets.
 * With whole-MB needs the test never fails (capacity - level IS the sum of what is held).  With a fractional need it can
 * fail by ONE ROUNDING: 2048 - fl(2048 - 100.3) < 100.3, so `yield RAM.put(100.3)` of the only holder WAITS.  The queue is
 * walked again (a) by every later ContainerPut.__init__ and (b) when a RAM get is PROCESSED (`_trigger_put` is the first
 * callback Get.__init__ registers, before the process's own resume): the level that get took lets the put through, and
 * the response leaves then (server.py:270-276). */
static void ram_trigger_put(sim_t* s, int sv) {
    srv_t* S = &s->srv[sv];
    while (S->ram_pq.n > 0) {
        int w = fifo_front(&S->ram_pq);
        double amount = s->reqs[w].ram;
        if (!(s->p->srv_ram_mb[sv] - S->ram_level >= amount)) break; /* _do_put returns None: `if not proceed: break` */
        fifo_pop(&S->ram_pq);
        S->ram_level += amount;
        sched(s, 0.0, PRIO_NORMAL, EV_RAM_PUT, w, 0); /* event.succeed() */
    }
}

/* Informational, like the starved case below: a refused put at the head of the put queue facing a waiter at the head of
 * the get queue that does not fit -- neither queue can ever move again (both are head-of-line blocked and nothing else
 * changes the level), so the server's RAM is dead-locked for good in the reference.  Nothing is changed here: the queues
 * simply never move; the engine reports the same flag (and stops keeping what queues up behind). */
static void ram_deadlock_check(sim_t* s, int sv) {
    srv_t* S = &s->srv[sv];
    if (S->ram_pq.n == 0 || S->ram_q.n == 0) return;
    if (S->ram_level >= s->reqs[fifo_front(&S->ram_q)].ram) return;
    if (s->p->srv_ram_mb[sv] - S->ram_level >= s->reqs[fifo_front(&S->ram_pq)].ram) return;
    s->flags |= AF_FLAG_RAM_STARVED;
}

/* tail of _handle_request after the core was given back (server.py:261-276) */
static void srv_finish(sim_t* s, int r) {
    req_t* R = &s->reqs[r];
    srv_t* S = &s->srv[R->server];
    if (R->in_io) {
        R->in_io = 0;
        S->io -= 1;
    }
    if (R->waiting_cpu) { /* defensive branch of the reference, never taken */
        R->waiting_cpu = 0;
        S->ready -= 1;
    }
    if (R->ram > 0.0) { /* `if total_ram:` */
        S->ram_in_use -= R->ram;
        fifo_push(&S->ram_pq, r); /* ContainerPut.__init__: append, then _trigger_put(None) */
        ram_trigger_put(s, R->server);
        if (S->ram_pq.n > 0) s->n_put_waits += 1; /* diagnostics: this put (or one ahead of it) was refused */
        ram_deadlock_check(s, R->server);
        return; /* yield RAM.put(total_ram): resumes at EV_RAM_PUT */
    }
    transport(s, r, s->p->srv_out_edge[R->server]);
}

/* The for-loop of _handle_request (server.py:197-255) from step `R->step`,
 * run until the generator yields. */
static void srv_continue(sim_t* s, int r) {
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
                /* cpu_req = CPU.get(1): ContainerGet.__init__ appends + triggers */
                R->got = 0;
                fifo_push(&S->cpu_q, r);
                cpu_trigger_get(s, sv);
                if (!R->got) { /* `if not cpu_req.triggered` */
                    R->waiting_cpu = 1;
                    S->ready += 1;
                }
                return; /* yield cpu_req */
            }
            sched(s, p->step_time[R->step], PRIO_NORMAL, EV_STEP_TIMEOUT, r, 0);
            return;
        }
        /* I/O step, server.py:235-255 */
        if (R->core_locked) {
            S->cpu_level += 1; /* CPU.put(1) succeeds at once */
            sched(s, 0.0, PRIO_NORMAL, EV_CPU_PUT_IO, r, 0);
            return; /* yield CPU.put(1) */
        }
        if (!R->in_io) {
            R->in_io = 1;
            S->io += 1;
        }
        sched(s, p->step_time[R->step], PRIO_NORMAL, EV_STEP_TIMEOUT, r, 0);
        return;
    }
    /* endpoint finished, server.py:257-259 */
    if (R->core_locked) {
        S->cpu_level += 1;
        sched(s, 0.0, PRIO_NORMAL, EV_CPU_PUT_END, r, 0);
        return;
    }
    srv_finish(s, r);
}

/* head of _handle_request (server.py:79-149), run by its Initialize event */
static void srv_init(sim_t* s, int r) {
    const af_plan_t* p = s->p;
    req_t* R = &s->reqs[r];
    const int sv = R->server;
    srv_t* S = &s->srv[sv];
    R->hops += 1; /* record_hop(SERVER) */
    uint32_t n_ep = p->srv_ep_begin[sv + 1] - p->srv_ep_begin[sv];
    uint32_t idx = S->arrivals++;
    uint32_t pick = 0;
    if (n_ep > 1) { /* rng.integers(0, n_ep), server.py:101 */
        pick = (uint32_t)(((uint64_t)orc_word0(s->seed, ORC_STREAM_SERVER(sv), idx) * n_ep) >> 32);
    }
    R->ep = (int)(p->srv_ep_begin[sv] + pick);
    R->ram = p->ep_ram[R->ep];
    R->step = p->ep_step_begin[R->ep];
    R->core_locked = 0;
    R->in_io = 0;
    R->waiting_cpu = 0;
    if (R->ram > 0.0) { /* server.py:146-149 */
        if (R->ram > p->srv_ram_mb[sv]) s->flags |= AF_FLAG_RAM_STARVED;
        fifo_push(&S->ram_q, r); /* ContainerGet.__init__ */
        ram_trigger_get(s, sv);
        ram_deadlock_check(s, sv);
        return; /* yield RAM.get(total_ram): resumes at EV_RAM_GOT */
    }
    srv_continue(s, r);
}

/* a forwarder process resumes with `state` (its box.get() event is processed) */
static void store_get_processed(sim_t* s, int r, int node) {
    const af_plan_t* p = s->p;
    req_t* R = &s->reqs[r];
    if (node == BOX_CLIENT) { /* ClientRuntime._forwarder, client.py:46-71 */
        R->hops += 1;
        if (R->hops > 3) {
            if (s->n_completed < s->clock_cap) {
                s->clock[2 * s->n_completed] = R->t0;
                s->clock[2 * s->n_completed + 1] = s->now;
            } else if (s->clock) {
                s->flags |= AF_FLAG_CLOCK_OVERFLOW;
            }
            s->n_completed += 1;
            req_free(s, r);
            /* yield completed_box.put(state): the client only asks for the next
             * message once that put event has been processed */
            sched(s, 0.0, PRIO_NORMAL, EV_CBOX_PUT, 0, 0);
            return;
        }
        transport(s, r, p->client_out_edge);
        forwarder_get(s, BOX_CLIENT);
        return;
    }
    if (node == BOX_LB) { /* LoadBalancerRuntime._forwarder, load_balancer.py:60-72 */
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
        transport(s, r, out);
        forwarder_get(s, BOX_LB);
        return;
    }
    /* ServerRuntime._dispatcher, server.py:303-313 */
    R->server = node - 2;
    sched(s, 0.0, PRIO_URGENT, EV_SRV_INIT, r, 0);
    forwarder_get(s, node);
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
    sched_abs(s, p->emark_time[s->emark_i], EV_EMARK);
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
    sched_abs(s, p->smark_time[s->smark_i], EV_SMARK);
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
static uint64_t g_last_ties, g_last_heap_events, g_last_put_waits;
uint64_t orc_last_put_waits(void) { return g_last_put_waits; } /* RAM puts that had to wait (tests: proves a case is in that regime) */
uint64_t orc_last_ties(void) { return g_last_ties; }
uint64_t orc_last_heap_events(void) { return g_last_heap_events; }

/* counts: uint64[8] indexed by af_count_slot.  clock: [clock_cap][2] f64 or
 * NULL.  samples: [n_series][tick_cap] 4-byte words or NULL.  Returns 0. */
int orc_simulate(const af_plan_t* plan, uint64_t seed, uint64_t clock_cap, double* clock,
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
    s->box = (box_t*)calloc(2 + plan->n_servers, sizeof(box_t));
    s->lb_order = (int*)calloc(plan->n_lb_edges ? plan->n_lb_edges : 1, sizeof(int));
    for (uint32_t v = 0; v < plan->n_servers; ++v) { /* build_containers: init full */
        s->srv[v].cpu_level = (int)plan->srv_cores[v];
        s->srv[v].ram_level = plan->srv_ram_mb[v];
    }
    s->lb_n = (int)plan->n_lb_edges;
    for (int i = 0; i < s->lb_n; ++i) s->lb_order[i] = plan->lb_edges[i];

    /* Initialize events run in process start order (simulation_runner.py:364-366):
     * edge timeline, server timeline, generator, client, servers, LB, collector.
     * Marks at t == 0 are applied inside the Initialize (dt == 0). */
    if (plan->n_edge_marks) {
        if (plan->emark_time[0] > 0.0) sched_abs(s, plan->emark_time[0], EV_EMARK);
        else apply_emarks(s);
    }
    if (plan->n_srv_marks) {
        if (plan->smark_time[0] > 0.0) sched_abs(s, plan->smark_time[0], EV_SMARK);
        else apply_smarks(s);
    }
    {
        double gap = next_gap(s);
        if (gap >= 0.0) sched(s, gap, PRIO_NORMAL, EV_GEN_TIMEOUT, -1, 0);
    }
    for (uint32_t b = 0; b < 2 + plan->n_servers; ++b) s->box[b].getter_waiting = 1;
    sched(s, plan->sample_period, PRIO_NORMAL, EV_TICK, -1, 0);

    const double T = plan->total_time;
    int have_prev_timed = 0;
    while (s->heap.n > 0) {
        if (!(s->heap.a[0].t < T)) break; /* stop event: URGENT at T, scheduled first */
        ev_t ev = heap_pop(&s->heap);
        if (is_timed(ev.kind)) {
            if (have_prev_timed && ev.t == s->now) s->n_ties += 1;
            have_prev_timed = 1;
        }
        s->now = ev.t;
        switch (ev.kind) {
            case EV_GEN_TIMEOUT: { /* RqsGeneratorRuntime._event_arrival, rqs_generator.py:101-119 */
                s->n_generated += 1;
                s->n_events += 1;
                int r = req_alloc(s);
                s->reqs[r].t0 = s->now;
                s->reqs[r].hops = 1; /* record_hop(GENERATOR) */
                transport(s, r, plan->gen_out_edge);
                double gap = next_gap(s); /* next(time_gaps) runs before the edge process starts */
                if (gap >= 0.0) sched(s, gap, PRIO_NORMAL, EV_GEN_TIMEOUT, -1, 0);
                break;
            }
            case EV_TICK:
                sample_tick(s);
                sched(s, plan->sample_period, PRIO_NORMAL, EV_TICK, -1, 0);
                break;
            case EV_EMARK: apply_emarks(s); break;
            case EV_SMARK: apply_smarks(s); break;
            case EV_EDGE_INIT: edge_init(s, ev.a); break;
            case EV_EDGE_TIMEOUT:
                s->n_events += 1;
                edge_timeout(s, ev.a);
                break;
            case EV_STORE_PUT: store_put_processed(s, ev.a); break;
            case EV_STORE_GET: store_get_processed(s, ev.a, ev.b); break;
            case EV_CBOX_PUT: forwarder_get(s, BOX_CLIENT); break; /* client.py:46-48 loop */
            case EV_SRV_INIT: srv_init(s, ev.a); break;
            case EV_RAM_GOT: { /* server.py:149; the get's first callback is the container's _trigger_put */
                req_t* R = &s->reqs[ev.a];
                ram_trigger_put(s, R->server);
                s->srv[R->server].ram_in_use += R->ram;
                srv_continue(s, ev.a);
                break;
            }
            case EV_CPU_GOT: { /* server.py:220-231 */
                req_t* R = &s->reqs[ev.a];
                if (R->waiting_cpu) {
                    R->waiting_cpu = 0;
                    s->srv[R->server].ready -= 1;
                }
                R->core_locked = 1;
                sched(s, plan->step_time[R->step], PRIO_NORMAL, EV_STEP_TIMEOUT, ev.a, 0);
                break;
            }
            case EV_STEP_TIMEOUT: /* the for-loop moves on to the next step */
                s->n_events += 1;
                s->reqs[ev.a].step += 1;
                srv_continue(s, ev.a);
                break;
            case EV_CPU_PUT_IO: { /* server.py:241-255: put processed -> grants, then resume */
                req_t* R = &s->reqs[ev.a];
                srv_t* Sv = &s->srv[R->server];
                cpu_trigger_get(s, R->server);
                R->core_locked = 0;
                if (!R->in_io) {
                    R->in_io = 1;
                    Sv->io += 1;
                }
                sched(s, plan->step_time[R->step], PRIO_NORMAL, EV_STEP_TIMEOUT, ev.a, 0);
                break;
            }
            case EV_CPU_PUT_END: { /* server.py:258-259 */
                req_t* R = &s->reqs[ev.a];
                cpu_trigger_get(s, R->server);
                R->core_locked = 0;
                srv_finish(s, ev.a);
                break;
            }
            case EV_RAM_PUT: { /* server.py:273-276 */
                req_t* R = &s->reqs[ev.a];
                ram_trigger_get(s, R->server);
                transport(s, ev.a, plan->srv_out_edge[R->server]);
                break;
            }
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
    g_last_ties = s->n_ties;
    g_last_heap_events = s->n_heap_events;
    g_last_put_waits = s->n_put_waits;
    for (uint32_t v = 0; v < plan->n_servers; ++v) {
        free(s->srv[v].cpu_q.a);
        free(s->srv[v].ram_q.a);
        free(s->srv[v].ram_pq.a);
    }
    for (uint32_t b = 0; b < 2 + plan->n_servers; ++b) free(s->box[b].items.a);
    free(s->box);
    free(s->srv);
    free(s->edge);
    free(s->lb_order);
    free(s->reqs);
    free(s->heap.a);
    return 0;
}

/* ---- spec probes for ctypes (oracle/rng_adapter.py, tests/test_rng_spec.py) */
double orc_x_uniform(uint64_t seed, uint32_t stream, uint32_t index, uint32_t j) {
    return orc_uniform(seed, stream, index, j);
}
uint32_t orc_x_word0(uint64_t seed, uint32_t stream, uint32_t index) {
    return orc_word0(seed, stream, index);
}
void orc_x_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                  uint32_t* out) {
    orc_philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}
double orc_x_log(double x) { return orc_log(x); }
double orc_x_exp(double x) { return orc_exp(x); }
double orc_x_norminv(double p) { return orc_norminv(p); }
int64_t orc_x_poisson(double mean, uint64_t seed, uint32_t stream, uint32_t index, uint32_t j0) {
    return orc_poisson(mean, seed, stream, index, j0);
}
double orc_x_variate(int dist, double mean, double sigma, uint64_t seed, uint32_t stream,
                     uint32_t index, uint32_t j0) {
    return orc_variate(dist, mean, sigma, seed, stream, index, j0);
}
uint32_t orc_tick_count(double period, double total_time) {
    /* collector.py:50-53: first tick at 0 + period, then now + period; a tick
     * is taken iff its time is < T (stop event URGENT at T). */
    uint32_t n = 0;
    double t = 0.0 + period;
    while (t < total_time) {
        n += 1;
        t = t + period;
    }
    return n;
}
