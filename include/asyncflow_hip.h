/* asyncflow_hip.h -- C ABI of the MI355X batched discrete-event engine.
 *
 * Drop-in boundary for ONE hot path of AsyncFlow (reference @ /root/reference):
 *
 *     SimulationRunner.run()  ->  simpy.Environment.run(until=T)
 *     (src/asyncflow/runtime/simulation_runner.py:349-376)
 *
 * executed over n independent scenarios (seed replicas / parameter-grid points)
 * on one MI355X.  The reference has NO FFI/plugin boundary for this path
 * (pure Python): the only seam is the Python call
 * `SimulationRunner(env=..., simulation_input=SimulationPayload).run()`.
 * The host-side mirror of that call is asyncflow_amd.SimulationRunner
 * (asyncflow_amd/runner.py); it lowers the validated payload to `af_plan_t`
 * and calls the entry points below through ctypes (INTEGRATION.md shows the
 * stub a reference maintainer would add).
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; pointers + sizes only;
 *  - every `const T*` inside af_plan_t / af_sweep_t is HOST memory, borrowed for
 *    the duration of the call (the engine copies what it needs to the device);
 *  - every pointer inside af_outputs_t is DEVICE memory owned by the caller
 *    (e.g. torch tensors); the engine never allocates or frees output buffers;
 *  - all entry points return 0 on success, a negative af_status otherwise, and
 *    leave a thread-local message retrievable through af_last_error();
 *  - one host thread per engine AT A TIME; calls are synchronous with respect to the
 *    returned buffers (the engine's own HIP streams, synchronised before return);
 *    different engines may be called from different host threads at once: an
 *    engine owns its streams, counter block and scratch buffers, the library keeps
 *    no mutable global state besides the RCCL symbol table af_comm_load fills, and
 *    the error message is per thread (tests/test_gpu_flow.py::
 *    test_sweeps_in_flight_on_host_threads_equal_their_lone_runs);
 *  - capacity overflow is reported per scenario in counts[AF_CNT_FLAGS] and is
 *    never a silent drop.
 */
#ifndef ASYNCFLOW_HIP_H
#define ASYNCFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AF_ABI_VERSION 7

/* ---- status codes ------------------------------------------------------ */
enum af_status {
    AF_OK = 0,
    AF_ERR_INVALID = -1,     /* malformed plan / sweep / outputs            */
    AF_ERR_NO_DEVICE = -2,   /* no HIP device, or device index out of range */
    AF_ERR_HIP = -3,         /* a HIP runtime call failed                   */
    AF_ERR_CAPACITY = -4,    /* requested capacities exceed what a wave can hold */
    AF_ERR_ABI = -5          /* caller built against another AF_ABI_VERSION */
};

/* ---- enumerations mirrored from src/asyncflow/config/constants.py ------- */
/* Distribution (constants.py:39-52) */
enum af_dist {
    AF_DIST_POISSON = 0,
    AF_DIST_NORMAL = 1,
    AF_DIST_LOG_NORMAL = 2,
    AF_DIST_EXPONENTIAL = 3,
    AF_DIST_UNIFORM = 4
};
/* SystemNodes that can be an edge target (constants.py:155-166) */
enum af_node_kind { AF_NODE_CLIENT = 0, AF_NODE_LB = 1, AF_NODE_SERVER = 2 };
/* LbAlgorithmsName (constants.py:144-148) */
enum af_lb_algo { AF_LB_ROUND_ROBIN = 0, AF_LB_LEAST_CONNECTIONS = 1 };
/* EndpointStepCPU / EndpointStepIO (constants.py:58-89); RAM steps are folded
 * into ep_ram at lowering time (server.py:106-110 sums them up front). */
enum af_step_kind { AF_STEP_CPU = 0, AF_STEP_IO = 1 };
/* SampledMetricName (constants.py:195-205) as a bit mask */
enum af_metric_bit {
    AF_METRIC_READY_QUEUE_LEN = 1u << 0,
    AF_METRIC_EVENT_LOOP_IO_SLEEP = 1u << 1,
    AF_METRIC_RAM_IN_USE = 1u << 2,
    AF_METRIC_EDGE_CONCURRENT_CONNECTION = 1u << 3
};

/* ---- the lowered scenario (shared by every scenario of a sweep) -------- */
typedef struct af_plan {
    uint32_t abi_version;  /* = AF_ABI_VERSION                                   */
    uint32_t struct_size;  /* = sizeof(af_plan_t)                                */

    /* SimulationSettings (schemas/settings/simulation.py:13-44) */
    double total_time;     /* total_simulation_time (s)                          */
    double sample_period;  /* sample_period_s (s)                                */
    uint32_t metrics_mask; /* af_metric_bit                                      */

    /* RqsGenerator (schemas/workload/rqs_generator.py:10-59) */
    uint32_t gen_users_dist;   /* AF_DIST_POISSON | AF_DIST_NORMAL               */
    double gen_users_mean;     /* avg_active_users.mean                          */
    double gen_users_sigma;    /* avg_active_users.variance (used AS sigma)      */
    double gen_rpm_mean;       /* avg_request_per_minute_per_user.mean           */
    double gen_window_s;       /* user_sampling_window                           */
    int32_t gen_out_edge;      /* edge index leaving the generator               */

    /* topology (schemas/topology/{nodes,edges,graph}.py) */
    uint32_t n_edges;
    uint32_t n_servers;
    int32_t client_out_edge;   /* edge index leaving the client                  */
    uint32_t has_lb;
    uint32_t lb_algo;          /* af_lb_algo                                     */
    uint32_t n_lb_edges;
    const int32_t* lb_edges;   /* [n_lb_edges] LB out-edges in payload order     */

    const uint8_t* edge_target_kind;  /* [n_edges] af_node_kind                  */
    const int32_t* edge_target_idx;   /* [n_edges] server index (or 0)           */
    const uint8_t* edge_dist;         /* [n_edges] af_dist of latency            */
    const double* edge_mean;          /* [n_edges] latency.mean                  */
    const double* edge_sigma;         /* [n_edges] latency.variance (AS sigma)   */
    const double* edge_dropout;       /* [n_edges] dropout_rate                  */

    const uint32_t* srv_cores;        /* [n_servers] cpu_cores                   */
    const double* srv_ram_mb;         /* [n_servers] ram_mb                      */
    const int32_t* srv_out_edge;      /* [n_servers] edge leaving the server     */
    const uint32_t* srv_ep_begin;     /* [n_servers+1] CSR into endpoints        */

    uint32_t n_endpoints;
    const uint32_t* ep_step_begin;    /* [n_endpoints+1] CSR into steps          */
    const double* ep_ram;             /* [n_endpoints] sum of necessary_ram      */

    uint32_t n_steps;
    const uint8_t* step_kind;         /* [n_steps] af_step_kind                  */
    const double* step_time;          /* [n_steps] cpu_time | io_waiting_time    */

    /* EventInjection timelines, pre-sorted exactly like
     * runtime/events/injection.py:142-151; *_time is the simulation clock at
     * which the mark is applied (relative waits re-accumulated, :181-188). */
    uint32_t n_edge_marks;
    const double* emark_time;         /* [n_edge_marks]                          */
    const int32_t* emark_edge;        /* [n_edge_marks] edge index               */
    const double* emark_delta;        /* [n_edge_marks] +spike_s start / -spike_s end */
    uint32_t n_srv_marks;
    const double* smark_time;         /* [n_srv_marks]                           */
    const int32_t* smark_lb_edge;     /* [n_srv_marks] LB out-edge index, -1 = not behind LB */
    const uint8_t* smark_down;        /* [n_srv_marks] 1 = SERVER_DOWN, 0 = SERVER_UP */
} af_plan_t;

/* ---- per-scenario inputs ------------------------------------------------ */
enum af_param {
    AF_PARAM_GEN_USERS_MEAN = 0,
    AF_PARAM_GEN_USERS_SIGMA = 1,
    AF_PARAM_GEN_RPM_MEAN = 2,
    AF_PARAM_EDGE_MEAN = 3,     /* index = edge  */
    AF_PARAM_EDGE_SIGMA = 4,    /* index = edge  */
    AF_PARAM_EDGE_DROPOUT = 5,  /* index = edge  */
    AF_PARAM_STEP_TIME = 6,     /* index = step  */
    /* ABI 4 -- sweeps over the sampling window, server resources and the injected events
     * (schemas/workload/rqs_generator.py:33-45, schemas/topology/nodes.py:58-69, schemas/events/injection.py:25-119).
     * Timeline columns address mark SLOTS of the plan's pre-sorted timelines: a scenario whose event
     * times sort differently passes its own (time, delta, edge) per slot, so any order is exact. */
    AF_PARAM_GEN_WINDOW = 7,    /* user_sampling_window (s)                                  */
    AF_PARAM_SRV_CORES = 8,     /* index = server; integral value within 1..65535            */
    AF_PARAM_SRV_RAM_MB = 9,    /* index = server                                            */
    AF_PARAM_EMARK_TIME = 10,   /* index = edge-mark slot: emark_time                        */
    AF_PARAM_EMARK_DELTA = 11,  /*                         emark_delta (+spike_s / -spike_s) */
    AF_PARAM_EMARK_EDGE = 12,   /*                         emark_edge (integral value)       */
    AF_PARAM_SMARK_TIME = 13,   /* index = server-mark slot: smark_time                      */
    AF_PARAM_SMARK_LB_EDGE = 14,/*                           smark_lb_edge (-1 = not behind the LB) */
    AF_PARAM_SMARK_DOWN = 15,   /*                           smark_down (0 / 1)              */
    AF_PARAM_COUNT_ = 16
};

typedef struct af_override {
    uint32_t param;        /* af_param                                         */
    uint32_t index;        /* entity index for the indexed params, else 0      */
    const double* values;  /* HOST [n_scenarios]                               */
} af_override_t;

typedef struct af_sweep {
    uint32_t n_scenarios;
    const uint64_t* seeds;            /* HOST [n_scenarios] Philox keys          */
    uint32_t n_overrides;
    const af_override_t* overrides;   /* HOST [n_overrides]                      */
    uint32_t draw_capacity;           /* upper bound on generated requests per scenario: the
                                         engine pre-generates that many random draws per
                                         stream (0 = outputs.clock_capacity)          */
} af_sweep_t;

/* ---- outputs ------------------------------------------------------------- */
/* counts[scenario][AF_CNT_*] */
enum af_count_slot {
    AF_CNT_GENERATED = 0,  /* RqsGeneratorRuntime.id_counter                     */
    AF_CNT_COMPLETED = 1,  /* len(ClientRuntime.rqs_clock)                       */
    AF_CNT_DROPPED = 2,    /* messages lost on edges (edge.py:78-86)             */
    AF_CNT_EVENTS = 3,     /* request-events: arrivals + deliveries + cpu/io step ends */
    AF_CNT_TICKS = 4,      /* sampler ticks taken (collector.py:50-66)           */
    AF_CNT_FLAGS = 5,      /* af_flag bits                                       */
    AF_CNT_MAX_LIVE = 6,   /* high-water mark of live requests                   */
    AF_CNT_MARKS = 7,      /* injection marks applied                            */
    AF_CNT_SLOTS = 8
};
enum af_flag {
    AF_FLAG_POOL_OVERFLOW = 1u << 0,   /* more live requests than request_capacity */
    AF_FLAG_FIFO_OVERFLOW = 1u << 1,   /* a CPU/RAM wait queue exceeded fifo_capacity */
    AF_FLAG_CLOCK_OVERFLOW = 1u << 2,  /* more completions than clock_capacity     */
    AF_FLAG_TICK_OVERFLOW = 1u << 3,   /* more ticks than tick_capacity            */
    AF_FLAG_RAM_STARVED = 1u << 4,     /* a server's RAM queue is blocked for good: a request needs more RAM than
                                          ram_mb (server.py:146-149), or -- fractional needs only -- a RAM put that
                                          simpy refuses by one rounding faces a waiter that does not fit, so that
                                          neither can ever move.  The engine does what the reference does (every
                                          later RAM request of that server waits for ever; tests/golden/ram_starved_t30,
                                          ram_put_deadlock_t20 are the reference's own output); informational */
    AF_FLAG_TIME_TIE = 1u << 5,        /* a zero-delay Timeout that is not a delivery to the
                                          client was created while other zero-time steps were
                                          pending: SimPy may order those steps differently.
                                          Needs a step time or an exponential / log-normal /
                                          uniform latency that does not advance the f64 clock
                                          (probability ~2^-53 per draw): instants shared by
                                          several timed events and zero-latency hops between
                                          servers (poisson / truncated normal) are NOT flagged,
                                          they follow SimPy's event order exactly
                                          (DESIGN.md "Ties"; informational) */
    AF_FLAG_DRAW_OVERFLOW = 1u << 6,   /* more arrivals than draw_capacity            */
    /* (bits 7..12 are internal to the engine and never visible in the outputs) */
    AF_FLAG_NEGATIVE_DELAY = 1u << 13  /* a message was sent with transit + spike < 0: the residue of overlapping
                                          spikes' += / -= (injection.py:191-198) under a zero transit time.  The
                                          reference raises there (simpy: ValueError "Negative delay", edge.py:107);
                                          the engine delivers at now + (transit + spike) and reports the scenario:
                                          asyncflow_amd.SimulationRunner raises the same ValueError for it */
    /* (bit 14 was AF_FLAG_RAM_PUT_BLOCKED in ABI 6: simpy's waiting Container.put is modelled since ABI 7, never reported) */
};

typedef struct af_outputs {
    /* RqsClock list (metrics/client.py:9-18): (start, finish) per completion,
     * in completion order.  DEVICE [n_scenarios][clock_capacity][2] f64. */
    uint32_t clock_capacity;
    double* clock;
    /* Sampled series (metrics/collector.py:50-66).
     * DEVICE [n_scenarios][tick_capacity][af_series_pitch(plan)] 4-byte words (one
     * 16-byte aligned row per tick; pitch = n_series rounded up to 4), series order:
     *   e in [0,n_edges):            edge_concurrent_connection (int32)
     *   n_edges + 3*s + 0:           ready_queue_len   of server s (int32)
     *   n_edges + 3*s + 1:           event_loop_io_sleep of server s (int32)
     *   n_edges + 3*s + 2:           ram_in_use        of server s (float32)
     * May be NULL (series not stored; ticks still counted). */
    uint32_t tick_capacity;
    uint32_t* samples;
    /* DEVICE [n_scenarios][AF_CNT_SLOTS] u32 */
    uint32_t* counts;
    /* Optional summary accumulated by the next-event kernel itself, for sweeps whose rqs_clock
     * (16 B per completion) is not wanted: per scenario a linear latency histogram over
     * [0, online_hist_max) whose last bin also takes the overflow (same binning as
     * af_summary_t.hist) and the completions per 1-s window (k-1, k] (analyzer.py:112-121).
     * DEVICE u32 arrays ZEROED BY THE CALLER; NULL = off.  Exact integer counts; percentiles read
     * from the histogram are accurate to one bin. */
    uint32_t online_hist_bins;
    double online_hist_max;
    uint32_t* online_hist;      /* [n_scenarios][online_hist_bins] */
    uint32_t online_rps_buckets;
    uint32_t* online_rps;       /* [n_scenarios][online_rps_buckets] */
} af_outputs_t;

#define AF_MAX_REQUEST_CAPACITY 65535u
#define AF_MAX_FIFO_CAPACITY 1048576u /* rounded up to a power of two by the engine.  (ABI <= 6: 16 384.  The reference's simpy
                                         Container queues have no bound -- server.py:146-149, 210-227 --; a saturated server's
                                         backlog grows with the horizon, and 2^20 waiters per server queue is what a scenario
                                         may cost in HBM: 32 B per slot and server and queue pair) */

typedef struct af_engine_options {
    uint32_t request_capacity;  /* live requests per scenario (0 = engine default, <= AF_MAX_REQUEST_CAPACITY) */
    uint32_t fifo_capacity;     /* waiters per server queue   (0 = engine default, <= AF_MAX_FIFO_CAPACITY)   */
    uint32_t force_global_state;/* 1 = keep per-scenario state in HBM even if it fits LDS */
    uint32_t lanes_per_wave;    /* scenarios per wavefront: power of two <= 64, 0 = auto
                                   (few scenarios are spread over many narrow waves)    */
    uint32_t draw_memory_mb;    /* HBM budget for the pre-generated draws of one chunk of the
                                   sweep, MiB (0 = min(160 GiB, 60 % of the free memory));
                                   larger sweeps run as several chunks                   */
    uint32_t expect_shared_instants; /* 1 = start with the kernel variant that replays SimPy's
                                   event order at instants shared by several timed events
                                   (~20 % slower).  0 = lean variant first; scenarios that
                                   meet such an instant are handed over to the other variant
                                   and the engine remembers it for its later runs          */
    /* Stage-parallel ("flow") kernel: one wavefront per scenario moves 64 requests per step through
     * the stations of a feed-forward request path (generator -> client -> [LB, round robin or least connections ->] servers
     * with IO* CPU* IO* endpoints -> client).  Bit-identical to the next-event kernels; a scenario
     * it cannot express (two events of one station at one instant, a list / tick-ring overflow) is
     * simulated again by them.  af_engine_flow_reason() tells why a plan is outside its range. */
    uint32_t flow_mode;         /* 0 = use it whenever the plan is in range -- plans whose servers need the
                                   event-by-event station (several endpoints per server, step programs that come
                                   back to the core queue) only for sweeps of <= 8 scenarios, above that the
                                   next-event kernels are faster for them; 1 = never; 2 = whenever in range  */
    uint32_t flow_list_entries; /* capacity of each station's message list: 64, 128 or 256
                                   (0 = from the expected number of messages in flight)         */
    uint32_t flow_ring_rows;    /* rows of the LDS ring of per-tick differences (power of two);
                                   0 = auto (a window of a few batches of arrivals + the time a
                                   request spends inside a server; edges slower than the ring
                                   reaches are handled by the receiving station);
                                   AF_FLOW_RING_IN_HBM = keep the differences in the sample rows
                                   in HBM (no limit on how far an interval reaches)            */
} af_engine_options_t;
#define AF_FLOW_RING_IN_HBM 0xFFFFFFFFu

typedef struct af_stats {
    double kernel_ms;           /* HIP-event time of the next-event kernel (af_des_kernel) */
    double pregen_ms;           /* HIP-event time of the draw pre-generation kernels       */
    double h2d_ms;              /* seeds/overrides upload                          */
    double summary_ms;          /* last af_engine_summarize (both kernels)         */
    uint64_t draw_bytes;        /* size of the pre-generated draw arrays (HBM)     */
    uint64_t state_bytes_per_scenario;
    uint32_t state_in_lds;      /* 1 = LDS-resident state, 0 = HBM-resident        */
    uint32_t lds_bytes_per_wave;
    uint32_t waves;             /* workgroups launched (one wavefront each)        */
    uint32_t lanes_per_wave;    /* scenarios carried by each wavefront             */
    uint32_t chunks;            /* sub-launches the sweep was split into           */
    uint32_t specialised_launches; /* next-event launches that used kernels from af_engine_set_kernels */
    uint32_t shared_instant_scenarios; /* scenarios in which two timed events shared an instant:
                                   simulated a second time by the kernel variant that replays
                                   SimPy's event-by-event order (0 when the engine started
                                   with that variant)                              */
    uint32_t request_capacity;
    uint32_t fifo_capacity;
    /* stage-parallel kernel (last af_engine_run) */
    double flow_kernel_ms;         /* HIP-event time of the stage-parallel kernel, 0 = not used      */
    uint32_t flow_scenarios;       /* scenarios it was launched on                                  */
    uint32_t flow_fallback;        /* ... of which handed back by the first launch, by reason:        */
    uint32_t flow_fallback_tie;    /*   two events of one station (or an event and a tick / mark) at one instant */
    uint32_t flow_fallback_list;   /*   more messages pending at a station than flow_list_entries   */
    uint32_t flow_fallback_ring;   /*   a request stayed in its server (or, fast-hop launches, a message on its edge)
                                        longer than the tick ring reaches                          */
    uint32_t flow_fallback_ram;    /*   RAM admission the recurrence cannot express                 */
    uint32_t flow_retried;         /* handed-back scenarios run again by the kernel's most tolerant instantiation
                                      (256-entry lists carrying send times, tick differences in HBM)             */
    uint32_t flow_to_next_event;   /* scenarios finally simulated by the next-event kernels                       */
    uint32_t flow_list_entries;    /* layout used                                                   */
    uint32_t flow_ring_rows;       /* (0 = differences kept in HBM)                                 */
    uint32_t flow_lds_bytes;       /* LDS per wavefront                                             */
    uint32_t jit_fallbacks;        /* launches that wanted plan-specialised kernels but ran the generic ones */
    double gather_ms;              /* last af_engine_gather (HIP events around the grouped all-gather)        */
    uint32_t pregen_group;         /* arrival pre-generation of the stage-parallel path: scenarios per workgroup of
                                      af_arrival_groups, 0 = the row kernel                                   */
    uint32_t summary_overlapped;   /* af_engine_run_summarized: scenarios whose analyzer ran on a second stream beside the
                                      last residency round of the stage-parallel kernel (0: everything after the run) */
    double summary_beside_ms;      /* ... HIP-event time of those analyzer kernels; summary_ms is then what was NOT hidden */
} af_stats_t;

typedef struct af_engine af_engine_t;

/* Replaces: SimulationRunner.__init__ + _build_* (simulation_runner.py:52-294).
 * device = AF_DEVICE_PLAN_ONLY makes a planning-only engine: it owns no device state and no HIP call is made for it, so it
 * can be created on a machine without a GPU.  It answers af_engine_jit_spec for sweeps of the stage-parallel kernel (the
 * spec is a pure function of plan, sweep and output shape), af_engine_flow_reason and af_engine_stats; every call that
 * would touch the device returns AF_ERR_NO_DEVICE.  Used to pre-build plan-specialised kernels where hipcc is (the build
 * machine) for a box where it may not be. */
#define AF_DEVICE_PLAN_ONLY (-1)
int af_engine_create(const af_plan_t* plan, int device, const af_engine_options_t* opts,
                     af_engine_t** out);
/* Plan-specialised kernels (optional).  The library contains generic next-event kernels; for long
 * sweeps a host may compile asyncflow_amd/csrc/engine.hip once more with the plan's shape as
 * compile-time constants (state offsets become instruction immediates, loops over edges / servers /
 * series unroll: ~8 % less kernel time on the 10k LB-2 sweep) and hand the code object over:
 *   af_engine_jit_spec   writes the hipcc -D flags that describe what af_engine_run(sweep, out) would
 *                        launch (NUL-terminated, deterministic: usable as a cache key); build with
 *                        hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off <flags>
 *   af_engine_set_kernels  loads the code object; launches whose spec differs from `spec` (another
 *                        sweep shape, the second pass over a subset) keep using the generic kernels.
 *                        NULL image: unload.  Results are bit-identical either way. */
int af_engine_jit_spec(af_engine_t* engine, const af_sweep_t* sweep, const af_outputs_t* out, char* buf, size_t cap);
int af_engine_set_kernels(af_engine_t* engine, const char* spec, const void* image, size_t size);

/* Batched analyzer on the device (replaces ResultsAnalyzer._process_event_metrics,
 * src/asyncflow/metrics/analyzer.py:83-126, for every scenario of a finished af_engine_run):
 * per-scenario latency statistics in LatencyKey order
 *   {total_requests, mean, median, std_dev, p95, p99, min, max}
 * (order statistics exact, numpy 'linear' interpolation; scenarios without completions: total 0,
 * the rest NaN), the 1-s throughput windows (k-1, k], k = 1..rps_buckets, an optional linear latency
 * histogram over [0, hist_max) whose last bin also takes the overflow, and the mean / maximum of
 * every sampled series.  All buffers are DEVICE memory owned by the caller; NULL skips an output. */
typedef struct af_summary_t {
    uint32_t n_scenarios;
    uint32_t rps_buckets;   /* floor(total_time)                                        */
    uint32_t hist_bins;     /* <= 8192                                                  */
    double hist_max;        /* seconds                                                  */
    double* stats;          /* [n][8] f64                                               */
    float* rps;             /* [n][rps_buckets] f32                                     */
    uint32_t* hist;         /* [n][hist_bins] u32                                       */
    double* series_mean;    /* [n][af_series_count] f64 (needs outputs.samples)         */
    uint32_t* series_max;   /* [n][af_series_count] u32                                 */
} af_summary_t;

/* `out` is the af_outputs_t the run filled (clock + counts are required, samples only for the
 * series outputs).  Synchronous like af_engine_run. */
int af_engine_summarize(af_engine_t* engine, const af_outputs_t* out, const af_summary_t* summary);

/* af_engine_run followed by af_engine_summarize, in one call and with the same results (replaces SimulationRunner.run +
 * ResultsAnalyzer.process_all_metrics, simulation_runner.py:349-376 + analyzer.py:75-81, for the whole sweep).
 * summary->n_scenarios must equal sweep->n_scenarios.  Where the sweep is ONE launch of the stage-parallel kernel, the analyzer
 * starts on a second stream as soon as the scenarios of the kernel's full residency rounds have finished (a counter the kernel's
 * waves bump; hipStreamWaitValue32) and runs beside its last, partial round (af_stats_t.summary_overlapped / summary_beside_ms);
 * scenarios it meets unfinished, and everything else, are analysed after the simulation, as the two separate calls would.
 * AF_NO_SUMMARY_OVERLAP=1 in the environment: always the latter (measurements). */
int af_engine_run_summarized(af_engine_t* engine, const af_sweep_t* sweep, const af_outputs_t* out, const af_summary_t* summary);

/* ---- the one collective of a multi-GPU sweep ---------------------------------------------
 * Scenarios shard over the GPUs of a node with no exchange during simulation; at the end every rank
 * contributes the summaries of ITS scenarios and receives everybody's: one grouped RCCL all-gather
 * over xGMI on the engine's stream.  The reference has nothing to replace here (single process, one
 * env per run: docs/api/high-level/runner.md:203-207); this is the Monte-Carlo roadmap item
 * (ROADMAP.md:23-29) across devices.
 *
 * RCCL is resolved at run time, NOT linked: af_comm_load(path) opens a specific librccl (a host that
 * already uses RCCL -- e.g. PyTorch, whose copy is torch/lib/librccl.so -- must name that one so that a
 * single RCCL lives in the process); with NULL the engine takes the RCCL already visible in the process,
 * then $ASYNCFLOW_RCCL_LIB, then librccl.so.1.  `comm` is a ncclComm_t: the host's own, or one made by
 * af_comm_init_rank from the 128-byte id rank 0 obtained with af_comm_unique_id and shared out of band. */
#define AF_COMM_ID_BYTES 128
int af_comm_load(const char* librccl_path);
int af_comm_unique_id(void* id_out /* AF_COMM_ID_BYTES */);
int af_comm_init_rank(const void* id, int world_size, int rank, int device, void** comm_out);
void af_comm_destroy(void* comm);
/* What RCCL itself says about a communicator: ncclCommCount / ncclCommUserRank (either output may be NULL).  A
 * multi-GPU bench line carries these next to WORLD_SIZE so that the first run on N GPUs certifies that RCCL saw N ranks. */
int af_comm_count(void* comm, int* world_size_out, int* rank_out);
/* All-gather of per-scenario summaries: every non-NULL array of `local` (n_scenarios rows: the rank's
 * shard, padded by the caller to the same n on every rank) into the array of the same name in `gathered`
 * (world_size * n_scenarios rows, rank order).  rps_buckets / hist_bins give the row lengths; series
 * arrays have af_series_count(plan) columns.  DEVICE memory owned by the caller; synchronous. */
int af_engine_gather(af_engine_t* engine, void* rccl_comm, int world_size, const af_summary_t* local,
                     const af_summary_t* gathered);

/* Replaces: _start_* + env.run(until=T) (simulation_runner.py:301-369) for
 * sweep->n_scenarios independent scenarios. */
int af_engine_run(af_engine_t* eng, const af_sweep_t* sweep, const af_outputs_t* out);
int af_engine_stats(const af_engine_t* eng, af_stats_t* stats);
/* Empty string: the stage-parallel kernel can run this engine's plan; otherwise why not (the plan then
 * always runs on the next-event kernels).  The pointer stays valid until the engine is destroyed. */
const char* af_engine_flow_reason(const af_engine_t* eng);
void af_engine_destroy(af_engine_t* eng);

/* Number of sampler ticks env.run(until=T) takes for (period, T): repeated f64
 * addition from 0, tick at t < T only (collector.py:50-53). */
uint32_t af_tick_count(double sample_period, double total_time);
uint32_t af_series_count(const af_plan_t* plan);  /* n_edges + 3*n_servers */
uint32_t af_series_pitch(const af_plan_t* plan);  /* row length of outputs.samples (multiple of 4) */

const char* af_last_error(void);
int af_abi_version(void);

/* Spec probes (device): evaluate the engine's own RNG/math on the GPU so the
 * parity tests can pin them against the oracle bit for bit.
 *   kind 0: uniform(seed,stream,index,j)
 *   kind 1: log(x)   kind 2: exp(x)   kind 3: norminv(x)   kind 4: sqrt(x)
 *   kind 5: x / y (x = in[i], y = in2[i])
 * `in`, `in2`, `out` are HOST arrays of n doubles (kind 0: in[i] = index,
 * in2[i] = stream * 65536 + j, both exact small integers). */
int af_probe_math(int device, int kind, uint64_t seed, const double* in, const double* in2,
                  double* out, size_t n);

/* Measurement probe (device): one launch of n_waves wavefronts, each filling its own contiguous region of
 * bytes_per_wave bytes exactly once with the store shape of
 *   pattern 0: wide coalesced stores (16 B per lane, 1 KB per instruction) -- the shape rocprofv3's WRITE_SIZE is
 *              calibrated for (MI355X_MICROARCH.md);
 *   pattern 1: the stage-parallel kernel's rqs_clock store: 16 B per lane over `lanes` consecutive lanes (0 = 57,
 *              the average batch of BASELINE config 2), batch after batch;
 *   pattern 2: its sampled-series store for a 12-word pitch: 4 B per lane over 60 lanes = five 48-byte rows.
 * Run under `rocprofv3 --pmc WRITE_SIZE` (scripts/calibrate_write_size.py) it tells how many counter units a byte
 * of each shape costs, i.e. whether WRITE_SIZE above the output size is traffic or counting granularity.
 * ms_out (may be NULL): HIP-event time of the launch. */
int af_probe_store(int device, int pattern, uint32_t n_waves, uint64_t bytes_per_wave, uint32_t lanes, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* ASYNCFLOW_HIP_H */
