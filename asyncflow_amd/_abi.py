"""ctypes mirror of include/asyncflow_hip.h (the C ABI of the HIP engine).

Field order and types must match the header exactly; tests/test_abi.py checks
``sizeof`` against the values the shared library reports and that every symbol
declared in the header is exported.
"""

from __future__ import annotations

import ctypes as C

AF_ABI_VERSION = 7

# af_status
MAX_REQUEST_CAPACITY = 65535   # include/asyncflow_hip.h AF_MAX_REQUEST_CAPACITY
MAX_FIFO_CAPACITY = 1 << 20    # AF_MAX_FIFO_CAPACITY (rounds 1-5: 16 384)
COMM_ID_BYTES = 128            # AF_COMM_ID_BYTES
AF_OK = 0
AF_ERR_INVALID = -1
AF_ERR_NO_DEVICE = -2
DEVICE_PLAN_ONLY = -1          # AF_DEVICE_PLAN_ONLY
AF_ERR_HIP = -3
AF_ERR_CAPACITY = -4
AF_ERR_ABI = -5

# af_dist (src/asyncflow/config/constants.py:39-52)
DIST_CODES = {"poisson": 0, "normal": 1, "log_normal": 2, "exponential": 3, "uniform": 4}
# af_node_kind
NODE_CLIENT, NODE_LB, NODE_SERVER = 0, 1, 2
# af_lb_algo (constants.py:144-148)
LB_CODES = {"round_robin": 0, "least_connection": 1}
# af_step_kind
STEP_CPU, STEP_IO = 0, 1
# af_metric_bit (constants.py:195-205)
METRIC_BITS = {
    "ready_queue_len": 1,
    "event_loop_io_sleep": 2,
    "ram_in_use": 4,
    "edge_concurrent_connection": 8,
}
# af_param
PARAM_CODES = {
    "gen_users_mean": 0,
    "gen_users_sigma": 1,
    "gen_rpm_mean": 2,
    "edge_mean": 3,
    "edge_sigma": 4,
    "edge_dropout": 5,
    "step_time": 6,
    "gen_window": 7,
    "srv_cores": 8,
    "srv_ram_mb": 9,
    "emark_time": 10,
    "emark_delta": 11,
    "emark_edge": 12,
    "smark_time": 13,
    "smark_lb_edge": 14,
    "smark_down": 15,
}
# af_count_slot
CNT_GENERATED, CNT_COMPLETED, CNT_DROPPED, CNT_EVENTS, CNT_TICKS, CNT_FLAGS, CNT_MAX_LIVE, CNT_MARKS = range(8)
CNT_SLOTS = 8
# af_flag
FLAG_POOL_OVERFLOW = 1
FLAG_FIFO_OVERFLOW = 2
FLAG_CLOCK_OVERFLOW = 4
FLAG_TICK_OVERFLOW = 8
FLAG_RAM_STARVED = 16
FLAG_TIME_TIE = 32
FLAG_DRAW_OVERFLOW = 64
FLAG_NEGATIVE_DELAY = 1 << 13
FLAG_NAMES = {
    FLAG_POOL_OVERFLOW: "request pool overflow (raise request_capacity)",
    FLAG_FIFO_OVERFLOW: "server wait-queue overflow (raise fifo_capacity)",
    FLAG_CLOCK_OVERFLOW: "rqs_clock capacity overflow (raise clock_capacity)",
    FLAG_TICK_OVERFLOW: "sample capacity overflow",
    FLAG_RAM_STARVED: "a server's RAM queue is blocked for good, as in the reference: a request needs more RAM than the server owns, "
                      "or a refused fractional RAM put faces a waiter that does not fit",
    FLAG_TIME_TIE: "a zero-delay timeout was created in the middle of a zero-time cascade (SimPy may order the pending steps differently)",
    FLAG_DRAW_OVERFLOW: "more arrivals than draw_capacity (raise clock_capacity)",
    FLAG_NEGATIVE_DELAY: "a message was sent with transit + spike < 0 (the reference raises ValueError 'Negative delay')",
}
FATAL_FLAGS = (
    FLAG_POOL_OVERFLOW | FLAG_FIFO_OVERFLOW | FLAG_CLOCK_OVERFLOW | FLAG_TICK_OVERFLOW | FLAG_DRAW_OVERFLOW
)

_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pu32 = C.POINTER(C.c_uint32)
_pu8 = C.POINTER(C.c_uint8)
_pu64 = C.POINTER(C.c_uint64)


class AfPlan(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("struct_size", C.c_uint32),
        ("total_time", C.c_double),
        ("sample_period", C.c_double),
        ("metrics_mask", C.c_uint32),
        ("gen_users_dist", C.c_uint32),
        ("gen_users_mean", C.c_double),
        ("gen_users_sigma", C.c_double),
        ("gen_rpm_mean", C.c_double),
        ("gen_window_s", C.c_double),
        ("gen_out_edge", C.c_int32),
        ("n_edges", C.c_uint32),
        ("n_servers", C.c_uint32),
        ("client_out_edge", C.c_int32),
        ("has_lb", C.c_uint32),
        ("lb_algo", C.c_uint32),
        ("n_lb_edges", C.c_uint32),
        ("lb_edges", _pi32),
        ("edge_target_kind", _pu8),
        ("edge_target_idx", _pi32),
        ("edge_dist", _pu8),
        ("edge_mean", _pd),
        ("edge_sigma", _pd),
        ("edge_dropout", _pd),
        ("srv_cores", _pu32),
        ("srv_ram_mb", _pd),
        ("srv_out_edge", _pi32),
        ("srv_ep_begin", _pu32),
        ("n_endpoints", C.c_uint32),
        ("ep_step_begin", _pu32),
        ("ep_ram", _pd),
        ("n_steps", C.c_uint32),
        ("step_kind", _pu8),
        ("step_time", _pd),
        ("n_edge_marks", C.c_uint32),
        ("emark_time", _pd),
        ("emark_edge", _pi32),
        ("emark_delta", _pd),
        ("n_srv_marks", C.c_uint32),
        ("smark_time", _pd),
        ("smark_lb_edge", _pi32),
        ("smark_down", _pu8),
    ]


class AfOverride(C.Structure):
    _fields_ = [("param", C.c_uint32), ("index", C.c_uint32), ("values", _pd)]


class AfSweep(C.Structure):
    _fields_ = [
        ("n_scenarios", C.c_uint32),
        ("seeds", _pu64),
        ("n_overrides", C.c_uint32),
        ("overrides", C.POINTER(AfOverride)),
        ("draw_capacity", C.c_uint32),
    ]


class AfOutputs(C.Structure):
    _fields_ = [
        ("clock_capacity", C.c_uint32),
        ("clock", C.c_void_p),
        ("tick_capacity", C.c_uint32),
        ("samples", C.c_void_p),
        ("counts", C.c_void_p),
        ("online_hist_bins", C.c_uint32),
        ("online_hist_max", C.c_double),
        ("online_hist", C.c_void_p),
        ("online_rps_buckets", C.c_uint32),
        ("online_rps", C.c_void_p),
    ]


class AfEngineOptions(C.Structure):
    _fields_ = [
        ("request_capacity", C.c_uint32),
        ("fifo_capacity", C.c_uint32),
        ("force_global_state", C.c_uint32),
        ("lanes_per_wave", C.c_uint32),
        ("draw_memory_mb", C.c_uint32),
        ("expect_shared_instants", C.c_uint32),
        ("flow_mode", C.c_uint32),
        ("flow_list_entries", C.c_uint32),
        ("flow_ring_rows", C.c_uint32),
    ]


class AfStats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("pregen_ms", C.c_double),
        ("h2d_ms", C.c_double),
        ("summary_ms", C.c_double),
        ("draw_bytes", C.c_uint64),
        ("state_bytes_per_scenario", C.c_uint64),
        ("state_in_lds", C.c_uint32),
        ("lds_bytes_per_wave", C.c_uint32),
        ("waves", C.c_uint32),
        ("lanes_per_wave", C.c_uint32),
        ("chunks", C.c_uint32),
        ("specialised_launches", C.c_uint32),
        ("shared_instant_scenarios", C.c_uint32),
        ("request_capacity", C.c_uint32),
        ("fifo_capacity", C.c_uint32),
        ("flow_kernel_ms", C.c_double),
        ("flow_scenarios", C.c_uint32),
        ("flow_fallback", C.c_uint32),
        ("flow_fallback_tie", C.c_uint32),
        ("flow_fallback_list", C.c_uint32),
        ("flow_fallback_ring", C.c_uint32),
        ("flow_fallback_ram", C.c_uint32),
        ("flow_retried", C.c_uint32),
        ("flow_to_next_event", C.c_uint32),
        ("flow_list_entries", C.c_uint32),
        ("flow_ring_rows", C.c_uint32),
        ("flow_lds_bytes", C.c_uint32),
        ("jit_fallbacks", C.c_uint32),
        ("gather_ms", C.c_double),
        ("pregen_group", C.c_uint32),
        ("summary_overlapped", C.c_uint32),
        ("summary_beside_ms", C.c_double),
    ]


FLOW_RING_IN_HBM = 0xFFFFFFFF   # AF_FLOW_RING_IN_HBM


class AfSummary(C.Structure):
    _fields_ = [
        ("n_scenarios", C.c_uint32),
        ("rps_buckets", C.c_uint32),
        ("hist_bins", C.c_uint32),
        ("hist_max", C.c_double),
        ("stats", C.c_void_p),
        ("rps", C.c_void_p),
        ("hist", C.c_void_p),
        ("series_mean", C.c_void_p),
        ("series_max", C.c_void_p),
    ]


#: every symbol include/asyncflow_hip.h declares
EXPORTED_SYMBOLS = (
    "af_engine_create",
    "af_engine_run",
    "af_engine_summarize",
    "af_engine_run_summarized",
    "af_engine_jit_spec",
    "af_engine_set_kernels",
    "af_engine_stats",
    "af_engine_flow_reason",
    "af_engine_gather",
    "af_comm_load",
    "af_comm_unique_id",
    "af_comm_init_rank",
    "af_comm_destroy",
    "af_comm_count",
    "af_engine_destroy",
    "af_tick_count",
    "af_series_count",
    "af_series_pitch",
    "af_last_error",
    "af_abi_version",
    "af_probe_math",
    "af_probe_store",
)


def declare(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes/restypes of the C ABI to a loaded library."""
    lib.af_engine_create.argtypes = [
        C.POINTER(AfPlan), C.c_int, C.POINTER(AfEngineOptions), C.POINTER(C.c_void_p),
    ]
    lib.af_engine_create.restype = C.c_int
    lib.af_engine_run.argtypes = [C.c_void_p, C.POINTER(AfSweep), C.POINTER(AfOutputs)]
    lib.af_engine_run.restype = C.c_int
    lib.af_engine_summarize.argtypes = [C.c_void_p, C.POINTER(AfOutputs), C.POINTER(AfSummary)]
    lib.af_engine_summarize.restype = C.c_int
    lib.af_engine_run_summarized.argtypes = [C.c_void_p, C.POINTER(AfSweep), C.POINTER(AfOutputs), C.POINTER(AfSummary)]
    lib.af_engine_run_summarized.restype = C.c_int
    lib.af_engine_jit_spec.argtypes = [C.c_void_p, C.POINTER(AfSweep), C.POINTER(AfOutputs), C.c_char_p, C.c_size_t]
    lib.af_engine_jit_spec.restype = C.c_int
    lib.af_engine_set_kernels.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    lib.af_engine_set_kernels.restype = C.c_int
    lib.af_engine_stats.argtypes = [C.c_void_p, C.POINTER(AfStats)]
    lib.af_engine_stats.restype = C.c_int
    lib.af_engine_flow_reason.argtypes = [C.c_void_p]
    lib.af_engine_flow_reason.restype = C.c_char_p
    lib.af_engine_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(AfSummary), C.POINTER(AfSummary)]
    lib.af_engine_gather.restype = C.c_int
    lib.af_comm_load.argtypes = [C.c_char_p]
    lib.af_comm_load.restype = C.c_int
    lib.af_comm_unique_id.argtypes = [C.c_void_p]
    lib.af_comm_unique_id.restype = C.c_int
    lib.af_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.af_comm_init_rank.restype = C.c_int
    lib.af_comm_destroy.argtypes = [C.c_void_p]
    lib.af_comm_destroy.restype = None
    lib.af_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.af_comm_count.restype = C.c_int
    lib.af_engine_destroy.argtypes = [C.c_void_p]
    lib.af_engine_destroy.restype = None
    lib.af_tick_count.argtypes = [C.c_double, C.c_double]
    lib.af_tick_count.restype = C.c_uint32
    lib.af_series_count.argtypes = [C.POINTER(AfPlan)]
    lib.af_series_count.restype = C.c_uint32
    lib.af_series_pitch.argtypes = [C.POINTER(AfPlan)]
    lib.af_series_pitch.restype = C.c_uint32
    lib.af_last_error.argtypes = []
    lib.af_last_error.restype = C.c_char_p
    lib.af_abi_version.argtypes = []
    lib.af_abi_version.restype = C.c_int
    lib.af_probe_math.argtypes = [C.c_int, C.c_int, C.c_uint64, _pd, _pd, _pd, C.c_size_t]
    lib.af_probe_math.restype = C.c_int
    lib.af_probe_store.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    lib.af_probe_store.restype = C.c_int
    return lib
