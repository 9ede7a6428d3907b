"""Timing of the reference's spike examples (event_inj_single_server.yml / heavy_inj_single_server.yml) on one GPU:
stage-parallel path (lookahead + long lists) against the next-event kernels.  python scripts/spike_examples.py [N]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from asyncflow_amd.runner import SimulationRunner
from asyncflow_amd.workloads import single_server_with_spike

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for heavy in (False, True):
    payload = single_server_with_spike(heavy=heavy)
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    for flow in (True, False):
        r = SimulationRunner(simulation_input=payload, seeds=seeds, flow=flow)
        r.run()
        t0 = time.perf_counter()
        res = r.run()
        wall = time.perf_counter() - t0
        st = res.engine_stats
        ev = int(res.request_events.sum())
        print(f"heavy={heavy} flow={flow} n={n} wall {wall * 1e3:.1f} ms kernel {st.kernel_ms:.1f} ms flow_ms {st.flow_kernel_ms:.1f} "
              f"events {ev:.3e} ev/s {ev / wall:.3e} list {st.flow_list_entries} ring {st.flow_ring_rows} lds {st.flow_lds_bytes} "
              f"fb {st.flow_fallback} (tie {st.flow_fallback_tie} list {st.flow_fallback_list} ring {st.flow_fallback_ring} ram {st.flow_fallback_ram}) "
              f"retried {st.flow_retried} to_next {st.flow_to_next_event}", flush=True)
