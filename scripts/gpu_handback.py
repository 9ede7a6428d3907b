"""What a hand-back costs: 8 LB-2 scenarios, T = 600 s, forced off the lean launch by a 2-row tick ring, are
re-simulated by the second-chance launch of the stage-parallel kernel (one WAVE per scenario: 256-entry lists with send times,
tick differences in HBM) -- against the same 8 scenarios on the next-event kernels (one LANE per scenario)."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.plan import lower  # noqa: E402
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from asyncflow_amd.workloads import lb_two_servers  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

p = lb_two_servers()
seeds = 0x5EED0000 + np.arange(8, dtype=np.uint64)
out = {}
for name, kw in (("lean", {}), ("forced_hand_back", {"flow_ring_rows": 2}), ("next_event", {"flow": False})):
    SimulationRunner(simulation_input=p, seeds=seeds, **kw).run()      # warm
    res = SimulationRunner(simulation_input=p, seeds=seeds, **kw).run()
    st = res.engine_stats
    out[name] = {"kernel_ms": float(st.kernel_ms), "flow_kernel_ms": float(st.flow_kernel_ms), "handed_back_by_first_launch": int(st.flow_fallback),
                 "retried_on_second_chance": int(st.flow_retried), "to_next_event": int(st.flow_to_next_event)}
    want = ol.simulate(lower(p), int(seeds[5]))
    out[name]["parity_scenario_5"] = bool(np.array_equal(res[5].rqs_clock, want.clock) and np.array_equal(res[5]._samples, want.samples))  # noqa: SLF001
print(json.dumps(out))
