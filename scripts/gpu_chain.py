"""Servers that feed servers on the stage-parallel kernel (round 4, FEAT_CHAIN): client -> LB -> {a1, a2} -> b -> client
(oracle/scenarios.py::shared_backend), replicas x T: kernel time on the stage-parallel kernel (generic and plan-specialised build)
vs the next-event kernels, hand-backs, parity of one scenario against the oracle."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.plan import lower  # noqa: E402
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
from oracle.scenarios import shared_backend  # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 120
p = shared_backend(horizon=T)
seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
out = {"scenarios": n, "horizon": T}
for name, kw in (("flow", {"specialise": False}), ("flow_specialised", {"specialise": True}), ("next_event", {"flow": False})):
    SimulationRunner(simulation_input=p, seeds=seeds[:64], **kw).run()       # warm
    res = SimulationRunner(simulation_input=p, seeds=seeds, **kw).run()
    st = res.engine_stats
    out[name] = {"kernel_ms": float(st.kernel_ms), "flow_kernel_ms": float(st.flow_kernel_ms), "flow_scenarios": int(st.flow_scenarios),
                 "handed_back": int(st.flow_to_next_event), "retried": int(st.flow_retried), "events": int(res.request_events.sum()),
                 "lds": int(st.flow_lds_bytes), "list": int(st.flow_list_entries), "reason": res.flow_reason}
    if name == "flow":
        want = ol.simulate(lower(p), int(seeds[7]))
        out["parity_scenario_7"] = bool(np.array_equal(res[7].rqs_clock, want.clock) and np.array_equal(res[7]._samples, want.samples))  # noqa: SLF001
out["speedup_vs_next_event"] = out["next_event"]["kernel_ms"] / max(out["flow_specialised"]["kernel_ms"], 1e-9)
print(json.dumps(out))
