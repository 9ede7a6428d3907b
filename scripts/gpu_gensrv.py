"""How fast do plans with several endpoints per server run on the stage-parallel kernel (general server station; round 4: shared instants resolved in the station)?
LB-2's topology with a second endpoint on both servers (and a step program that comes back to the core), replicas x T:
kernel time on the stage-parallel kernel (general server station) vs the next-event kernels, hand-backs, parity of one scenario."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.plan import lower  # noqa: E402
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from asyncflow_amd.workloads import lb_two_servers_two_endpoints  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 120
p = lb_two_servers_two_endpoints(horizon=T)
seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
out = {}
for name, kw in (("flow", {"flow": "always", "specialise": False}), ("flow_specialised", {"specialise": True}), ("next_event", {"flow": False})):
    SimulationRunner(simulation_input=p, seeds=seeds[:64], **kw).run()       # warm
    res = SimulationRunner(simulation_input=p, seeds=seeds, **kw).run()
    st = res.engine_stats
    out[name] = {"kernel_ms": float(st.kernel_ms), "flow_scenarios": int(st.flow_scenarios), "handed_back": int(st.flow_to_next_event),
                 "flow_kernel_ms": float(st.flow_kernel_ms), "why": {k: int(getattr(st, "flow_fallback_" + k)) for k in ("tie", "list", "ring", "ram")}, "events": int(res.request_events.sum()), "lds": int(st.flow_lds_bytes)}
    if name == "flow":
        want = ol.simulate(lower(p), int(seeds[7]))
        out["parity_scenario_7"] = bool(np.array_equal(res[7].rqs_clock, want.clock) and np.array_equal(res[7]._samples, want.samples))  # noqa: SLF001
out["speedup"] = out["next_event"]["kernel_ms"] / out["flow_specialised"]["kernel_ms"]
print(json.dumps(out))
