"""Launch order of the stage-parallel kernel on a sweep over the load (round 3): the 100 x 100 users x RTT grid exactly as
`expand_grid` writes it out -- users ascending with the index -- through SimulationRunner, T = 600 s, with the engine's
heaviest-first order and (AF_FLOW_ORDER_OFF=1, a second process) without.  Prints the flow kernel's time."""
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from asyncflow_amd.workloads import grid_users_rtt, lb_two_servers  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
a, b = grid_users_rtt(side)
p = lb_two_servers(horizon=600)
seeds = 0x5EED0000 + np.arange(a.size, dtype=np.uint64)
sweep = {"rqs_input.avg_active_users.mean": a, "topology_graph.edges[*].latency.mean": b}
best = None
for _ in range(3):
    res = SimulationRunner(simulation_input=p, seeds=seeds, sweep=sweep, specialise=True).run()
    st = res.engine_stats
    best = st.flow_kernel_ms if best is None else min(best, st.flow_kernel_ms)
print(json.dumps({"order_off": os.environ.get("AF_FLOW_ORDER_OFF"), "flow_kernel_ms": best, "pregen_ms": st.pregen_ms, "pregen_group": st.pregen_group,
                  "flow_scenarios": st.flow_scenarios, "handed_back": st.flow_to_next_event, "events": int(res.request_events.sum())}))
