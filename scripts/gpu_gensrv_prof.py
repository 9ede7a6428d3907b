"""Section profile (AF_FLOW_PROF) of the general-server form of the stage-parallel kernel on the two-endpoint LB-2."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from asyncflow_amd.workloads import _endpoint, lb_two_servers  # noqa: E402

n, T = int(sys.argv[1]), int(sys.argv[2])
p = lb_two_servers(horizon=T)
for s in p["topology_graph"]["nodes"]["servers"]:
    s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015), ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
res = SimulationRunner(simulation_input=p, seeds=seeds, specialise=True).run()
st = res.engine_stats
print("kernel_ms", float(st.flow_kernel_ms), "specialised", int(st.specialised_launches), "jit_fallbacks", int(st.jit_fallbacks), "events", int(res.request_events.sum()), "lds", int(st.flow_lds_bytes))
