import sys, numpy as np
sys.path.insert(0, '.')
from asyncflow_amd import _abi
from asyncflow_amd.runner import SimulationRunner
from asyncflow_amd.workloads import lb_two_servers
from oracle.scenarios import shared_backend
for name, p in (("chain", shared_backend(horizon=120)), ("lb2", lb_two_servers(horizon=120))):
    seeds = 0x5EED0000 + np.arange(512, dtype=np.uint64)
    res = SimulationRunner(simulation_input=p, seeds=seeds, specialise=True).run()
    st = res.engine_stats
    c = res.counts
    print(name, "rounds/scenario", c[:, _abi.CNT_MAX_LIVE].mean(), "generated", c[:, 0].mean(), "events", c[:, 3].mean(), "spec launches", st.specialised_launches, "ring rows", st.flow_ring_rows, "flow ms", st.flow_kernel_ms)
