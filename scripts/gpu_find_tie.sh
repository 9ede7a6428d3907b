cd $GRAFT_REPO_ROOT
AF_DEBUG=1 python - <<'PY' 2>&1 | grep -v "flow launch" | tail -12
import numpy as np, time
from asyncflow_amd.runner import SimulationRunner
from asyncflow_amd.workloads import lb_two_servers
seeds = 0x5EED0000 + np.arange(131072, dtype=np.uint64)
t0 = time.time()
res = SimulationRunner(simulation_input=lb_two_servers(), seeds=seeds, collect_clock=False, collect_samples=False,
                       online_summary={"hist_bins": 1024, "hist_max": 0.256}).run()
st = res.engine_stats
print("wall %.2f s kernel_ms %.1f flow_ms %.1f scen %d handed back %d (tie %d list %d ring %d ram %d) retried %d to_next_event %d" % (
    time.time() - t0, st.kernel_ms, st.flow_kernel_ms, st.flow_scenarios, st.flow_fallback, st.flow_fallback_tie, st.flow_fallback_list,
    st.flow_fallback_ring, st.flow_fallback_ram, st.flow_retried, st.flow_to_next_event))
PY
