cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/gputests.log 2>&1; tail -5 gpurun_out/gputests.log
python - <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from asyncflow_amd.runner import SimulationRunner
from asyncflow_amd.workloads import lb_two_servers
n = 10000
for algo in ("round_robin", "least_connection"):
    payload = lb_two_servers(algo=algo)
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    r = SimulationRunner(simulation_input=payload, seeds=seeds)
    r.run()
    t0 = time.perf_counter(); res = r.run(); wall = time.perf_counter() - t0
    st = res.engine_stats; ev = int(res.request_events.sum())
    print(f"{algo}: wall {wall*1e3:.1f} ms flow_ms {st.flow_kernel_ms:.1f} ev/s {ev/wall:.3e} list {st.flow_list_entries} lds {st.flow_lds_bytes} fb {st.flow_fallback} to_next {st.flow_to_next_event}", flush=True)
PY
