# least-connections timing + A/B against a baseline library (asyncflow_amd/csrc/libasyncflow_hip_base.so, if present)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
lc() { python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from asyncflow_amd.runner import SimulationRunner
from asyncflow_amd.workloads import lb_two_servers, fanout8
n = 10000
for name, payload in (("lb2 rr", lb_two_servers()), ("lb2 lc", lb_two_servers(algo="least_connection")), ("fanout8 lc", fanout8())):
    if name == "fanout8 lc":
        payload["topology_graph"]["nodes"]["load_balancer"]["algorithms"] = "least_connection"
    seeds = 0x5EED0000 + np.arange(n, dtype=np.uint64)
    r = SimulationRunner(simulation_input=payload, seeds=seeds)
    r.run()
    t0 = time.perf_counter(); res = r.run(); wall = time.perf_counter() - t0
    st = res.engine_stats; ev = int(res.request_events.sum())
    print(f"[{os.environ.get('ASYNCFLOW_HIP_LIB','')[-12:]}] {name}: wall {wall*1e3:.1f} ms flow_ms {st.flow_kernel_ms:.1f} ev/s {ev/wall:.3e} list {st.flow_list_entries} lds {st.flow_lds_bytes} fb {st.flow_fallback} to_next {st.flow_to_next_event}", flush=True)
PY
}
if [ "$1" = "tests" ]; then ( timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/gputests.log 2>&1; tail -3 gpurun_out/gputests.log; fi
lc
if [ -f asyncflow_amd/csrc/libasyncflow_hip_base.so ]; then ASYNCFLOW_HIP_LIB=$PWD/asyncflow_amd/csrc/libasyncflow_hip_base.so lc; fi
