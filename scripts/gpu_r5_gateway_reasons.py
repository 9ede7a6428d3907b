"""Why does the first launch hand scenarios of the servers-in-front-of-the-LB family back?  (reasons per payload)"""
import random
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.runner import SimulationRunner  # noqa: E402
from oracle.scenarios import gateway_lb  # noqa: E402

for k in range(24):
    rng = random.Random(99000 + k)
    kw = dict(front=rng.choice((1, 2)), algo=rng.choice(("round_robin", "least_connection")), users=rng.choice((60, 150, 300)),
              horizon=10, general=rng.random() < 0.4, backend=rng.random() < 0.5, spike=rng.random() < 0.5)
    res = SimulationRunner(simulation_input=gateway_lb(**kw), seeds=np.arange(8, dtype=np.uint64) + 1000 * k + 7, on_negative_delay="flag").run()
    st = res.engine_stats
    print(k, kw, "fallback", st.flow_fallback, "tie", st.flow_fallback_tie, "list", st.flow_fallback_list, "ring", st.flow_fallback_ring,
          "ram", st.flow_fallback_ram, "retried", st.flow_retried, "next", st.flow_to_next_event, "lds", st.flow_lds_bytes,
          "entries", st.flow_list_entries, "rows", st.flow_ring_rows, flush=True)
