"""750 fuzzed cases of the general server station on the wave emulator (tests/hostcheck): 250 two-endpoint LB-2 seeds at
T = 120 s, 250 tie storms (dyadic step times, Poisson hops, tight RAM), 250 random topologies -- in the second-chance form
(256-/1024-entry lists with send times) and in the compact first-launch form.  Every case is exact or handed back; `_run`
raises on any difference from the oracle.  Prints the tallies (profiles/r04/fuzz_general_servers.txt)."""
import collections
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from asyncflow_amd.workloads import _endpoint, lb_two_servers  # noqa: E402
from oracle.scenarios import random_payload, tie_storm  # noqa: E402
from tests.test_flow_hostcheck import _run  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250


def report(horizon: int) -> dict:
    p = lb_two_servers(horizon=horizon)
    for s in p["topology_graph"]["nodes"]["servers"]:
        s["endpoints"].append(_endpoint("/report", [("io_db", 0.004), ("ram", 64), ("cpu_bound_operation", 0.0015),
                                                     ("io_wait", 0.006), ("cpu_bound_operation", 0.0005)]))
    return p


def tally(name: str, cases, **kw) -> None:
    c: collections.Counter = collections.Counter()
    for payload, seed in cases:
        st, info = _run(payload, seed, **kw)
        c[st if st == "exact" else "handed back: " + ", ".join(sorted(info))] += 1
    print(f"{name:58s} {dict(sorted(c.items()))}", flush=True)


robust = dict(ipl=1, ring_rows=0, robust=True, long_list_entries=1024)
compact = dict(ipl=1, ring_rows=32)
lb2 = [(report(120), 0x5EED0000 + k) for k in range(n)]
storms = [(tie_storm(random.Random(7000 + k), horizon=8), 3 * k + 1) for k in range(n)]
topo = [(random_payload(random.Random(31000 + k), horizon=8), 17 * k) for k in range(n)]
for form, kw in (("second-chance form", robust), ("compact first-launch form", compact)):
    tally(f"two-endpoint LB-2, T = 120 s, {form}", lb2, **kw)
    tally(f"tie storms, {form}", storms, **kw)
    tally(f"random topologies, {form}", topo, **kw)
print("0 different (a difference raises)")
