"""profiles/rNN/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round2.sh.

    python scripts/make_traffic_json.py profiles/r02

HBM bytes per launch and kernel = 2 x FETCH_SIZE + WRITE_SIZE, KB = 1024 B (gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section; WRITE_SIZE is uncalibrated for narrow rows).
"""
import collections
import csv
import json
import sys

out = sys.argv[1]


def short(k: str) -> str:
    for name in ("af_flow_kernel", "af_pregen_arrivals", "af_pregen_edges", "af_summary_kernel", "af_series_kernel", "af_des_kernel", "af_jit"):
        if name in k:
            return name
    return k[:30]


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
    return acc, n


f, fn = per_kernel(f"{out}/pmc3.csv")
w, wn = per_kernel(f"{out}/pmc4.csv")
stats = {r["Name"]: r for r in csv.DictReader(open(f"{out}/kernel_stats_trace.csv"))}


def avg_ms(sub):
    for k, r in stats.items():
        if sub in k:
            return float(r["AverageNs"]) / 1e6
    return None


def entry(k):
    fe = f[k].get("FETCH_SIZE", 0.0) / max(fn[(k, "FETCH_SIZE")], 1)
    wr = w[k].get("WRITE_SIZE", 0.0) / max(wn[(k, "WRITE_SIZE")], 1)
    return {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "read_bytes": 2 * fe * 1024, "write_bytes": wr * 1024,
            "bytes_per_launch": 2 * fe * 1024 + wr * 1024, "avg_ms": avg_ms(k.replace("af_", ""))}


tj = {"kernel": "af_flow_kernel",
      "command": "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --generic-kernels (10 000 LB-2 replicas, T = 600 s, full outputs)",
      "correction": "gfx950: FETCH_SIZE x2 (calibrated for wide coalesced reads), KB = 1024 B (MI355X_MICROARCH.md, HBM / rocprofv3 "
                    "section); separate --pmc passes for FETCH_SIZE and WRITE_SIZE; WRITE_SIZE uncalibrated for rows narrower than a request"}
tj.update(entry("af_flow_kernel"))
tj["other_kernels"] = {k: entry(k) for k in ("af_pregen_arrivals", "af_summary_kernel", "af_series_kernel") if k in f or k in w}
json.dump(tj, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(tj, indent=1))
