"""profiles/rNN/binding_<label>.json (+ traffic_<label>.json) from the passes of scripts/profile_round5.sh.

    python scripts/make_binding_json.py <dir> [label]      label: c2 | c3 | c4 | c5 | gensrv (files pmc<i>_<label>.csv, ...);
                                                           without a label: rounds 3-4's file names (pmc<i>.csv, binding.json)

binding.json: what limits the dominant kernel (instruction issue, not HBM) -- per launch: VALU / SALU / LDS / branch
wave-instructions, VALU wave-instructions per request-event, SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x kernel cycles),
lane utilisation, wave lifetime; plus the run's own FETCH_SIZE / WRITE_SIZE (traffic.json keeps round 2's format).
`sources_sha1` is the digest of the kernel sources the profile was taken on: bench.py marks the numbers stale when the
tree differs.  Counter units: SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles, FETCH_SIZE / WRITE_SIZE in KB of
1024 B with FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
from __future__ import annotations

import collections
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
out = Path(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else ""
sfx = f"_{label}" if label else ""
N_SIMD = 1024.0
KERNELS = ("af_flow_jit", "af_flow_kernel", "af_pregen_arrivals", "af_arrival", "af_pregen_edges", "af_summary_kernel", "af_series_kernel", "af_des_kernel", "af_jit")


def short(k: str) -> str:
    for name in KERNELS:
        if name in k:
            return name
    return k[:30]


def sources_sha1() -> str:
    h = hashlib.sha1()
    for name in ("engine.hip", "af_flow.hpp", "af_flow_host.hpp", "af_core.hpp", "af_math.hpp", "af_plan_pack.hpp", "af_summary.hpp", "af_pregen.hpp"):
        h.update((ROOT / "asyncflow_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()


def per_kernel(path: Path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    if not path.exists():
        return acc, n
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(short(r["Kernel_Name"]), r["Counter_Name"])] += 1
    return acc, n


counters: dict = collections.defaultdict(dict)
for i in (1, 2, 3, 4, 5):
    acc, n = per_kernel(out / f"pmc{i}{sfx}.csv")
    for k, d in acc.items():
        for c, v in d.items():
            counters[k][c] = v / max(n[(k, c)], 1)            # per launch (dispatch)
stats = {short(r["Name"]): r for r in csv.DictReader(open(out / f"kernel_stats_trace{sfx}.csv"))}
bench = json.loads((out / f"bench_unprofiled{sfx}.log").read_text().strip().splitlines()[-1])
dom = "af_flow_jit" if "af_flow_jit" in counters else "af_flow_kernel" if "af_flow_kernel" in counters else max(counters, key=lambda k: counters[k].get("SQ_WAVE_CYCLES", 0.0))
c = counters[dom]
avg_ns = float(stats[dom]["AverageNs"]) if dom in stats else float("nan")
events = float(bench["events_per_step"]) / max(int(bench["config"].get("slices_per_step", 1)), 1)
waves = c.get("SQ_WAVES", float("nan"))
gui = c.get("GRBM_GUI_ACTIVE")
clock_ghz = gui / 8.0 / avg_ns if gui else 2.4                 # GRBM_GUI_ACTIVE: busy clocks summed over the 8 XCDs
kernel_cycles = avg_ns * clock_ghz
binding = {
    "kernel": dom,
    "command": f"python bench.py --config {bench['config']['baseline_config']} --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --no-diagnostics ({bench['config']['workload']})",
    "launches_per_step": int(bench["config"].get("slices_per_step", 1)),
    "sources_sha1": sources_sha1(),
    "binding": "valu_issue",
    "kernel_avg_ms_trace": avg_ns / 1e6,
    "clock_ghz_from_GRBM_GUI_ACTIVE": clock_ghz if gui else None,
    "request_events_per_launch": events,
    "waves_per_launch": waves,
    "wave_insts_per_launch": {k: c.get("SQ_INSTS_" + k) for k in ("VALU", "SALU", "LDS", "BRANCH", "VMEM_RD", "VMEM_WR", "SMEM")},
    "valu_wave_insts_per_request_event": c.get("SQ_INSTS_VALU", float("nan")) / events,
    "all_wave_insts_per_request_event": sum(c.get("SQ_INSTS_" + k, 0.0) for k in ("VALU", "SALU", "LDS", "BRANCH", "VMEM_RD", "VMEM_WR")) / events,
    "valu_issue_frac": c.get("SQ_ACTIVE_INST_VALU", float("nan")) * 4.0 / (N_SIMD * kernel_cycles),
    "valu_issue_frac_note": "SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1 024 SIMDs x kernel cycles): share of the launch during which a SIMD issues VALU",
    "valu_lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", float("nan")) / (64.0 * c.get("SQ_ACTIVE_INST_VALU", float("nan"))),
    "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", float("nan")) / c.get("SQ_WAVE_CYCLES", float("nan")),
    "wave_lifetime_ms": c.get("SQ_WAVE_CYCLES", float("nan")) * 4.0 / waves / (clock_ghz * 1e6) if waves == waves else None,
    "lds_bank_conflict_frac": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
    "FETCH_SIZE_KB": c.get("FETCH_SIZE"),
    "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
    "hbm_read_bytes": 2.0 * c["FETCH_SIZE"] * 1024.0 if "FETCH_SIZE" in c else None,
    "l2_write_bytes": c["WRITE_SIZE"] * 1024.0 if "WRITE_SIZE" in c else None,
    "raw_counters_per_launch": dict(c),
    "other_kernels": {k: {"avg_ms": float(stats[k]["AverageNs"]) / 1e6 if k in stats else None, **v} for k, v in counters.items() if k != dom},
}
# ---- what one count of SQ_ACTIVE_INST_VALU is worth: scripts/microbench/valu_calibration.hip (VERDICT r4 "What's weak" 5)
cal_path = out / "valu_calibration.json"
if not cal_path.exists():
    found = sorted((ROOT / "profiles").glob("r*/valu_calibration.json"))
    cal_path = found[-1] if found else None
if cal_path is not None and c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_INSTS_VALU"):
    cal = json.loads(cal_path.read_text())
    per = cal.get("simd_cycles_per_valu_inst", {})
    cnt = cal.get("simd_cycles_per_active_count", {})
    insts_per_simd = c["SQ_INSTS_VALU"] / N_SIMD
    active_per_simd = c["SQ_ACTIVE_INST_VALU"] / N_SIMD
    binding["valu_calibration"] = {
        "source": str(cal_path.relative_to(ROOT)) if cal_path.is_relative_to(ROOT) else str(cal_path),
        "simd_cycles_per_valu_inst": per, "simd_cycles_per_active_count": cnt,
        "active_counts_per_valu_inst_this_kernel": c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
        "active_counts_per_valu_inst_calibration": cal.get("active_counts_per_valu_inst"),
        # VALU-busy share of the launch if every instruction cost what the calibration kernels' instruction costs a SIMD
        "busy_frac_if_all_insts_cost_like": {k: insts_per_simd * v / kernel_cycles for k, v in per.items()},
        # ... and by the counter's own counts, each worth what it is worth in a kernel that is VALU-bound by construction
        "busy_frac_by_active_counts_worth": {k: active_per_simd * v / kernel_cycles for k, v in cnt.items()},
        "reading": cal.get("reading"),
    }
    lo = min(binding["valu_calibration"]["busy_frac_by_active_counts_worth"].values()) if cnt else None
    hi = max(binding["valu_calibration"]["busy_frac_by_active_counts_worth"].values()) if cnt else None
    if lo is not None:
        binding["valu_busy_frac_calibrated"] = 0.5 * (lo + hi)
        binding["valu_busy_frac_bracket"] = [lo, hi]
        binding["valu_busy_frac_is"] = ("SQ_ACTIVE_INST_VALU counts per SIMD x (SIMD cycles one count is worth in a kernel whose VALU is busy "
                                        "by construction: valu_calibration.hip, 8 waves per SIMD) / kernel cycles; bracket = the v_add_u32 and "
                                        "the v_fma_f64 calibration, value = its middle")
(out / f"binding{sfx}.json").write_text(json.dumps(binding, indent=1))
traffic = {"kernel": dom, "command": binding["command"], "sources_sha1": binding["sources_sha1"],
           "correction": "gfx950: FETCH_SIZE x2, KB = 1024 B; WRITE_SIZE counts L2 write requests exactly on the kernel's store shapes "
                         "(profiles/r03/write_calibration.json) -- it includes the kernel's scratch stores, which stay in L2",
           "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
           "read_bytes": binding["hbm_read_bytes"], "write_bytes": binding["l2_write_bytes"],
           "bytes_per_launch": (binding["hbm_read_bytes"] or 0.0) + (binding["l2_write_bytes"] or 0.0), "avg_ms": avg_ns / 1e6}
(out / f"traffic{sfx}.json").write_text(json.dumps(traffic, indent=1))
print(json.dumps({k: v for k, v in binding.items() if k not in ("raw_counters_per_launch", "other_kernels")}, indent=1))
