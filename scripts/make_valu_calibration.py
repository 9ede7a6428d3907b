"""profiles/rNN/valu_calibration.json from the runs of scripts/microbench/valu_calibration.hip (scripts/profile_round5.sh cal).

    python scripts/make_valu_calibration.py <dir>

Reads <dir>/cal_stdout.json (the program's own hipEvent timings and the VALU wave-instruction counts it KNOWS),
<dir>/cal_counters.csv (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE of the same program).  Per (kernel, waves per SIMD), second launch of each:

* insts_counter / insts_known           -- is SQ_INSTS_VALU a count of wave-instructions?  (must be 1.00)
* active_per_inst                       -- SQ_ACTIVE_INST_VALU per VALU wave-instruction
* simd_cycles_per_inst                  -- launch cycles x 1 024 SIMDs / instructions: what one instruction COSTS a SIMD
                                           when the VALU is the only thing the kernel uses (cycles = GRBM_GUI_ACTIVE / 8 XCDs)
* active_x4_frac                        -- SQ_ACTIVE_INST_VALU x 4 / (1 024 x launch cycles): rounds 3-4's "valu_issue_frac"
The kernels with >= 4 waves per SIMD of independent chains keep every VALU busy for the whole launch BY CONSTRUCTION:
`saturated` holds their numbers, and `reading` says what one count of SQ_ACTIVE_INST_VALU is worth.
"""
from __future__ import annotations

import collections
import csv
import json
import sys
from pathlib import Path

out = Path(sys.argv[1])
N_SIMD = 1024.0
known = json.loads((out / "cal_stdout.json").read_text())
rows = collections.defaultdict(dict)          # dispatch id -> {counter: value, "kernel":, "grid":, "ns":}
for r in csv.DictReader(open(out / "cal_counters.csv")):
    if "cal_" not in r["Kernel_Name"]:
        continue
    d = rows[int(r["Dispatch_Id"])]
    d["kernel"] = r["Kernel_Name"].split("(")[0]
    d["grid"] = int(r["Grid_Size"])
    d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by_key = collections.defaultdict(list)
for disp in sorted(rows):
    d = rows[disp]
    by_key[(d["kernel"], d["grid"] // 64)].append(d)
table = []
for launch in known["launches"]:
    ds = by_key.get((launch["kernel"], launch["waves"]))
    if not ds:
        continue
    d = ds[-1]                                  # the timed (second) launch
    cycles = d["GRBM_GUI_ACTIVE"] / 8.0         # busy clocks summed over the 8 XCDs
    insts = d["SQ_INSTS_VALU"]
    table.append({
        "kernel": launch["kernel"], "waves_per_simd": launch["waves_per_simd"], "waves": launch["waves"],
        "ms_hip_events_unprofiled": launch["ms"], "ms_under_pmc": d["ns"] / 1e6, "clock_ghz": cycles / d["ns"],
        "insts_known": launch["known_valu_wave_insts"], "insts_counter": insts, "insts_counter_over_known": insts / launch["known_valu_wave_insts"],
        "active_per_inst": d["SQ_ACTIVE_INST_VALU"] / insts,
        "simd_cycles_per_inst": cycles * N_SIMD / insts,
        "active_x4_frac": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cycles),
        "busy_cycles_over_cycles": d.get("SQ_BUSY_CYCLES", float("nan")) / cycles,
        "lanes_per_inst_THREAD_CYCLES": d.get("SQ_THREAD_CYCLES_VALU", float("nan")) / d["SQ_ACTIVE_INST_VALU"],
        "wave_cycles_x4_per_wave": d.get("SQ_WAVE_CYCLES", float("nan")) * 4.0 / max(d.get("SQ_WAVES", float("nan")), 1.0),
    })
sat = {t["kernel"]: t for t in table if t["waves_per_simd"] == 8}
sat4 = {t["kernel"]: t for t in table if t["waves_per_simd"] == 4}
res = {"program": "scripts/microbench/valu_calibration.hip", "reps": known["reps"], "launches": table,
       "saturated_8_waves_per_simd": sat, "saturated_4_waves_per_simd": sat4}
if "cal_fma_f64" in sat and "cal_add_u32" in sat:
    f, u = sat["cal_fma_f64"], sat["cal_add_u32"]
    res["simd_cycles_per_valu_inst"] = {"v_fma_f64": f["simd_cycles_per_inst"], "v_add_u32": u["simd_cycles_per_inst"]}
    res["active_counts_per_valu_inst"] = {"v_fma_f64": f["active_per_inst"], "v_add_u32": u["active_per_inst"]}
    res["active_x4_frac_of_a_saturated_valu"] = {"v_fma_f64": f["active_x4_frac"], "v_add_u32": u["active_x4_frac"]}
    # one count of SQ_ACTIVE_INST_VALU in SIMD cycles, per mix: cycles the VALU was provably busy / counts
    res["simd_cycles_per_active_count"] = {"v_fma_f64": f["simd_cycles_per_inst"] / f["active_per_inst"],
                                           "v_add_u32": u["simd_cycles_per_inst"] / u["active_per_inst"]}
    res["reading"] = (
        "A kernel of 8 waves per SIMD of independent chains keeps the VALU busy for the whole launch, so 1 024 x launch cycles "
        "ARE its VALU-busy cycles.  SQ_ACTIVE_INST_VALU x 4 / that = "
        f"{f['active_x4_frac']:.3f} (v_fma_f64) and {u['active_x4_frac']:.3f} (v_add_u32): "
        "the x4 reading is right where the value is 1.00 and over- / under-states VALU busy time by that factor elsewhere; "
        f"one VALU wave-instruction costs a SIMD {f['simd_cycles_per_inst']:.2f} (v_fma_f64) / {u['simd_cycles_per_inst']:.2f} (v_add_u32) cycles "
        f"and is {f['active_per_inst']:.3f} / {u['active_per_inst']:.3f} counts.")
(out / "valu_calibration.json").write_text(json.dumps(res, indent=1))
print(json.dumps({k: v for k, v in res.items() if k != "launches"}, indent=1))
for t in table:
    print(f"{t['kernel']:26s} wps {t['waves_per_simd']}: insts ctr/known {t['insts_counter_over_known']:.4f}  active/inst {t['active_per_inst']:.3f}  "
          f"SIMD cycles/inst {t['simd_cycles_per_inst']:.3f}  active x4 frac {t['active_x4_frac']:.3f}  lanes/inst {t['lanes_per_inst_THREAD_CYCLES']:.1f}  {t['clock_ghz']:.2f} GHz")
