"""Static VALU instruction mix of a plan-specialised af_flow_jit, priced with the calibrated issue costs.

    python scripts/valu_mix.py <config> [profiles/r05/valu_calibration.json] [binding json to annotate]

No GPU needed: a planning-only engine writes the spec of `bench.py --config C`, hipcc builds the code object (or the
cache has it), llvm-objdump disassembles it.  Every VALU instruction of af_flow_jit is put in one of two issue classes
(scripts/microbench/valu_cost.hip measured ~40 opcodes, profiles/r03/valu_cost.txt; valu_calibration.hip prices the
classes in SIMD cycles):
  full rate   v_add_u32 / v_sub / v_xor / v_and / v_or / v_mov_b32 / v_add_f32 / v_lshlrev ... VOP1/VOP2 32-bit ALU
  quarter     everything 64-bit (f64, u64, b64), every VOP3-only 32-bit form (v_add3, v_lshl_add, v_bfe, v_mad_*,
              v_mul_lo/hi), compares, v_cndmask with an SGPR-pair mask, v_readlane / v_readfirstlane / DPP / mbcnt
A STATIC count is not the dynamic mix (loops, divergent regions); it says which class dominates the code the kernel
spends its time in, which is what the reconciliation of `valu_issue_frac` needs (VERDICT r4 "What's weak" 5).
"""
from __future__ import annotations

import collections
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

import bench  # noqa: E402
from asyncflow_amd import jit  # noqa: E402
from asyncflow_amd.engine import PLAN_ONLY, Engine  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cal_path = Path(sys.argv[2]) if len(sys.argv) > 2 else sorted((ROOT / "profiles").glob("r*/valu_calibration.json"))[-1]
args = bench.make_parser().parse_args(["--config", str(cfg)])
args.horizon = None
wl = bench.build_workload(cfg, 0, 1, 0, None)
shape = bench.rank_shape(wl, args)
eng = Engine(shape["plan"], PLAN_ONLY, **shape["engine_kw"])
hi = min(shape["slice"], shape["n"])
over = [(c, i, np.ascontiguousarray(v[:hi])) for c, i, v, _ in shape["over"]]
spec = eng.jit_spec(shape["seeds"][:hi], over, clock_ptr=8, clock_capacity=shape["clock_cap"], samples_ptr=8,
                    tick_capacity=shape["ticks"], counts_ptr=8, draw_capacity=shape["clock_cap"])
eng.close()
image = jit.code_object(spec)
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.NamedTemporaryFile(suffix=".hsaco") as f, tempfile.NamedTemporaryFile(suffix=".elf") as elf:
    f.write(image)
    f.flush()
    # (hipcc --genco writes an offload bundle: the gfx950 ELF is one of its entries)
    subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={f.name}",
                    f"--output={elf.name}", "--unbundle"], check=True)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", elf.name], capture_output=True, text=True, check=True).stdout

FULL = re.compile(r"^v_(add|sub|subrev)_(u32|i32|f32|co_u32)$|^v_(addc|subb|subbrev)_co_u32$|^v_(xor|and|or|not|mov)_b32$|"
                  r"^v_(lshlrev|lshrrev|ashrrev)_b32$|^v_(min|max)_(u32|i32|f32)$|^v_(mul|fma|mac|fmac)_f32$|^v_cndmask_b32$|^v_accvgpr")
counts: collections.Counter = collections.Counter()
classes: collections.Counter = collections.Counter()
other: collections.Counter = collections.Counter()
in_kernel = False
for line in dis.splitlines():
    if line.endswith(">:"):
        in_kernel = "af_flow_jit" in line
        continue
    if not in_kernel:
        continue
    m = re.match(r"^\s+([a-z0-9_]+)\s", line)
    if not m:
        continue
    op = m.group(1)
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if op.startswith("v_"):
        vop3 = op.endswith("_e64") or op.endswith("_dpp") or op.endswith("_e64_dpp") or op.endswith("_sdwa")
        # v_cndmask_b32_e64 takes an SGPR-pair mask (quarter rate in valu_cost.txt); e32 reads vcc
        full = bool(FULL.match(base)) and not (vop3 and base != "v_mov_b32" and not base.endswith("_f32"))
        classes["full" if full else "quarter"] += 1
        counts[base] += 1
    else:
        other[op.split("_")[0]] += 1

cal = json.loads(cal_path.read_text())
c_full, c_quarter = cal["simd_cycles_per_valu_inst"]["v_add_u32"], cal["simd_cycles_per_valu_inst"]["v_fma_f64"]
n = sum(classes.values())
avg = (classes["full"] * c_full + classes["quarter"] * c_quarter) / max(n, 1)
res = {"config": cfg, "kernel": "af_flow_jit", "static_valu_instructions": n, "static_class_counts": dict(classes),
       "static_quarter_rate_share": classes["quarter"] / max(n, 1), "simd_cycles": {"full": c_full, "quarter": c_quarter},
       "static_mix_simd_cycles_per_valu_inst": avg, "other_static_instructions": dict(other),
       "top_valu_opcodes": counts.most_common(25), "calibration": str(cal_path.relative_to(ROOT)) if cal_path.is_absolute() and ROOT in cal_path.parents else str(cal_path)}
if len(sys.argv) > 3:
    bp = Path(sys.argv[3])
    bj = json.loads(bp.read_text())
    raw = bj["raw_counters_per_launch"]
    cycles = bj["kernel_avg_ms_trace"] * 1e6 * (bj.get("clock_ghz_from_GRBM_GUI_ACTIVE") or 2.4)
    busy = raw["SQ_INSTS_VALU"] / 1024.0 * avg / cycles
    bj["valu_static_mix"] = res
    bj["valu_busy_frac_calibrated"] = busy
    bj["valu_busy_frac_is"] = ("VALU wave-instructions per SIMD (SQ_INSTS_VALU / 1 024) x the SIMD cycles one instruction costs, averaged over the "
                               "kernel's STATIC instruction mix with the calibrated costs of the two issue classes (valu_calibration.hip: "
                               f"{c_full:.2f} cycles full rate, {c_quarter:.2f} quarter rate), / kernel cycles.  SQ_ACTIVE_INST_VALU is a COUNT of "
                               "VALU instructions (1.000 per instruction in both calibration kernels), not busy time: rounds 3-4's 'x 4' priced "
                               "every instruction at 4 cycles")
    # what binds: the VALU when it is busy most of the launch, else the waits the resident waves do not hide
    bj["binding"] = ("valu_issue" if busy >= 0.6 else
                     f"latency: {bj.get('wait_any_frac_of_wave_cycles', float('nan')):.0%} of the wave cycles in s_waitcnt (dependent LDS round trips) at "
                     f"the occupancy the launch's LDS allows; VALU busy only {busy:.0%}")
    bp.write_text(json.dumps(bj, indent=1))
    res["valu_busy_frac_calibrated"] = busy
print(json.dumps(res, indent=1))
