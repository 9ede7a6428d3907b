"""Disassembly of the plan-specialised af_flow_jit of `bench.py --config C` (no GPU needed): python scripts/jit_disasm.py C out.s"""
from __future__ import annotations

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

cfg = int(sys.argv[1])
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
print(spec, file=sys.stderr)
image = jit.code_object(spec)
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.NamedTemporaryFile(suffix=".hsaco") as f, tempfile.NamedTemporaryFile(suffix=".elf") as elf:
    f.write(image)
    f.flush()
    subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={f.name}",
                    f"--output={elf.name}", "--unbundle"], check=True)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", elf.name], capture_output=True, text=True, check=True).stdout
Path(sys.argv[2]).write_text(dis)
