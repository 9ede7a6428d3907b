"""WRITE_SIZE / FETCH_SIZE calibration on the store shapes of the stage-parallel kernel (GPU box).

    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/cal -o c -- python scripts/calibrate_write_size.py run
    python scripts/calibrate_write_size.py report /tmp/cal profiles/r03/write_calibration.json

`run` launches af_probe_store (include/asyncflow_hip.h) once per pattern: every wave fills its own contiguous region
exactly once, so the bytes stored are known; `report` divides the counter by them.
"""
from __future__ import annotations

import csv
import ctypes as C
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

N_WAVES = 8192
PATTERNS = {   # kernel name -> (pattern id, bytes per wave, lanes)
    "af_probe_store_wide": (0, 1_048_576, 0),
    "af_probe_store_pairs16": (1, 57 * 16 * 1330, 57),       # ~1 330 batches of 57 completions: one LB-2 scenario
    "af_probe_store_rows48": (2, 48 * 11_999 // 240 * 240, 0),   # the 11 999 sample rows of one LB-2 scenario
}


def run() -> None:
    from asyncflow_amd.engine import load_library

    lib = load_library()
    for name, (pat, nbytes, lanes) in PATTERNS.items():
        ms = C.c_double()
        rc = lib.af_probe_store(0, pat, N_WAVES, nbytes, lanes, C.byref(ms))
        assert rc == 0, lib.af_last_error()
        print(f"{name}: {N_WAVES} waves x {nbytes} B = {N_WAVES * nbytes / 1e9:.3f} GB in {ms.value:.3f} ms "
              f"({N_WAVES * nbytes / ms.value / 1e6:.0f} GB/s)")


def stored_bytes(name: str) -> float:
    pat, nbytes, lanes = PATTERNS[name]
    words = nbytes // 4 & ~3
    if pat == 0:
        return N_WAVES * (words // 256 * 256 + max(0, (words % 256) // 4 * 4)) * 4.0
    if pat == 1:
        return N_WAVES * (words // (4 * lanes)) * 4 * lanes * 4.0
    return N_WAVES * (words // 60) * 60 * 4.0


def report(src: str, dst: str) -> None:
    out = {"n_waves": N_WAVES, "unit": "WRITE_SIZE is reported in KB of 1024 B (MI355X_MICROARCH.md)", "patterns": {}}
    for path in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            for name in PATTERNS:
                if name in r["Kernel_Name"] and r["Counter_Name"] in ("WRITE_SIZE", "FETCH_SIZE"):
                    e = out["patterns"].setdefault(name, {"stored_bytes": stored_bytes(name)})
                    e[r["Counter_Name"] + "_KB"] = e.get(r["Counter_Name"] + "_KB", 0.0) + float(r["Counter_Value"])
    for name, e in out["patterns"].items():
        if "WRITE_SIZE_KB" in e:
            e["counter_bytes_per_stored_byte"] = e["WRITE_SIZE_KB"] * 1024.0 / e["stored_bytes"]
    Path(dst).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
