"""Device-to-host rate of this box for the full outputs of a sweep (the PCIe-inclusive rate; never `value`).

    python scripts/measure_d2h.py [GiB per copy, default 2]

Copies a device buffer to pinned and to pageable host memory, prints one JSON line with both rates and what BASELINE config
2's 18.2 GB of full outputs (rqs_clock + sampled series of 10 000 scenarios) would add to the 0.048 s step."""
import json
import sys
import time

import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
dev = torch.empty(n, dtype=torch.uint8, device="cuda:0")
dev.fill_(7)
out = {}
for kind in ("pinned", "pageable"):
    host = torch.empty(n, dtype=torch.uint8, pin_memory=(kind == "pinned"))
    host.copy_(dev)   # (first touch of the pages)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 4
    for _ in range(reps):
        host.copy_(dev, non_blocking=(kind == "pinned"))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    assert int(host[-1]) == 7
    out[kind + "_GBps"] = n / dt / 1e9
    del host
full = 18.18e9   # config 2: WRITE_SIZE of the flow kernel = every output word (profiles/r05/binding_c2.json)
for kind in ("pinned", "pageable"):
    out[f"config2_full_outputs_s_{kind}"] = full / (out[kind + "_GBps"] * 1e9)
    out[f"config2_events_per_s_with_d2h_{kind}"] = 5.472e9 / (0.0477 + out[f"config2_full_outputs_s_{kind}"])
out["GiB_per_copy"] = gib
print(json.dumps(out))
