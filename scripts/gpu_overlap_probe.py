"""Can the analyzer of a finished part of a sweep hide under the stage-parallel kernel of the next part?  (HBM-bound beside
VALU-bound.)  BASELINE config 2, 10 000 replicas: the step as it is (run, then summarize) against the same work as
run(A) -> [run(B) on the engine's stream || summarize(A) on a second engine's stream] -> summarize(B), for several splits.
Prints one JSON line.  Measurement only: nothing in the product calls this."""
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from asyncflow_amd.engine import Engine  # noqa: E402

args = bench.make_parser().parse_args(["--config", "2"])
args.horizon = None
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = bench.build_workload(2, 0, 1, 0, None)
sw = bench.RankSweep(wl, dev, args)
sw.prepare()
shape = bench.rank_shape(wl, args)
eng2 = Engine(sw.plan, dev.index, **shape["engine_kw"])
n = sw.n


def run_part(lo, hi):
    seeds = sw.seeds[lo:hi]
    over = [(c, i, np.ascontiguousarray(v[lo:hi])) for c, i, v, _ in sw.over]
    return sw.eng.run(seeds, over, specialise=True, clock_ptr=sw.clock[lo:].data_ptr(), clock_capacity=sw.clock_cap,
                      samples_ptr=sw.samples[lo:].data_ptr(), tick_capacity=sw.ticks, counts_ptr=sw.counts[lo:hi].data_ptr(),
                      draw_capacity=sw.clock_cap)


def summarize_part(eng, lo, hi):
    return eng.summarize(hi - lo, clock_ptr=sw.clock[lo:].data_ptr(), clock_capacity=sw.clock_cap,
                         samples_ptr=sw.samples[lo:].data_ptr(), tick_capacity=sw.ticks, counts_ptr=sw.counts[lo:hi].data_ptr(),
                         stats_ptr=sw.s_stats[lo:hi].data_ptr(), rps_ptr=sw.s_rps[lo:hi].data_ptr(), rps_buckets=sw.T,
                         hist_ptr=sw.s_hist[lo:hi].data_ptr(), hist_bins=256, hist_max=sw.hist_max,
                         series_mean_ptr=sw.s_mean[lo:hi].data_ptr(), series_max_ptr=sw.s_max[lo:hi].data_ptr())


def timed(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def plain():
    run_part(0, n)
    summarize_part(sw.eng, 0, n)


def split_serial(cuts):
    def f():
        lo = 0
        for hi in [*cuts, n]:
            run_part(lo, hi)
            summarize_part(sw.eng, lo, hi)
            lo = hi
    return f


def split_overlap(cuts):
    def f():
        lo, th = 0, None
        for hi in [*cuts, n]:
            run_part(lo, hi)                 # (returns when the part's kernels are done: af_engine_run synchronises)
            if th is not None:
                th.join()
            th = threading.Thread(target=summarize_part, args=(eng2, lo, hi))
            th.start()                       # ... under the next part's kernels
            lo = hi
        th.join()
    return f


out = {"plain_ms": timed(plain)}
ref = (sw.s_stats.clone(), sw.s_rps.clone(), sw.s_mean.clone(), sw.counts.clone())
for cuts in ([8192], [4096, 8192], [5000], [2500, 5000, 7500], [6000], [7168]):
    key = "+".join(map(str, cuts))
    out[f"serial_{key}_ms"] = timed(split_serial(cuts))
    out[f"overlap_{key}_ms"] = timed(split_overlap(cuts))
    same = all(torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b)
               for a, b in zip(ref, (sw.s_stats, sw.s_rps, sw.s_mean, sw.counts)))
    out[f"same_{key}"] = bool(same)
out["plain_again_ms"] = timed(plain)
print(json.dumps(out), flush=True)
