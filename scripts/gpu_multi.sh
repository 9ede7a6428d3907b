cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/multi.log 2>&1; echo "rc=$?" >> gpurun_out/multi.log; tail -25 gpurun_out/multi.log
AF_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --generic-kernels --scenarios 2000 > gpurun_out/bench_dist1.log 2>&1; echo "rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_dist1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("n_gpus", d["n_gpus"], "gather_ms", d["gather_ms"], d["gather_path"], "value %.3e"%d["value"])
else: print(open("gpurun_out/bench_dist1.log").read()[-2000:])
PY
timeout 300 python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --generic-kernels --scenarios 1000 | tail -c 300
