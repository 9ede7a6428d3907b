#!/bin/bash
# quick instruction-mix counters of the flow kernel for any bench flags: bash scripts/gpu_pmc_quick.sh <tag> <bench flags...>
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG; mkdir -p $OUT
KERN="af_flow|af_pregen|af_summary|af_series"
pass() { i=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pq$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --generic-kernels $EXTRA > $OUT/bench_under_pmc$i.log 2>&1; f=$(find /tmp/pq$i -name "*counter_collection.csv" | head -1); head -1 $f > $OUT/pmc$i.csv; grep -E "$KERN" $f >> $OUT/pmc$i.csv; }
EXTRA="$@"
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
python - <<PY
import csv,collections
for i in (1,2):
    rows=list(csv.DictReader(open("$OUT/pmc%d.csv"%i)))
    acc=collections.defaultdict(float)
    for r in rows:
        if "af_flow" in r.get("Kernel_Name",""):
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(i, dict(acc))
PY
grep "^{" $OUT/bench_under_pmc1.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('flow_ms', d['flow_kernel_ms'], 'events', d['events_per_step'], 'scen', d['config']['scenarios_rank0'])"
