#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
pass() { i=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pq$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --no-diagnostics > $O/bench_under_pmc$i.log 2>&1; f=$(find /tmp/pq$i -name "*counter_collection.csv" | head -1); head -1 $f > $O/pmc$i.csv; grep -E "af_arrival|af_pregen" $f >> $O/pmc$i.csv; }
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM
python - <<'PY'
import csv, collections
for i in (1,2):
    acc=collections.defaultdict(dict)
    for r in csv.DictReader(open(f"gpurun_out/r3p/pmc{i}.csv")):
        k = 'chain' if 'chain' in r['Kernel_Name'] else 'units' if 'units' in r['Kernel_Name'] else 'other'
        acc[k][r['Counter_Name']] = float(r['Counter_Value'])
    for k,v in acc.items(): print(i, k, v)
PY
