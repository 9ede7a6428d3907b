#!/bin/bash
# round 3: WRITE_SIZE calibration, FEAT_FAR on LB-2, PC sampling of the plan-specialised flow kernel (beta feature: short timeout)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
python scripts/calibrate_write_size.py run > $O/cal_plain.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/calw -o c -- python scripts/calibrate_write_size.py run > $O/cal_write.log 2>&1
python scripts/calibrate_write_size.py report /tmp/calw $O/write_calibration.json >> $O/cal_write.log 2>&1
AF_FLOW_FORCE_FAR=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_far.log 2>&1
export ASYNCFLOW_JIT_EXTRA_FLAGS="-gline-tables-only"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > $O/bench_g.log 2>&1
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 2000 \
   --output-format csv -d /tmp/pcs -o s -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity-check > $O/pcs_run.log 2>&1
echo "pcs rc=$?" >> $O/pcs_run.log
ls -la /tmp/pcs/* >> $O/pcs_run.log 2>&1
python - <<'PY' > $O/pcs_summary.txt 2>&1
import csv, glob, collections, sys
files = glob.glob('/tmp/pcs/**/*pc_sampling*.csv', recursive=True)
print(files)
for f in files:
    by_line = collections.Counter(); by_inst = collections.Counter(); n = 0
    with open(f) as fh:
        rd = csv.DictReader(fh)
        print(rd.fieldnames)
        for r in rd:
            n += 1
            by_line[r.get('Instruction_Comment', '')] += 1
            by_inst[(r.get('Instruction', '').split() or ['?'])[0]] += 1
    print('samples', n)
    print('--- by source line')
    for k, v in by_line.most_common(150): print(f'{v:8d} {100.0 * v / max(n, 1):6.2f}%  {k}')
    print('--- by opcode')
    for k, v in by_inst.most_common(60): print(f'{v:8d} {100.0 * v / max(n, 1):6.2f}%  {k}')
PY
tail -1 $O/bench_far.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('far', d['flow_kernel_ms'], d['ms_per_step'])"
tail -1 $O/bench_g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('g', d['flow_kernel_ms'], d['ms_per_step'])"
cat $O/cal_plain.log; tail -30 $O/cal_write.log; head -60 $O/pcs_summary.txt; tail -5 $O/pcs_run.log
