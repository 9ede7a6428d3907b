#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -n -i -B2 -A12 "pc.sampl\|PC Sampl" $O/avail.txt | head -80
export ASYNCFLOW_JIT_EXTRA_FLAGS="-gline-tables-only"
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check > $O/bench_g.log 2>&1
try() { tag=$1; shift
  timeout 300 rocprofv3 --pc-sampling-beta-enabled "$@" --output-format csv -d /tmp/pcs_$tag -o s -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity-check > $O/pcs_$tag.log 2>&1
  echo "pcs $tag rc=$?"; ls -la /tmp/pcs_$tag 2>&1 | head; }
try st --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576
try ht --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1
python - <<'PY' > $O/pcs_summary.txt 2>&1
import csv, glob, collections, sys
files = glob.glob('/tmp/pcs_*/**/*pc_sampling*.csv', recursive=True)
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
    for k, v in by_line.most_common(200): print(f'{v:8d} {100.0 * v / max(n, 1):6.2f}%  {k}')
    print('--- by opcode')
    for k, v in by_inst.most_common(60): print(f'{v:8d} {100.0 * v / max(n, 1):6.2f}%  {k}')
PY
head -40 $O/pcs_summary.txt; tail -3 $O/pcs_st.log $O/pcs_ht.log
