cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05h; mkdir -p $OUT
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.log 2>&1; echo "rc=$?" >> $OUT/bench_driver_cmd.log
for c in 2 3 4 5 6; do L=c$c; [ $c = 6 ] && L=gensrv; python bench.py --config $c --no-cpu-baseline --no-diagnostics --steps 3 --warmup 1 > $OUT/bench_unprofiled_$L.log 2>&1; grep '^{' $OUT/bench_unprofiled_$L.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],2), d['roofline']['binding']['stale'], d['roofline'].get('traffic'), d['parity_spot_check']['ok'])"; done
grep '^{' $OUT/bench_driver_cmd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['binding']['stale'], d['parity_spot_check']['ok'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
