cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05e; mkdir -p $OUT
B="python bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-diagnostics --no-parity-check"
show() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), 'pregen', round(d['pregen_ms'],2), 'summary', round(d['summary_ms'],3), 'jitfb', d['config']['flow']['jit_fallbacks'])"; }
for n in 2048 4096 6144 8192 10000 12288; do $B --scenarios $n > $OUT/tail_n$n.log 2>&1; show $OUT/tail_n$n.log n=$n; done
for w in 12 14 15; do AF_WAVES_PER_CU=$w $B > $OUT/tail_wpc$w.log 2>&1; show $OUT/tail_wpc$w.log wpc=$w; done
