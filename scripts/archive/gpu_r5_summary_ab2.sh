cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05g; mkdir -p $OUT
B="python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics --no-parity-check"
for rep in 1 2 3; do for w in 4 8; do AF_SUMMARY_WPE=$w $B > $OUT/sumab2_wpe${w}_$rep.log 2>&1; printf "wpe=$w rep=$rep "; grep '^{' $OUT/sumab2_wpe${w}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), 'summary', round(d['summary_ms'],3))"; done; done
