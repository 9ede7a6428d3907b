cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05e; mkdir -p $OUT
show() { grep '^{' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), d['parity_spot_check']['ok'], 'jitfb', d['config']['flow']['jit_fallbacks'], 'handed', d.get('handed_back'))"; }
for c in 3 5 6 4 2; do
  B="python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-diagnostics"
  $B > $OUT/keys_c${c}_new.log 2>&1; show $OUT/keys_c${c}_new.log "c$c keys-per-call"
  ASYNCFLOW_JIT_EXTRA_FLAGS="-DAF_PHILOX_KEYS_HOISTED" $B > $OUT/keys_c${c}_old.log 2>&1; show $OUT/keys_c${c}_old.log "c$c hoisted"
done
