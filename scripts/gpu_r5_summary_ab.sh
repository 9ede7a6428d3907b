#!/bin/bash
# The analyzer's two kernels (af_engine_summarize): register budget of the latency kernel (AF_SUMMARY_WPE = waves per SIMD: 4 / 6 / 8,
# i.e. two / three / four scenarios per CU) x the series kernel beside it on a second stream or after it (AF_SUMMARY_SERIAL=1).
# BASELINE config 2, same box, interleaved; then the analyzer's GPU tests on every form (bit-equal order statistics or it is not a form).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r05e; mkdir -p $OUT
for rep in 1 2; do
  for wpe in 4 6 8; do
    for serial in 1 0; do
      if [ $serial = 1 ]; then export AF_SUMMARY_SERIAL=1; else unset AF_SUMMARY_SERIAL; fi
      AF_SUMMARY_WPE=$wpe python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-diagnostics --no-parity-check > $OUT/sum_wpe${wpe}_serial${serial}_$rep.log 2>&1
    done
  done
done
unset AF_SUMMARY_SERIAL
for rep in 1 2; do for wpe in 4 6 8; do for serial in 1 0; do f=sum_wpe${wpe}_serial${serial}_$rep; printf "%-26s" $f
  grep '^{' $OUT/$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'flow', round(d['flow_kernel_ms'],2), 'pregen', round(d['pregen_ms'],2), 'summary', round(d['summary_ms'],3))"; done; done; done | tee $OUT/summary_ab.txt
for wpe in 6 8; do
  AF_SUMMARY_WPE=$wpe timeout 600 python -m pytest tests/test_gpu_analyzer.py -m gpu -x -q > $OUT/analyzer_tests_wpe$wpe.log 2>&1; tail -2 $OUT/analyzer_tests_wpe$wpe.log
done
timeout 600 python -m pytest tests/test_gpu_analyzer.py -m gpu -x -q > $OUT/analyzer_tests_default.log 2>&1; tail -2 $OUT/analyzer_tests_default.log
