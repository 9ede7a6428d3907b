#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root).
#   bash scripts/profile_round.sh          -> gpurun_out/prof_final/*
# Counters are collected in their own passes (never with trace domains).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_final; mkdir -p $OUT
python bench.py --steps 2 --warmup 1 > $OUT/bench_unprofiled.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.log 2>&1
cp /tmp/pt/t_kernel_stats.csv $OUT/kernel_stats_trace.csv 2>/dev/null || cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_trace.csv
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1); head -1 $f > $OUT/kernel_trace_af.csv; grep -E "af_des|af_jit|af_pregen|af_summary|af_series" $f >> $OUT/kernel_trace_af.csv
pass() { i=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pp$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc$i.log 2>&1; f=$(find /tmp/pp$i -name "*counter_collection.csv" | head -1); head -1 $f > $OUT/pmc$i.csv; grep -E "af_des|af_jit|af_pregen|af_summary|af_series" $f >> $OUT/pmc$i.csv; }
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
pass 3 FETCH_SIZE
pass 4 WRITE_SIZE
pass 5 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum
ls -la $OUT
