cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof2
run() { i=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pp$i -o p -- python bench.py --replicas 2048 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pp$i.log 2>&1; f=$(find /tmp/pp$i -name "*counter_collection.csv" | head -1); grep af_des $f | awk -F, '{print $(NF-3), $(NF-2)}' >> gpurun_out/prof2/pmc.txt; tail -1 /tmp/pp$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', d['kernel_ms'], 'events', d['events_per_step'])" >> gpurun_out/prof2/pmc.txt; }
run 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_WAVE_CYCLES
run 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
run 3 SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32
cat gpurun_out/prof2/pmc.txt
