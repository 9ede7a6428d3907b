cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_quick; mkdir -p $OUT
KERN="af_flow|af_pregen"
pass() { i=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pq$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --generic-kernels $EXTRA > $OUT/log$i.txt 2>&1; f=$(find /tmp/pq$i -name "*counter_collection.csv" | head -1); head -1 $f > $OUT/pmc$i.csv; grep -E "$KERN" $f >> $OUT/pmc$i.csv; }
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass 2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM
python - <<PY
import csv,collections
for i in (1,2):
    rows=list(csv.DictReader(open("$OUT/pmc%d.csv"%i)))
    for kern in ("af_flow","af_pregen"):
        acc=collections.defaultdict(float)
        for r in rows:
            if kern in r.get("Kernel_Name",""):
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
        print(i, kern, {k: "%.3e"%v for k,v in acc.items()})
PY
