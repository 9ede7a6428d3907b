cd $GRAFT_REPO_ROOT
bash scripts/profile_round2.sh r02 > gpurun_out/prof_r02.out 2>&1
tail -12 gpurun_out/prof_r02.out | cut -c1-400
bash scripts/gpu_default_bench.sh 2>&1 | tail -60
bash scripts/gpu_ab.sh "|--config 3" "|--config 4 --scenarios 20000" "|--config 5" "|--scenarios 65536" "|--no-flow --steps 1 --warmup 0"
