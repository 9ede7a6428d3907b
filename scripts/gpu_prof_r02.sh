cd $GRAFT_REPO_ROOT
bash scripts/profile_round2.sh r02 > gpurun_out/prof_r02.out 2>&1
tail -22 gpurun_out/prof_r02.out
bash scripts/gpu_ab.sh "|--online-summary --scenarios 131072" "|--scenarios 65536" "|--config 5" "|--config 4 --scenarios 20000"
