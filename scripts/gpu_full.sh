cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/gputests.log
tail -12 gpurun_out/gputests.log
bash scripts/gpu_ab.sh "|--config 3" "|--config 4 --scenarios 10000" "|--config 5 --scenarios 6250" "|--config 1"
