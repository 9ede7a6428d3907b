cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/gputests.log
tail -6 gpurun_out/gputests.log
bash scripts/gpu_ab.sh "|" "|--config 3"
