cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_gpu_suite.log
tail -5 gpurun_out/r6_gpu_suite.log
