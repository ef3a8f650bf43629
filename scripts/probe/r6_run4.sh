cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_discriminator.py tests/test_gpu_graph.py tests/test_gpu_train_step.py -m gpu -x -q 2>&1 | tail -12
