cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_discriminator.py -m gpu -x -q -k "parity or blur or resblock" 2>&1 | tail -3
