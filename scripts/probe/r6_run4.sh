cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_discriminator.py -m gpu -x -q -s -k "parity_subconv" 2>&1 | tail -25
