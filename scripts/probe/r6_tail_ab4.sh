#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_generator.py -x -q -m gpu 2>&1 | tail -3
bash scripts/probe/r6_tail_ab2.sh
