#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_generator.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2; do
for m in main side; do
  CIPS_INR_TAIL=$m python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'])"
done
done
