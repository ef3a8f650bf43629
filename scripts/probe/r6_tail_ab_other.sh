#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do
for m in main side; do
  for cfg in "--img-size 128 --batch 8" "--img-size 256 --batch 4 --num-steps 12 --freeze --diffaug" "--img-size 256 --batch 4 --num-steps 24 --freeze --diffaug --no-aux" ""; do
    CIPS_INR_TAIL=$m python scripts/bench_full_step.py --steps 3 --warmup 1 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', '$cfg', d['ms_D_step'], d['ms_G_step'], d['ms_step'])"
  done
done
done
