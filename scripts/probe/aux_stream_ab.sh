#!/bin/bash
# A/B of the discriminator's side streams (cips3d_amd.discriminator.AUX_SIDE_STREAM; the ResBlock-skip stream of round 5 was removed with its switch), full GAN step at C2, two rounds
cd "$(dirname "$0")/../.." || exit 1
for rnd in 1 2; do for v in 0 1; do
  echo "AUX=$v round $rnd: $(CIPS_SIDE=$v timeout 300 python - <<PY 2>&1 | tail -1 | grep -o '"ms_[A-Za-z_]*": [0-9.]*' | tr '\n' ' '
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from cips3d_amd import discriminator
discriminator.AUX_SIDE_STREAM = os.environ["CIPS_SIDE"] == "1"
sys.argv = ["bench_full_step.py", "--steps", "6", "--warmup", "3"]
runpy.run_path("scripts/bench_full_step.py", run_name="__main__")
PY
)"
done; done
