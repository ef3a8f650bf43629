#!/bin/bash
# A/B of the fused SIREN backward schedules on one box: parity tests, then timings of both kernels (interleaved rounds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_kernels.py -x -q -k "siren or march" 2>&1 | tail -5
for r in 1 2; do
  for v in 0 1; do
    echo "== CIPS_SIREN_BWD_V4=$v round $r"
    CIPS_SIREN_BWD_V4=$v REPS=10 python scripts/bench_siren.py 2>&1 | grep -v Warning
  done
done
python scripts/bench_march.py 2>&1 | tail -6
echo "== phase timestamps, v4"
CIPS_SIREN_BWD_V4=1 CIPS_X3_PROF=1 REPS=3 python scripts/bench_siren.py 2>&1 | grep -E "wave 0|wave 3|G1 " | head -4
} > gpurun_out/r4_siren_ab.log 2>&1
cat gpurun_out/r4_siren_ab.log
