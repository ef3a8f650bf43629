#!/bin/bash
# the C2 step's own per-dispatch clock and MFMA-busy (one counter pass each for eager launches and hipGraph replay)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out
for mode in eager graph; do
  extra=""; [ $mode = eager ] && extra="--no-graph"
  rm -rf gpurun_out/pmc_step_$mode
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $REPO/gpurun_out/pmc_step_$mode -o p -- \
     python $REPO/bench.py --steps 4 --warmup 2 $extra --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs > $REPO/gpurun_out/pmc_step_$mode.log 2>&1)
  c=$(find gpurun_out/pmc_step_$mode -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmc_step_$mode -name "*kernel_trace.csv" | head -1)
  echo "== $mode launches (bench.py --steps 4 --warmup 2: with the capture's own warm-ups the trace holds more than 6 steps; per-step columns divide by the step count of the LAST argument)" 
  python scripts/probe/step_clock.py $c $k 6
  grep -h "^{" gpurun_out/pmc_step_$mode.log | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('   bench line under the counter pass:', l['ms_per_step'], 'ms / step,', l['config']['launch'])"
  rm -rf gpurun_out/pmc_step_$mode
done
