#!/bin/bash
# A/B of the INR head tail placement (CIPS_INR_TAIL=main|side) on the captured headline step + an eager timeline by queue
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do
for m in main side; do
  CIPS_INR_TAIL=$m python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'])"
done
done
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python scripts/probe/step_timeline.py $f 3 > gpurun_out/r6_step_timeline_side.txt
grep -n "composite_bwd\|siren_bwd_x4\|cores\|torgb_bwd_w\|glin_bwd_w_kernel<16>\|siren_bwd_finalize" gpurun_out/r6_step_timeline_side.txt | head -20
