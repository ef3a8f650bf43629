#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py --steps 3 --warmup 1 --no-graph > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
cp $f gpurun_out/r6_step_kernel_trace.csv; python scripts/probe/step_timeline.py $f 3 > gpurun_out/r6_step_timeline.txt
head -3 gpurun_out/r6_step_timeline.txt; wc -l gpurun_out/r6_step_timeline.txt
