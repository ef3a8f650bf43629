#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/c5t; 
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5t -- python scripts/bench_full_step.py --steps 2 --warmup 1 --img-size 256 --batch 4 --num-steps 12 --freeze --diffaug > /tmp/c5.log 2>&1
f=$(find /tmp/c5t -name "*kernel_trace.csv" | head -1)
python scripts/probe/trace_by_grid.py $f 3 > gpurun_out/r6_c5_fullstep_by_grid.txt
head -60 gpurun_out/r6_c5_fullstep_by_grid.txt
tail -2 /tmp/c5.log | cut -c1-300
