cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/probe/d_step_only.py 5 > gpurun_out/r6_dstep_fold.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dstep -o p -- python $GRAFT_REPO_ROOT/scripts/probe/d_step_only.py 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_dstep.log 2>&1)
f=$(find gpurun_out/prof_dstep -name "*kernel_trace.csv" | head -1)
python scripts/probe/trace_by_grid.py $f 5 > gpurun_out/r6_dstep_by_grid_fold.txt
rm -rf gpurun_out/prof_dstep
python scripts/bench_full_step.py --steps 6 --warmup 2 > gpurun_out/r6_fullstep_fold.txt 2>&1
grep -h "D step\|ms_step" gpurun_out/r6_dstep_fold.txt gpurun_out/r6_fullstep_fold.txt
