cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dstep -o p -- python $GRAFT_REPO_ROOT/scripts/probe/d_step_only.py 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_dstep.log 2>&1)
python scripts/probe/trace_busy.py $(find gpurun_out/prof_dstep -name "*kernel_trace.csv" | head -1) 0.3
grep "D step" gpurun_out/prof_dstep.log
rm -rf gpurun_out/prof_dstep
