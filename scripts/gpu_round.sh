#!/bin/bash
# One gpurun call: parity tests per file (a crash in one must not hide the others), smoke, bench,
# rocprof kernel trace.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
for f in tests/test_gpu_kernels.py tests/test_gpu_generator.py tests/test_gpu_discriminator.py tests/test_gpu_train_step.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -s --tb=short > gpurun_out/$n.log 2>&1
  echo "$n exit $?" | tee -a gpurun_out/summary.txt
  tail -5 gpurun_out/$n.log
done
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/summary.txt
tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?" | tee -a gpurun_out/summary.txt
tail -3 gpurun_out/bench.log
if [ -n "$PROFILE" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OLDPWD/gpurun_out/prof.log 2>&1)
  echo "prof exit $?" | tee -a gpurun_out/summary.txt
  find gpurun_out/prof -name "*kernel_stats*" | head
fi
