#!/bin/bash
# round 5, GPU call 2: the full GPU suite on the cleaned-up tree, bench line, kernel trace, march A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -2 > $O/device.txt; nproc >> $O/device.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/r5_gpu_suite.log 2>&1; echo "suite exit $?"; tail -5 $O/r5_gpu_suite.log
for t in 1 3 1 3; do
  TRIG=$t python - <<'PY' 2>&1 | tail -1 | sed "s/^/trig=$t /" | tee -a $O/c2_march_ab.txt
import os, sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from cips3d_amd import ops
ops.TRIG_MODE = int(os.environ["TRIG"])
import bench_march
bench_march.main()
PY
done
timeout 900 python bench.py > $O/r5_bench_c2.json 2> $O/r5_bench_c2.err; echo "bench exit $?"; tail -c 1500 $O/r5_bench_c2.json; tail -3 $O/r5_bench_c2.err
bash scripts/prof.sh r5_c2 > /dev/null 2>&1; head -30 $O/prof_r5_c2.txt
