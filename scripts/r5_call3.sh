#!/bin/bash
# round 5, GPU call 3: chain launch of the head forward — parity, stagger sweep, step time
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -s --tb=short -k "chain or inr_head" > $O/c3_chain_tests.log 2>&1; echo "chain tests exit $?"; tail -4 $O/c3_chain_tests.log
timeout 600 python scripts/bench_chain.py > $O/c3_bench_chain.txt 2>&1; cat $O/c3_bench_chain.txt | tail -12
for c in 1 0 1 0; do
  CH=$c python - <<'PY' 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})" | sed "s/^/chain=$c /" | tee -a $O/c3_bench_ab.txt
import os, sys
sys.path.insert(0, ".")
from cips3d_amd import ops
ops.INR_CHAIN = os.environ["CH"] == "1"
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-exact", "--no-full-step", "--no-other-configs"]
import bench
bench.main()
PY
done
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_train_step.py tests/test_gpu_formats.py -m gpu -x -q --tb=short > $O/c3_gen.log 2>&1; echo "generator/train/formats exit $?"; tail -3 $O/c3_gen.log
