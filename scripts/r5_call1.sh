#!/bin/bash
# round 5, GPU call 1: fp16 operand planes in the SIREN forward — probe, parity, free-running gradients A/B, timing A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -2 > $O/device.txt; nproc >> $O/device.txt
./scripts/probe/f16_subnormal_mfma_probe > $O/r5_f16_subnormal_probe.txt 2>&1; cat $O/r5_f16_subnormal_probe.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -k "siren or march or composite or resample" > $O/c1_kernels.log 2>&1; echo "kernels exit $?"; grep -E "sigma: oracle|passed|failed" $O/c1_kernels.log | tail -8
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_train_step.py -m gpu -q --tb=short > $O/c1_generator.log 2>&1; echo "generator exit $?"; tail -3 $O/c1_generator.log
timeout 600 python -m pytest tests/test_gpu_real_configs.py -m gpu -q -s --tb=short -k "c2_headline or c2_flat or c2_full" > $O/c1_real.log 2>&1; echo "real exit $?"; grep -E "FREE|tightest|passed|failed|worst" $O/c1_real.log | tail
timeout 1500 python scripts/free_running_parity.py > $O/c1_free.log 2>&1; echo "free exit $?"; grep -E "FREE-RUNNING|assertion" $O/c1_free.log
for t in 1 3 1 3; do
  TRIG=$t python - <<'PY' 2>&1 | tail -1 | sed "s/^/trig=$t /" | tee -a $O/c1_march_ab.txt
import os, sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from cips3d_amd import ops
ops.TRIG_MODE = int(os.environ["TRIG"])
import bench_march
bench_march.main()
PY
done
for t in 1 3 1; do
  TRIG=$t python - <<'PY' 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})" | sed "s/^/trig=$t /" | tee -a $O/c1_bench_ab.txt
import os, sys
sys.path.insert(0, ".")
from cips3d_amd import ops
ops.TRIG_MODE = int(os.environ["TRIG"])
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-exact", "--no-full-step", "--no-other-configs"]
import bench
bench.main()
PY
done
