cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_generator.py -m gpu -x -q -s -k "f32_all" 2>&1 | grep -E "f32_all\]|passed|failed|Error|assert" | tail -30
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-full-step --no-other-configs 2>gpurun_out/b.err | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step']); print(l['f32_mfma_head_and_siren_forward']); print(l['f32_all'])"
tail -3 gpurun_out/b.err
