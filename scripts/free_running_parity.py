"""Free-running gradient parity of the NeRF side: the C1 and C3 gradient tests of
tests/test_gpu_real_configs.py with fine-sample placement and relu-clamp branches left to the product, in the default
forward (fp16 operand planes) and — A/B on the same oracle run — with the round-1..4 bf16 planes (ops.TRIG_MODE bit 1).
Writes one JSON record (-> profiles/r5_parity_free_running.json).  Runs the CPU oracle on the GPU box's host cores."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_real_configs as T   # noqa: E402

CASES = [("C1 b=4 r32 S=12+12 aux, nerf_noise 0.2", 4, 32, 12, True, True, 0.2, 31),
         ("C3 geometry b=2 r128 S=12+12, aux, nerf_noise 0.1", 2, 128, 12, True, True, 0.1, 1283),
         ("r128 S=24 flat b=2 aux, nerf_noise 0.1", 2, 128, 24, False, True, 0.1, 1284)]
T.FREE_AB = [("fp16 planes (default)", 1), ("bf16 planes (rounds 1-4)", 3)]
for c in CASES[:int(os.environ.get("NCASES", "3"))]:
    try:
        T._g_forward_backward_vs_oracle(*c, pin_fine=False, pin_clamp=False, tol=1.0, free_bar=1e-3)
    except AssertionError as e:
        print("assertion:", c[0], str(e)[:300])
dst = os.path.join(ROOT, "gpurun_out", "r5_parity_free_running.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    json.dump({"what": "relative L2 error of free-running gradients vs the fp32 CPU oracle (INR-head LeakyReLU gates pinned, "
                       "fine-sample placement and relu-clamp branches the product's own)", "cases": T.FREE_RUNNING}, f, indent=1)
print(json.dumps(T.FREE_RUNNING, indent=1))
