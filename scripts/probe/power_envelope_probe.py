"""What the board's power management leaves of the matrix peak: each case loops one kernel for ~3 s while a thread samples rocm-smi
(socket power, sclk); delivered TFLOP/s from event timing.  Cases: head NT GEMM main loop (random / zero operands), NT forward
epilogue, grouped K-major weight-gradient GEMM (random / zero), a streaming kernel for the idle-matrix-pipe reference."""
import os, sys, json, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = 32, 4096, 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04; g = torch.randn(B, n, C, device=d)
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False); gP, _ = ops.split_planes(g, want_t=False)
zx, _ = ops.split_planes(torch.zeros_like(x), want_t=False); zw, _ = ops.split_planes(torch.zeros_like(w), want_t=False)
oP = ops.Planes.empty(B, n, C, device=d); bits = torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8)
dW1 = torch.empty(B, C, C, device=d); dW2 = torch.empty(B, C, C, device=d)
big = torch.empty(256 << 20, device=d); big2 = torch.empty_like(big)
F = 2.0 * B * n * C * C
cases = [
    ("streaming copy 1 GiB (matrix pipe idle)", lambda: big2.copy_(big), 0.0),
    ("NT main loop, random operands", lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C), F),
    ("NT main loop, ZERO operands", lambda: ops.gemm_x3(zx, zw, n, C, C, C, C, B, n * C, C * C), F),
    ("NT forward epilogue, random", lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, act=1, mask_out=bits, gate_bits=2), F),
    ("K-major grouped x2, random", lambda: ops.gemm_x3_km_grouped([(xP, gP, dW1), (gP, xP, dW2)], C, C, n, C, C, B, n * C, n * C), 2 * F),
    ("K-major grouped x2, ZERO", lambda: ops.gemm_x3_km_grouped([(zx, zx, dW1), (zx, zx, dW2)], C, C, n, C, C, B, n * C, n * C), 2 * F),
]


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        j = json.loads(out[out.index("{"):])
        c = next(iter(j.values()))
        pw = [float(v) for k, v in c.items() if "ower" in k and "(W)" in k]
        sc = [v for k, v in c.items() if k.lower().startswith("sclk")]
        return (pw[0] if pw else None), (sc[0] if sc else None)
    except Exception as e:                               # noqa
        return None, repr(e)[:60]


try:
    print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout.strip().splitlines()[-3:])
except Exception as e:
    print("showmaxpower:", e)
print("idle:", smi())
SECS = float(os.environ.get("SECS", 3.0))
for name, fn, flops in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()
    th = threading.Thread(target=lambda: [samples.append(smi()) for _ in iter(lambda: stop.is_set(), True)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # per-launch times of the first and the last 50 launches show the clock sagging under sustained load
    t_first = t_last = None
    th.start(); t0 = time.time(); nl = 0; e0.record()
    while time.time() - t0 < SECS:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        b.record(); b.synchronize(); nl += 50
        t = a.elapsed_time(b) / 50 * 1e3
        t_first = t if t_first is None else t_first
        t_last = t
    e1.record(); torch.cuda.synchronize(); stop.set(); th.join()
    us = e0.elapsed_time(e1) / nl * 1e3
    pw = [p for p, _ in samples if p is not None]
    sc = [s for _, s in samples]
    tf = flops / us / 1e6 if flops else 0.0
    print(f"{name:42s} {us:7.1f} us/launch (first 50: {t_first:6.1f}, last 50: {t_last:6.1f})  {tf:6.1f} TFLOP/s = {tf / 833.3:.3f} of the x3 roof   "
          f"power {min(pw) if pw else 0:.0f}..{max(pw) if pw else 0:.0f} W (n={len(pw)})   sclk {sc[:1]} .. {sc[-1:]}", flush=True)
    time.sleep(2.0)
