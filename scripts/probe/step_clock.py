"""Per-dispatch shader clock and MFMA-busy of the benchmarked C2 step from ONE rocprofv3 counter pass
(--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES over bench.py): for every dispatch
    sclk = GRBM_GUI_ACTIVE / 8 XCDs / duration,   busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
then per kernel (the largest by time) the launch-time-weighted means, and the time-weighted clock of the whole step.
usage: step_clock.py <counter_collection.csv> <kernel_trace.csv> <steps in the trace>"""
import collections, csv, sys
cc = list(csv.DictReader(open(sys.argv[1])))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
steps = float(sys.argv[3])
disp = collections.defaultdict(dict)
for r in cc:
    disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])      # calls, us, gui, busy
tot_us = tot_gui = 0.0
for did, c in disp.items():
    t = kt.get(did)
    if t is None or "GRBM_GUI_ACTIVE" not in c:
        continue
    us = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    n = c["name"].replace("(anonymous namespace)::", "")
    n = n.split("(")[0] if not n.startswith("void at::") else n[:70]
    a = agg[n]
    a[0] += 1; a[1] += us; a[2] += c["GRBM_GUI_ACTIVE"]; a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    tot_us += us; tot_gui += c["GRBM_GUI_ACTIVE"]
print(f"# {len(disp)} dispatches, {tot_us / 1e3 / steps:.2f} ms of kernel time per step (counter pass: launches serialised); "
      f"time-weighted shader clock of the step {tot_gui / 8 / tot_us / 1e3:.3f} GHz")
print(f"{'calls/step':>10} {'ms/step':>8} {'avg_us':>8} {'sclk GHz':>9} {'MFMA busy':>10} {'busy x GHz':>11}  kernel")
for n, (c, us, gui, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    ghz = gui / 8 / us / 1e3
    b = busy / (1024 * gui / 8) if gui else 0.0
    print(f"{c / steps:10.1f} {us / 1e3 / steps:8.3f} {us / c:8.1f} {ghz:9.3f} {b:10.3f} {b * ghz:11.3f}  {n}")
