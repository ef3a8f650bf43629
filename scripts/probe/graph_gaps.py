"""Idle time between the kernels of one graph-replayed bench step: reads a rocprofv3 kernel trace (csv) of
`bench.py --steps 3 --warmup 1 ...` (graph mode) and reports, for the last step, kernel time, gaps, and the gaps by size."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# the last step: walk back from the end until 1/4 of ... simpler: take the last N kernels where N = kernels per step (from the
# repeating pattern: count launches of the SIREN backward)
idx = [i for i, r in enumerate(rows) if "siren_bwd_x3_kernel" in r[2]]
per = idx[-1] - idx[-2]
last = rows[idx[-1] - per + 1: idx[-1] + 1]          # one full period ending with the SIREN backward
# extend to the period's remaining tail (kernels after siren bwd belong to the same step) by rotating: use period between the
# two last siren_bwd launches instead
last = rows[idx[-2] + 1: idx[-1] + 1]
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
gaps = [last[i + 1][0] - last[i][1] for i in range(len(last) - 1)]
pos = [g for g in gaps if g > 0]
print(f"kernels {len(last)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  sum of positive gaps {sum(pos)/1e6:.3f} ms  overlapped {-sum(g for g in gaps if g < 0)/1e6:.3f} ms")
hist = collections.Counter(min(int(g / 1000), 20) for g in pos)
print("gap histogram (us: count):", dict(sorted(hist.items())))
big = sorted(((gaps[i], last[i][2][:50], last[i + 1][2][:50]) for i in range(len(gaps))), reverse=True)[:12]
for g, a, b in big:
    print(f"{g/1e3:8.1f} us  after {a}  before {b}")
