"""GPU busy time (union of kernel intervals) against the wall clock spanned by a rocprofv3 kernel_trace.csv: how much of the span has NO kernel
running (host-bound gaps).  usage: trace_busy.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = rows[int(len(rows) * skip):]                   # drop construction / warm-up
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, last = 0, rows[0][0], rows[0][1], rows[0][2]
gaps, named = [], []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        named.append((s - cur_e, last, n))
        cur_s, cur_e, last = s, e, n
    else:
        if e > cur_e:
            cur_e, last = e, n
busy += cur_e - cur_s
span = t1 - t0
gaps.sort(reverse=True)
print(f"{len(rows)} launches over {span / 1e6:.2f} ms: some kernel running {busy / 1e6:.2f} ms ({busy / span:.1%}), idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps "
      f"(largest {[round(g / 1e3, 1) for g in gaps[:8]]} us; gaps > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms)")

import collections
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in named:
    if g > 15000:
        agg[(a, b)][0] += 1; agg[(a, b)][1] += g
print("gaps > 15 us by (kernel before -> kernel after):")
for (a, b), (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {c:4d} x {tot / c / 1e3:7.1f} us = {tot / 1e6:6.2f} ms   {a} -> {b}")
