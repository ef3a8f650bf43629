"""Aggregate a rocprofv3 kernel_trace.csv by (kernel name, grid size): which SHAPES the time of a kernel family sits in.
usage: trace_by_grid.py <kernel_trace.csv> [divide_by]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    n = n.split("(")[0] if not n.startswith("void at::") else n[:90]
    g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    a = agg[(n, g)]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"# {len(rows)} launches, {tot/1e3/div:.2f} ms per step (divide_by {div:g})")
print(f"{'calls':>7} {'ms/step':>8} {'avg_us':>8}  kernel  (workgroups x, grid y, grid z)")
for (n, g), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
    print(f"{c/div:7.1f} {us/1e3/div:8.3f} {us/c:8.1f}  {n} {g}")
