"""Timeline of one eager headline step from a rocprofv3 kernel trace: every launch of the LAST step with start (us from the step's first
launch), duration, queue, and how much of the step is covered only by 'small' kernels.  usage: step_timeline.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at every siren_march / first get_zs normal_ kernel: use the march kernel as the anchor
idx = [i for i, r in enumerate(rows) if "siren_march_x3_kernel" in r["Kernel_Name"]]
idx = idx[-nsteps:]
step = rows[idx[-2]:idx[-1]]            # one step-long window, from a march launch to the next
t0 = int(step[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in step)
print(f"# {len(step)} launches, {(end - t0) / 1e3:.1f} us first start -> last end")
qs = sorted({r["Queue_Id"] for r in step})
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:70]
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{st / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{qs.index(r['Queue_Id'])}  {n}")
