"""Do kernels of the two head chains (CIPS_INR_CHUNKS=2) actually run concurrently?  Reads a rocprofv3 kernel-trace csv and
reports, for the head GEMM kernels, how much of their busy time overlaps another kernel's."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
gem = [e for e in ev if "gemm_bf16x3" in e[2]]
t0 = gem[len(gem) // 2][0]
win = [e for e in gem if t0 <= e[0] < t0 + 20_000_000]
tot = sum(e[1] - e[0] for e in win)
ov = 0
for i, a in enumerate(win):
    for b in win[i + 1:]:
        if b[0] >= a[1]: break
        ov += min(a[1], b[1]) - b[0]
span = win[-1][1] - win[0][0]
print(f"{len(win)} GEMM launches in a {span/1e6:.2f} ms window: kernel time {tot/1e6:.2f} ms, pairwise overlap {ov/1e6:.2f} ms")
for e in win[:24]:
    print(f"  {(e[0]-win[0][0])/1e3:9.1f} us  dur {(e[1]-e[0])/1e3:7.1f} us  {e[2][:70]}")
