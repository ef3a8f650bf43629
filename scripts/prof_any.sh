#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command: scripts/prof_any.sh <tag> <script.py> [args...]
cd "$(dirname "$0")/.." || exit 1
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
SCRIPT=$REPO/$1; shift
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o p -- python $SCRIPT "$@" > $REPO/gpurun_out/prof_$TAG.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" "$TAG" <<'PY'
import csv, sys
f, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
out = open(f"gpurun_out/prof_{tag}.txt", "w")
out.write(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name\n")
for r in rows[:60]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    if len(n) > 110: n = n[:107] + "..."
    out.write(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:12.1f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}  {n}\n")
out.close()
print(open(f"gpurun_out/prof_{tag}.txt").read())
PY
tail -3 gpurun_out/prof_$TAG.log
