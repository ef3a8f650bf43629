#!/bin/bash
# PMC pass over a microbench (counters in their own run, kernel-trace only).  usage: pmc.sh TAG counters...; BENCH=script.py
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-pmc}; shift
COUNTERS="$@"
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d $REPO/gpurun_out/pmc_$TAG -o p -- python $REPO/scripts/${BENCH:-bench_gemm.py} > $REPO/gpurun_out/pmc_$TAG.log 2>&1)
f=$(find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in cs.items():
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
tail -2 gpurun_out/pmc_$TAG.log
