#!/bin/bash
# Round-6 tracked profiles (copied from gpurun_out/ into profiles/ afterwards).
#  stats: rocprofv3 --kernel-trace --stats of the eager bench step at C2 and of the full GAN step at C2;
#  pmc:   HBM traffic of the roofline kernel, MFMA-busy of the step's kernels (counters in their own runs, kernel-trace only).
# usage: scripts/r6_profiles.sh [stats|pmc|all]
cd "$(dirname "$0")/.." || exit 1
WHAT=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
B="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact --no-graph --no-full-step --no-other-configs"
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  scripts/prof_any.sh r6_c2 bench.py $B > /dev/null
  scripts/prof_any.sh r6_fullstep_c2 scripts/bench_full_step.py --steps 2 --warmup 1 > /dev/null
  for t in r6_c2 r6_fullstep_c2; do echo "== $t"; head -16 gpurun_out/prof_$t.txt; tail -2 gpurun_out/prof_$t.log; done
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  scripts/pmc_roofline.sh > gpurun_out/r6_roofline_pmc.log 2>&1; cp gpurun_out/roofline_pmc.json gpurun_out/r6_roofline_pmc.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $REPO/gpurun_out/r6_pmcmfma -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact --no-graph --no-full-step --no-other-configs > $REPO/gpurun_out/r6_pmcmfma.log 2>&1)
  python - <<'PY'
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r6_pmcmfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if any(t in k for t in ("gemm_bf16x3", "siren_", "composite", "modfc", "torgb")):
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = {}
for k, cs in agg.items():
    mm = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    gui = mm.get("GRBM_GUI_ACTIVE", 0.0)
    rows[k] = {"launches": n, **{c: round(v, 1) for c, v in mm.items()}, "launch_kcycles": round(gui / 8 / 1e3, 1),
               "mfma_busy_frac": round(mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * 8 / (gui * 1024), 4) if gui else None}
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES over 2 eager bench steps at C2 "
                     "(+1 warm-up); per-kernel means over the launches; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": rows}, open("gpurun_out/r6_mfma_busy.json", "w"), indent=1)
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:12]:
    print(k[:70], v["launches"], v["launch_kcycles"], v["mfma_busy_frac"])
PY
  tail -2 gpurun_out/r6_pmcmfma.log
fi
