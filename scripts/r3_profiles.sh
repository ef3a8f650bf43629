#!/bin/bash
# Round-3 tracked profiles (copied from gpurun_out/ into profiles/ afterwards): rocprofv3 --kernel-trace --stats of the
# bench step at C2 / r128 / r256 (frozen and not) and of the full GAN step at C2 / C4 / C5; then PMC passes (kernel-trace
# only, counters in their own runs): HBM traffic of the roofline kernel and of the fused ray-march, MFMA-busy of the step.
# (round 3: C2, C2 hierarchical and the full GAN step at C2; the other geometries keep their round-2 tables)
# usage: scripts/r3_profiles.sh [stats|pmc|all]
cd "$(dirname "$0")/.." || exit 1
WHAT=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
B="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact --no-graph --no-full-step"
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  scripts/prof_any.sh r3_c2 bench.py $B > /dev/null
  scripts/prof_any.sh r3_c2_hier bench.py $B --hier > /dev/null
  scripts/prof_any.sh r3_fullstep_c2 scripts/bench_full_step.py --steps 2 --warmup 1 > /dev/null
  for t in r3_c2 r3_c2_hier r3_fullstep_c2; do
    echo "== $t"; head -12 gpurun_out/prof_$t.txt; tail -2 gpurun_out/prof_$t.log
  done
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  # (1) HBM traffic of the roofline kernel (forward flavour of the head GEMM)
  scripts/pmc_roofline.sh > gpurun_out/r3_roofline_pmc.log 2>&1; cp gpurun_out/roofline_pmc.json gpurun_out/r3_roofline_pmc.json
  # (2) HBM traffic of the fused ray-march under no_grad (per-ray bytes vs the algorithmic 4 S + 132)
  for pass in "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    set -- $pass; tag=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmcmarch_$tag -o p -- python $REPO/scripts/march_kernel.py > $REPO/gpurun_out/pmcmarch_$tag.log 2>&1)
  done
  # (3) MFMA-busy of every kernel of the bench step (one eager step is enough)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $REPO/gpurun_out/pmcmfma -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact --no-graph --no-full-step > $REPO/gpurun_out/pmcmfma.log 2>&1)
  python - <<'PY'
import csv, glob, json, collections
# ---- march traffic ----
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcmarch_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "siren_march" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in vals.items()}
rd = (m.get("TCC_EA0_RDREQ_sum", 0) - m.get("TCC_EA0_RDREQ_32B_sum", 0)) * 128 + m.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
wr = m.get("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (m.get("TCC_EA0_WRREQ_sum", 0) - m.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32
rays = 32 * 4096
out = {"kernel": "siren_march_x3_kernel<true> (fused rays + SIREN + composite, no_grad), C2: b=32, 64x64 rays, S=24",
       "counters_mean_per_launch": m, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "rays": rays,
       "hbm_bytes_per_ray": (rd + wr) / rays, "algorithmic_bytes_per_ray": 4 * 24 + 132,
       "method": "rocprofv3 --kernel-trace --pmc, separate read / write passes; reads = (RDREQ - RDREQ_32B) x 128 B + RDREQ_32B x 32 B "
                 "(gfx950: wide requests are 128 B, MI355X_MICROARCH.md §HBM), writes = WRREQ_64B x 64 B + rest x 32 B; first launch dropped"}
json.dump(out, open("gpurun_out/r3_march_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
# ---- MFMA busy ----
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcmfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if any(t in k for t in ("gemm_bf16x3", "siren_", "composite", "modfc", "torgb")):
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = {}
for k, cs in agg.items():
    mm = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    gui = mm.get("GRBM_GUI_ACTIVE", 0.0)
    # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the chip's 1024 SIMDs (= 32 x the number of
    # v_mfma_f32_32x32x16_bf16 issued: checked against the GEMMs' MFMA counts); GRBM_GUI_ACTIVE is summed over the 8 XCDs,
    # so the launch lasted GUI_ACTIVE / 8 shader cycles: busy fraction = MFMA_BUSY / (1024 x GUI_ACTIVE / 8)
    rows[k] = {"launches": n, **{c: round(v, 1) for c, v in mm.items()},
               "launch_kcycles": round(gui / 8 / 1e3, 1),
               "mfma_busy_frac": round(mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * 8 / (gui * 1024), 4) if gui else None}
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES over "
                     "2 eager bench steps at C2 (+1 warm-up); per-kernel means over the launches; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": rows}, open("gpurun_out/r3_mfma_busy.json", "w"), indent=1)
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:16]:
    print(k[:70], v)
PY
  tail -2 gpurun_out/pmcmfma.log
fi
