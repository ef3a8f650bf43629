cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for l in libcips3d_hip_base libv3_43 libv3_53 libv3_54 libv3_55; do
  echo "== $l" >> gpurun_out/v3_heavy_ab.txt
  python scripts/bench_v3_heavy.py cips3d_amd/lib/$l.so 2>&1 | grep " us " >> gpurun_out/v3_heavy_ab.txt
done; done
cat gpurun_out/v3_heavy_ab.txt
