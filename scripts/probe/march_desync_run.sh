#!/bin/bash
# GPU side of the march desync probe (libraries built by march_desync.sh here, shipped with the snapshot)
out=gpurun_out/r6_march_desync.txt; : > $out
for round in 1 2; do
for L in lib_tuning lib_tuning_prio; do
  for dz in 0 1500 3000 4500 6000 7500 9000; do
    CIPS_X3_MDESYNC=$dz python scripts/probe/march_desync.py cips3d_amd/$L/libcips3d_hip.so 2>&1 | grep "^lib" >> $out
  done
done
done
cat $out
