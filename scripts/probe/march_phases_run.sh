#!/bin/bash
out=gpurun_out/r6_march_phases.txt; : > $out
L=cips3d_amd/lib_tuning/libcips3d_hip.so
python scripts/probe/march_phases.py $L 2>&1 | grep -v Warn >> $out
CIPS_X3_MONE=1 python scripts/probe/march_phases.py $L 2>&1 | grep -v Warn >> $out
CIPS_X3_MDESYNC=6000 python scripts/probe/march_phases.py $L 2>&1 | grep -v Warn >> $out
CIPS_X3_MDESYNC=6000 python scripts/probe/march_phases.py cips3d_amd/lib_tuning_prio/libcips3d_hip.so 2>&1 | grep -v Warn >> $out
cat $out
