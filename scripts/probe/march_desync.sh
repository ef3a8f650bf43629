#!/bin/bash
# Probe: the fused march's two waves per SIMD started out of phase (CIPS_X3_MDESYNC shader cycles) with / without raised
# priority for the MFMA phases (-DCIPS_X3_PRIO).  Builds two probe libraries (lib_tuning, lib_tuning_prio).
cd "$(dirname "$0")/../.." || exit 1
for v in "" prio; do
python - <<PY
from cips3d_amd import build
build.HIPCC_EXTRA = ['-DCIPS_TUNING'] + (['-DCIPS_X3_PRIO'] if "$v" else [])
build.LIBDIR = 'lib_tuning' + ('_prio' if "$v" else '')
print('probe build:', build.build(force=False))
PY
done
