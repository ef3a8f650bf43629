#!/bin/bash
# Probe build: the library with its tuning aids compiled in (-DCIPS_TUNING: phase skipping, store suppression, start-phase
# skew, in-kernel timestamps, read from CIPS_X3_* environment variables by common.h: cips_tune_env).  Results of such a build are WRONG BY DESIGN when a
# variable is set; it overwrites cips3d_amd/lib/libcips3d_hip.so — rebuild the product with
# `python -m cips3d_amd.build --force` afterwards (tests/test_abi.py fails on a tuning build, on purpose).
cd "$(dirname "$0")/../.." || exit 1
python -c "
import os
from cips3d_amd import build
build.HIPCC_EXTRA = ['-DCIPS_TUNING']
build.build(force=True)
print('tuning build written; rebuild with: python -m cips3d_amd.build --force')
"
