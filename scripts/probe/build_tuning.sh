#!/bin/bash
# Probe build: the library with its tuning aids compiled in (-DCIPS_TUNING: phase skipping, store suppression, start-phase
# skew, in-kernel timestamps, read from CIPS_X3_* environment variables by common.h: cips_tune_env).  Results of such a build
# are WRONG BY DESIGN when a variable is set.  It is written to cips3d_amd/lib_tuning/ (own objects, own .so): the product
# library is never touched; a probe script loads it by setting cips3d_amd._lib.LIB_PATH before the first call.
cd "$(dirname "$0")/../.." || exit 1
python -c "
from cips3d_amd import build
build.HIPCC_EXTRA = ['-DCIPS_TUNING']
build.LIBDIR = 'lib_tuning'
print('tuning build:', build.build(force=True))
"
