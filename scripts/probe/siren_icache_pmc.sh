#!/bin/bash
# instruction-cache counters of the SIREN backward kernels (scripts/probe/siren_bwd_occupancy_probe.py: b = 1 ... 32 at P = 98304)
cd "$(dirname "$0")/../.." || exit 1
export BENCH=probe/siren_bwd_occupancy_probe.py
bash scripts/pmc.sh ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE > gpurun_out/ic1.txt 2>&1
bash scripts/pmc.sh ic2 SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES > gpurun_out/ic2.txt 2>&1
bash scripts/pmc.sh ic3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 > gpurun_out/ic3.txt 2>&1
grep -A5 "siren_bwd" gpurun_out/ic1.txt gpurun_out/ic2.txt gpurun_out/ic3.txt | grep -v "^--"
