#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export BENCH=bench_siren_split.py
bash scripts/pmc.sh tl1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL > gpurun_out/tl1.txt 2>&1
bash scripts/pmc.sh tl2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY > gpurun_out/tl2.txt 2>&1
bash scripts/pmc.sh tl3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY > gpurun_out/tl3.txt 2>&1
bash scripts/pmc.sh tl4 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM > gpurun_out/tl4.txt 2>&1
for f in tl1 tl2 tl3 tl4; do grep -A5 "siren_bwd_tail_kernel<true>(TailArg grid=393216\|x4_kernel<true, true>(Bwd grid=196608" gpurun_out/$f.txt | grep -v "^--"; done
