#!/bin/bash
# counter passes over scripts/probe/head_gemm_pmc_probe.py (each --pmc set in its own run, kernel-trace only)
cd "$(dirname "$0")/../.." || exit 1
export BENCH=probe/head_gemm_pmc_probe.py
bash scripts/pmc.sh hg_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL > gpurun_out/hg_lds.txt 2>&1
bash scripts/pmc.sh hg_inst SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 > gpurun_out/hg_inst.txt 2>&1
bash scripts/pmc.sh hg_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES > gpurun_out/hg_busy.txt 2>&1
bash scripts/pmc.sh hg_bar SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU > gpurun_out/hg_bar.txt 2>&1
cat gpurun_out/hg_lds.txt gpurun_out/hg_inst.txt gpurun_out/hg_busy.txt gpurun_out/hg_bar.txt
