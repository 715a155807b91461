#!/bin/bash
# PMC counters of the v2 matching filter kernel (variant 4, stage 3): what holds the matrix pipe at ~48 %?
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S="python tools/sweep_match.py --images 300 --rounds 2 --variants 43"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_r1 -o m -- $S > gpurun_out/pmc_r1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r2 -o m -- $S > gpurun_out/pmc_r2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_r3 -o m -- $S > gpurun_out/pmc_r3.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_WAVE32_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT --output-format csv -d gpurun_out/pmc_r4 -o m -- $S > gpurun_out/pmc_r4.log 2>&1
for d in pmc_r1 pmc_r2 pmc_r3 pmc_r4; do ls gpurun_out/$d/*counter_collection.csv 2>/dev/null | head -2; tail -2 gpurun_out/$d.log | cut -c1-300; done
