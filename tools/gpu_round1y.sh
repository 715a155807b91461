#!/bin/bash
# PMC counters of the BA kernels at C5 (what holds chol_update_mfma at ~15 TF/s?)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S="python bench_ba.py c5"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_y1 -o m -- $S > gpurun_out/pmc_y1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_y2 -o m -- $S > gpurun_out/pmc_y2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_y3 -o m -- $S > gpurun_out/pmc_y3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmc_y4 -o m -- $S > gpurun_out/pmc_y4.log 2>&1
ls gpurun_out/pmc_y*/
