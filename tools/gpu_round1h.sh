#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python -m pytest tests/test_matching_gpu.py -m gpu -q -p no:cacheprovider -x -k "43 and (adversarial or parity or golden)" 2>&1 | tail -5 > gpurun_out/pytest_h.log
tail -3 gpurun_out/pytest_h.log
grep -q "passed" gpurun_out/pytest_h.log || exit 1
grep -q "failed" gpurun_out/pytest_h.log && exit 1
S="python tools/sweep_match.py --images 300 --rounds 2 --variants 43"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h -o v4 -- $S > gpurun_out/prof_h.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_h1 -o m -- $S > gpurun_out/pmc_h1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_h2 -o m -- $S > gpurun_out/pmc_h2.log 2>&1
timeout 120 python tools/sweep_match.py --images 80 --desc 8000 --rounds 2 --variants 43 > gpurun_out/sweep_big.log 2>&1
head -4 gpurun_out/prof_h/v4_kernel_stats.csv | cut -c1-160; grep -a '"variant"' gpurun_out/prof_h.log gpurun_out/sweep_big.log
