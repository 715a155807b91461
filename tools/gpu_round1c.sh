#!/bin/bash
# third GPU session: instruction-rate microbenchmarks, full GPU test suite, PMC counter passes on the matching kernel
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 tools/_build/ubench > gpurun_out/ubench.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest_gpu_all.log
B="python bench.py --images 400 --steps 1 --warmup 0 --no-cpu-baseline --no-ba"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc1 -o m -- $B > gpurun_out/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2 -o m -- $B > gpurun_out/pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc3 -o m -- $B > gpurun_out/pmc3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmc4 -o m -- $B > gpurun_out/pmc4.log 2>&1
cat gpurun_out/ubench.log; tail -5 gpurun_out/pytest_gpu_all.log; ls -R gpurun_out/pmc1 | head; tail -3 gpurun_out/pmc1.log
