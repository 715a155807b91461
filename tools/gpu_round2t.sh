#!/bin/bash
# HBM traffic of the matching kernels with the final batch pipeline (PMC pass only: --kernel-trace + --pmc, nothing else)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmc_2t -o m -- python bench.py --images 400 --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming > gpurun_out/pmc_2t.log 2>&1
ls gpurun_out/pmc_2t/ | head; tail -1 gpurun_out/pmc_2t.log | cut -c1-200
