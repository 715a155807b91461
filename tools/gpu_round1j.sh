#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py > gpurun_out/bench_j.log 2> gpurun_out/bench_j.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_j -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba > gpurun_out/prof_j.log 2>&1
cat gpurun_out/bench_j.log | cut -c1-1800; head -8 gpurun_out/prof_j/b_kernel_stats.csv | cut -c1-150
