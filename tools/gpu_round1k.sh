#!/bin/bash
# full GPU parity suite + headline bench + kernel-trace profile of the same bench command (v2 matching kernel default)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 ) > gpurun_out/pytest_k.log 2>&1
timeout 400 python bench.py > gpurun_out/bench_k.log 2> gpurun_out/bench_k.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba > gpurun_out/prof_k.log 2>&1
cat gpurun_out/pytest_k.log; cat gpurun_out/bench_k.log | cut -c1-2500; head -10 gpurun_out/prof_k/b_kernel_stats.csv | cut -c1-150
