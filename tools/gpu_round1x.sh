#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/pytest_x.log 2>&1
( MVGX_BA_TWO_LEVEL_MIN_N=1 timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_x2.log 2>&1
for t in 2048 100000 1; do
  MVGX_BA_TWO_LEVEL_MIN_N=$t timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_x_t$t.json 2> gpurun_out/bench_ba_c5_x.err
  MVGX_BA_TWO_LEVEL_MIN_N=$t timeout 300 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_x_t$t.json 2>> gpurun_out/bench_ba_c5_x.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_x5.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_x.log gpurun_out/pytest_x2.log; grep -o '"lm_iteration_ms": [0-9.]*' gpurun_out/bench_ba_c5_x_t*.json gpurun_out/bench_ba_x_t*.json; head -6 gpurun_out/prof_x5/ba5_kernel_stats.csv | cut -c1-140
