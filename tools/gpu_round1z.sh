#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_z.log 2>&1
( MVGX_BA_TWO_LEVEL_MIN_N=1 MVGX_BA_UPDATE128_MIN_TILES=1 timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_z2.log 2>&1
for t in 128 100000 16; do
  MVGX_BA_UPDATE128_MIN_TILES=$t timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_z_t$t.json 2> gpurun_out/bench_ba_c5_z.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_z5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_z5.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_z.log gpurun_out/pytest_z2.log; grep -o '"lm_iteration_ms": [0-9.]*' gpurun_out/bench_ba_c5_z_t*.json; head -7 gpurun_out/prof_z5/ba5_kernel_stats.csv | cut -c1-150
