#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/pytest_w.log 2>&1
( MVGX_BA_BIG_UPDATE_TILES=2 timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_w2.log 2>&1
for t in 24 12 1000; do
  MVGX_BA_BIG_UPDATE_TILES=$t timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_w_t$t.json 2> gpurun_out/bench_ba_c5_w.err
done
timeout 300 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_w.json 2>> gpurun_out/bench_ba_c5_w.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_w5.log 2>&1
B="python bench.py --images 400 --steps 1 --warmup 0 --no-cpu-baseline --no-ba"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_w1 -o m -- $B > gpurun_out/pmc_w1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w2 -o m -- $B > gpurun_out/pmc_w2.log 2>&1
tail -3 gpurun_out/pytest_w.log gpurun_out/pytest_w2.log; grep -o '"lm_iteration_ms": [0-9.]*' gpurun_out/bench_ba_c5_w_t*.json gpurun_out/bench_ba_w.json; head -6 gpurun_out/prof_w5/ba5_kernel_stats.csv | cut -c1-140
