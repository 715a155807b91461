#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 2; do
  MVGX_BA_CHOL_MODE=$m timeout 300 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_u_m$m.json 2> gpurun_out/bench_ba_u_m$m.err
  MVGX_BA_CHOL_MODE=$m timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_u_m$m.json 2>> gpurun_out/bench_ba_u_m$m.err
done
( time timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/pytest_u.log 2>&1
grep -o '"lm_iteration_ms": [0-9.]*' gpurun_out/bench_ba_u_m*.json gpurun_out/bench_ba_c5_u_m*.json; tail -3 gpurun_out/pytest_u.log
