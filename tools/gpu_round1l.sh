#!/bin/bash
# BA v2 (sorted Schur products + Gram blocks + MFMA-f64 Cholesky): parity under the three path switches, C3 + C5 bench,
# kernel-trace profile of the C3 BA bench
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/pytest_ba_l0.log 2>&1
( MVGX_BA_LEGACY=2 timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/pytest_ba_l2.log 2>&1
( MVGX_BA_LEGACY=1 timeout 300 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/pytest_ba_l1.log 2>&1
timeout 300 python bench_ba.py > gpurun_out/bench_ba_l.json 2> gpurun_out/bench_ba_l.err
timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_l.json 2> gpurun_out/bench_ba_c5_l.err
MVGX_BA_LEGACY=3 timeout 200 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_legacy_l.json 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l -o ba -- python bench_ba.py c3 --no-cpu > gpurun_out/prof_l.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_l5.log 2>&1
( time timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 ) > gpurun_out/pytest_l_all.log 2>&1
cat gpurun_out/pytest_ba_l0.log gpurun_out/pytest_ba_l2.log gpurun_out/pytest_ba_l1.log; cut -c1-1500 gpurun_out/bench_ba_l.json; cut -c1-1200 gpurun_out/bench_ba_c5_l.json; tail -3 gpurun_out/bench_ba_c5_l.err; cut -c1-600 gpurun_out/bench_ba_legacy_l.json
head -25 gpurun_out/prof_l/ba_kernel_stats.csv | cut -c1-160; head -25 gpurun_out/prof_l5/ba5_kernel_stats.csv | cut -c1-160; tail -5 gpurun_out/pytest_l_all.log
