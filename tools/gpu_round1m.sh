#!/bin/bash
# BA (Z = L^-1 Y products, panel-wise diagonal factor, chunked Gram kernels): full GPU parity suite, C3 + C5 bench, profiles
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > gpurun_out/pytest_m.log 2>&1
timeout 300 python bench_ba.py > gpurun_out/bench_ba_m.json 2> gpurun_out/bench_ba_m.err
timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_m.json 2> gpurun_out/bench_ba_c5_m.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_m -o ba -- python bench_ba.py c3 --no-cpu > gpurun_out/prof_m.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_m5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_m5.log 2>&1
tail -12 gpurun_out/pytest_m.log; cut -c1-900 gpurun_out/bench_ba_m.json; echo; cut -c1-900 gpurun_out/bench_ba_c5_m.json; tail -3 gpurun_out/bench_ba_c5_m.err
