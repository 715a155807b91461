#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/pytest_t.log 2>&1
timeout 300 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_t.json 2> gpurun_out/bench_ba_t.err
timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_t.json 2> gpurun_out/bench_ba_c5_t.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t -o ba -- python bench_ba.py c3 --no-cpu > gpurun_out/prof_t.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_t5.log 2>&1
tail -5 gpurun_out/pytest_t.log; cut -c1-400 gpurun_out/bench_ba_t.json; echo; cut -c1-400 gpurun_out/bench_ba_c5_t.json
