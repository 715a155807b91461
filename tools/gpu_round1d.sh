#!/bin/bash
# fourth GPU session: adapter (drop-in) tests, multi-rank BA tests, default bench line
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_adapter_gpu.py tests/test_ba_multirank_gpu.py tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/pytest_d.log
timeout 900 python bench.py > gpurun_out/bench_d.log 2> gpurun_out/bench_d.err
tail -40 gpurun_out/pytest_d.log; cat gpurun_out/bench_d.log; tail -5 gpurun_out/bench_d.err
