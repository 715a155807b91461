#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 400 python -m pytest tests/test_hamming_gpu.py tests/test_adapter_gpu.py tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider --durations=8 -k "hamming or track_filters or degenerate or Hamming or golden_and_reference or descriptor_lengths or akaze or duplicates or error_behaviour" 2>&1 | tail -25 ) > gpurun_out/pytest_2e.log 2>&1
cat gpurun_out/pytest_2e.log
timeout 200 python bench_hamming.py > gpurun_out/bench_hamming_2e.json 2> gpurun_out/bench_hamming_2e.err
cat gpurun_out/bench_hamming_2e.json; tail -3 gpurun_out/bench_hamming_2e.err
