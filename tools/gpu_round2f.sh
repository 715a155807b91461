#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 400 python -m pytest tests/test_l2f_gpu.py tests/test_hamming_gpu.py tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider --durations=5 -k "l2f or float or hamming or golden_and_reference or ragged or akaze or error_behaviour or descriptor_lengths or duplicates" 2>&1 | tail -25 ) > gpurun_out/pytest_2f.log 2>&1
cat gpurun_out/pytest_2f.log
timeout 200 python bench_hamming.py l2f > gpurun_out/bench_l2f_2f.json 2> gpurun_out/bench_l2f_2f.err
cat gpurun_out/bench_l2f_2f.json; tail -2 gpurun_out/bench_l2f_2f.err
