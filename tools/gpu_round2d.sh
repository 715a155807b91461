#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
nproc; python -c "import os;print(os.cpu_count(), len(os.sched_getaffinity(0)))"
timeout 120 python tools/time_match_small.py > gpurun_out/time_small_2d.log 2>&1
cat gpurun_out/time_small_2d.log
( time timeout 400 python -m pytest tests/test_matching_gpu.py -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -25 ) > gpurun_out/pytest_2d.log 2>&1
cat gpurun_out/pytest_2d.log
