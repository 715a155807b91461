#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 600 python -m pytest tests/test_matching_gpu.py tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/pytest_2b.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/pytest_2b.log
timeout 300 python bench.py --no-ba --no-cpu-baseline --overlap 0 --steps 3 --warmup 1 > gpurun_out/bench_2b_ov0.json 2> gpurun_out/bench_2b_ov0.err
timeout 300 python bench.py --no-ba --no-cpu-baseline --overlap 1 --steps 3 --warmup 1 > gpurun_out/bench_2b_ov1.json 2> gpurun_out/bench_2b_ov1.err
cat gpurun_out/bench_2b_ov0.json gpurun_out/bench_2b_ov1.json
