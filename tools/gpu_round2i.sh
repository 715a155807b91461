#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/time_ba_create.py > gpurun_out/create_2i.log 2>&1
cat gpurun_out/create_2i.log
( timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) > gpurun_out/pytest_2i.log 2>&1
cat gpurun_out/pytest_2i.log
