#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/time_adapter_match.py > gpurun_out/adapter_match_2j.log 2>&1
grep -v "^INFO" gpurun_out/adapter_match_2j.log
( timeout 200 python -m pytest tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_2j.log 2>&1
cat gpurun_out/pytest_2j.log
