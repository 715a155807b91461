#!/bin/bash
# fifth GPU session: matching kernel v2 (filter + verify) parity + A/B sweep
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_matching_gpu.py tests/test_adapter_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/pytest_e.log
timeout 400 python tools/sweep_match.py --images 300 --rounds 5 --variants 1,3,41,42,43 --out gpurun_out/sweep_e.json > gpurun_out/sweep_e.log 2>&1
tail -40 gpurun_out/pytest_e.log; cat gpurun_out/sweep_e.log | tail -8
