#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( timeout 200 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider -k "reject_loop or track_filters" --durations=3 2>&1 | tail -8 ) > gpurun_out/pytest_2r.log 2>&1
cat gpurun_out/pytest_2r.log
