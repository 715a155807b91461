#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) > gpurun_out/pytest_2a.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/pytest_2a.log
