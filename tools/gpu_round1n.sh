#!/bin/bash
# full GPU parity suite with the new BA features (functors, control points, priors) through C ABI + adapter
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/pytest_n.log 2>&1
tail -40 gpurun_out/pytest_n.log
