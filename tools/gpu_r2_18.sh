#!/bin/bash
# round 2, call 18: phase stamps of the 64 x 64 factor-and-invert kernel
mkdir -p gpurun_out/r2_18
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MVGX_BA_FACTOR_DEBUG=1 timeout 300 python tools/ba_one_iteration.py > gpurun_out/r2_18/stamps.log 2>&1; grep "factor kernel" gpurun_out/r2_18/stamps.log
MVGX_BA_FACTOR_DEBUG=1 MVGX_BA_SOLVER=dense timeout 300 python tools/ba_one_iteration.py > gpurun_out/r2_18/stamps_dense.log 2>&1; grep "factor kernel" gpurun_out/r2_18/stamps_dense.log
