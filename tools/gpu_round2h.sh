#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/time_ba_create.py > gpurun_out/create_2h.log 2>&1
cat gpurun_out/create_2h.log
