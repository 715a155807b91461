#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MVGX_BA_CREATE_TIMING=1 timeout 300 python tools/time_adapter_ba.py --ref > gpurun_out/adapter_ba_2m.log 2>&1
grep -v "^INFO\|^$" gpurun_out/adapter_ba_2m.log | tail -30
