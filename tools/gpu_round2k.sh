#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MVGX_ADAPTER_TIMING=1 timeout 300 python tools/time_adapter_match.py > gpurun_out/adapter_match_2k.log 2>&1
grep -v "^INFO" gpurun_out/adapter_match_2k.log | tail -40
