#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pin in 0 1; do
MVGX_ADAPTER_PINNED_RESULTS=$pin MVGX_ADAPTER_TIMING=1 timeout 300 python tools/time_adapter_match.py > gpurun_out/adapter_match_2l_pin$pin.log 2>&1
echo "== pinned_results=$pin"; grep -v "^INFO" gpurun_out/adapter_match_2l_pin$pin.log | tail -16
done
