#!/bin/bash
# kernel-level profile of matching variant 4 (filter + verify)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v4 -o v4 -- python tools/sweep_match.py --images 300 --rounds 3 --variants 43 > gpurun_out/prof_v4.log 2>&1
cat gpurun_out/prof_v4/v4_kernel_stats.csv | cut -c1-160; tail -3 gpurun_out/prof_v4.log
