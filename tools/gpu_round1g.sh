#!/bin/bash
# matching v2 iteration: quick parity subset + profile
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_matching_gpu.py -m gpu -q -p no:cacheprovider -x -k "41 or 43 or oneshot" 2>&1 | tail -15 > gpurun_out/pytest_g.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v4 -o v4 -- python tools/sweep_match.py --images 300 --rounds 3 --variants 41,43 > gpurun_out/prof_v4.log 2>&1
tail -5 gpurun_out/pytest_g.log; head -4 gpurun_out/prof_v4/v4_kernel_stats.csv | cut -c1-160; grep -a '"variant"' gpurun_out/prof_v4.log
