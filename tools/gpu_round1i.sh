#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python -m pytest tests/test_matching_gpu.py -m gpu -q -p no:cacheprovider -x -k "43 and (adversarial or parity or golden)" 2>&1 | tail -5 > gpurun_out/pytest_i.log
tail -3 gpurun_out/pytest_i.log
grep -q "passed" gpurun_out/pytest_i.log || exit 1
grep -q "failed" gpurun_out/pytest_i.log && exit 1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_i -o v4 -- python tools/sweep_match.py --images 300 --rounds 2 --variants 43 > gpurun_out/prof_i.log 2>&1
head -4 gpurun_out/prof_i/v4_kernel_stats.csv | cut -c1-160; grep -a '"variant"' gpurun_out/prof_i.log
