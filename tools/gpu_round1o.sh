#!/bin/bash
# AoS Jacobian records + adapter split: full parity suite, C3/C5 bench + kernel stats
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/pytest_o.log 2>&1
timeout 300 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_o.json 2> gpurun_out/bench_ba_o.err
timeout 300 python bench_ba.py c5 > gpurun_out/bench_ba_c5_o.json 2> gpurun_out/bench_ba_c5_o.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o -o ba -- python bench_ba.py c3 --no-cpu > gpurun_out/prof_o.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o5 -o ba5 -- python bench_ba.py c5 > gpurun_out/prof_o5.log 2>&1
tail -12 gpurun_out/pytest_o.log; cut -c1-700 gpurun_out/bench_ba_o.json; echo; cut -c1-700 gpurun_out/bench_ba_c5_o.json
