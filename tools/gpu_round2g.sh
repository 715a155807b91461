#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_2g.log 2>&1; tail -3 gpurun_out/smoke_2g.log
timeout 400 python bench.py > gpurun_out/bench_2g.json 2> gpurun_out/bench_2g.err; tail -c 600 gpurun_out/bench_2g.json; tail -2 gpurun_out/bench_2g.err
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_2g -o bench -- python bench.py --no-cpu-baseline --no-ba-c5 --steps 2 > gpurun_out/bench_2g_prof.json 2> gpurun_out/bench_2g_prof.err
find gpurun_out/prof_2g -name "*kernel_stats*" | head
( time timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 2>&1 | tail -25 ) > gpurun_out/pytest_2g.log 2>&1
tail -22 gpurun_out/pytest_2g.log
