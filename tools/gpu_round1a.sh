#!/bin/bash
# first GPU session: parity tests, variant sweep, bench line, rocprof kernel trace
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 300 python tools/sweep_match.py --images 300 --rounds 5 --out gpurun_out/sweep_300.json > gpurun_out/sweep.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench.log 2> gpurun_out/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_match -o match -- python bench.py --images 400 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
ls -R gpurun_out/prof_match | head -30
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/sweep.log | tail -5; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err
