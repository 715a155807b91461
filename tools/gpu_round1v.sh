#!/bin/bash
# full GPU suite + headline bench.py + kernel-trace profile of the bench + HBM traffic counters of the filter kernel
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/pytest_v.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/bench_v.json 2> gpurun_out/bench_v.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba > gpurun_out/prof_v.log 2>&1
B="python bench.py --images 400 --steps 1 --warmup 0 --no-cpu-baseline --no-ba"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_v1 -o m -- $B > gpurun_out/pmc_v1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmc_v2 -o m -- $B > gpurun_out/pmc_v2.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_v.log 2>&1
tail -6 gpurun_out/pytest_v.log; cut -c1-1500 gpurun_out/bench_v.json; tail -3 gpurun_out/bench_v.err; tail -3 gpurun_out/smoke_v.log; head -6 gpurun_out/prof_v/b_kernel_stats.csv | cut -c1-200
