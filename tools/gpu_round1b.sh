#!/bin/bash
# second GPU session: BA parity tests + smoke + BA bench + rocprof of BA
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_ba.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 600 python bench_ba.py > gpurun_out/bench_ba.log 2> gpurun_out/bench_ba.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba -o ba -- python -c "
import bench_ba, json
print(json.dumps(bench_ba.ba_bench_record(0, 1, cpu=False)))" > gpurun_out/prof_ba.log 2>&1
ls -R gpurun_out/prof_ba | head
tail -30 gpurun_out/pytest_ba.log; cat gpurun_out/smoke.log | tail -5; cat gpurun_out/bench_ba.log; tail -5 gpurun_out/bench_ba.err
