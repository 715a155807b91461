#!/bin/bash
# round 3, call 6: the geometric filter on the MI355X - tests against the compiled reference, bench record
mkdir -p gpurun_out/r3_06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_06
timeout 900 python -m pytest tests/test_geofilter_gpu.py -m gpu -q -x > $O/pytest_geofilter.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_geofilter.log
timeout 600 python bench_geofilter.py > $O/bench_geofilter.json 2> $O/bench_geofilter.err; cat $O/bench_geofilter.json | cut -c1-2500; tail -3 $O/bench_geofilter.err
