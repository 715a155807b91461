#!/bin/bash
# round 2, call 2: streaming + multi-device contexts (matching, BA) on the one GPU of the box; adapter end to end
mkdir -p gpurun_out/r2_02
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_02
timeout 1200 python -m pytest tests/test_matching_gpu.py tests/test_adapter_gpu.py tests/test_ba_multirank_gpu.py tests/test_ba_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python bench.py --no-ba --no-hamming --no-cpu-baseline --steps 5 > $O/bench_stream.json 2> $O/bench_stream.err
timeout 300 python bench.py --no-ba --no-hamming --no-cpu-baseline --steps 5 --collect > $O/bench_collect.json 2> $O/bench_collect.err
for f in stream collect; do python -c "
import json,sys; r=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mean_launch_ms'])"; done
MVGX_ADAPTER_TIMING=1 timeout 600 python tools/time_adapter_match.py > $O/adapter_match.log 2>&1; grep -v "^\[mvgx" $O/adapter_match.log | tail -5
MVGX_DEVICES=0,0 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_2ctx.log 2>&1; grep -v "^\[mvgx" $O/adapter_match_2ctx.log | tail -5
