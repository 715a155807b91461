#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python bench.py --no-ba --no-hamming --no-cpu-baseline > gpurun_out/bench_2o.json 2> gpurun_out/bench_2o.err
python -c "
import json;d=json.load(open('gpurun_out/bench_2o.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
( timeout 200 python -m pytest tests/test_matching_gpu.py tests/test_adapter_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > gpurun_out/pytest_2o.log 2>&1
cat gpurun_out/pytest_2o.log
timeout 100 python tools/time_adapter_match.py 2>&1 | grep -v INFO | tail -4
