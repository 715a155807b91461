#!/bin/bash
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 300 python -m pytest tests/test_matching_gpu.py -m gpu -q -p no:cacheprovider -k "variant4 or 4- or -4 or sink or drop_in or error" 2>&1 | tail -8 ) > gpurun_out/pytest_2c.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/pytest_2c.log
for bp in 131072 65536 32768; do
timeout 200 python bench.py --no-ba --no-cpu-baseline --batch-pairs $bp --steps 3 --warmup 1 > gpurun_out/bench_2c_bp$bp.json 2> gpurun_out/bench_2c_bp$bp.err
python -c "
import json;d=json.load(open('gpurun_out/bench_2c_bp$bp.json'));print($bp,d['value'],d['ms_per_step'],d['roofline']['mean_launch_ms'],d['roofline']['launches'],d['roofline']['frac'])"
done
