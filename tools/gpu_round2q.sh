#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 100000 1; do
MVGX_BA_TWO_LEVEL_MIN_N=$t timeout 100 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_c3_2q_t$t.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/bench_ba_c3_2q_t$t.json'));print($t,d['lm_iteration_ms'],d['phases']['solve_ms'],d['final_rmse'])"
done
