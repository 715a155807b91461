#!/bin/bash
# round 3, call 12: BA tests + bench scenes after moving the symbolic phase to create and dropping the final cost pass
mkdir -p gpurun_out/r3_12
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_12
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ba.log
for s in c3 c5; do
timeout 300 python bench_ba.py $s --no-cpu > $O/ba_$s.json 2> $O/ba_$s.err; python -c "
import json; r=json.load(open('$O/ba_$s.json')); print('$s', r['lm_iteration_ms'], r['iterations'], r['solve_ms'], r['final_rmse'], r['create_s_host_structure_plus_upload'], r['roofline'])"
done
python tools/create_phase_table.py c5 5 | tail -4
