#!/bin/bash
# round 3, call 30: state-of-the-tree verification - the full GPU suite, smoke(), the driver's bench command, and the bench's main
# pass under rocprofv3 --kernel-trace --stats (side records off) for the per-kernel summary
mkdir -p gpurun_out/${CALL_DIR:-r3_30}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_30}
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_all.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
(time timeout 1500 python bench.py) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"+os.environ.get("CALL_DIR","r3_30")+"/"
r=json.loads(open(O+"bench.json").read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mean_launch_ms'], r['parity']['identical'])
for k in ('ba','ba_c5_single_gpu'):
    b=r[k]; print(k, b['lm_iteration_ms'], b['iterations'], b['cpu_baseline'].get('rmse_diff_vs_reference'), b['phases'], b['create_s_host_structure_plus_upload'], b.get('create_s_first_call_in_process'), b['roofline'].get('traffic_over_algorithmic'))
print({k: r[k].get('value') for k in ('hamming','l2_float','l2_uint8_144','geometric_filter') if k in r})
print(r['geometric_filter'].get('parity'), r['l2_uint8_144'].get('parity'))
PY
tail -3 $O/bench.err
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --no-ba --no-hamming --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -8 $O/bench_kernel_stats.csv
python -c "
import json; r=json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1]); print('under rocprof: value', r['value'], 'mean_launch_ms', r['roofline']['mean_launch_ms'])"
rm -rf $O/prof
