#!/bin/bash
# round 2, call 12: the driver's bench command under rocprofv3 --kernel-trace --stats (kernel summary for profiles/), then plain
mkdir -p gpurun_out/r2_12
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_12
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2_12/prof_bench/bench_kernel_stats.csv")))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
PY
(time timeout 1200 python bench.py) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r2_12/bench.json").read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mean_launch_ms'], r['parity']['identical'])
print('ba', r['ba']['lm_iteration_ms'], r['ba']['cpu_baseline'].get('rmse_diff_vs_reference'), 'c5', r['ba_c5_single_gpu']['lm_iteration_ms'], r['ba_c5_single_gpu']['cpu_baseline'].get('rmse_diff_vs_reference'))
print({k: r[k].get('value') for k in ('hamming','l2_float','l2_uint8_144') if k in r})
PY
tail -3 $O/bench.err
