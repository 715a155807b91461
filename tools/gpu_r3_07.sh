#!/bin/bash
# round 3, call 7: state-of-the-tree verification - the full GPU suite, smoke(), the driver's bench command
mkdir -p gpurun_out/r3_07
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_07
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
(time timeout 1500 python bench.py) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r3_07/bench.json").read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mean_launch_ms'], r['parity']['identical'])
print('ba', r['ba']['lm_iteration_ms'], r['ba']['cpu_baseline'].get('rmse_diff_vs_reference'), r['ba']['phases'], r['ba']['create_s_host_structure_plus_upload'])
print('c5', r['ba_c5_single_gpu']['lm_iteration_ms'], r['ba_c5_single_gpu']['cpu_baseline'].get('rmse_diff_vs_reference'), r['ba_c5_single_gpu']['phases'], r['ba_c5_single_gpu']['create_s_host_structure_plus_upload'])
print({k: r[k].get('value') for k in ('hamming','l2_float','l2_uint8_144','geometric_filter') if k in r})
print(r['geometric_filter'].get('parity'), r['l2_uint8_144'].get('parity'))
PY
tail -3 $O/bench.err
