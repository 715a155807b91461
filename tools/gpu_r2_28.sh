#!/bin/bash
# round 2, call 28: the window-fold epilogue of the filter kernel (kDbg bit 3): parity tests, then the bench headline with and without
mkdir -p gpurun_out/r2_28
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_28
timeout 900 python -m pytest tests/test_matching_gpu.py tests/test_real_images.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in 0 16 8 0 16; do
  MVGX_MATCH_FILTER=$v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ba --no-hamming > $O/bench_f$v.json 2> $O/bench_f$v.err
  python - <<PY
import json
r=json.loads(open("$O/bench_f$v.json").read().strip().splitlines()[-1])
print("filter variant $v:", "%.4g" % r['value'], "ms/step %.2f" % r['ms_per_step'], "frac %.4f" % r['roofline']['frac'], "launch ms %.3f" % r['roofline']['mean_launch_ms'], "matches", r['config']['matches_rank0'])
PY
done
