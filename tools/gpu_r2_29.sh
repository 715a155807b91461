#!/bin/bash
# round 2, call 29: does the per-workgroup start-up / tail of the filter kernel matter? the same 2.0e12 descriptor pairs as longer
# database streams per workgroup (descriptors per image 2000 / 4000 / 8000 -> 8 / 16 / 32 LDS windows per workgroup)
mkdir -p gpurun_out/r2_29
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_29
for cfg in "1000 2000" "500 4000" "250 8000" "2000 1000"; do
  set -- $cfg
  timeout 300 python bench.py --images $1 --desc $2 --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-hamming > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - <<PY
import json
r=json.loads(open("$O/bench_$1x$2.json").read().strip().splitlines()[-1])
print("$1 images x $2:", "%.4g pairs/s" % r['value'], "ms/step %.2f" % r['ms_per_step'], "kernel frac %.4f" % r['roofline']['frac'], "launch ms %.3f" % r['roofline']['mean_launch_ms'], "launches", r['roofline']['launches'])
PY
done
