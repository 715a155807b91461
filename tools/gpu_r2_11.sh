#!/bin/bash
# round 2, call 11: point-group size experiment (42 points / 2 workgroups per CU, 32 / 3, 24 / 4)
mkdir -p gpurun_out/r2_11
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_11
for v in "" _g32 _g24; do
  for cfg in c3x c5; do
    MVGX_LIB_PATH=$GRAFT_REPO_ROOT/openmvg_amd/lib/libmvgx_hip$v.so timeout 300 python bench_ba.py $cfg --no-cpu > $O/ba_${cfg}$v.json 2> $O/ba_${cfg}$v.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_11/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3f" % r["lm_iteration_ms"], r["iterations"], "%.6f" % r["final_rmse"], "schur %.3f" % r["phases"]["schur_ms"])
    except Exception as e: print(f, "ERR", e)
PY
