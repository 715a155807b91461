#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python bench_ba.py c3 --no-cpu > gpurun_out/bench_ba_c3_2n.json 2> gpurun_out/bench_ba_c3_2n.err
timeout 200 python bench_ba.py c5 --no-cpu > gpurun_out/bench_ba_c5_2n.json 2> gpurun_out/bench_ba_c5_2n.err
python - <<'PY'
import json
for n in ("c3","c5"):
    d=json.load(open(f"gpurun_out/bench_ba_{n}_2n.json"))
    print(n, d["lm_iteration_ms"], d["iterations"], d["phases"], d["reduced_solve"], d["create_s_host_structure_plus_upload"])
PY
tail -2 gpurun_out/bench_ba_c3_2n.err
