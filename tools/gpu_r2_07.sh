#!/bin/bash
# round 2, call 7: where the filter kernel's time goes (parts removed), plus the MFMA ceiling on descriptor-like operands
mkdir -p gpurun_out/r2_07
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_07
timeout 600 python tools/filter_breakdown.py 400 > $O/filter_breakdown.json 2> $O/filter_breakdown.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r2_07/filter_breakdown.json").read().strip().splitlines()[-1])
for x in r["runs"]: print(x["debug_filter"], x["what"].ljust(34), "%.2f ms" % x["kernel_ms"], "%.0f TOPS" % x["tops"], "%.3f" % x["frac_of_5000"])
PY
