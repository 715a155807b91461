#!/bin/bash
# round 3, call 36: the adapter harness rebuilt from the current adapter objects (calls 25 - 35 ran the adapter tests / timings with a
# harness linked at 18:13, before the indexed geometric filter, the device hashing stage and the threaded scene flattening):
# adapter GPU tests, cascade-hashing TU end to end, Adjust() by size with phases
mkdir -p gpurun_out/r3_36
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_36
timeout 900 python -m pytest tests/test_adapter_gpu.py tests/test_geofilter_gpu.py tests/test_cascade.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 600 python tools/time_adapter_cascade.py > $O/adapter_cascade.jsonl 2> $O/err.log; cat $O/adapter_cascade.jsonl
timeout 600 python tools/time_adapter_ba_sizes.py --no-ref > $O/adapter_ba_sizes.jsonl 2>> $O/err.log; cat $O/adapter_ba_sizes.jsonl
MVGX_ADAPTER_TIMING=1 python - <<PY 2>&1 | tail -8
import sys; sys.path.insert(0,".")
from openmvg_amd import synth
from tests import _oracle
sc = synth.ba_scene(n_cams=200, n_points=100000, track_len=10, model=3, n_intr_groups=1, seed=0xAD1A + 200)
for _ in range(3):
    rc, st, *_ = _oracle.ref_ba_adjust(sc, lib=_oracle.adapter()); print("Adjust ms", st[2]*1e3, flush=True)
PY
