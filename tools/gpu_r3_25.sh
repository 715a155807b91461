#!/bin/bash
# round 3, call 25: device hashing stage of CASCADE_HASHING_L2 - tests, end-to-end Cascade_Hashing_Matcher_Regions::Match timing
mkdir -p gpurun_out/r3_25
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_25
timeout 900 python -m pytest tests/test_cascade.py tests/test_adapter_gpu.py tests/test_capi_symbols.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/time_adapter_cascade.py > $O/adapter_cascade.jsonl 2> $O/err.log; cat $O/adapter_cascade.jsonl; tail -2 $O/err.log
