#!/bin/bash
# round 2, call 8: cascade hashing on the device (tests, adapter end to end), l2u8 record
mkdir -p gpurun_out/r2_08
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_08
timeout 900 python -m pytest tests/test_cascade.py tests/test_adapter_gpu.py tests/test_l2u8_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 1500 python tools/time_adapter_cascade.py > $O/adapter_cascade.jsonl 2> $O/adapter_cascade.err; cat $O/adapter_cascade.jsonl
timeout 300 python bench_hamming.py l2u8 > $O/l2u8.json 2> $O/l2u8.err; cut -c1-600 $O/l2u8.json
