#!/bin/bash
# rebuild everything that travels to the GPU box (HIP library, oracle, adapter objects, adapter harness) before a gpurun call
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /tmp/prep_gpu.log 2>&1 || { tail -30 /tmp/prep_gpu.log; exit 1; }
ls -la openmvg_amd/lib/libmvgx_hip.so tests/native/_build/*.so | awk '{print $6, $7, $8, $9}'
