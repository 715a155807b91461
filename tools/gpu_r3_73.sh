#!/bin/bash
# homography model of the geometric filter: GPU tests, throughput + parity count against the compiled reference
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_73}; mkdir -p $O
timeout 1200 python -m pytest tests/test_geofilter_h.py tests/test_geofilter_gpu.py tests/test_capi_symbols.py -q -m gpu 2>&1 | tail -15 | tee $O/pytest_geofilter_h.log
timeout 600 python tools/geofilter_h_run.py 20000 250 3000 2>&1 | tail -3 | tee $O/geofilter_h_run.json
