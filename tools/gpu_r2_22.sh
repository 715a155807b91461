#!/bin/bash
# round 2, call 22: the N = 8 workload (BASELINE configs[3]: 10 000 images, 5.0e7 image pairs) on ONE GPU, one pass, to check the
# large-scale path (pair list, work lists, streaming) before the driver's multi-GPU run
mkdir -p gpurun_out/r2_22
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_22
timeout 900 python tools/with_peak_rss.py python bench.py --images 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming > $O/bench_10000.json 2> $O/bench_10000.err
echo "rc=$?"; tail -c 1500 $O/bench_10000.json; grep -E "peak_rss|elapsed" $O/bench_10000.err
