#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_2p.log 2>&1; tail -2 gpurun_out/smoke_2p.log
timeout 400 python bench.py > gpurun_out/bench_2p.json 2> gpurun_out/bench_2p.err
python -c "
import json;d=json.load(open('gpurun_out/bench_2p.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'],d['ba']['lm_iteration_ms'],d['ba_c5_single_gpu']['lm_iteration_ms'],d['hamming']['value'],d['l2_float']['value'])"
