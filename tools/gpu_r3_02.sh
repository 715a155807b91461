#!/bin/bash
# round 3, call 2: the fused point-group pass (no stored Jacobian for grouped points) - BA tests on the GPU, C3 / C5 timings,
# kernel timeline of one C5 iteration, HBM traffic of one C5 iteration (FETCH_SIZE / WRITE_SIZE passes)
mkdir -p gpurun_out/r3_02
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_02
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ba.log
timeout 300 python bench_ba.py c3 --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err; python -c "
import json; r=json.load(open('$O/ba_c3.json')); print('c3', r['lm_iteration_ms'], r['iterations'], r['final_rmse'], r['phases'], r['create_s_host_structure_plus_upload'])"
timeout 300 python bench_ba.py c5 > $O/ba_c5.json 2> $O/ba_c5.err; python -c "
import json; r=json.load(open('$O/ba_c5.json')); print('c5', r['lm_iteration_ms'], r['iterations'], r['final_rmse'], r['phases'], r['create_s_host_structure_plus_upload'])"
tail -3 $O/ba_c5.err
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ba -- python $R/tools/ba_iterations.py c5 4 > $O/prof.log 2>&1)
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T > $O/ba_c5_iteration_timeline.txt 2>&1; head -70 $O/ba_c5_iteration_timeline.txt
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/ba_c5_kernel_stats.csv \;
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -o ba -- python $R/tools/ba_iterations.py c3 4 > $O/prof3.log 2>&1)
T=$(find $O/prof3 -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T > $O/ba_c3_iteration_timeline.txt 2>&1; tail -3 $O/ba_c3_iteration_timeline.txt
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -o m -- python $R/tools/ba_iterations.py c5 3 > $O/pmc_rd.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -o m -- python $R/tools/ba_iterations.py c5 3 > $O/pmc_wr.log 2>&1)
python tools/pmc_kernels.py $O/pmc_rd --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/ba_iterations.py c5 3" > $O/ba_c5_pmc_fetch.json 2> $O/s1.err
python tools/pmc_kernels.py $O/pmc_wr --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python tools/ba_iterations.py c5 3" > $O/ba_c5_pmc_write.json 2> $O/s2.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_02/"
for f in ("ba_c5_pmc_fetch.json","ba_c5_pmc_write.json"):
    try:
        j=json.load(open(O+f)); print(f, j.get("window"), {k:v for k,v in j.items() if k.startswith("hbm")})
    except Exception as e: print(f,"failed",e)
PY
rm -rf $O/prof $O/prof3 $O/pmc_rd $O/pmc_wr
