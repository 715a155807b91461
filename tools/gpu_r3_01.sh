#!/bin/bash
# round 3, call 1: the measurement holes VERDICT r2 names, on the round-2 kernels (the "before" of this round's BA rework):
#  (a) HBM traffic of one C5 LM iteration (FETCH_SIZE / WRITE_SIZE in separate passes), (b) MFMA-busy + clock pass of the final
#  l2_filter_kernel, (c) the new l2_uint8_144 workload (matches, cpu_baseline, parity)
mkdir -p gpurun_out/r3_01
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_01
R=$GRAFT_REPO_ROOT
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -o m -- python $R/tools/ba_iterations.py c5 3 > $O/pmc_rd.log 2>&1)
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -o m -- python $R/tools/ba_iterations.py c5 3 > $O/pmc_wr.log 2>&1)
python tools/pmc_kernels.py $O/pmc_rd --window 'ba_linearize_kernel<true>' --note "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/ba_iterations.py c5 3" > $O/ba_c5_pmc_fetch.json 2> $O/s1.err
python tools/pmc_kernels.py $O/pmc_wr --window 'ba_linearize_kernel<true>' --note "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python tools/ba_iterations.py c5 3" > $O/ba_c5_pmc_write.json 2> $O/s2.err
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_01/"
for f in ("ba_c5_pmc_fetch.json","ba_c5_pmc_write.json"):
    try:
        j=json.load(open(O+f)); print(f, j.get("window"), {k:v for k,v in j.items() if k.startswith("hbm")})
    except Exception as e: print(f,"failed",e)
PY
tail -2 $O/pmc_rd.log
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming"
(cd /tmp; timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- $B > $O/pmc_mfma.log 2>&1)
python tools/pmc_kernels.py $O/pmc_mfma --note "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming" > $O/match_filter_mfma_busy.json 2> $O/s3.err
find $O/pmc_mfma -name "*kernel_trace.csv" -exec cp {} $O/match_filter_mfma_busy_kernel_trace.csv \;
python tools/filter_busy_summary.py $O/match_filter_mfma_busy.json $O/match_filter_mfma_busy_kernel_trace.csv > $O/match_filter_mfma_busy_summary.json 2> $O/s4.err; cat $O/match_filter_mfma_busy_summary.json | head -30
tail -2 $O/pmc_mfma.log
timeout 300 python bench_hamming.py l2u8 > $O/bench_l2u8.json 2> $O/bench_l2u8.err; cut -c1-1500 $O/bench_l2u8.json
rm -rf $O/pmc_rd $O/pmc_wr $O/pmc_mfma $O/match_filter_mfma_busy_kernel_trace.csv
