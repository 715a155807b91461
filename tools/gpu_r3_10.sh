#!/bin/bash
# round 3, call 10: kernel timeline of one LM iteration (c5, c3) with the current kernels; HBM traffic of one iteration of both
# bench scenes (FETCH_SIZE / WRITE_SIZE passes) -> profiles/round3_ba_iteration_traffic.json (bench_ba.py's roofline.traffic)
mkdir -p gpurun_out/${CALL_DIR:-r3_10}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_10}
R=$GRAFT_REPO_ROOT
for s in c5 c3; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$s -o ba -- python $R/tools/ba_iterations.py $s 4 --warm > $O/prof_$s.log 2>&1)
  T=$(find $O/prof_$s -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T 1 > $O/ba_${s}_iteration_timeline.txt 2>&1
  find $O/prof_$s -name "*kernel_stats.csv" -exec cp {} $O/ba_${s}_kernel_stats.csv \;
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd_$s -o m -- python $R/tools/ba_iterations.py $s 3 --warm > $O/pmc_rd_$s.log 2>&1)
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr_$s -o m -- python $R/tools/ba_iterations.py $s 3 --warm > $O/pmc_wr_$s.log 2>&1)
  python tools/pmc_kernels.py $O/pmc_rd_$s --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/ba_iterations.py $s 3" > $O/ba_${s}_pmc_fetch.json 2> $O/s1.err
  python tools/pmc_kernels.py $O/pmc_wr_$s --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python tools/ba_iterations.py $s 3" > $O/ba_${s}_pmc_write.json 2> $O/s2.err
  rm -rf $O/prof_$s $O/pmc_rd_$s $O/pmc_wr_$s
done
python tools/ba_traffic_from_pmc.py $O/ba_c3_pmc_fetch.json $O/ba_c3_pmc_write.json $O/ba_c5_pmc_fetch.json $O/ba_c5_pmc_write.json > $O/ba_iteration_traffic.json
python -c "
import json; j=json.load(open('$O/ba_iteration_traffic.json'))
for k,v in j.items(): print(k, v['hbm_bytes_per_iteration'], v['read_bytes_x2'], v['written_bytes'])"
cat $O/ba_c5_iteration_timeline.txt
