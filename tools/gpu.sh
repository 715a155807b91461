#!/bin/bash
# ONE parameterised script for the MI355X calls of a round (replaces the one-shot tools/gpu_rN_MM.sh files of rounds 1 - 3, which
# stay in git history):   gpurun --timeout S -- 'CALL=r4_03 bash tools/gpu.sh suite bench geopmc'
# Every mode writes under gpurun_out/$CALL/ (merged back into the repository's gpurun_out/; what is kept is copied to profiles/).
# rocprofv3 rules of the pool: counters in their own runs, only --kernel-trace beside --pmc, never with sys / hip / hsa trace domains.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${CALL:-r4_xx}; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$R"

pmc() {   # pmc <tag> "<counters>" <command...>   -> $O/pmc_<tag>/ (csv)
  local tag=$1 ctr=$2; shift 2
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$O/pmc_$tag" -o m -- "$@" > "$O/pmc_$tag.log" 2>&1)
}

for mode in "$@"; do
  echo "=== $mode"
  case $mode in
  suite)      # the driver's round-end checks: GPU tests + smoke
    timeout 1500 python -m pytest tests -m gpu -q -x > "$O/pytest_gpu_all.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu_all.log"
    timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log" ;;
  tests:*)    # tests:<pytest args>, e.g. tests:tests/test_real_images.py
    sel=${mode#tests:}; sel=${sel//+/ }   # ('+' separates several paths / arguments)
    timeout 1500 python -m pytest $sel -m gpu -q -s > "$O/pytest_sel.log" 2>&1; echo "pytest rc=$?"; tail -8 "$O/pytest_sel.log" ;;
  bench)      # the driver's command
    (time timeout 1500 python bench.py) > "$O/bench.json" 2> "$O/bench.err"; tail -4 "$O/bench.err"
    echo "stdout lines: $(wc -l < "$O/bench.json"), last line bytes: $(tail -1 "$O/bench.json" | wc -c)"
    cp gpurun_out/bench_side.json "$O/bench_full.json" 2>/dev/null; cp gpurun_out/bench_stderr.log "$O/bench_reference_stderr.log" 2>/dev/null
    python tools/bench_summary.py "$O/bench_full.json" ;;
  benchprof)  # rocprofv3 --kernel-trace --stats of the headline leg (same command, side records off)
    (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bench" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-ba --no-hamming > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err")
    find "$O/prof_bench" -name "*kernel_stats.csv" -exec cp {} "$O/bench_kernel_stats.csv" \; ; rm -rf "$O/prof_bench"
    head -6 "$O/bench_kernel_stats.csv" | cut -c1-160 ;;
  matchpmc)   # HBM traffic of one pass over the bench workload (filter kernel): TCC request counters, the guide's x2 on reads
    pmc match "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming
    python tools/pmc_traffic_summary.py "$O/pmc_match" 499500 "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming" > "$O/match_traffic_pmc.json" 2> "$O/matchpmc.err"
    python -c "import json;print(json.load(open('$O/match_traffic_pmc.json')).get('filter_kernel'))"; rm -rf "$O/pmc_match" ;;
  matchbusy)  # matrix-pipe busy cycles + sustained clock of the filter kernel
    pmc busy "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming
    python tools/pmc_kernels.py "$O/pmc_busy" --note "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 0 (headline leg)" > "$O/match_filter_busy_pmc.json" 2> "$O/matchbusy.err"
    python tools/filter_busy_summary.py "$O/match_filter_busy_pmc.json" $(find "$O/pmc_busy" -name "*kernel_trace.csv" | head -1) > "$O/match_filter_mfma_busy.json" 2>> "$O/matchbusy.err"
    grep -E "busy_frac|clock_ghz|valu_per_mfma|busy_x_clock" "$O/match_filter_mfma_busy.json"; rm -rf "$O/pmc_busy" ;;
  geopmc)     # SQ counters of the geometric-filter kernel (F, H and E), two passes each
    for m in f h e; do
      pmc geo_a_$m "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" python "$R/tools/geofilter_run.py" 20000 250 $m
      pmc geo_b_$m "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" python "$R/tools/geofilter_run.py" 20000 250 $m
      python tools/pmc_kernels.py "$O/pmc_geo_a_$m" --note "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python tools/geofilter_run.py 20000 250 $m" > "$O/geofilter_${m}_pmc_a.json" 2> "$O/geo_a_$m.err"
      python tools/pmc_kernels.py "$O/pmc_geo_b_$m" --note "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -- python tools/geofilter_run.py 20000 250 $m" > "$O/geofilter_${m}_pmc_b.json" 2> "$O/geo_b_$m.err"
      tail -1 "$O/pmc_geo_a_$m.log"; rm -rf "$O/pmc_geo_a_$m" "$O/pmc_geo_b_$m"
    done
    python tools/geofilter_pmc_summary.py "$O" ;;
  batrace)    # kernel timeline of one LM iteration, both scenes
    for s in c3 c5; do
      (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$s" -o t -- python "$R/tools/ba_iterations.py" $s 4 --warm > "$O/trace_$s.log" 2>&1)
      python tools/ba_timeline.py $(find "$O/trace_$s" -name "*kernel_trace.csv" | head -1) > "$O/ba_${s}_iteration_timeline.txt" 2>&1; tail -4 "$O/ba_${s}_iteration_timeline.txt"; rm -rf "$O/trace_$s"
    done ;;
  batraffic)  # HBM bytes of one LM iteration (FETCH_SIZE and WRITE_SIZE in separate passes)
    for s in c3 c5; do
      pmc fetch_$s "FETCH_SIZE" python "$R/tools/ba_iterations.py" $s 3 --warm
      pmc write_$s "WRITE_SIZE" python "$R/tools/ba_iterations.py" $s 3 --warm
      python tools/pmc_kernels.py "$O/pmc_fetch_$s" --window ba_cam_gram_kernel --note "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/ba_iterations.py $s 3 --warm" > "$O/ba_${s}_iteration_pmc_fetch.json" 2>> "$O/batraffic.err"
      python tools/pmc_kernels.py "$O/pmc_write_$s" --window ba_cam_gram_kernel --note "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python tools/ba_iterations.py $s 3 --warm" > "$O/ba_${s}_iteration_pmc_write.json" 2>> "$O/batraffic.err"
      rm -rf "$O/pmc_fetch_$s" "$O/pmc_write_$s"
    done
    python tools/ba_traffic_from_pmc.py "$O/ba_c3_iteration_pmc_fetch.json" "$O/ba_c3_iteration_pmc_write.json" "$O/ba_c5_iteration_pmc_fetch.json" "$O/ba_c5_iteration_pmc_write.json" > "$O/ba_iteration_traffic.json" 2>> "$O/batraffic.err"; tail -3 "$O/batraffic.err"; head -c 600 "$O/ba_iteration_traffic.json" ;;
  bawaves)    # SQ wave counters of one LM iteration per kernel
    for s in c5 c3; do
      pmc wa_$s "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" python "$R/tools/ba_iterations.py" $s 3 --warm
      pmc wb_$s "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES" python "$R/tools/ba_iterations.py" $s 3 --warm
      python tools/pmc_kernels.py "$O/pmc_wa_$s" --window ba_cam_gram_kernel --note "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -- python tools/ba_iterations.py $s 3 --warm" > "$O/ba_${s}_pmc_waves_a.json" 2>> "$O/bawaves.err"
      python tools/pmc_kernels.py "$O/pmc_wb_$s" --window ba_cam_gram_kernel --note "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES -- python tools/ba_iterations.py $s 3 --warm" > "$O/ba_${s}_pmc_waves_b.json" 2>> "$O/bawaves.err"
      rm -rf "$O/pmc_wa_$s" "$O/pmc_wb_$s"
    done ;;
  baiter)     # LM iteration time of both scenes, three repetitions (tools/ba_iterations.py)
    for rep in 1 2 3; do for s in c3 c5; do echo "$s $(python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a "$O/ba_iterations.txt"; done; done ;;
  baphases)   # phases of an LM iteration (HIP events around them: a little slower than the plain solve), three repetitions, optional env:... in front
    for rep in 1 2 3; do for s in c3 c5; do echo "$s $(MVGX_BA_PHASE_TIMING=1 python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a "$O/ba_phases.txt"; done; done ;;
  bastamps)   # MVGX_BA_GROUP_DEBUG phase stamps of ba_point_group_kernel per mode (thread 0 of every workgroup, shader clocks), both scenes
    for s in c3 c5; do echo "== $s" | tee -a "$O/ba_group_stamps.txt"; MVGX_BA_GROUP_DEBUG=1 python tools/ba_iterations.py $s 8 --warm 2>&1 | grep -E "point-group kernel|iter_ms" | tee -a "$O/ba_group_stamps.txt"; done ;;
  baab)       # same-box A/B of a BA change: tools/_build/libmvgx_prev.so (tools/build_prev_lib.sh) against the tree, alternating
    for rep in 1 2 3; do for s in c3 c5; do
      echo "tree $s $(python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a "$O/ba_ab.txt"
      echo "prev $s $(MVGX_LIB_PATH=$R/tools/_build/libmvgx_prev.so python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a "$O/ba_ab.txt"
    done; done ;;
  balibab)    # same-box A/B of BA builds: LIBS = "tree <name> ..." (tools/_build/libmvgx_<name>.so), alternating, both scenes
    for rep in 1 2 3; do for s in c3 c5; do for which in ${LIBS:-tree prev}; do
      lib=""; [ $which != tree ] && lib=$R/tools/_build/libmvgx_$which.so
      echo "$which $s $(MVGX_LIB_PATH=$lib python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a "$O/ba_lib_ab.txt"
    done; done; done ;;
  geolibab)   # same-box A/B of geometric-filter builds: LIBS = "tree <name> ...", the three models, alternating
    for rep in 1 2 3; do for m in f h e; do for which in ${LIBS:-tree prev}; do
      lib=""; [ $which != tree ] && lib=$R/tools/_build/libmvgx_$which.so
      echo "$which $(MVGX_LIB_PATH=$lib python tools/geofilter_run.py 20000 250 $m 2>/dev/null | head -1)" | tee -a "$O/geo_lib_ab.txt"
    done; done; done ;;
  matchab)    # the filter kernel on the two MFMA shapes, alternating, headline leg only
    for rep in 1 2 3; do for shape in ${SHAPES:-16 32}; do
      python bench.py --filter-shape $shape --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-hamming 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'filter_shape': $shape, 'value': r['value'], 'frac': r['roofline']['frac'], 'mean_launch_ms': r['roofline']['mean_launch_ms'], 'ms_per_step': r['ms_per_step'], 'kernel': r['roofline']['kernel']}))" | tee -a "$O/match_filter_shape_ab.jsonl"
    done; done ;;
  matchlibab) # same-box A/B of a matching change: tools/_build/libmvgx_exp.so (an experimental build) against the tree, alternating, headline leg; DESC / IMAGES: the set
    for rep in 1 2 3; do for which in ${LIBS:-tree exp}; do
      lib=""; [ $which != tree ] && lib=$R/tools/_build/libmvgx_$which.so
      MVGX_LIB_PATH=$lib python bench.py ${FSHAPE:+--filter-shape $FSHAPE} --images ${IMAGES:-1000} --desc ${DESC:-2000} --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-hamming 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'library': '$which${FSHAPE:+ shape $FSHAPE}', 'descriptors_per_image': ${DESC:-2000}, 'value': r['value'], 'frac': r['roofline']['frac'], 'mean_launch_ms': r['roofline']['mean_launch_ms'], 'ms_per_step': r['ms_per_step']}))" | tee -a "$O/match_lib_ab.jsonl"
    done; done ;;
  descsweep)  # the filter kernel's fraction of the i8 peak by descriptors per image (the per-workgroup start and the per-image finish amortise)
    for dsc in 1000 2000 4000 8000; do
      n=$((2000000 / dsc)); [ $n -gt 1000 ] && n=1000
      python bench.py --images $n --desc $dsc --steps 3 --warmup 1 --no-cpu-baseline --no-ba --no-hamming 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'images': $n, 'descriptors_per_image': $dsc, 'value': r['value'], 'frac': r['roofline']['frac'], 'mean_launch_ms': r['roofline']['mean_launch_ms'], 'ms_per_step': r['ms_per_step']}))" | tee -a "$O/match_frac_by_descriptors.jsonl"
    done ;;
  env:*)      # env:NAME=VALUE applies to the modes that follow
    export "${mode#env:}" ;;
  run:*)      # run:<script under tools/ or repo-relative python file with args, '+' for spaces>
    cmd=${mode#run:}; cmd=${cmd//+/ }
    timeout 1200 python $cmd 2>&1 | tee "$O/run_$(echo "$cmd" | tr -c 'A-Za-z0-9_.' '_' | cut -c1-60).log" | tail -15 ;;
  fuzzother:*)  # fuzzother:<seconds>[:seed] - the Hamming / float / uint8-144 matchers against the reference's own
    IFS=: read -r _ so seed <<< "$mode"; seed=${seed:-1}
    timeout $((so + 600)) python tools/fuzz_gpu.py other "$so" "$seed" 2>&1 | grep -v "^INFO" | tee "$O/fuzz_other_seed$seed.txt" | tail -5 ;;
  fuzz:*)     # fuzz:<seconds match>:<seconds ba>[:seed] - randomised parity campaign against the compiled reference (tools/fuzz_gpu.py)
    IFS=: read -r _ sm sb seed <<< "$mode"; seed=${seed:-1}
    timeout $((sm + 600)) python tools/fuzz_gpu.py match "$sm" "$seed" 2>&1 | grep -v "^INFO" | tee "$O/fuzz_match_seed$seed.txt" | tail -5
    timeout $((sb + 900)) python tools/fuzz_gpu.py ba "$sb" "$seed" 2>&1 | grep -v "^INFO" | tee "$O/fuzz_ba_seed$seed.txt" | tail -8 ;;
  *) echo "unknown mode $mode" ;;
  esac
done
