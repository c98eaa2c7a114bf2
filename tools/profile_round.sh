#!/bin/bash
# The round's rocprofv3 evidence, taken over the TIMED STEPS of the bench schedules at HEAD:
#   tools/profile_round.sh <tag> [serial-only|no-pmc]   (through gpurun; results under gpurun_out/<tag>_*; copy what is judged into profiles/)
# 1. kernel trace of the PIPELINED schedule (three HIP streams: what `value` is timed in) -> <tag>_kernel_stats_steady.csv
# 2. kernel trace of `--serial` (one HIP stream, nothing else on the chip)               -> <tag>_kernel_stats_serial.csv
#    (tools/trace_summary.py: per-kernel dispatch durations between the two --pmc-markers; bench.py quotes both as *_rocprof_*)
# 3. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes over the pipelined schedule -> HBM-side traffic per launch (tools/pmc_summary.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
what=${2:-all}      # all | serial-only | no-pmc | T5 (only the 5-frames-per-call captures)
STEPS=40
ARGS="--steps $STEPS --warmup 10 --pmc-markers --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check --no-guarded --no-t5"
mkdir -p $R/gpurun_out/$tag
# what the summaries record as `# bench_config:` (bench.py: a capture of another configuration is stale); must match bench_config_string() of $ARGS
export ADK_PROFILE_CONFIG="streams=256 stages=2 frames_per_step=1 precision=split16 guard=default rvq=${ADK_BENCH_RVQ:-tx}"
if [ "$what" != "serial-only" ] && [ "$what" != "T5" ]; then
  timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/trace -o p --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/${tag}_trace.log 2>&1
  echo "trace rc=$?"
  python $R/tools/trace_summary.py $R/gpurun_out/$tag/trace $R/gpurun_out/${tag}_kernel_stats_steady.csv $STEPS | head -40
fi
if [ "$what" != "T5" ]; then
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/trace_serial -o p --output-format csv -- python $R/bench.py $ARGS --serial > $R/gpurun_out/${tag}_trace_serial.log 2>&1
echo "serial trace rc=$?"
python $R/tools/trace_summary.py $R/gpurun_out/$tag/trace_serial $R/gpurun_out/${tag}_kernel_stats_serial.csv $STEPS | head -40
fi
if [ "$what" = "all" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$tag/pmc_$c -o p --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
    echo "$c rc=$?"
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/$tag $R/gpurun_out/${tag}_pmc_traffic.csv | head -40
fi
if [ "$what" = "all" ] || [ "$what" = "T5" ]; then
  # T5: the same captures at the reference streamer's default chunk (5 frames per stream per call, demoStream.py:28) -- where the north-star's
  # named kernel runs as conv_up16 on 5x the rows: <tag>_kernel_stats_T5_{steady,serial}.csv, <tag>_pmc_traffic_T5.csv
  export ADK_PROFILE_CONFIG="streams=256 stages=2 frames_per_step=5 precision=split16 guard=default rvq=${ADK_BENCH_RVQ:-tx}"
  S5=100   # (the in-step figure of the 5-frames-per-call kernel from >= 100 launches, not 16)
  A5="--frames-per-step 5 --steps $S5 --warmup 6 --preroll 16 --pmc-markers --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check --no-guarded --no-t5"
  mkdir -p $R/gpurun_out/${tag}_T5
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_T5/trace -o p --output-format csv -- python $R/bench.py $A5 > $R/gpurun_out/${tag}_T5_trace.log 2>&1
  echo "T5 trace rc=$?"
  python $R/tools/trace_summary.py $R/gpurun_out/${tag}_T5/trace $R/gpurun_out/${tag}_kernel_stats_T5_steady.csv $S5 | head -14
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_T5/trace_serial -o p --output-format csv -- python $R/bench.py $A5 --serial > $R/gpurun_out/${tag}_T5_trace_serial.log 2>&1
  echo "T5 serial trace rc=$?"
  python $R/tools/trace_summary.py $R/gpurun_out/${tag}_T5/trace_serial $R/gpurun_out/${tag}_kernel_stats_T5_serial.csv $S5 | head -14
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${tag}_T5/pmc_$c -o p --output-format csv -- python $R/bench.py $A5 > $R/gpurun_out/${tag}_T5_pmc_$c.log 2>&1
    echo "T5 $c rc=$?"
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/${tag}_T5 $R/gpurun_out/${tag}_pmc_traffic_T5.csv | head -14
  find $R/gpurun_out/${tag}_T5 -name "*.db" -delete 2>/dev/null
  find $R/gpurun_out/${tag}_T5 -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
fi
find $R/gpurun_out/$tag -name "*.db" -delete 2>/dev/null
find $R/gpurun_out/$tag -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
