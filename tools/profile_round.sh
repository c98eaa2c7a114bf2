#!/bin/bash
# The round's rocprofv3 evidence, taken over the TIMED STEPS of the pipelined bench schedule (three HIP streams) at HEAD:
#   tools/profile_round.sh <tag>      (through gpurun; results under gpurun_out/<tag>_*; copy what is judged into profiles/)
# 1. kernel trace -> per-kernel steady-state durations (tools/trace_summary.py)
# 2. FETCH_SIZE and WRITE_SIZE in two separate --pmc passes -> HBM-side traffic per launch (tools/pmc_summary.py)
# All three passes run `bench.py --pmc-markers` in the SAME schedule the headline is timed in (no --serial).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
STEPS=40
ARGS="--steps $STEPS --warmup 10 --pmc-markers --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check"
mkdir -p $R/gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/$tag/trace -o p --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/${tag}_trace.log 2>&1
echo "trace rc=$?"
python $R/tools/trace_summary.py $R/gpurun_out/$tag/trace $R/gpurun_out/${tag}_kernel_stats_steady.csv $STEPS | head -40
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$tag/pmc_$c -o p --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python $R/tools/pmc_summary.py $R/gpurun_out/$tag $R/gpurun_out/${tag}_pmc_traffic.csv | head -40
find $R/gpurun_out/$tag -name "*.db" -delete 2>/dev/null
find $R/gpurun_out/$tag -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
