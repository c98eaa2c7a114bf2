#!/bin/bash
# Round 6, session 28: conv_up16's output stores write-through (sc1, tools/dbg/up1) against default (up0) at 5 frames per call: rocprofv3 over the serial schedule, alternating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export ADK_PROFILE_CONFIG="streams=256 stages=2 frames_per_step=5 precision=split16 guard=default rvq=tx"
A5="--frames-per-step 5 --steps 60 --warmup 6 --preroll 16 --pmc-markers --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check --no-guarded --no-t5"
for r in 1 2; do for v in 0 1; do
  rm -rf /tmp/t5_$v
  ADK_LIB_PATH=$R/tools/dbg/up$v/libaudiodec_hip.so timeout 300 rocprofv3 --kernel-trace -d /tmp/t5_$v -o p --output-format csv -- python $R/bench.py $A5 --serial > /dev/null 2>&1
  python $R/tools/trace_summary.py /tmp/t5_$v $R/gpurun_out/r6s28_${v}_$r.csv 60 > /dev/null 2>&1
  echo "sc1=$v round $r: $(grep up16 $R/gpurun_out/r6s28_${v}_$r.csv | cut -d, -f1-8)"
done; done
