#!/bin/bash
# round 5, session 24: the single-stream step of the FINAL build (few-streams lowering + conv_gv16) by per-op events and a rocprofv3 kernel trace:
# the "after" of profiles/r5_single_stream_latency.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s24
ADK_SPLIT16=1 ADK_VOCODER_STAGES=1 ADK_GUARD=0 timeout 200 python $R/tools/op_profile.py vctk_v1 1 1 2>&1 | grep -v "^Load" > $R/gpurun_out/s24_ops_B1.txt; echo "ops rc=$?"
cat $R/gpurun_out/s24_ops_B1.txt
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/s24/trace1 -o p --output-format csv -- python $R/tools/single_stream_steps.py 1 40 > $R/gpurun_out/s24_trace1.log 2>&1; echo "trace rc=$?"
tail -1 $R/gpurun_out/s24_trace1.log
python $R/tools/trace_summary.py $R/gpurun_out/s24/trace1 $R/gpurun_out/s24_kernel_stats_B1.csv 40 | head -40
find $R/gpurun_out/s24 -name "*.db" -delete 2>/dev/null
