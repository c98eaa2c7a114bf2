#!/bin/bash
# round 5, session 4: where a single-stream step spends its 0.75 ms -- per-op events and a rocprofv3 kernel trace (kernel time vs gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s4
ADK_SPLIT16=1 ADK_VOCODER_STAGES=1 ADK_GUARD=0 timeout 200 python $R/tools/op_profile.py vctk_v1 1 1 2>&1 | grep -v "^Load" > $R/gpurun_out/s4_ops_B1.txt; echo "ops rc=$?"
cat $R/gpurun_out/s4_ops_B1.txt
for B in 1 32; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/s4/trace$B -o p --output-format csv -- python $R/tools/single_stream_steps.py $B 40 > $R/gpurun_out/s4_trace$B.log 2>&1; echo "trace rc=$?"
  tail -1 $R/gpurun_out/s4_trace$B.log
  python $R/tools/trace_summary.py $R/gpurun_out/s4/trace$B $R/gpurun_out/s4_kernel_stats_B$B.csv 40 | head -34
done
find $R/gpurun_out/s4 -name "*.db" -delete 2>/dev/null
