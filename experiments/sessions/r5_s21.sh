#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for B in 1; do
ADK_RB16_HELPERS=0 timeout 200 python tools/rb16_trace.py $B 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s21_trace_B$B.log; cat gpurun_out/s21_trace_B$B.log
done
