#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for B in 1 16; do
ADK_TRACE_LIB=$GRAFT_REPO_ROOT/tools/dbg/t1/libaudiodec_hip.so timeout 200 python tools/rb16_trace.py $B 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s5_rb16_trace_B$B.log; echo "== trace B=$B rc=$?"
cat gpurun_out/s5_rb16_trace_B$B.log
done
