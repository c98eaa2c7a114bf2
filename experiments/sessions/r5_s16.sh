#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for h in 0 4; do
ADK_RB16_HELPERS=$h timeout 200 python tools/rb16_trace.py 1 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s16_trace_h$h.log; echo "== helpers $h rc=$?"
grep -A7 "voc.stage1\|voc.stage2\|enc.block2" gpurun_out/s16_trace_h$h.log | grep "==\|conv 0\|conv 1\|conv 4\|stage-in"
done
