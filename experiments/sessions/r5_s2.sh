#!/bin/bash
# round 5, session 2: the balanced 128-channel chain -- bit identity, phase timeline, A/B through the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "balanced or benched or chains" ) > gpurun_out/s2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s2_tests.log; tail -12 gpurun_out/s2_tests.log
for v in 1 0; do
  ADK_RB16_BALANCE=$v timeout 300 python tools/rb16_trace.py 256 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s2_rb16_trace_balance$v.log; echo "trace $v rc=$?"
  grep -A8 "voc.stage1\|enc.block2" gpurun_out/s2_rb16_trace_balance$v.log | head -40
done
bash tools/ab_session.sh s2bal ADK_RB16_BALANCE 0 1 2 2>&1 | grep -v "^encoder\|^decoder.*rvq" | head -60
