#!/bin/bash
# Round 6, session 10: conv_wt16 correctness (against the stream-K kernel; shadow bit-identity with it off; the benched configuration against the oracle) + finer phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_b256.py -q -m gpu -k "wide_tile or shadow_rings or benched" ) > gpurun_out/r6s10_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r6s10_tests.log
for cfg in "128 2" "128 3" "256 2"; do timeout 300 python tools/wt16_trace.py 256 $cfg 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s10_wt16_trace.log; done
