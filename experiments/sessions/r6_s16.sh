#!/bin/bash
# Round 6, session 16: conv_ou16 eight-wave form with the A fragments of GEMM 1 requested one step ahead: fused-vs-unfused test + phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units" ) > gpurun_out/r6s16_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6s16_tests.log
for b in 256 1; do ADK_OU16_V=3 timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s16_trace.log; done
