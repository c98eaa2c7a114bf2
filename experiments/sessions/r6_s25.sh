#!/bin/bash
# Round 6, session 25: conv_ou16 without a workgroup barrier between the GEMMs (LDS flags; -DADK_OU16_FLAGS=1, tools/dbg/ou1) against the barrier form
# (tools/dbg/ou0): tests with the product build (flags on), phase clocks alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units or benched" ) > gpurun_out/r6s25_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6s25_tests.log
for r in 1 2 3; do for v in 0 1; do for b in 256 1; do
  echo "flags=$v streams=$b round $r" | tee -a gpurun_out/r6s25_trace.log
  ADK_OU16_TRACE_LIB=$GRAFT_REPO_ROOT/tools/dbg/ou$v/libaudiodec_hip.so timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -E "median workgroup|GEMM|epilogue|span" | tee -a gpurun_out/r6s25_trace.log
done; done; done
