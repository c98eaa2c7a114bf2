#!/bin/bash
# Round 6, session 2: conv_ou16 variants under the phase clock (tools/ou16_trace.py): V=1 register-fed, V=2 DMA-fed with / without the
# hand-placed A-fragment prefetch of GEMM 1 (tools/dbg/ou1 = product flags + stamps, tools/dbg/ou2 = -DADK_OU16_APF=0)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units" ) > gpurun_out/r6s2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6s2_tests.log
for cfg in "1 ou1" "2 ou1" "2 ou2"; do set -- $cfg; for b in 256 1; do
  echo "== trace ADK_OU16_V=$1 lib=$2 streams=$b"; ADK_OU16_V=$1 ADK_OU16_TRACE_LIB=$GRAFT_REPO_ROOT/tools/dbg/$2/libaudiodec_hip.so timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s2_trace.log
done; done
