#!/bin/bash
# Round 6, session 9: phase clocks of conv_wt16 (tile rows x stage buffers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "128 2" "128 3" "128 4" "256 2" "256 3"; do timeout 300 python tools/wt16_trace.py 256 $cfg 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s9_wt16_trace.log; done
