#!/bin/bash
# single-stream (B = 1) per-op profile: where the 1.04 ms per frame goes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
ADK_SPLIT16=1 python tools/op_profile.py vctk_v1 1 1 > gpurun_out/r2p_ops_b1.txt 2>gpurun_out/r2p_ops_b1.err
ADK_SPLIT16=1 python tools/op_profile.py vctk_v1 8 1 > gpurun_out/r2p_ops_b8.txt 2>>gpurun_out/r2p_ops_b1.err
tail -3 gpurun_out/r2p_ops_b1.err
cat gpurun_out/r2p_ops_b1.txt
