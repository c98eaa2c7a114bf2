#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{ for s in s2 s3 e1 e0; do LD_LIBRARY_PATH=tools/bin/rldbg16 $K conv $s 4 256 50; done; } > gpurun_out/r3q_rl_wg.log 2>&1
cat gpurun_out/r3q_rl_wg.log
