#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{ for s in s0 s1 e3 o1 d3; do LD_LIBRARY_PATH=tools/bin/dbg32 $K conv $s 4 256 50; done; } > gpurun_out/r3i_wg.log 2>&1
cat gpurun_out/r3i_wg.log
