#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{ for s in s2 s3; do ADK_CONV_RP16=1 LD_LIBRARY_PATH=tools/bin/rpdbg timeout 60 $K conv $s 4 512 50; done; } > gpurun_out/r3t_rp_trace.log 2>&1
cut -c1-160 gpurun_out/r3t_rp_trace.log
