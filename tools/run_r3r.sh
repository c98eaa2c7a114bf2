#!/bin/bash
# pipelined rows kernel (ADK_CONV_RP16=1) against conv_rl16: time, bits (out#), numerics vs the direct kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s2 s2d1 s3 s3d1; do
  for B in 256 128 512; do
    echo "== $s B=$B rl16 / rp16"
    timeout 60 $K conv $s 4 $B 100 1
    ADK_CONV_RP16=1 timeout 60 $K conv $s 4 $B 100 1
  done
done
} > gpurun_out/r3r_rp16.log 2>&1
grep -E "^==|^conv" gpurun_out/r3r_rp16.log | sed 's/(algorithmic[^)]*)//; s/TF.*TB\/s//' | cut -c1-190
