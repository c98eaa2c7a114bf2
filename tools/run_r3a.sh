#!/bin/bash
# tile-aligned stream-K ranges: on / off per layer and stream count, with a numerics check against the direct kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 e3 e2 d1 d2 d3 up0 up1 up2 in p o0 o1 o2 r3 r2; do
  for B in 1 32 64 256; do
    echo "== $s B=$B: legacy / aligned / aligned with a 256-workgroup cap"
    ADK_CONV_ALIGNED=0 $K conv $s 4 $B 100
    $K conv $s 4 $B 100 1
    echo -n "cap256: "; ADK_CONV_G=256 $K conv $s 4 $B 100
    echo -n "cap256 legacy: "; ADK_CONV_ALIGNED=0 ADK_CONV_G=256 $K conv $s 4 $B 100
  done
done
} > gpurun_out/r3a_aligned.log 2>&1
grep -c "^conv\|cap256" gpurun_out/r3a_aligned.log
