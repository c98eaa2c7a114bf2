#!/bin/bash
# stream-K: cap on the workgroups sharing one tile, across stream counts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s1 e3 e2 d1 d2 d3 up0 up1 up2 in p o0 o1 r3; do
  for B in 1 32 64 256; do
    echo "== $s B=$B: max split 0 / 3 / 4 / 5 / 6 / 8"
    for ms in 0 3 4 5 6 8; do ADK_CONV_MAX_SPLIT=$ms $K conv $s 4 $B 100; done
  done
done
} > gpurun_out/r2s_max_split.log 2>&1
grep -c conv gpurun_out/r2s_max_split.log
