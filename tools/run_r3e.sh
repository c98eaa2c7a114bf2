#!/bin/bash
# 8-wave 128x128 tiles of the split stream-K kernel (ADK_CONV_CFG=6) against the default pick
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 up0 up1 d2 d3 o0; do
  for B in 256 64; do
    echo "== $s B=$B default / cfg6 / cfg6 max split 3 / 4 / 8"
    $K conv $s 4 $B 100
    ADK_CONV_CFG=6 $K conv $s 4 $B 100 1
    for ms in 3 4 8; do ADK_CONV_CFG=6 ADK_CONV_MAX_SPLIT=$ms $K conv $s 4 $B 100; done
  done
done
} > gpurun_out/r3e_w8.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3e_w8.log | sed 's/default.*//') <(grep "^conv" gpurun_out/r3e_w8.log | awk '{print $7}' | paste -d' ' - - - - -) <(grep "max|d|" gpurun_out/r3e_w8.log | sed 's/.*max|d| vs impl 1 = //')
