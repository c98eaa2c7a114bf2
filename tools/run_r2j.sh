#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
echo "== bk16 128x128 register-staged (auto pick) vs 64-wide (ADK_CONV_BK16=0) =="
for s in s0 s0d1 s1 s1d1 e2 up1 d1; do
  timeout 60 $K conv $s 4 256 100 2; ADK_CONV_BK16=0 timeout 60 $K conv $s 4 256 100
done
echo "== forced on layers the heuristic skips =="
for s in e3 up0 up2 d2 d3 o0 o1 r2 r3; do timeout 60 $K conv $s 9 256 100 2; done
echo "== split caps =="
for sp in 1 2 8; do for s in s0 s1 e2; do ADK_BK16_SPLIT=$sp timeout 60 $K conv $s 4 256 100; done; done
echo "== other stream counts (forced) =="
for b in 64 128 512; do for s in s0 s1; do timeout 60 $K conv $s 9 $b 100 2; ADK_CONV_BK16=0 timeout 60 $K conv $s 4 $b 100; done; done
} > gpurun_out/r2j_kbench.log 2>&1
cat gpurun_out/r2j_kbench.log
