#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
echo "== up3 streamer, conversion interleaved =="
$K conv up3 7 256 300 2; $K conv up3 7 256 300; $K conv up3 7 1 300; $K conv up3 7 1024 100
echo "== stream-K split kernel: prefetch distance 2 (default) vs 1 =="
for s in s0 s0d1 s1 s1d1 e3 e2 up0 up1 up2 d0 d1 d2 d3 o0 o1 o2 r2 r3 in p; do
  timeout 60 $K conv $s 4 256 100 2; ADK_SK16_PD=1 timeout 60 $K conv $s 4 256 100
done
echo "== single stream =="
for s in s0 s1 e3 up0 d3 p; do timeout 60 $K conv $s 4 1 100 2; ADK_SK16_PD=1 timeout 60 $K conv $s 4 1 100; done
} > gpurun_out/r2d_kbench.log 2>&1
cat gpurun_out/r2d_kbench.log
