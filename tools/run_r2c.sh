#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
echo "== up3 streamer (12 waves) =="
$K conv up3 7 256 300 2; $K conv up3 7 1 300; $K conv up3 7 64 300; $K conv up3 7 512 200; $K conv up3 7 1024 100
echo "== rvq =="
$K rvq 256 200; $K rvq 512 100
echo "== gk16 128x128 (default pick) vs old =="
for s in s0 s0d1 s1 s1d1 e2 up1 up2 d1 o1; do
  timeout 60 $K conv $s 4 256 100 2; ADK_CONV_GK16=0 timeout 60 $K conv $s 4 256 100
done
echo "== gk16 split caps on s0 / s1 / e2 =="
for sp in 2 3 6 8; do for s in s0 s1 e2; do ADK_GK16_SPLIT=$sp timeout 60 $K conv $s 4 256 100; done; done
echo "== forced shapes: 256x128 (2), 128x256 (3) with the coalesced slabs =="
for s in s0 s1; do ADK_CONV_GK16=2 timeout 60 $K conv $s 4 256 100; ADK_CONV_GK16=3 timeout 60 $K conv $s 4 256 100; done
echo "== forced 128x128 on the small-tile-count layers =="
for s in e3 up0 d2 d3 o0; do ADK_CONV_GK16=4 timeout 60 $K conv $s 4 256 100 2; done
} > gpurun_out/r2c_kbench.log 2>&1
cat gpurun_out/r2c_kbench.log
