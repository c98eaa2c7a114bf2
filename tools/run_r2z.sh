#!/bin/bash
# stream-K at 256 streams: workgroup counts that put whole tiles (or exact halves) on a workgroup
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1; do for G in 0 120 240 360 480 512; do echo "== $s G=$G"; ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
for s in e3; do for G in 0 80 160 240 320 400 480; do echo "== $s G=$G"; ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
for s in e2; do for G in 0 200 400 512; do echo "== $s G=$G"; ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
for s in s1 s1d1; do for G in 0 200 296 304 400 512; do echo "== $s G=$G"; ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
for s in s1 s1d1; do for G in 0 200 400 600; do echo "== $s cfg2 G=$G"; ADK_CONV_CFG=2 ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
for s in o1 up1 up2 d1 r2; do for G in 0 104 200 256 400; do echo "== $s G=$G"; ADK_CONV_MAX_SPLIT=0 ADK_CONV_G=$G $K conv $s 4 256 100; done; done
} > gpurun_out/r2z_G.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r2z_G.log) <(grep "^conv" gpurun_out/r2z_G.log | awk '{print $5, $7, $8}')
