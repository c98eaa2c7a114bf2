#!/bin/bash
# gk16 with the conversion knocked out (= what a pre-split activation ring would cost): tools/bin/gkdbg1 built with -DADK_GK16_DBG=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 up1 d2; do
  echo "== $s: sk16 / gk16 full (128x128, 256x128, 128x256) / gk16 no conversion (same three)"
  $K conv $s 4 256 100
  for g in 4 2 3; do ADK_CONV_GK16=$g $K conv $s 8 256 100; done
  for g in 4 2 3; do ADK_CONV_GK16=$g LD_LIBRARY_PATH=tools/bin/gkdbg1 $K conv $s 8 256 100; done
done
} > gpurun_out/r2o_gk_knockout.log 2>&1
cat gpurun_out/r2o_gk_knockout.log
