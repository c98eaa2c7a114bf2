#!/bin/bash
# tile shapes of the split stream-K kernel under tile-aligned ranges, 256 streams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 o0 o1 up0 up1 d2 d3; do
  for c in -1 0 1 2 3; do
    echo "== $s cfg=$c"; if [ $c -lt 0 ]; then $K conv $s 4 256 100; else ADK_CONV_CFG=$c $K conv $s 4 256 100; fi
  done
done
} > gpurun_out/r3d_cfg.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3d_cfg.log) <(grep "^conv" gpurun_out/r3d_cfg.log | awk '{print $5, $7, $8}')
