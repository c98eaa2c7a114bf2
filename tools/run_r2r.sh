#!/bin/bash
# deep layers at few streams: how many K chunks per workgroup should the stream-K schedule hand out?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s1 e3 e2 d1 d2 d3 up0 up1 up2 in p o0 o1 r3 r2; do
  for B in 1 8 32; do
    echo "== $s B=$B: min units 2 / 4 / 8 / 16 / 1000"
    for mu in 2 4 8 16 1000; do ADK_CONV_MIN_UNITS=$mu $K conv $s 4 $B 100; done
  done
done
} > gpurun_out/r2r_min_units.log 2>&1
grep -c conv gpurun_out/r2r_min_units.log
