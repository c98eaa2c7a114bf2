#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s1 e3 up0 d3 o2; do
  echo "-- $s: full / no X loads (4) / one W chunk (8) / both (12)"
  $K conv $s 4 256 100
  for d in 4 8 12; do LD_LIBRARY_PATH=tools/bin/dbg$d $K conv $s 4 256 100; done
done
} > gpurun_out/r2i_kbench2.log 2>&1
cat gpurun_out/r2i_kbench2.log
