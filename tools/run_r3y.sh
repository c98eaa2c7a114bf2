#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for cfg in "vctk_v1 1" "vctk_sym 1" "vctk_sym 32" "vctk_v1 64"; do
  for g in 0 1; do
    echo "== $cfg graph=$g"; SIDE=1 ADK_GRAPH=$g ADK_SPLIT16=1 python tools/hiccup.py $cfg 600 2>/dev/null | tail -2
  done
done
} > gpurun_out/r3y_graph_small.log 2>&1
cat gpurun_out/r3y_graph_small.log
