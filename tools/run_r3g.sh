#!/bin/bash
# per-iteration timeline of the split stream-K kernel (debug build -DADK_SK16_DBG=16)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s1 e3 o1; do
  LD_LIBRARY_PATH=tools/bin/dbg16 $K conv $s 4 256 50
done
LD_LIBRARY_PATH=tools/bin/dbg16 $K conv s0 4 1 50
} > gpurun_out/r3g_timeline.log 2>&1
cat gpurun_out/r3g_timeline.log
