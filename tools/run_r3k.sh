#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 e2; do
  for sh in 50 47 45 42 40 36; do echo "== $s share=$sh"; ADK_CONV_OWNER_SHARE=$sh $K conv $s 4 256 200 1; done
done
} > gpurun_out/r3k_share.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3k_share.log) <(grep "^conv" gpurun_out/r3k_share.log | awk '{print $7}') <(grep "max|d|" gpurun_out/r3k_share.log | sed 's/.*max|d| vs impl 1 = //; s/ (|ref.*flags/ f/')
