#!/bin/bash
# more stream-K workgroups than slots: exact halves of the 300 tiles of the 128-channel grouped conv (600 workgroups, two rounds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s1 s1d1 s0; do
  echo "== $s default / oversub 20 % / 50 % / 100 %"
  $K conv $s 4 256 100 1
  for o in 20 50 100; do ADK_CONV_OVERSUB=$o $K conv $s 4 256 100 1; done
done
} > gpurun_out/r4d_oversub.log 2>&1
grep -E "^==|^conv" gpurun_out/r4d_oversub.log | sed 's/(algorithmic[^)]*)//; s/TF.*TB\/s//; s/max|d| vs impl 1 = //; s/(|ref|max [0-9.]*, nonfinite 0)//' | cut -c1-150
