#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s3 s3d1 s2; do
  for B in 256 512; do
    echo "== $s B=$B rl16 / rp16 80 KB / 40 KB / 27 KB buffers"
    timeout 60 $K conv $s 4 $B 100 1
    for kb in 80 40 27; do ADK_RP16_BUF_KB=$kb ADK_CONV_RP16=1 timeout 60 $K conv $s 4 $B 100 1; done
  done
done
} > gpurun_out/r4a_rp16.log 2>&1
grep -E "^==|^conv" gpurun_out/r4a_rp16.log | sed 's/(algorithmic[^)]*)//; s/TF.*TB\/s//; s/max|d| vs impl 1 = //; s/(|ref|max [0-9.]*, nonfinite 0)//' | cut -c1-150
