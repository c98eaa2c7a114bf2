#!/bin/bash
# small / medium stream counts: stream-K vs rows-in-LDS (with shorter time tiles) on the layers the rows kernel supports
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in e0 e1 r0 r1 s2 s2d1 s3 s3d1; do
  for B in 1 8 32 64 128; do
    echo "== $s B=$B: auto / stream-K / rows (default tt, 32, 64, 128)"
    $K conv $s 4 $B 100
    $K conv $s 6 $B 100
    $K conv $s 5 $B 100
    for tt in 32 64 128; do ADK_RL16_TT=$tt $K conv $s 5 $B 100; done
  done
done
} > gpurun_out/r2q_small_batch.log 2>&1
grep -c conv gpurun_out/r2q_small_batch.log
