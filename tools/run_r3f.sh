#!/bin/bash
# stream-K main loop without register copies (ping-pong operand sets, activations two chunks ahead): timing + numerics vs the direct kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 d1 d2 d3 up0 up1 up2 in p o0 o1 o2 o3 r3 r2; do
  for B in 256 32 1; do
    echo "== $s B=$B split16 / f32"
    $K conv $s 4 $B 100 1
    $K conv $s 2 $B 100 1
  done
done
} > gpurun_out/r3f_pingpong.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3f_pingpong.log | sed 's/split16.*//') <(grep "^conv" gpurun_out/r3f_pingpong.log | awk '{print $7}' | paste -d' ' - -) <(grep "^conv" gpurun_out/r3f_pingpong.log | sed 's/.*max|d| vs impl 1 = //; s/ (|ref.*flags/ f/' | paste -d' ' - -)
