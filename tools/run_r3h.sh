#!/bin/bash
# interleaved iteration of the split stream-K kernel: timing + numerics (vs the direct kernel) + timeline; ilv0 = same code, three phases
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 d1 d2 d3 up0 up1 up2 in p o0 o1 o2 o3 r3; do
  for B in 256 32; do
    echo "== $s B=$B interleaved / three phases"
    $K conv $s 4 $B 100 1
    LD_LIBRARY_PATH=tools/bin/ilv0 $K conv $s 4 $B 100
  done
done
for s in s0 s1 e3; do LD_LIBRARY_PATH=tools/bin/dbg16 $K conv $s 4 256 50; done
} > gpurun_out/r3h_ilv.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3h_ilv.log | sed 's/interleaved.*//') <(grep "^conv" gpurun_out/r3h_ilv.log | head -76 | awk '{print $7}' | paste -d' ' - -) <(grep "max|d|" gpurun_out/r3h_ilv.log | sed 's/.*max|d| vs impl 1 = //; s/ (|ref.*flags/ f/')
grep -A30 "timeline" gpurun_out/r3h_ilv.log | grep -E "^conv|mean|it  [2-9]:|it 1[0-2]:" | head -40
