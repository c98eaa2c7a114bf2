#!/bin/bash
# stream-K fix-up: sc1 reads without the acquire fence, owner halves on the later-dispatched blocks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 e2 e3 d3 d2 up0 o0 in p s1; do
  for B in 256 32; do
    echo "== $s B=$B both / fence + plain reads / owners first"
    $K conv $s 4 $B 200 1
    LD_LIBRARY_PATH=tools/bin/nosc1 $K conv $s 4 $B 200
    LD_LIBRARY_PATH=tools/bin/noown $K conv $s 4 $B 200
  done
done
LD_LIBRARY_PATH=tools/bin/dbg32 $K conv s0 4 256 50
} > gpurun_out/r3j_fixup.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3j_fixup.log | sed 's/both.*//') <(grep "^conv" gpurun_out/r3j_fixup.log | head -66 | awk '{print $7}' | paste -d' ' - - -) <(grep "max|d|" gpurun_out/r3j_fixup.log | sed 's/.*max|d| vs impl 1 = //; s/ (|ref.*flags/ f/')
grep -A40 "per-workgroup" gpurun_out/r3j_fixup.log | head -44
