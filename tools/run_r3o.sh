#!/bin/bash
# tile shapes again, with the copy-free interleaved loop (vector-memory bytes per MFMA now matter)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=tools/bin/kbench
{
for s in s0 s0d1 s1 s1d1 e3 e2 d2 d3 up0 o0; do
  for c in -1 0 2 3 6; do
    echo "== $s cfg=$c"; if [ $c -lt 0 ]; then $K conv $s 4 256 100; else ADK_CONV_CFG=$c $K conv $s 4 256 100; fi
  done
done
} > gpurun_out/r3o_cfg.log 2>&1
paste -d' ' <(grep "==" gpurun_out/r3o_cfg.log) <(grep "^conv" gpurun_out/r3o_cfg.log | awk '{print $5, $7, $8}')
