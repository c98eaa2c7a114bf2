#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K=tools/bin/kbench
{ for s in e3 e2 e1 e0 r3 r1; do $K conv $s 4 256 200 1; done; } > gpurun_out/r3n_elu.log 2>&1
cat gpurun_out/r3n_elu.log | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "elu or conv_kernels_agree or causal_conv" 2>&1 | tail -8
