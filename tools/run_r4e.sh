#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_b256.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r4e_tests.log
cat gpurun_out/r4e_tests.log
