#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "noaddl" 2>&1 | tail -12 > gpurun_out/r4b_tests.log
cat gpurun_out/r4b_tests.log
