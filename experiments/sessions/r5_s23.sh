#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_b256.py tests/test_offline.py tests/test_cabi.py -q -m gpu -x ) > gpurun_out/s23_tests.log 2>&1; tail -6 gpurun_out/s23_tests.log; grep -n "^E " gpurun_out/s23_tests.log | head
