#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for B in 1 8 32 64; do python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_b256.py -q -m gpu -x -k "fixture or chains" ) > gpurun_out/s19_tests.log 2>&1; tail -2 gpurun_out/s19_tests.log
