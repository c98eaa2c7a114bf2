#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_b256.py -q -m gpu -x -k "fixture or chains or configs_2_and_3 or benched" ) > gpurun_out/s13_tests.log 2>&1; tail -3 gpurun_out/s13_tests.log
for B in 1 8 32 64; do python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
bash tools/ab_multi.sh s13 X 1 1 2>&1 | tail -2
