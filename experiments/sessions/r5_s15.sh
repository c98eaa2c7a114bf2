#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
t() { echo "== $*"; for B in 1 8 21 32; do env "$@" python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done; }
t ADK_RB16_HELPERS=0
t ADK_RB16_HELPERS=4
t ADK_RB16_HELPERS=8
t ADK_RB16_HELPERS=2
t ADK_RB16_HELPERS=4 ADK_RB16_HELPER_BLOCKS=128
t ADK_RB16_HELPERS=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_b256.py -q -m gpu -x -k "fixture or chains" ) > gpurun_out/s15_tests.log 2>&1; tail -3 gpurun_out/s15_tests.log
