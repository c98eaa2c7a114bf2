#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
t() { echo "== $*"; for B in 1 8 32; do env "$@" python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done; }
t ADK_RVQ_HELPERS=0
t ADK_RVQ_HELPERS=2
t ADK_RVQ_HELPERS=4
t ADK_RVQ_HELPERS=1
t ADK_RVQ_HELPERS=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -q -m gpu -x -k "rvq or fixture" ) > gpurun_out/s17_tests.log 2>&1; tail -3 gpurun_out/s17_tests.log
