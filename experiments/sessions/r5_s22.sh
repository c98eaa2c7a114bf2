#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x ) > gpurun_out/s22_tests.log 2>&1; tail -6 gpurun_out/s22_tests.log
for B in 1 2 6 8 32; do python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
echo "-- ADK_GV16=0"
for B in 1 2 6; do ADK_GV16=0 python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
