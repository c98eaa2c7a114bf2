#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_b256.py -q -m gpu -x -k "fixture or chains or configs_2_and_3 or extra" ) > gpurun_out/s6_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s6_tests.log; tail -8 gpurun_out/s6_tests.log
for B in 1 8 32 64; do python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
echo "-- ADK_RB16_PF_FEW=0 ADK_RB16_RING_FEW=3 (round-4 prefetch depths)"
for B in 1 8 32 64; do ADK_RB16_PF_FEW=0 ADK_RB16_RING_FEW=3 python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
echo "-- ADK_RB16_FEW128=256"
for B in 64 85; do ADK_RB16_FEW128=256 python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
for B in 85; do python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
