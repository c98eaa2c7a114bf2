#!/bin/bash
# Round 6, session 4: the lazy guard of direct calls (audiodec_amd/lazy_guard.py) -- its GPU tests, the guard tests around it, then the whole suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_lazy_guard.py tests/test_gpu_pipeline_guard.py -q -m gpu -x ) > gpurun_out/r6s4_guard.log 2>&1; echo "guard tests rc=$?"; tail -25 gpurun_out/r6s4_guard.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "overflow" ) > gpurun_out/r6s4_overflow.log 2>&1; echo "overflow tests rc=$?"; tail -25 gpurun_out/r6s4_overflow.log
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r6s4_tests.log 2>&1; echo "suite rc=$?"; tail -30 gpurun_out/r6s4_tests.log
