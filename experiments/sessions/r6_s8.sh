#!/bin/bash
# Round 6, session 8: the wide-tile kernel conv_wt16 -- correctness against the stream-K kernel, then per-op times off / on
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "wide_tile or shadow_rings or benched" ) > gpurun_out/r6s8_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r6s8_tests.log
timeout 600 python tools/wt16_bench.py 256 2 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r6s8_wt16_bench.log
