#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_pipeline_guard.py -q -m gpu -x -k "three_frames" ) > gpurun_out/s10.log 2>&1
grep -n "Error\|assert\|^E " gpurun_out/s10.log | head -40
