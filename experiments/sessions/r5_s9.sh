#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python tools/rvq_hot_cold.py 2>&1 | grep -v "^Load" | tee gpurun_out/s9_rvq_hot_cold.txt
