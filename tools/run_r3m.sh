#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "streamk_schedules" 2>&1 | tail -15 > gpurun_out/r3m_tests.log
cat gpurun_out/r3m_tests.log
