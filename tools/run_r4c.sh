#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/r4c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4c_tests.log
tail -6 gpurun_out/r4c_tests.log
