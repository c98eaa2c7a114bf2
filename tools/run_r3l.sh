#!/bin/bash
# after the copy-free, interleaved stream-K loop: full GPU suite + bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -q -m gpu --durations=8 -x ) > gpurun_out/r3l_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3l_tests.log
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r3l_ops.csv ) > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err
echo "bench rc=$?" >> gpurun_out/r3l_bench.err
tail -15 gpurun_out/r3l_tests.log; tail -c 1500 gpurun_out/r3l_bench.json
