#!/bin/bash
# round 2, call A: new parity tests first (fail-fast off so everything reports), then bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py tests/test_gpu_soak.py -q -m gpu -x --durations=10 ) > gpurun_out/r2a_newtests.log 2>&1
echo "newtests rc=$?" >> gpurun_out/r2a_newtests.log
( time timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_b256.py --deselect tests/test_gpu_soak.py --durations=5 ) > gpurun_out/r2a_oldtests.log 2>&1
echo "oldtests rc=$?" >> gpurun_out/r2a_oldtests.log
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r2a_ops.csv ) > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.err
tail -3 gpurun_out/r2a_newtests.log; tail -3 gpurun_out/r2a_oldtests.log; tail -c 1500 gpurun_out/r2a_bench.json
