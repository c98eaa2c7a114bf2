#!/bin/bash
# the bench lines again (bench.py changed: the guard-mode legs take the median of three regions); traces / PMC passes of this build are in profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/gpu_session.sh r5 bench 2>&1 | tail -3 | cut -c1-300
( timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_20_steps.json 2> gpurun_out/r5_bench_20_steps.err ); echo "bench20 rc=$?"
