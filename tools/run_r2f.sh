#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "graph" --durations=5 ) > gpurun_out/r2f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2f_tests.log
( time timeout 600 python bench.py --steps 200 --warmup 20 --graph 1 --no-cpu-baseline --no-extra-configs ) > gpurun_out/r2f_bench_graph.json 2> gpurun_out/r2f_bench_graph.err
echo "bench graph rc=$?" >> gpurun_out/r2f_bench_graph.err
( time timeout 600 python bench.py --steps 200 --warmup 20 --graph 0 --no-cpu-baseline --no-extra-configs ) > gpurun_out/r2f_bench_eager.json 2> gpurun_out/r2f_bench_eager.err
tail -6 gpurun_out/r2f_tests.log; python - <<'PY'
import json
for f in ("gpurun_out/r2f_bench_graph.json","gpurun_out/r2f_bench_eager.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["latency_ms"], d.get("self_check",{}).get("ok"), d.get("other_precision"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r2f_bench_graph.err
