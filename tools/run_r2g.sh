#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo skip tests > gpurun_out/r2g_tests.log
echo "tests rc=$?" >> gpurun_out/r2g_tests.log
( time timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --dump-ops gpurun_out/r2g_ops.csv ) > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err
echo "bench rc=$?" >> gpurun_out/r2g_bench.err
( time ADK_FUSE=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision ) > gpurun_out/r2g_bench_nofuse.json 2> gpurun_out/r2g_bench_nofuse.err
tail -8 gpurun_out/r2g_tests.log; python - <<'PY'
import json
for f in ("gpurun_out/r2g_bench.json","gpurun_out/r2g_bench_nofuse.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["latency_ms"], d.get("self_check",{}).get("ok"), d.get("roofline_convtr"), d.get("kernels"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r2g_bench.err
