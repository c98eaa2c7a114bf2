#!/bin/bash
# round 5, session 1: the deferred guard on the GPU (new tests + the pipelined parity tests) and a first bench line with it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 800 python -m pytest tests/test_gpu_pipeline_guard.py tests/test_gpu_b256.py -q -m gpu -x -k "guard or replay or benched or shadow or graph" --durations=8 ) > gpurun_out/s1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/s1_tests.log; tail -25 gpurun_out/s1_tests.log
( time timeout 500 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --dump-ops gpurun_out/s1_ops.csv ) > gpurun_out/s1_quick.json 2> gpurun_out/s1_quick.err
echo "bench rc=$?" >> gpurun_out/s1_quick.err; tail -5 gpurun_out/s1_quick.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s1_quick.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "guard_depth", "guard_stats", "unguarded", "guard_synchronous", "latency_ms")})
    print("T5", {k: d["roofline_convtr_T5"].get(k) for k in ("frac", "pipeline_frames_per_s", "self_check")})
    print("roof", {k: d["roofline"].get(k) for k in ("kernel", "frac", "frac_source", "frac_events_pipelined", "frac_events_serial")})
    print("self_check", d.get("self_check"))
except Exception as e:
    print("no json:", e)
PY
