#!/bin/bash
# Round 6, session 7: flag words in pinned host memory (ABI 14: a post is an event, no 1-thread kernel) -- every guard test, then what the guard costs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_lazy_guard.py tests/test_gpu_pipeline_guard.py tests/test_offline.py tests/test_gpu_parity.py -q -m gpu -x -k "guard or overflow or flags or error or wire or index" ) > gpurun_out/r6s7_guard.log 2>&1; echo "guard tests rc=$?"; tail -8 gpurun_out/r6s7_guard.log
bash experiments/sessions/r6_s6.sh
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-t5 --no-self-check > gpurun_out/r6s7_bench.json 2> gpurun_out/r6s7_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6s7_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "launches/step", d["summary"]["launches_per_step"])
for k in ("unguarded", "guard_direct_calls", "guard_synchronous"):
    print(k, {a: b for a, b in d.get(k, {}).items() if a != "what"})
print(d["summary"]["latency_ms"])
PY
