#!/bin/bash
# Round 6, session 5: lazy-guard GPU tests again + the bench with its guard legs (value / unguarded / guard_direct_calls / guard_synchronous)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_lazy_guard.py -q -m gpu -x ) > gpurun_out/r6s5_guard.log 2>&1; echo "guard tests rc=$?"; tail -12 gpurun_out/r6s5_guard.log
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-t5 --no-self-check > gpurun_out/r6s5_bench.json 2> gpurun_out/r6s5_bench.err; echo "bench rc=$?"
tail -c 2600 gpurun_out/r6s5_bench.json; tail -3 gpurun_out/r6s5_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6s5_bench.json").read().strip().splitlines()[-1])
for k in ("unguarded", "guard_direct_calls", "guard_synchronous"):
    print(k, {a: b for a, b in d.get(k, {}).items() if a != "what"})
print("host", d.get("host_ms_per_step_of_each_rank", {}).get("issue"), d.get("guard_stats"))
PY
