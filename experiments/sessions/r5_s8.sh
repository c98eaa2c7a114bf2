#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check --no-op-profile"
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $ARGS "$@" > gpurun_out/s8_$name.json 2> gpurun_out/s8_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s8_$name.json").read().strip().splitlines()[-1])
    print("$name: value", d["value"], "ms/step", d["ms_per_step"], "host", (d.get("host_ms_per_step_of_each_rank") or {}).get("issue"))
except Exception as e:
    print("$name: no line:", e); print(open("gpurun_out/s8_$name.err").read()[-600:])
PY
}
run base_a X=1 --
run groups2 X=1 -- --groups 2
run groups2_q8 GPU_MAX_HW_QUEUES=8 -- --groups 2
run q8 GPU_MAX_HW_QUEUES=8 --
run stages12 X=1 -- --stages 1,2
run stages1 X=1 -- --stages 1
run guard_off X=1 -- --guard off
run base_b X=1 --
