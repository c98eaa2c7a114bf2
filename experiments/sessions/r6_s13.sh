#!/bin/bash
# Round 6, session 13: CU-masked HIP streams -- the transmitter program and the two receiver programs on disjoint XCD sets --
# against the shared chip, alternating on one box (bench.py quick form)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check --no-op-profile"
for r in 1 2; do for m in "" "xcd:3,3,2" "block:3,3,2" "xcd:3,2,3" "xcd:8,8,8" "xcd:6,6,6"; do
  ADK_PIPE_CUMASK=$m timeout 600 python bench.py $ARGS > gpurun_out/r6s13.json 2> gpurun_out/r6s13.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s13.json").read().strip().splitlines()[-1])
    print("mask '$m' round $r: value", d["value"], "ms/step", d["ms_per_step"])
except Exception as e:
    print("mask '$m' round $r: no line:", e); print(open("gpurun_out/r6s13.err").read()[-600:])
PY
done; done 2>&1 | tee gpurun_out/r6s13_cu_masks.log
