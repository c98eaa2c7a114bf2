#!/bin/bash
# Round 6, session 29: conv_ou16's output stores write-through (sc1, tools/dbg/st16) against default (st0) inside the three-stream schedule: bench value, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2 3 4; do for v in 0 16; do
  ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/st$v/libaudiodec_hip.so timeout 600 python bench.py $ARGS > gpurun_out/r6s29_v${v}_$r.json 2> gpurun_out/r6s29_v${v}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s29_v${v}_$r.json").read().strip().splitlines()[-1])
    n = d["summary"]["north_star_kernel"]
    print("aux=$v round $r: value", d["value"], "single", d["summary"]["latency_ms"]["single_stream"], "batch", d["summary"]["latency_ms"]["batch"], "ou16 events serial", n["frac_events_serial"], "pipelined us", d["summary"]["north_star_kernel_launch_us"])
except Exception as e:
    print("no line:", e)
PY
done; done
