#!/bin/bash
# Round 6, session 24: where the runtime keeps kernel arguments (HIP_FORCE_DEV_KERNARG = 0 / 1): bench A/B, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2; do for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 600 python bench.py $ARGS > gpurun_out/r6s24_v${v}_$r.json 2> gpurun_out/r6s24_v${v}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s24_v${v}_$r.json").read().strip().splitlines()[-1])
    print("HIP_FORCE_DEV_KERNARG=$v round $r: value", d["value"], "single", d["summary"]["latency_ms"]["single_stream"], "batch", d["summary"]["latency_ms"]["batch"], "ou16 events serial", d["summary"]["north_star_kernel"]["frac_events_serial"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s24_v${v}_$r.err").read()[-1500:])
PY
done; done
timeout 300 python bench.py $ARGS > gpurun_out/r6s24_default.json 2> /dev/null; python -c "
import json; d = json.loads(open('gpurun_out/r6s24_default.json').read().strip().splitlines()[-1]); print('unset: value', d['value'], 'single', d['summary']['latency_ms']['single_stream'])"
