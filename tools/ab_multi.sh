#!/bin/bash
# tools/ab_multi.sh <tag> <ENV_NAME> <rounds> <value> [<value> ...]: the quick bench alternately with ENV_NAME set to each value, `rounds` times.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; name=$2; rounds=$3; shift 3
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    f=$(echo "$v" | tr '/ ,' '___' | tail -c 40)
    env $name=$v timeout 300 python bench.py $ARGS > gpurun_out/${tag}_${f}_$r.json 2> gpurun_out/${tag}_${f}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_${f}_$r.json").read().strip().splitlines()[-1])
    print("$name=$v round $r: value", d["value"], "ms/step", d["ms_per_step"], "batch latency", d["latency_ms"].get("encode_decode_at_batch_median"))
except Exception as e:
    print("$name=$v round $r: no line:", e); print(open("gpurun_out/${tag}_${f}_$r.err").read()[-800:])
PY
  done
done
