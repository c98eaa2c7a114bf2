#!/bin/bash
# A/B of a library knob through the bench, in one GPU-box session:  tools/ab_session.sh <tag> <ENV_NAME> <value_a> <value_b> [rounds]
# Runs the quick bench (no CPU legs / extra configs / other precision / self-check) alternately with ENV_NAME=value_a and =value_b,
# `rounds` times each (default 2), and prints the headline, the serial per-op times of the chain launches and the batch latency.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; name=$2; va=$3; vb=$4; rounds=${5:-2}
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in $(seq 1 $rounds); do
  for v in $va $vb; do
    f=$(echo "$v" | tr '/ ' '__' | tail -c 40)          # (a value may be a path: ADK_LIB_PATH)
    env $name=$v timeout 600 python bench.py $ARGS --dump-ops gpurun_out/${tag}_ops_${f}_$r.csv > gpurun_out/${tag}_${f}_$r.json 2> gpurun_out/${tag}_${f}_$r.err
    echo "== $name=$v round $r rc=$?"
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_${f}_$r.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch latency", d["latency_ms"].get("encode_decode_at_batch_median"), "single", d["latency_ms"].get("encode_decode_single_stream_median"))
    print({k: (v["ms_per_step_serial"], v["ms_per_step"]) for k, v in d["kernels"].items() if k.startswith("conv_rb16")})
except Exception as e:
    print("no line:", e); print(open("gpurun_out/${tag}_${f}_$r.err").read()[-1500:])
PY
    grep -E "conv_rb16|rvq" gpurun_out/${tag}_ops_${f}_$r.csv | cut -d, -f1-3,10,13
  done
done
