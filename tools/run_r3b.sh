#!/bin/bash
# 256-stream pipeline with tile-aligned stream-K ranges: workgroup cap per launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
for v in "256 1" "256 0" "384 1" "512 1" "480 1" "256 1" "512 0"; do
  set -- $v
  ADK_BENCH_WORKGROUPS=$1 ADK_CONV_ALIGNED=$2 python bench.py $X > gpurun_out/r3b_wg$1_al$2_$RANDOM.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"), d["latency_ms"].get("encode_decode_single_stream_median"))
    except Exception as e: print(f,"ERR",e)
PY
