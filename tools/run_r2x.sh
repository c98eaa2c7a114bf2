#!/bin/bash
# 3-stream pipeline at 256 streams: cap on stream-K workgroups per tile
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
for ms in 5 0 3 4 6 5 0; do
  ADK_CONV_MAX_SPLIT=$ms python bench.py $X > gpurun_out/r2x_ms${ms}_$RANDOM.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2x_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"), d["latency_ms"].get("encode_decode_single_stream_median"))
    except Exception as e: print(f,"ERR",e)
PY
