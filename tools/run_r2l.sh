#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
python bench.py $X > gpurun_out/r2l_base.json 2>/dev/null
ADK_CONV_BK16=1 python bench.py $X > gpurun_out/r2l_bk16.json 2>/dev/null
ADK_CONV_GK16=1 python bench.py $X > gpurun_out/r2l_gk16.json 2>/dev/null
python bench.py $X --stages 1 > gpurun_out/r2l_stages1.json 2>/dev/null
python bench.py $X --frames-per-step 2 > gpurun_out/r2l_fps2.json 2>/dev/null
python bench.py $X --frames-per-step 5 > gpurun_out/r2l_fps5.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2l_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"))
    except Exception as e: print(f,"ERR",e)
PY
