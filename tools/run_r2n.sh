#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
python bench.py $X > gpurun_out/r2n_base.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lbs/libaudiodec_hip.so python bench.py $X > gpurun_out/r2n_lbs.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lbs/libaudiodec_hip.so ADK_BENCH_WORKGROUPS=384 python bench.py $X > gpurun_out/r2n_lbs_wg384.json 2>/dev/null
python bench.py $X > gpurun_out/r2n_base2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2n_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"))
    except Exception as e: print(f,"ERR",e)
PY
