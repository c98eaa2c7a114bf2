#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
for g in 1 2 4; do python bench.py $X --groups $g > gpurun_out/r3z_g$g.json 2>/dev/null; done
python - <<'PY'
import json
for g in (1,2,4):
    try:
        d=json.loads(open(f"gpurun_out/r3z_g{g}.json").read().strip().splitlines()[-1]); print(g, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"))
    except Exception as e: print(g, "ERR", e)
PY
