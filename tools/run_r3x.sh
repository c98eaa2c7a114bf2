#!/bin/bash
# HIP-graph replay at few streams: single-stream latency and SURVEY configs 1-3 with ADK_GRAPH=1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 100 --warmup 10 --no-cpu-baseline --no-self-check --no-other-precision --no-op-profile"
python bench.py $X > gpurun_out/r3x_nograph.json 2>/dev/null
ADK_GRAPH=1 python bench.py $X --graph 1 > gpurun_out/r3x_graph.json 2>gpurun_out/r3x_graph.err
python - <<'PY'
import json
for f in ("nograph","graph"):
    try:
        d=json.loads(open(f"gpurun_out/r3x_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["latency_ms"]["encode_decode_single_stream_median"], d["latency_ms"]["encode_decode_at_batch_median"], json.dumps(d["extra_configs"])[:420])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r3x_graph.err
