#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r3w_ops.csv ) > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3w_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["latency_ms"]["encode_decode_at_batch_median"], d["latency_ms"]["encode_decode_single_stream_median"], d["roofline"]["frac"], d["self_check"]["ok"])
print(json.dumps(d["extra_configs"]))
PY
tail -3 gpurun_out/r3w_bench.err
