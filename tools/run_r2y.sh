#!/bin/bash
# 256 streams: where to cut the vocoder into concurrently running programs, and how many persistent workgroups each stream-K launch takes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
X="--steps 300 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
for st in 2 1,2 1,2,3 1 3 1,3; do
  for wg in 256 192 128; do
    ADK_BENCH_WORKGROUPS=$wg python bench.py $X --stages $st > gpurun_out/r2y_st${st}_wg${wg}.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2y_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"))
    except Exception as e: print(f,"ERR",e)
PY
