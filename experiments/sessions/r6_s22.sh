#!/bin/bash
# Round 6, session 22 (23: + the ring write in front of the Cin = 1 conv in that conv's launch, switched together): conv_oc16 (last conv_out + LeakyReLU + output conv + tanh as one launch): parity, bit-identity against the two-launch form, per-op table, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_b256.py tests/test_gpu_parity.py -q -m gpu -x -k "fused_residual_units or benched or residual_chains or two_stage or reference_fixture or batched_streams or chunked or range_overflow" ) > gpurun_out/r6s${S:-22}_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r6s${S:-22}_tests.log
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2; do for v in 0 1; do export ADK_CONV_CIN1W=$v
  ADK_CONV_OC16=$v timeout 600 python bench.py $ARGS --dump-ops gpurun_out/r6s${S:-22}_ops_v${v}_$r.csv > gpurun_out/r6s${S:-22}_v${v}_$r.json 2> gpurun_out/r6s${S:-22}_v${v}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s${S:-22}_v${v}_$r.json").read().strip().splitlines()[-1])
    print("ADK_CONV_OC16=$v round $r: value", d["value"], "single", d["summary"]["latency_ms"]["single_stream"], "batch", d["summary"]["latency_ms"]["batch"], "launches", d["summary"]["launches_per_step"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s${S:-22}_v${v}_$r.err").read()[-1500:])
PY
  grep -E "blocks.3.conv_out|output_conv|encoder,ring_write|encoder.conv," gpurun_out/r6s${S:-22}_ops_v${v}_$r.csv | cut -d, -f1-3,10,13
done; done
