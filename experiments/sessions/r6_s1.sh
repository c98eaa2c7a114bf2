#!/bin/bash
# Round 6, session 1: the LDS-DMA-fed conv_ou16 (ADK_OU16_V=2, default) against the register-fed kernel of round 3 (ADK_OU16_V=1)
#   1. bit-identity + parity tests that run through it   2. phase timelines of both (debug build tools/dbg/ou1)   3. A/B through the quick bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units or benched" ) > gpurun_out/r6s1_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r6s1_tests.log
for v in 1 2; do for b in 256 1; do
  echo "== trace ADK_OU16_V=$v streams=$b"; ADK_OU16_V=$v timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s1_trace_v$v.log
done; done
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1; do for v in 1 2; do
  ADK_OU16_V=$v timeout 600 python bench.py $ARGS --dump-ops gpurun_out/r6s1_ops_v${v}_$r.csv > gpurun_out/r6s1_v${v}_$r.json 2> gpurun_out/r6s1_v${v}_$r.err
  echo "== ADK_OU16_V=$v round $r rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s1_v${v}_$r.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], d["summary"]["latency_ms"], d["summary"]["north_star_kernel"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s1_v${v}_$r.err").read()[-1500:])
PY
  grep -E "conv_out,conv_ou16|upsamples.3" gpurun_out/r6s1_ops_v${v}_$r.csv | cut -d, -f1-3,10,13
done; done
