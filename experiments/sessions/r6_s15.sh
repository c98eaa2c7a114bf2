#!/bin/bash
# Round 6, session 15: conv_ou16 on eight waves / 16 x 16 x 32 MFMAs (ADK_OU16_V=3, default) against the four-wave form (ADK_OU16_V=2): tests, phase clocks, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units or benched" ) > gpurun_out/r6s15_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r6s15_tests.log
( ADK_OU16_V=2 timeout 600 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units" ) > gpurun_out/r6s15_tests_v2.log 2>&1; echo "tests (four-wave form) rc=$?"; tail -2 gpurun_out/r6s15_tests_v2.log
for v in 2 3; do for b in 256 1; do
  ADK_OU16_V=$v timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s15_trace.log
done; done
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2; do for v in 2 3; do
  ADK_OU16_V=$v timeout 600 python bench.py $ARGS --dump-ops gpurun_out/r6s15_ops_v${v}_$r.csv > gpurun_out/r6s15_v${v}_$r.json 2> gpurun_out/r6s15_v${v}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s15_v${v}_$r.json").read().strip().splitlines()[-1])
    print("ADK_OU16_V=$v round $r: value", d["value"], "single", d["summary"]["latency_ms"]["single_stream"], "batch", d["summary"]["latency_ms"]["batch"], "ou16 events serial frac", d["summary"]["north_star_kernel"]["frac_events_serial"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s15_v${v}_$r.err").read()[-1500:])
PY
  grep -E "conv_out,conv_ou16" gpurun_out/r6s15_ops_v${v}_$r.csv | cut -d, -f1-3,10,13
done; done
