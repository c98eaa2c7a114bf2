#!/bin/bash
# round 5, session 3: loop variants of the balanced 128-channel chain (B fragments a step ahead, lockstep interval, weight prefetch distance):
# phase timelines of the debug builds t1..t4, then the product builds through the bench against the two-stream split
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for n in 1 2 3 4; do
  ADK_TRACE_LIB=$GRAFT_REPO_ROOT/tools/dbg/t$n/libaudiodec_hip.so timeout 200 python tools/rb16_trace.py 256 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s3_trace_t$n.log; echo "== trace t$n rc=$?"
  grep -A8 "voc.stage1" gpurun_out/s3_trace_t$n.log | head -10
done
for dp in "0,0,0" "5.5,7.5,11" "3,4,6"; do
  ADK_RB16_BALANCE=0 ADK_RB16_DEPHASE=$dp ADK_TRACE_LIB=$GRAFT_REPO_ROOT/tools/dbg/t1/libaudiodec_hip.so timeout 200 python tools/rb16_trace.py 256 1 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/s3_trace_dephase_$dp.log; echo "== trace dephase $dp rc=$?"
  grep "^== " gpurun_out/s3_trace_dephase_$dp.log
done
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check --no-op-profile"
run() {   # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS > gpurun_out/s3_$name.json 2> gpurun_out/s3_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/s3_$name.json").read().strip().splitlines()[-1])
    print("$name: value", d["value"], "ms/step", d["ms_per_step"], "batch latency", d["latency_ms"].get("encode_decode_at_batch_median"))
except Exception as e:
    print("$name: no line:", e); print(open("gpurun_out/s3_$name.err").read()[-600:])
PY
}
run base0_a ADK_RB16_BALANCE=0
run b1 ADK_RB16_BALANCE=1
run b2 ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/b2/libaudiodec_hip.so
run b3 ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/b3/libaudiodec_hip.so
run b4 ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/b4/libaudiodec_hip.so
run dp_a ADK_RB16_BALANCE=0 ADK_RB16_DEPHASE=5.5,7.5,11
run dp_b ADK_RB16_BALANCE=0 ADK_RB16_DEPHASE=3,4,6
run base0_b ADK_RB16_BALANCE=0
