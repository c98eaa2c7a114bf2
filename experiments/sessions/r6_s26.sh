#!/bin/bash
# Round 6, session 26: conv_ou16 flags (tools/dbg/f1) against the barrier (tools/dbg/f0) by the per-op HIP events of the serial schedule and by rocprofv3, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2 3; do for v in 0 1; do
  ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/f$v/libaudiodec_hip.so timeout 600 python bench.py $ARGS > gpurun_out/r6s26_v${v}_$r.json 2> gpurun_out/r6s26_v${v}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s26_v${v}_$r.json").read().strip().splitlines()[-1])
    n = d["summary"]["north_star_kernel"]
    print("flags=$v round $r: value", d["value"], "ou16 events serial frac", n["frac_events_serial"], "-> us incl. one event record", round(29.66e6 / (n["frac_events_serial"] * 8e12) * 1e6, 2))
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s26_v${v}_$r.err").read()[-800:])
PY
done; done
cd /tmp
for v in 0 1 0 1; do
  rm -rf /tmp/rp_$v; ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/f$v/libaudiodec_hip.so timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --serial $ARGS --no-op-profile > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("/tmp/rp_$v/**/p_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "ou16" in r["Name"]: print("flags=$v rocprofv3 --serial:", r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
PY
done
