#!/bin/bash
# Round 6, session 11: conv_wt16 inside the three-stream pipeline: off / 128 rows x 2 buffers (two workgroups per CU) / 128 rows x 3 buffers (one per CU), alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2; do for cfg in "0 128 2" "1 128 2" "1 128 3" "1 256 2"; do set -- $cfg
  ADK_WT16=$1 ADK_WT16_TM=$2 ADK_WT16_NB=$3 timeout 600 python bench.py $ARGS > gpurun_out/r6s11_$1_$2_$3_$r.json 2> gpurun_out/r6s11_$1_$2_$3_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r6s11_$1_$2_$3_$r.json").read().strip().splitlines()[-1])
    k = d["kernels"]
    print("wt16=$1 rows=$2 buffers=$3 round $r: value", d["value"], "ms/step", d["ms_per_step"], "batch", d["summary"]["latency_ms"]["batch"], {n: (v["ms_per_step_serial"], v["ms_per_step"]) for n, v in k.items() if n.startswith(("conv_wt16", "conv_sk16<64"))})
except Exception as e:
    print("no line:", e); print(open("gpurun_out/r6s11_$1_$2_$3_$r.err").read()[-800:])
PY
done; done
