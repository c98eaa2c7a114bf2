#!/bin/bash
# round 5, session 25: conv_gv16 over several 32-column tiles (ADK_GV16_MAXN = 32 (default) | 64 | 256): parity of the kernel, then the
# quick bench (headline, batch latency, serial per-op times of the small matrix-core launches) and host-synchronised steps at 64 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "few_column" ) > gpurun_out/s25_tests.log 2>&1; tail -4 gpurun_out/s25_tests.log; grep -n "^E " gpurun_out/s25_tests.log | head
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for r in 1 2; do
  for v in 32 256 64; do
    [ "$r" = 2 ] && [ "$v" = 64 ] && continue
    ADK_GV16_MAXN=$v timeout 600 python bench.py $ARGS --dump-ops gpurun_out/s25_ops_${v}_$r.csv > gpurun_out/s25_${v}_$r.json 2> gpurun_out/s25_${v}_$r.err
    echo "== ADK_GV16_MAXN=$v round $r rc=$?"
    python - <<PY
import json, csv
try:
    d = json.loads(open("gpurun_out/s25_${v}_$r.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "batch latency", d["latency_ms"].get("encode_decode_at_batch_median"), "single", d["latency_ms"].get("encode_decode_single_stream_median"))
    rows = list(csv.DictReader(open("gpurun_out/s25_ops_${v}_$r.csv")))
    for fam in ("conv_gv16", "conv_sk16<64x64>", "conv_sk16<32x128>", "conv_rb16"):
        sel = [x for x in rows if x["kernel"].startswith(fam)]
        print(f"  {fam}: {len(sel)} launches, serial {sum(float(x['us']) for x in sel):.1f} us, pipelined {sum(float(x['us_pipelined']) for x in sel):.1f} us")
    if "$r" == "1":
        for x in rows:
            if x["kernel"].startswith("conv_gv16"): print("    ", x["prog"], x["op"], x["us"], x["us_pipelined"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/s25_${v}_$r.err").read()[-1500:])
PY
  done
done
for v in 32 64; do echo "-- 64 streams, ADK_GV16_MAXN=$v"; ADK_GV16_MAXN=$v python tools/single_stream_steps.py 64 60 2>&1 | tail -1; done
