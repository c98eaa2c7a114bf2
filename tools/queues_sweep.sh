#!/bin/bash
# Does the number of hardware queues (GPU_MAX_HW_QUEUES, ROCm default 4) limit schedules with more than three HIP streams?
# tools/queues_sweep.sh: quick bench with the vocoder cut at 2 / at 1 and 2 / at 1, 2 and 3 and the RVQ search on its own stream, at 4 and 8 queues.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check --no-op-profile"
run() {   # label, env..., -- bench args
  label=$1; shift
  out=gpurun_out/queues_${label}.json
  env "$@" timeout 200 python bench.py $ARGS $EXTRA > $out 2> gpurun_out/queues.err
  python -c "
import json; d=json.loads(open('$out').read().strip().splitlines()[-1]); print('$label:', d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'])" 2>/dev/null || { echo "$label: no line"; tail -3 gpurun_out/queues.err; }
}
for r in 1 2; do
  EXTRA="--stages 2"     run q4_cut2_$r      GPU_MAX_HW_QUEUES=4
  EXTRA="--stages 2"     run q8_cut2_$r      GPU_MAX_HW_QUEUES=8
  EXTRA="--stages 1,2"   run q8_cut12_$r     GPU_MAX_HW_QUEUES=8
  EXTRA="--stages 2"     run q8_cut2_rvqown_$r GPU_MAX_HW_QUEUES=8 ADK_BENCH_RVQ=own
  EXTRA="--stages 1,2,3" run q8_cut123_$r    GPU_MAX_HW_QUEUES=8
done
