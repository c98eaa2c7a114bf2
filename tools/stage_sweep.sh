#!/bin/bash
# pipeline throughput for different vocoder program cuts (bench.py --stages): tools/stage_sweep.sh "2" "1" "1,2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check --no-op-profile"
for st in "$@"; do
  for r in 1 2; do
    timeout 300 python bench.py $ARGS --stages "$st" > gpurun_out/stages_$(echo $st | tr ',' '_')_$r.json 2> gpurun_out/stages.err
    python -c "
import json; d=json.loads(open('gpurun_out/stages_$(echo $st | tr ',' '_')_$r.json').read().strip().splitlines()[-1]); print('stages $st run $r:', d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'])"
  done
done
