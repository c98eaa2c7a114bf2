#!/bin/bash
# Knock-outs of conv_gk16 (ADK_GK16_DBG bits; results are garbage, only the per-op times count): serial per-op times of the stage-0 convs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 20 --warmup 5 --preroll 8 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for d in "$@"; do
  ADK_GK16=1 ADK_GK16_DBG=$d timeout 300 python bench.py $ARGS --dump-ops gpurun_out/gk_dbg_$d.csv > gpurun_out/gk_dbg_$d.json 2> gpurun_out/gk_dbg_$d.err
  echo "== dbg $d rc=$?"; grep -E "blocks.0.convs|conv_out|upsamples.1" gpurun_out/gk_dbg_$d.csv | cut -d, -f2,3,10 | tr '\n' ' '; echo
done
