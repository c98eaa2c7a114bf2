#!/bin/bash
# per-op serial times of the wide stream-K launches under forced tile configs (ADK_CONV_CFG) / chunk depth, shadows on, conv_gk16 off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 20 --warmup 5 --preroll 8 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
for v in "$@"; do
  env $v timeout 300 python bench.py $ARGS --dump-ops gpurun_out/cfg_$(echo $v | tr '= ' '__').csv > /dev/null 2> gpurun_out/cfg.err
  echo "== $v rc=$?"; grep -E "blocks.0.convs1.0|blocks.0.convs1.2|blocks.0.conv_out|upsamples.[01],|conv_blocks.3.res_units.0|conv_blocks.3.conv|conv_blocks.2.conv|projector" gpurun_out/cfg_$(echo $v | tr '= ' '__').csv | cut -d, -f2,3,10 | tr '\n' ' '; echo
done
