#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== new"; N_ITER=40 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | awk '{ if ($2 > 2 || $8 > 2) print NR": "$0 }'
echo "== new, HIP_ENABLE_DEFERRED_LOADING=0"; HIP_ENABLE_DEFERRED_LOADING=0 N_ITER=40 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | awk '{ if ($2 > 2 || $8 > 2) print NR": "$0 }'
echo "== ADK_RL16_FEW=0"; N_ITER=40 ADK_RL16_FEW=0 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | awk '{ if ($2 > 2 || $8 > 2) print NR": "$0 }'
echo "== f32"; N_ITER=40 ADK_SPLIT16=0 python tools/cfg1_time.py 2>/dev/null | awk '{ if ($2 > 2 || $8 > 2) print NR": "$0 }'
} > gpurun_out/r2w_cfg1.log 2>&1
cat gpurun_out/r2w_cfg1.log
