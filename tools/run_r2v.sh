#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== new"; ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | tail -4
echo "== ADK_RL16_FEW=0"; ADK_RL16_FEW=0 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | tail -3
echo "== ADK_CONV_MAX_SPLIT=0"; ADK_CONV_MAX_SPLIT=0 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | tail -3
echo "== ADK_RVQ_V1=1"; ADK_RVQ_V1=1 ADK_SPLIT16=1 python tools/cfg1_time.py 2>/dev/null | tail -3
} > gpurun_out/r2v_cfg1.log 2>&1
cat gpurun_out/r2v_cfg1.log
