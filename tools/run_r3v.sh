#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
{
ADK_SPLIT16=1 python tools/hiccup.py vctk_sym 32 1500 2>/dev/null | tail -1
NOGC=1 ADK_SPLIT16=1 python tools/hiccup.py vctk_sym 32 1500 2>/dev/null | tail -1
ADK_SPLIT16=1 python tools/hiccup.py vctk_sym 64 1500 2>/dev/null | tail -1
ADK_SPLIT16=1 python tools/hiccup.py vctk_v1 256 800 2>/dev/null | tail -1
} > gpurun_out/r3v_hiccup.log 2>&1
cat gpurun_out/r3v_hiccup.log
