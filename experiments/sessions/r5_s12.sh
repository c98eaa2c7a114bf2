#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
t() { echo "== $*"; for B in 1 32; do env "$@" python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done; }
t X=1
t ADK_RB16_RING=0
t ADK_CONV_MAX_SPLIT=3
t ADK_CONV_MAX_SPLIT=8
t ADK_CONV_MIN_UNITS=1
t ADK_CONV_MIN_UNITS=4
t ADK_SK16_KD=2
t ADK_CONV_OCC=1
t X=1
