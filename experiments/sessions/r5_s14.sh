#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
t() { echo "== $*"; for B in 1 32; do env "$@" python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done; }
t X=1
for n in f1 f2 f3 f4 f5; do t ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/$n/libaudiodec_hip.so; done
t X=1
