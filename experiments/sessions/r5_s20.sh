#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for r in 1 2; do for h in 0 4; do echo "== round $r ADK_RB16_HELPERS=$h"; for B in 1 8 16 32; do ADK_RB16_HELPERS=$h python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done; done; done
