#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for w in -1 1 0; do
  echo "== ADK_RB16_WARM=$w"
  for B in 1 8 32; do ADK_RB16_WARM=$w python tools/single_stream_steps.py $B 60 2>&1 | tail -1; done
done
