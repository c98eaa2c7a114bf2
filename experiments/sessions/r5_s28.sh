#!/bin/bash
# round 5, session 28: HIP-graph replay of the programs at ONE stream (host-synchronised steps): does replaying the 37 launches as graphs shorten the frame?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for r in 1 2; do
  for v in 0 1; do echo "-- ADK_GRAPH=$v"; ADK_GRAPH=$v python tools/single_stream_steps.py 1 100 2>&1 | tail -1; done
done
