#!/bin/bash
# Round 6, session 12: the stage-0 convs on the stream-K kernel's other tile shapes (ADK_CONV_CFG forces one for every layer; only the stage-0 figures are read)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for cfg in 2 0 1 3 6; do echo "== ADK_CONV_CFG=$cfg (2 = 64x64 default, 0 = 128x64, 1 = 128x128 4 waves, 3 = 64x128, 6 = 128x128 8 waves)"; ADK_WT16=0 ADK_CONV_CFG=$cfg timeout 300 python tools/wt16_bench.py 256 1 2>&1 | grep "wt16=0"; done
