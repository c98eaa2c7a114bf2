#!/bin/bash
# Round 6, session 14: knock-out -- what would conv_sk16<64x64> cost on the stage-0 / encoder-block-3 convs if the two waves that multiply the same
# 32 rows did not BOTH fetch the weight fragments (tools/dbg/skko = -DADK_SK16_DBG=64: the wn == 1 waves fetch none; results are garbage)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p tools/bin
hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/kbench.cpp -L audiodec_amd -laudiodec_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/audiodec_amd" -o tools/bin/kbench
hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/kbench.cpp -L tools/dbg/skko -laudiodec_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/tools/dbg/skko" -o tools/bin/kbench_ko
for r in 1 2; do for sh in s0 s0d1 e3 o0 up1 d3; do
  echo "== $sh product:"; tools/bin/kbench conv $sh 6 256 300 2>&1 | tail -1
  echo "== $sh knock-out:"; tools/bin/kbench_ko conv $sh 6 256 300 2>&1 | tail -1
done; done
