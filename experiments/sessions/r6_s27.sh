#!/bin/bash
# Round 6, session 27: cache policy of conv_ou16's output stores (aux of raw_buffer_store: 0 default, 2 nt, 16 sc1, 17 sc0 sc1, 18 sc1 nt) by rocprofv3
# over the serial schedule, each variant twice, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do for v in 0 2 16 17 18; do
  ADK_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbg/st$v/libaudiodec_hip.so bash tools/profile_round.sh r6s27_${v}_$r serial-only > /dev/null 2>&1
  echo "aux=$v round $r: $(grep ou16 gpurun_out/r6s27_${v}_${r}_kernel_stats_serial.csv | cut -d, -f1-8)  | step sum $(grep 'summed kernel time' gpurun_out/r6s27_${v}_${r}_kernel_stats_serial.csv | sed 's/.*summed kernel time \([0-9.]*\).*/\1/')"
done; done
