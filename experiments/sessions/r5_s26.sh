#!/bin/bash
# round 5, session 26: conv_up16 with two chunks per workgroup (second chunk prefetched) for launches of >= 1024 chunks: parity, then the launch alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "upsampling_streamer" ) > gpurun_out/s26_tests.log 2>&1; tail -4 gpurun_out/s26_tests.log; grep -n "^E " gpurun_out/s26_tests.log | head
for r in 1 2; do
  for v in 1 2; do ADK_UP16_TPW=$v python tools/up16_time.py 256:5 256:10 64:40 2>&1 | grep -v amdgpu.ids; done
done
