#!/bin/bash
# The round's final capture, one gpurun call:  tools/final_round.sh <tag>
#  1. rocprofv3 kernel traces (pipelined + serial) and PMC passes of the timed steps (tools/profile_round.sh) -> copied into profiles/ ON THE BOX
#  2. bench.py (100 steps, all legs): its roofline.frac now comes from the trace of this very build;  3. bench.py in the driver's form (20 / 5)
#  4. the phase timeline of the north-star's named kernel (debug library tools/dbg/ou1, built beforehand: python tools/ou16_trace.py --build)
#  5. the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1
bash tools/profile_round.sh $tag
cp gpurun_out/${tag}_kernel_stats_steady.csv gpurun_out/${tag}_kernel_stats_serial.csv gpurun_out/${tag}_pmc_traffic.csv profiles/
cp gpurun_out/${tag}_kernel_stats_T5_steady.csv gpurun_out/${tag}_kernel_stats_T5_serial.csv gpurun_out/${tag}_pmc_traffic_T5.csv profiles/
bash tools/gpu_session.sh $tag bench
( timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_20_steps.json 2> gpurun_out/${tag}_bench_20_steps.err ); echo "bench20 rc=$?"
( python tools/ou16_trace.py 256; python tools/ou16_trace.py 1 ) 2>&1 | grep -v "^Load\|amdgpu.ids" > gpurun_out/${tag}_ou16_trace.log; echo "trace rc=$?"
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1 ); echo "smoke rc=$?"; tail -n 2 gpurun_out/${tag}_smoke.log
[ "${ADK_FINAL_SKIP_TESTS:-0}" = 1 ] || bash tools/gpu_session.sh $tag tests     # (a re-capture of an already tested build on another box skips the suite)
