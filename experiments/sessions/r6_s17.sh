#!/bin/bash
# Round 6, session 17 (18: W1 staged by column block, 19: W2 requested behind block 5, idle eighth wave): the eight-wave conv_ou16 as the only product form: fused-vs-two-launch test,
# benched-configuration test, phase clocks, one short bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "fused_residual_units or benched" ) > gpurun_out/r6s${S:-17}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6s${S:-17}_tests.log
for b in 256 1; do timeout 300 python tools/ou16_trace.py $b 2>&1 | grep -v "^Load\|amdgpu.ids" | tee -a gpurun_out/r6s${S:-17}_trace.log; done
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --no-guarded --no-t5 --no-self-check"
timeout 600 python bench.py $ARGS > gpurun_out/r6s${S:-17}_bench.json 2> gpurun_out/r6s${S:-17}_bench.err; tail -c 1500 gpurun_out/r6s${S:-17}_bench.json
