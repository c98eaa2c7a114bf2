#!/bin/bash
# One GPU-box session: tools/gpu_session.sh <tag> [what ...]   (run through gpurun; logs under gpurun_out/<tag>_*)
#   chains   the residual-chain bit-identity tests + the benched-configuration parity test
#   tests    the whole GPU suite
#   bench    python bench.py (default run) with the per-op table
#   quick    python bench.py without CPU legs / extra configs / other precision
#   kbench   build tools/bin/kbench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    chains)
      ( time timeout 900 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "chains or benched" ) > gpurun_out/${tag}_chains.log 2>&1
      echo "chains rc=$?" >> gpurun_out/${tag}_chains.log; tail -15 gpurun_out/${tag}_chains.log ;;
    tests)
      ( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/${tag}_tests.log 2>&1
      echo "tests rc=$?" >> gpurun_out/${tag}_tests.log; tail -25 gpurun_out/${tag}_tests.log ;;
    bench)
      ( time timeout 900 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/${tag}_ops.csv ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
      echo "bench rc=$?" >> gpurun_out/${tag}_bench.err; tail -3 gpurun_out/${tag}_bench.err; head -c 1500 gpurun_out/${tag}_bench.json; echo ;;
    quick)
      ( time timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-other-precision --dump-ops gpurun_out/${tag}_ops.csv ) > gpurun_out/${tag}_quick.json 2> gpurun_out/${tag}_quick.err
      echo "quick rc=$?" >> gpurun_out/${tag}_quick.err; tail -3 gpurun_out/${tag}_quick.err; head -c 1200 gpurun_out/${tag}_quick.json; echo; cat gpurun_out/${tag}_ops.csv ;;
    kbench)
      mkdir -p tools/bin && hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/kbench.cpp -L audiodec_amd -laudiodec_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/audiodec_amd" -o tools/bin/kbench ;;
    *) echo "unknown step $what" ;;
  esac
done
