#!/bin/bash
# profiles of the final round-2 state: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes over bench.py, then the full test suite and bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2k
cd $R
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r2k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2k_tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2k/stats -o p --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --no-self-check > $R/gpurun_out/r2k_stats.log 2>&1
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r2k/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --serial --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check > $R/gpurun_out/r2k_pmc_$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/r2k gpurun_out/r2k_pmc_traffic.csv | head -40
ls gpurun_out/r2k/stats | head; find gpurun_out/r2k/stats -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r2k/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r2k_kernel_stats.csv && head -30 gpurun_out/r2k_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r2k/stats/*/*.db 2>/dev/null
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r2k_ops.csv ) > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err
echo "bench rc=$?" >> gpurun_out/r2k_bench.err
tail -4 gpurun_out/r2k_tests.log; tail -c 700 gpurun_out/r2k_bench.json
