#!/bin/bash
# final round-2 state: full GPU suite, kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes (steady-state launches marked), bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3p
cd $R
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r3p_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3p_tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3p/stats -o p --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --no-self-check > $R/gpurun_out/r3p_stats.log 2>&1
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r3p/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --serial --pmc-markers --no-cpu-baseline --no-other-precision --no-op-profile --no-extra-configs --no-self-check > $R/gpurun_out/r3p_pmc_$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/r3p gpurun_out/r3p_pmc_traffic.csv | head -30
f=$(find gpurun_out/r3p/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3p_kernel_stats.csv && head -24 gpurun_out/r3p_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r3p/stats gpurun_out/r3p/pmc_*/*.db gpurun_out/r3p/pmc_*/*/*.db 2>/dev/null
cp gpurun_out/r3p_pmc_traffic.csv profiles/r2_pmc_traffic.csv      # so that the bench below reads the passes just taken
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r3p_ops.csv ) > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err
echo "bench rc=$?" >> gpurun_out/r3p_bench.err
tail -4 gpurun_out/r3p_tests.log; tail -c 600 gpurun_out/r3p_bench.json
