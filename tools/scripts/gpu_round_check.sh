set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r1e_tests.log
cat gpurun_out/r1e_tests.log
./tools/bin/mfma_f16_probe > gpurun_out/f16_probe.txt 2>&1; cat gpurun_out/f16_probe.txt
timeout 600 python tools/config_bench.py --out gpurun_out/configs.json > gpurun_out/configs.log 2>&1; tail -30 gpurun_out/configs.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --groups 2 > gpurun_out/bench_groups2.json 2> gpurun_out/bench_groups2.err; cut -c1-400 gpurun_out/bench_groups2.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1e -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1e_bench.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_r1e_bench.log | cut -c1-600
ls gpurun_out/prof_r1e | head
