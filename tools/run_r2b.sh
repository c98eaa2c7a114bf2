#!/bin/bash
# round 2, call B: micro-benchmarks (no torch) of the new kernels with cross-checks, then the tests that cover them, then bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K=tools/bin/kbench
{
echo "== rvq =="
for r in 1 32 256 512; do $K rvq $r 200; ADK_RVQ_V1=1 $K rvq $r 200; done
echo "== up3 (north-star kernel): streamer / rows-in-LDS / stream-K, checked against the f32 stream-K kernel =="
$K conv up3 7 256 300 2; $K conv up3 5 256 300 2; $K conv up3 6 256 300 2
$K conv up3 7 1 300 2; $K conv up3 7 64 300 2; $K conv up3 7 1024 100 2
echo "== deep layers: big-tile kernel (auto) vs first-round stream-K (ADK_CONV_GK16=0), both checked against the f32 stream-K kernel =="
for s in s0 s0d1 s1 s1d1 e3 e2 up0 up1 up2 d1 d2 d3 o0 o1 r2 r3 in; do
  timeout 60 $K conv $s 4 256 100 2; ADK_CONV_GK16=0 timeout 60 $K conv $s 4 256 100; 
done
echo "== deep layers at other stream counts =="
for b in 1 16 64; do for s in s0 s1 e3 up0; do timeout 60 $K conv $s 4 $b 100 2; ADK_CONV_GK16=0 timeout 60 $K conv $s 4 $b 100; done; done
} > gpurun_out/r2b_kbench.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_b256.py tests/test_gpu_parity.py -q -m gpu -x --durations=5 ) > gpurun_out/r2b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2b_tests.log
( time timeout 600 python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/r2b_ops.csv ) > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
echo "bench rc=$?" >> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_kbench.log | tail -80; tail -5 gpurun_out/r2b_tests.log; tail -c 600 gpurun_out/r2b_bench.json
