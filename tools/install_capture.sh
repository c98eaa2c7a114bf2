#!/bin/bash
# Copy what tools/final_round.sh <tag> left under gpurun_out/ into profiles/ (the files the docs and bench.py's frac_source cite).
tag=${1:-r6}
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/${tag}_kernel_stats_steady.csv gpurun_out/${tag}_kernel_stats_serial.csv gpurun_out/${tag}_pmc_traffic.csv gpurun_out/${tag}_kernel_stats_T5_steady.csv \
   gpurun_out/${tag}_kernel_stats_T5_serial.csv gpurun_out/${tag}_pmc_traffic_T5.csv gpurun_out/${tag}_ops.csv profiles/
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench_latest.json
cp gpurun_out/${tag}_bench_20_steps.json profiles/${tag}_bench_20_steps.json
cp gpurun_out/${tag}_ou16_trace.log profiles/${tag}_ou16_trace_final_capture.log
cp gpurun_out/${tag}_tests.log profiles/${tag}_gpu_tests.log
for f in chain_pipeline end_to_end_chain_encoder_split16 end_to_end_f32 end_to_end_split16 same_latent; do cp gpurun_out/soak_$f.json profiles/${tag}_soak_$f.json; done
