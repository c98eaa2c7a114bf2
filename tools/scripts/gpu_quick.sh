mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or batched or chunked" 2>&1 | tail -3
for prec in f32 split16; do
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --precision $prec > gpurun_out/bench_$prec.json 2> gpurun_out/bench_$prec.err; tail -2 gpurun_out/bench_$prec.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$prec.json'))
print('$prec', d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'], d['roofline']['kernel'], d['roofline']['achieved'], 'flags', d['device_error_flags'])
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY
done
