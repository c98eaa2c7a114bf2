mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for prec in f32 split16; do
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --precision $prec > gpurun_out/bench_$prec.json 2> gpurun_out/bench_$prec.err; tail -2 gpurun_out/bench_$prec.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$prec.json'))
print('$prec', d['value'], d['ms_per_step'], d['latency_ms'], d['roofline']['kernel'], d['roofline']['achieved'], 'flags', d['device_error_flags'])
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY
done
