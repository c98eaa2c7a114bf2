mkdir -p gpurun_out
timeout 300 python tools/conv_bench.py --shape s0,s1,s2,e2,e3,up0,up3,d3,p --impl 6 --check 2>&1 | grep -v amdgpu.ids
echo ---- f32 stream-K
timeout 300 python tools/conv_bench.py --shape s0,s1,s2,e2,e3,up0,up3,d3,p --impl 2 2>&1 | grep -v amdgpu.ids
echo ---- cfg sweep split16 stream-K
timeout 300 python tools/conv_bench.py --shape s0,s1,e2,e3 --impl 6 --cfg=0,2,3,4 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split16 or conv_kernels or pipeline" 2>&1 | tail -8
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --precision split16 > gpurun_out/bench_split16.json 2> gpurun_out/bench_split16.err; tail -3 gpurun_out/bench_split16.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_split16.json'))
print(d['value'], d['ms_per_step'], d['latency_ms'], d['roofline']['kernel'], d['roofline']['achieved'])
for k,v in d.get('kernels',{}).items(): print(k,v)
PY
