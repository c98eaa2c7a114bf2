set -x
mkdir -p gpurun_out
timeout 300 python tools/conv_bench.py --shape s3,s2,e0,e1 --impl 4 --check 2>&1 | tail -12
timeout 300 python tools/conv_bench.py --shape s3,s2,e0,e1 --impl 3 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split16 or conv_kernels or pipeline" 2>&1 | tail -8
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --precision split16 > gpurun_out/bench_split16.json 2> gpurun_out/bench_split16.err; tail -3 gpurun_out/bench_split16.err; cut -c1-200 gpurun_out/bench_split16.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_split16.json'))
print(d['value'], d['ms_per_step'], d['latency_ms'], d['roofline']['kernel'], d['roofline']['achieved'])
for k,v in d.get('kernels',{}).items(): print(k,v)
PY
