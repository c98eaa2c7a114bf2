timeout 200 python tools/conv_bench.py --shape s0,s1,e2,s3,s2 --impl 4 --check 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['latency_ms'], d['device_error_flags'])
for k,v in d.get('kernels',{}).items(): print('  ',k,v)"
