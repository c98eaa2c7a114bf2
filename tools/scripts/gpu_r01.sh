timeout 200 python tools/conv_bench.py --shape r0,r1 --impl 5 --check 2>&1 | grep -v amdgpu
timeout 200 python tools/conv_bench.py --shape r0,r1 --impl 6 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline or conv_kernels" 2>&1 | tail -3
for rep in 1 2; do timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encoder_kernels_at_batch'], d['device_error_flags'])"; done
