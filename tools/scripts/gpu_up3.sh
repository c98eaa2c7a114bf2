timeout 300 python tools/conv_bench.py --shape up3 --impl 5 --check 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/conv_bench.py --shape up3 --impl 6 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "convtranspose or pipeline or two_stage" 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline_convtr'], d['device_error_flags'])"
