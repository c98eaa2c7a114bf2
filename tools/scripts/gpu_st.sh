timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_stage" 2>&1 | tail -3
for st in 2 1,2 1,2,3 1,3; do
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision --no-op-profile --stages $st 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('stages $st', d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['device_error_flags'])"; done
