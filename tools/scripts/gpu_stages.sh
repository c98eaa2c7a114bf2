timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_stage or two_hip" 2>&1 | tail -4
for st in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision --stages $st 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('stages', $st, d['value'], d['ms_per_step'], d['latency_ms'])"
done
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision --stages 2 --precision f32 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('f32 stages 2', d['value'], d['ms_per_step'])"
