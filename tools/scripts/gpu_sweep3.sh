for kd in 1 2; do for mu in 2 4; do
ADK_SK16_KD=$kd ADK_CONV_MIN_UNITS=$mu timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('KD', $kd, 'min_units', $mu, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'], d['device_error_flags'])"; done; done
ADK_BENCH_WORKGROUPS=256 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile --precision f32 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('f32 wg256', d['value'], d['ms_per_step'])"
ADK_BENCH_WORKGROUPS=0 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile --precision f32 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('f32 wg512', d['value'], d['ms_per_step'])"
