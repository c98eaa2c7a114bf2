for kd in 1 2; do for occ in 1 2; do
ADK_SK16_KD=$kd ADK_CONV_OCC=$occ timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('KD', $kd, 'occ', $occ, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'])"; done; done
