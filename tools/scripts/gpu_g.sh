for g in 128 192 256 320 384; do
ADK_SK16_KD=1 ADK_CONV_G=$g timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('G', $g, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'])"; done
